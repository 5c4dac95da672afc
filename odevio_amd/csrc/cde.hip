// Neural-CDE pose path (reference src/models/PoseCDE.py:76-103, CDEFunc in src/models/ODEFunc.py:44-83;
// torchcde 0.2.5 -> torchdiffeq 0.2.3 in the reference).
//
//   f(t, z) = reshape(tanh(W_L act(... act(W_1 z + b_1)) + b_L), [B, H, C]) . dX/dt(t),   C = H + 1
//
// The last Linear has H*C outputs (2.1 M parameters at H = 128, 1.08 G = 4.3 GB at H = 1024) and is the only part with
// real traffic; everything else in a vector-field evaluation is microseconds.  This file is built around that:
//
//  * `cde_stream_kernel` - the last layer as a pure HBM weight stream.  One persistent workgroup per CU walks whole
//    h-groups of C weight rows; every wave streams its own 16-row blocks (64 KB contiguous at H = 1024) through a
//    wave-private LDS ring filled by LDS-DMA (global_load_lds_dwordx4: full 256-byte row segments, no staging
//    registers), 28 KB in flight per wave = 112 KB per CU, counted vmcnt waits and NO barrier in the loop.  The batch
//    activations x [16][H] sit in REGISTERS in MFMA-operand order (a 1-wave-per-SIMD kernel has 512 of them), so LDS
//    carries nothing but the stream.  Products on the fp32 MFMA (v_mfma_f32_16x16x4_f32: an exact fmaf chain, 1/3 of
//    the MFMA pipe at the HBM rate for 16 batch rows); bias + tanh + the contraction with dX/dt are fused into the
//    epilogue, so the [B, H, C] tensor never exists; dX/dt is computed on the fly from the observations.
//  * dX/dt of the RECTILINEAR control path (torchcde linear_interpolation_coeffs(rectilinear=0)) is sparse by
//    construction: even pieces move only the time channel, odd pieces only the features.  On an even piece all but ONE
//    of the C rows per h are multiplied by an exact zero, so only those H rows are evaluated (`cde_hidden_kernel` with a
//    row stride): 4 MB instead of 4.3 GB, bit-identical to adding the zeros.
//  * the controller of the adaptive solver lives on the device (`cde_ctl_*`, struct CdeCtl): error norm, accept / reject,
//    next step size, clipping at the knots, dense output, FSAL / re-evaluation after a jump.  No scalar travels to the
//    host inside a step.
#include <algorithm>
#include <cmath>

#include "../../include/odevio.h"
#include "cde.h"
#include "common.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float cde_act(float v, int act) {
  switch (act) {
    case 0: return tanhf(v);
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.01f * v;
    case 3: return v > 20.f ? v : log1pf(expf(v));
    default: return v;
  }
}

// Is this kernel's work wanted?  (fixed-grid solvers: always; adaptive: not after `done`, and the re-evaluation of f
// behind a jump only when the controller asked for it)
__device__ __forceinline__ bool cde_wanted(const CdeWhen& wh) {
  if (!wh.ctl) return true;
  if (wh.ctl->done) return false;
  return !wh.only_on_jump || wh.ctl->need_jump_eval != 0;
}
__device__ __forceinline__ int cde_seg(const CdeWhen& wh) { return wh.ctl ? wh.ctl->seg_stage[wh.slot] : wh.seg; }

// out[b][n] = act(sum_k x[b][k] W[n][k] + bias[n]);  one wave per output column, lanes stride K.  Any K, N: used once per
// forward for the initial layer (K = C = H + 1), and for hidden sizes that are not multiples of 128.
__global__ __launch_bounds__(256) void cde_linear_kernel(CdeWhen wh, const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int B, int K, int ldx, int N, int act) {
  if (!cde_wanted(wh)) return;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* w = W + (size_t)n * K;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(x[(size_t)b * ldx + k], w[k], s);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[(size_t)b * N + n] = cde_act(s + bias[n], act);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny GEMM on the fp32 MFMA: out[b][n] = epi(sum_k x[b][k] W[n * w_stride + k] + bias[n * b_stride]) for 16 output
// columns per 4-wave workgroup; the waves split K (H / 4 each) and combine through LDS in wave order (deterministic).
//   mode 0: epi = act                          (CDEFunc hidden layers: w_stride = H, b_stride = 1)
//   mode 1: epi = tanh(.) * dXdt[b][0]         (last layer on an EVEN piece of the control path: only channel 0 moves, so
//                                               row h*C of each h-group is the only one with a non-zero multiplier;
//                                               w_stride = C*H, b_stride = C; dXdt[b][0] = tau_{i+1} - tau_i from obs)
// MFMA 16x16x4: lane (r = lane & 15, q = lane >> 4) feeds A[r][k = q] and B[k = q][n = r].  A lane loads 4 consecutive k
// (one float4) of its row and hands element j to MFMA j, whose k-set is then {16 s + 4 q' + j}: any bijection of k works
// as long as the x fragment uses the same one.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cde_hidden_kernel(CdeWhen wh, const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out, int B, int H,
                                                         size_t w_stride, int b_stride, int act, int mode,
                                                         const float* __restrict__ obs, int L, int C) {
  __shared__ float red[4][16][17];
  if (!cde_wanted(wh)) return;
  int seg = 0;
  if (mode == 1) {
    seg = cde_seg(wh);
    if (seg & 1) return;   // odd piece: the streaming kernel's job
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int kq = H >> 2;                         // K range of this wave
  const float* wrow = W + (size_t)(n0 + r) * w_stride + wave * kq + 4 * q;
  for (int b0 = 0; b0 < B; b0 += 16) {
    const int nb = min(16, B - b0);
    const float* xrow = x + (size_t)(b0 + min(r, nb - 1)) * H + wave * kq + 4 * q;   // rows past nb re-read the last one (ignored)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // every load of the wave's K range in flight at once (H <= 1024: at most 16 k-steps per wave, clamped addresses past
    // the range), then the MFMAs: one round trip to L2 instead of one per loop trip (7.6 -> ~4 us per launch at H = 1024)
    f32x4 wv[16], xv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = min(16 * i, kq - 16);
      wv[i] = *reinterpret_cast<const f32x4*>(wrow + k);
      xv[i] = *reinterpret_cast<const f32x4*>(xrow + k);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (16 * i < kq) {   // wave-uniform
        if (i & 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][j], xv[i][j], acc1, 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][j], xv[i][j], acc0, 0, 0, 0);
        }
      }
    }
    // D: column (= batch row) = lane & 15, row (= output column) = 4 * (lane >> 4) + e
    __syncthreads();   // red free
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][4 * q + e][r] = acc0[e] + acc1[e];
    __syncthreads();
    const int nl = tid >> 4, bb = tid & 15;   // 256 threads = 16 columns x 16 batch rows
    if (bb < nb) {
      const float s = (red[0][nl][bb] + red[1][nl][bb]) + (red[2][nl][bb] + red[3][nl][bb]);
      const int n = n0 + nl;
      float v = s + bias[(size_t)n * b_stride];
      if (mode == 0) v = cde_act(v, act);
      else {
        const float* o0 = obs + ((size_t)(b0 + bb) * L + (seg >> 1)) * C;
        v = tanhf(v) * (o0[C] - o0[0]);
      }
      out[(size_t)(b0 + bb) * H + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The last layer as an HBM weight stream (odd pieces of the control path: every feature channel moves).
//   out[b][h] = sum_{c >= 1} tanh(W[h*C + c] . x[b] + bias[h*C + c]) * (x_{i+1}[b][c] - x_i[b][c])      (c = 0: multiplier 0)
// Work: workgroup g takes the h-groups g, g + gridDim.x, ...; inside a group wave w takes the 16-row blocks w, w + 4, ...
// A block is 16 rows x H columns = NP pieces of 16 rows x 64 columns (4 KB, 256 B per row); a piece is 4 DMA instructions
// (4 rows each).  LDS image of a piece: row r at r * 256 with its sixteen 16-byte chunks XOR-permuted by r (applied on
// the SOURCE address: the DMA writes lane-linearly), so the MFMA fragment reads (lane = row + 16 * k-quarter,
// ds_read_b128) touch 16 different bank quads per lane group: conflict-free.
// Ring: 8 slots per wave, piece p + 7 is requested before piece p is multiplied; `s_waitcnt vmcnt(28)` = all but the 7
// youngest pieces have landed.  Past the last block the stream re-requests its last piece (unused) so that the counts
// stay exact; nothing is ever requested outside the weight matrix.
// BF16 = false: fp32 weights, the parity path.  BF16 = true (--dtype bf16 / fp16): the last layer stored as bf16 (half the
// stream) and multiplied on the bf16 MFMA (v_mfma_f32_16x16x32_bf16) against x held as two bf16 pieces, fp32
// accumulation: 1/4 of the MFMA time of the fp32 path, so the kernel stays a pure stream (widening the weights to fp32
// for the fp32 MFMA left it MFMA-bound: 500 us per evaluation against 356); bias, tanh, the contraction with dX/dt, the
// state and the controller stay fp32.  Outside the 1e-4 claim (the weights carry 8 significant bits).
// ---------------------------------------------------------------------------------------------------------------------
#define CS_SLOTS 8
#define CS_AHEAD 7
#define CS_PIECE 4096
#define CS_WAVE_LDS (CS_SLOTS * CS_PIECE)
#define CS_LDS (4 * CS_WAVE_LDS + 4096)   // rings + per-wave partial sums [8 h-groups][4 waves][16]... see red

template <int H, bool BF16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void cde_stream_kernel(
    CdeWhen wh, const float* __restrict__ x, const void* __restrict__ Wv, const float* __restrict__ bias,
    const float* __restrict__ obs, float* __restrict__ out, int B, int L, int C) {
  constexpr int WB = BF16 ? 2 : 4;          // bytes per stored weight
  constexpr int PK = 256 / WB;              // k-columns per piece (a piece is 16 rows x 256 bytes): 64 (fp32) or 128 (bf16)
  constexpr int NP = H / PK;                // pieces per 16-row block
  constexpr int NS = H / 16;                // float4 x fragments per lane over the whole K range
  const unsigned char* W = reinterpret_cast<const unsigned char*>(Wv);
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  if (!cde_wanted(wh)) return;
  const int seg = cde_seg(wh);
  if (!(seg & 1)) return;         // even piece: cde_hidden_kernel mode 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the cursors below are wave-uniform (SGPRs, s_cbranch)
  const int r = lane & 15, q = lane >> 4;
  const int nblk = (C + 15) >> 4;
  float* red = reinterpret_cast<float*>(lds + 4 * CS_WAVE_LDS);   // [wave][16]
  unsigned char* ring = lds + wave * CS_WAVE_LDS;
  const int obs_i = seg >> 1;

  // DMA lane geometry: instruction d of a piece covers rows 4d .. 4d+3; lane l -> row 4d + (l >> 4), LDS slot l & 15,
  // source chunk (l & 15) ^ row
  const int d_row = lane >> 4, d_slot = lane & 15;
  // fragment read offsets inside a piece for the 4 k-steps it holds: chunk (4 st + q) ^ r of row r
  int f_off[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) f_off[st] = r * 256 + (((4 * st + q) ^ r) & 15) * 16;

  for (int b0 = 0; b0 < B; b0 += CDE_BT) {
    const int nb = min(CDE_BT, B - b0);
    // ---- x fragments of the whole K range, in registers: xv[s] = x[b0 + r][16 s + 4 q .. +3].  Staged through LDS (the
    // ring is idle here): one coalesced round trip to L2 for the workgroup instead of NS dependent ones per lane.  Rows
    // past nb re-read row nb - 1: their MFMA columns are computed and never stored.
    f32x4 xv[BF16 ? 1 : NS];
    bf16x8 xh[BF16 ? NS / 2 : 1], xl[BF16 ? NS / 2 : 1];
    {
      constexpr int XLD = H + 4;   // row stride in floats: the +4 spreads the 16 rows of a fragment read over the bank quads
      float* xs = reinterpret_cast<float*>(lds);
      __syncthreads();
      for (int i = tid; i < CDE_BT * (H / 4); i += 256) {
        const int row = i / (H / 4), c4 = i - row * (H / 4);
        *reinterpret_cast<f32x4*>(xs + row * XLD + 4 * c4) =
            *reinterpret_cast<const f32x4*>(x + (size_t)(b0 + min(row, nb - 1)) * H + 4 * c4);
      }
      __syncthreads();
      // fp32 storage: fragment s covers k = 16 s + 4 q .. +3 (one 16-byte chunk of the row = 4 weights).
      // bf16 storage: a 16-byte chunk holds 8 weights = the 8 k-values one lane feeds to v_mfma_f32_16x16x32_bf16
      // (k = 32 s + 8 q .. +7).  x is carried as TWO bf16 pieces x = xh + xl (16 significant bits, the residual exact):
      // an x rounded to 8 bits differs from stage to stage of a Runge-Kutta step, which the adaptive solver's error
      // estimate reads as error - measured: 49 instead of 20 steps per window with a single piece; rounded WEIGHTS are a
      // consistent change of f and cost no steps.
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (BF16) {
          if (s < NS / 2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xs + r * XLD + 32 * s + 8 * q);
            const f32x4 b = *reinterpret_cast<const f32x4*>(xs + r * XLD + 32 * s + 8 * q + 4);
            bf16x8 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              h[j] = (__bf16)a[j]; h[4 + j] = (__bf16)b[j];
              l[j] = (__bf16)(a[j] - (float)h[j]); l[4 + j] = (__bf16)(b[j] - (float)h[4 + j]);   // the residual is exact
            }
            xh[s] = h;
            xl[s] = l;
          }
        } else {
          xv[s] = *reinterpret_cast<const f32x4*>(xs + r * XLD + 16 * s + 4 * q);
        }
      }
      __syncthreads();   // everyone has its fragments: the ring may be filled
    }
    // ---- the two cursors over this wave's (h, block) items: `c*` is multiplied, `d*` is requested (7 pieces ahead)
    int ch = blockIdx.x, cblk = wave;
    int dh = ch, dblk = cblk, dpc = 0;
    const bool have_work = ch < H && cblk < nblk;
    const unsigned char* d_ptr[4];   // per DMA instruction: source of this lane for piece 0 of the block being requested
    auto d_setup = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int row = 4 * d + d_row;
        const int c = min(dblk * 16 + row, C - 1);   // rows past C re-read the last row (masked in the epilogue)
        d_ptr[d] = W + ((size_t)dh * C + c) * H * WB + ((d_slot ^ row) & 15) * 16;
      }
    };
    auto d_issue = [&](int slot) __attribute__((always_inline)) {
      unsigned char* dst = ring + slot * CS_PIECE;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        __builtin_amdgcn_global_load_lds((gptr_t)(d_ptr[d] + dpc * 256), (lptr_t)(dst + d * 1024), 16, 0, 0);
      // advance the request cursor; behind the last block it stays on the last piece (re-requested, unused)
      if (dpc + 1 < NP) {
        ++dpc;
      } else {
        int nh = dh, nblk_ = dblk + 4;
        if (nblk_ >= nblk) { nblk_ = wave; nh = dh + gridDim.x; }
        if (nh < H) { dh = nh; dblk = nblk_; dpc = 0; d_setup(); }
      }
    };
    float part = 0.f;     // this lane's share of out[b0 + r][ch]
    if (have_work) {
      d_setup();
      unsigned head = 0, tail = 0;   // pieces multiplied / requested so far; ring slot = count & 7
#pragma unroll
      for (int p = 0; p < CS_AHEAD; ++p) d_issue(tail++ & (CS_SLOTS - 1));
      while (ch < H) {
        // epilogue operands of this block, requested BEFORE its remaining DMAs: by the time the block's last piece has
        // landed (counted wait) these older loads are complete too
        float e_bias[4], e_g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = min(cblk * 16 + 4 * q + e, C - 1);
          e_bias[e] = bias[(size_t)ch * C + c];
          const float* o0 = obs + ((size_t)(b0 + min(r, nb - 1)) * L + obs_i) * C + c;
          e_g[e] = o0[C] - o0[0];
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
          d_issue(tail++ & (CS_SLOTS - 1));                      // -> the slot multiplied in the previous iteration
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CS_AHEAD) : "memory");
          const unsigned char* pb = ring + (head++ & (CS_SLOTS - 1)) * CS_PIECE;
          f32x4 wv[4];   // all four fragment reads of the piece in flight before the first MFMA (one wave per SIMD: nothing else hides the LDS latency)
#pragma unroll
          for (int st = 0; st < 4; ++st) wv[st] = *reinterpret_cast<const f32x4*>(pb + f_off[st]);
          // all four reads complete before anything below is issued: the MFMAs need them anyway, and the next piece's
          // DMA - which the scheduler is free to interleave with these MFMAs - refills exactly this slot
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            if (BF16) {
              // 8 bf16 weights per lane and read = one operand of the bf16 MFMA (K = 32 per instruction), fp32 accumulate
              const bf16x8 wh = __builtin_bit_cast(bf16x8, wv[st]);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[4 * pc + st], acc1, 0, 0, 0);   // low piece of x
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[4 * pc + st], acc0, 0, 0, 0);   // high piece
            } else {
              const f32x4 xs = xv[4 * pc + st];
              if (st & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[st][j], xs[j], acc1, 0, 0, 0);
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[st][j], xs[j], acc0, 0, 0, 0);
              }
            }
          }
        }
        // D: column (= batch row) = lane & 15, row (= weight row inside the block) = 4 * (lane >> 4) + e
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = cblk * 16 + 4 * q + e;
          if (c >= 1 && c < C && r < nb) part = fmaf(tanhf(acc0[e] + acc1[e] + e_bias[e]), e_g[e], part);
        }
        // next item of this wave
        cblk += 4;
        if (cblk >= nblk) {
          // h-group finished for this wave: fold the k-quarters and hand the 16 batch sums to the workgroup's combiner
          float s = part;
          s += __shfl_xor(s, 16, 64);
          s += __shfl_xor(s, 32, 64);
          if (lane < 16) red[wave * 16 + lane] = s;
          part = 0.f;
          // Waves finish a group at different times; the combine needs all four.  One barrier per h-group (4.2 MB of
          // stream at H = 1024): raw s_barrier, the DMAs in flight stay in flight.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (tid < nb) out[(size_t)(b0 + tid) * H + ch] = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();   // red is free again
          cblk = wave;
          ch += gridDim.x;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unused look-ahead DMAs must not outlive the workgroup's LDS
    }
  }
}

// Generic-size fallback of the two last-layer kernels (any H % 16 == 0, both kinds of piece): register-staged, one
// workgroup per h.  Used for hidden sizes the streaming kernel is not instantiated for.
__global__ __launch_bounds__(256) void cde_last_generic_kernel(CdeWhen wh, const float* __restrict__ x, const float* __restrict__ W,
                                                               const float* __restrict__ bias, const float* __restrict__ obs,
                                                               float* __restrict__ out, int B, int H, int C, int L) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [CDE_BT][H + 8] then [4 waves][CDE_BT] reduction scratch
  if (!cde_wanted(wh)) return;
  const int seg = cde_seg(wh);
  const int obs_i = seg >> 1;
  const bool even = !(seg & 1);   // even piece: only the time channel (c = 0) moves; odd: only the feature channels
  const int ldx = H + 8;
  float* red = xs + CDE_BT * ldx;
  const int h = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nblk = (C + 15) >> 4;
  const float* Wh = W + (size_t)h * C * H;
  const float* bh = bias + (size_t)h * C;
  for (int b0 = 0; b0 < B; b0 += CDE_BT) {
    const int nb = min(CDE_BT, B - b0);
    __syncthreads();
    for (int i = tid; i < CDE_BT * H; i += 256) {
      const int bb = i / H, k = i - bb * H;
      xs[bb * ldx + k] = bb < nb ? x[(size_t)(b0 + bb) * H + k] : 0.f;
    }
    __syncthreads();
    float part = 0.f;
    const float* xrow = xs + r * ldx + 4 * q;
    for (int blk = wave; blk < nblk; blk += 4) {
      const int c_ld = min(blk * 16 + r, C - 1);
      const float* wrow = Wh + (size_t)c_ld * H + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < H; k += 16) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + k);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = blk * 16 + 4 * q + e;
        if (c < C && r < nb && (even ? c == 0 : c >= 1)) {
          const float* o0 = obs + ((size_t)(b0 + r) * L + obs_i) * C + c;
          part = fmaf(tanhf(acc[e] + bh[c]), o0[C] - o0[0], part);
        }
      }
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lane < 16) red[wave * CDE_BT + lane] = part;
    __syncthreads();
    if (tid < nb) out[(size_t)(b0 + tid) * H + h] = (red[tid] + red[CDE_BT + tid]) + (red[2 * CDE_BT + tid] + red[3 * CDE_BT + tid]);
  }
}

// out = [y +] scale * sum_j coef[j] * k_j   (and the same values into the controller's y1 buffer when mirror_y1 is set:
// the last dopri5 stage is evaluated AT y1)
__global__ void cde_combine_kernel(const CdeCtl* __restrict__ ctl, int only_on_jump, const float* __restrict__ y0, const float* __restrict__ y1,
                                   int y_sel, const float* __restrict__ kbase, CdeCoefs cf, int nk, float scale, int scale_sel,
                                   float* __restrict__ out, int mirror_y1, int n) {
  if (ctl) {
    if (ctl->done || (only_on_jump && !ctl->need_jump_eval)) return;
    if (scale_sel == 1) scale = ctl->dtf;
    else if (scale_sel == 2) scale = ctl->h0;
  }
  const float* y = y_sel < 0 ? nullptr : y0;
  if (y_sel == 1) y = ctl->yi ? y1 : y0;
  float* out2 = mirror_y1 ? const_cast<float*>(ctl->yi ? y0 : y1) : nullptr;
  // the coefficient handed to each stage is fl(a_ij * dt), like the oracle's `ks[j] * (aij * dtf)`
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = (float)(cf.c[j] * (double)scale);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float acc = 0.f;
    bool first = true;
    for (int j = 0; j < nk; ++j) {
      if (cf.c[j] == 0.0) continue;
      const float term = kbase[(size_t)j * n + i] * c[j];
      acc = first ? term : acc + term;
      first = false;
    }
    const float v = (y ? y[i] : 0.f) + acc;
    out[i] = v;
    if (out2) out2[i] = v;
  }
}

__global__ void cde_emit_copy_kernel(const float* __restrict__ src, float* __restrict__ sol, int B, int H, int P, int p) {
  const int n = B * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int b = i / H, hh = i - b * H;
    sol[((size_t)b * P + p) * H + hh] = src[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-side controller of torchdiffeq's adaptive dopri5 (restated in DESIGN.md section 3.5; the host version of
// round 1 moved here term by term): time in double, the state fp32, f sees fp32 time.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float f32_prev(float t) { return nextafterf(t, t - 1.0f); }
__device__ __forceinline__ float f32_next(float t) { return nextafterf(t, t + 1.0f); }
__device__ __forceinline__ int seg_of(float t, int n_knots) {
  int seg = (int)ceil((double)t) - 1;   // t on a knot belongs to the piece on its left (torch.bucketize)
  return max(0, min(seg, n_knots - 2));
}
__constant__ double kDP_C[7] = {0., 1 / 5., 3 / 10., 4 / 5., 8 / 9., 1., 1.};

__global__ void cde_ctl_init_kernel(CdeCtl* ctl, const double* t_out, int n_out, int n_knots, int max_steps) {
  if (threadIdx.x || blockIdx.x) return;
  CdeCtl c = {};
  c.t_begin = c.tcur = c.tprev = t_out[0];
  c.n_out = n_out;
  c.p_next = 1;
  c.n_knots = n_knots;
  c.max_steps = max_steps;
  c.done = n_out <= 1;
  // jump points: the knots 0 .. n_knots-1 of the control path that lie after t_begin
  c.jump_next = INFINITY;
  for (int kn = 0; kn < n_knots; ++kn)
    if ((double)kn > c.t_begin) { c.jump_next = (double)kn; break; }
  c.t_stage[0] = (float)c.t_begin;
  c.seg_stage[0] = seg_of((float)c.t_begin, n_knots);
  *ctl = c;
}

// rms(num / (atol + rtol * |ref|)) over n elements, one 1024-thread workgroup; same summation tree as round 1's
// cde_rms_kernel (the step decisions of every tested configuration were pinned against the oracle with it)
template <int MODE>   // 0: num = a, ref = y0;  1: num = a - b, ref = y0;  2: num = dtf * sum e_j k_j, ref = max(|y0|, |y1|)
__device__ __forceinline__ float block_rms(const float* a, const float* b, const float* y0, const float* y1, const float* ec, const bool* ez,
                                           int nk, float atol, float rtol, int n, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    float num;
    if (MODE == 2) {
      float acc = 0.f;
      bool first = true;
      for (int j = 0; j < nk; ++j) {
        if (ez[j]) continue;   // a zero tableau entry contributes no term (same association as the oracle)
        const float term = a[(size_t)j * n + i] * ec[j];
        acc = first ? term : acc + term;
        first = false;
      }
      num = acc;
    } else {
      num = MODE == 1 ? a[i] - b[i] : a[i];
    }
    const float ref = MODE == 2 ? fmaxf(fabsf(y0[i]), fabsf(y1[i])) : fabsf(y0[i]);
    const float z = num / (atol + rtol * ref);
    s = fmaf(z, z, s);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < 16; ++w) t += red[w];
  return sqrtf(t / (float)n);
}

__global__ __launch_bounds__(1024) void cde_init_step_kernel(CdeCtl* ctl, int phase, const float* __restrict__ y0, const float* __restrict__ f0,
                                                             const float* __restrict__ f1, float atol, float rtol, int n) {
  __shared__ float red[16];
  if (ctl->done) return;
  if (phase == 1) {
    const float d0 = block_rms<0>(y0, nullptr, y0, nullptr, nullptr, nullptr, 0, atol, rtol, n, red);
    const float d1 = block_rms<0>(f0, nullptr, y0, nullptr, nullptr, nullptr, 0, atol, rtol, n, red);
    if (threadIdx.x == 0) {
      const float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : (float)(0.01 * (double)d0 / (double)d1);
      ctl->d0 = d0; ctl->d1 = d1; ctl->h0 = h0;
      const float tp = (float)((float)ctl->t_begin + h0);
      ctl->t_stage[7] = tp;
      ctl->seg_stage[7] = seg_of(tp, ctl->n_knots);
    }
  } else {
    const float r2 = block_rms<1>(f1, f0, y0, nullptr, nullptr, nullptr, 0, atol, rtol, n, red);
    if (threadIdx.x == 0) {
      const float h0 = ctl->h0, d1 = ctl->d1;
      const float d2 = fabsf(r2 / h0);
      float h1;
      if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = fmaxf(1e-6f, h0 * 1e-3f);
      else h1 = (float)pow((double)(0.01f / fmaxf(d1, d2)), (double)(1.0f / 5.0f));
      ctl->d2 = d2;
      ctl->dt = fmin(100.0 * (double)h0, (double)h1);
    }
  }
}

__global__ void cde_ctl_begin_kernel(CdeCtl* ctl) {
  if (threadIdx.x || blockIdx.x || ctl->done) return;
  ctl->need_jump_eval = 0;
  if (++ctl->n_steps > ctl->max_steps) {
    ctl->status = ODEVIO_ERR_MAX_STEPS;
    ctl->done = 1;
    return;
  }
  double step = ctl->dt, t1 = ctl->tcur + step;
  int on_jump = 0;
  if (ctl->tcur < ctl->jump_next && ctl->jump_next < ctl->tcur + step) {
    on_jump = 1;
    t1 = ctl->jump_next;
    step = t1 - ctl->tcur;
  }
  const float dtf = (float)step;
  ctl->step = step; ctl->t1 = t1; ctl->on_jump = on_jump; ctl->dtf = dtf;
  for (int i = 1; i < 7; ++i) {
    const float ti = (i == 6) ? f32_prev((float)t1) : (float)((float)ctl->tcur + kDP_C[i] * dtf);
    ctl->t_stage[i] = ti;
    ctl->seg_stage[i] = seg_of(ti, ctl->n_knots);
  }
}

__global__ __launch_bounds__(1024) void cde_err_ratio_kernel(CdeCtl* ctl, const float* __restrict__ ya, const float* __restrict__ yb,
                                                             const float* __restrict__ kbase, CdeCoefs e, float atol, float rtol, int n) {
  __shared__ float red[16];
  if (ctl->done) return;
  const float* y = ctl->yi ? yb : ya;
  const float* y1 = ctl->yi ? ya : yb;
  float ec[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ec[j] = (float)(e.c[j] * (double)ctl->dtf);
  bool ez[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ez[j] = e.c[j] == 0.0;
  const float ratio = block_rms<2>(kbase, nullptr, y, y1, ec, ez, 7, atol, rtol, n, red);
  if (threadIdx.x == 0) {
    ctl->ratio = ratio;
    ctl->accept = ratio <= 1.0f ? 1 : 0;
  }
}

// accepted step: ymid, dense-output polynomial (torchdiffeq _interp_fit), every output time inside (tcur, t1] -> sol,
// FSAL (k0 = k6) unless the step ended on a jump (then f is re-evaluated on the far side by the kernels that follow)
__global__ void cde_step_finish_kernel(const CdeCtl* __restrict__ ctl, const double* __restrict__ t_out, const float* __restrict__ ya,
                                       const float* __restrict__ yb, float* __restrict__ kbase, CdeCoefs mid, float* __restrict__ co,
                                       float* __restrict__ sol, int B, int H, int n_out) {
  if (ctl->done || !ctl->accept) return;
  const int n = B * H;
  const float* y = ctl->yi ? yb : ya;
  const float* y1 = ctl->yi ? ya : yb;
  const float dt = ctl->dtf;
  float mc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) mc[j] = (float)(mid.c[j] * (double)dt);
  const double tprev = ctl->tcur, tcur = ctl->t1;
  const int on_jump = ctl->on_jump;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float acc = 0.f;
    bool first = true;
    for (int j = 0; j < 7; ++j) {
      if (mid.c[j] == 0.0) continue;
      const float term = kbase[(size_t)j * n + i] * mc[j];
      acc = first ? term : acc + term;
      first = false;
    }
    const float a0 = y[i], a1 = y1[i], am = a0 + acc, g0 = kbase[i], g1 = kbase[(size_t)6 * n + i];
    const float c0 = a0;
    const float c1 = dt * g0;
    const float c2 = dt * (g1 - 4.f * g0) - 11.f * a0 - 5.f * a1 + 16.f * am;
    const float c3 = dt * (5.f * g0 - 3.f * g1) + 18.f * a0 + 14.f * a1 - 32.f * am;
    const float c4 = 2.f * dt * (g1 - g0) - 8.f * (a1 + a0) + 16.f * am;
    co[i] = c0; co[n + i] = c1; co[2 * n + i] = c2; co[3 * n + i] = c3; co[4 * n + i] = c4;
    const int b = i / H, hh = i - b * H;
    for (int p = ctl->p_next; p < n_out && t_out[p] <= tcur; ++p) {
      const float xx = (float)((t_out[p] - tprev) / (tcur - tprev));
      float v = c0 + xx * c1, xp = xx;
      xp = xp * xx; v = v + xp * c2;
      xp = xp * xx; v = v + xp * c3;
      xp = xp * xx; v = v + xp * c4;
      sol[((size_t)b * n_out + p) * H + hh] = v;
    }
    if (!on_jump) kbase[i] = g1;
  }
}

__global__ void cde_ctl_update_kernel(CdeCtl* ctl, const double* __restrict__ t_out) {
  if (threadIdx.x || blockIdx.x || ctl->done) return;
  const float ratio = ctl->ratio;
  if (ctl->accept) {
    ++ctl->n_acc;
    ctl->have_interp = 1;
    ctl->tprev = ctl->tcur;
    ctl->tcur = ctl->t1;
    ctl->yi ^= 1;
    while (ctl->p_next < ctl->n_out && t_out[ctl->p_next] <= ctl->tcur) ++ctl->p_next;
    if (ctl->on_jump) {
      if (ctl->jump_next + 1.0 <= (double)(ctl->n_knots - 1)) ctl->jump_next += 1.0;
      const float tn = f32_next((float)ctl->tcur);   // f on the far side of the discontinuity
      ctl->t_stage[0] = tn;
      ctl->seg_stage[0] = seg_of(tn, ctl->n_knots);
      ctl->need_jump_eval = 1;
    }
  }
  // _optimal_step_size
  double factor;
  if (ratio == 0.f) factor = 10.0;
  else {
    const double dfac = ratio < 1.0f ? 1.0 : 0.2;
    factor = fmin(10.0, fmax(0.9 / pow((double)ratio, 0.2), dfac));
  }
  ctl->dt = ctl->step * factor;
  // every output is out: nothing after this step is wanted, not even the re-evaluation behind a jump
  if (ctl->p_next >= ctl->n_out) {
    ctl->need_jump_eval = 0;
    ctl->done = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
void cde_launch_linear(const float* x, int ldx, const float* W, const float* bias, float* out, int B, int K, int N, int act, hipStream_t st) {
  hipLaunchKernelGGL(cde_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, st, CdeWhen{nullptr, 0, 0, 0}, x, W, bias, out, B, K, ldx, N, act);
}

static bool cde_fast_size(int H) { return H == 128 || H == 256 || H == 512 || H == 1024; }

void cde_launch_hidden(const CdeWhen& wh, const float* x, const float* W, const float* bias, float* out, int B, int H, int act, hipStream_t st) {
  if (H % 128 == 0)
    hipLaunchKernelGGL(cde_hidden_kernel, dim3(H / 16), dim3(256), 0, st, wh, x, W, bias, out, B, H, (size_t)H, 1, act, 0,
                       (const float*)nullptr, 0, 0);
  else
    hipLaunchKernelGGL(cde_linear_kernel, dim3((H + 3) / 4), dim3(256), 0, st, wh, x, W, bias, out, B, H, H, H, act);
}

template <int H, bool BF16>
static hipError_t launch_stream_t(const CdeModel& m, const CdeWhen& wh, const float* x, const float* obs, int B, int L, float* out, hipStream_t st) {
  static unsigned long long attr_mask = 0;
  const hipError_t e = once_per_device(attr_mask, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(cde_stream_kernel<H, BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS);
  });
  if (e != hipSuccess) return e;
  const int grid = std::min(m.n_cu > 0 ? m.n_cu : 256, H);
  const void* W = BF16 ? m.w_last16 : (const void*)m.w[m.n_hidden];
  hipLaunchKernelGGL((cde_stream_kernel<H, BF16>), dim3(grid), dim3(256), CS_LDS, st, wh, x, W, m.b[m.n_hidden], obs, out, B, L, m.C);
  return hipSuccess;
}
// fp32 weights (the parity path), or the bf16 copy of the last layer when the plan carries one (reduced-precision mode)
template <int H>
static hipError_t launch_stream(const CdeModel& m, const CdeWhen& wh, const float* x, const float* obs, int B, int L, float* out, hipStream_t st) {
  return m.w_last16 ? launch_stream_t<H, true>(m, wh, x, obs, B, L, out, st) : launch_stream_t<H, false>(m, wh, x, obs, B, L, out, st);
}

int cde_launch_last(const CdeModel& m, const CdeWhen& wh, const float* x, const float* obs, int B, int L, float* out, hipStream_t st) {
  const int H = m.H, C = m.C;
  const float* W = m.w[m.n_hidden];
  const float* bias = m.b[m.n_hidden];
  hipError_t e = hipSuccess;
  if (cde_fast_size(H)) {
    // even pieces: H rows (one per h-group); returns at once on an odd piece
    hipLaunchKernelGGL(cde_hidden_kernel, dim3(H / 16), dim3(256), 0, st, wh, x, W, bias, out, B, H, (size_t)C * H, C, 0, 1, obs, L, C);
    // odd pieces: the weight stream; returns at once on an even piece
    switch (H) {
      case 128: e = launch_stream<128>(m, wh, x, obs, B, L, out, st); break;
      case 256: e = launch_stream<256>(m, wh, x, obs, B, L, out, st); break;
      case 512: e = launch_stream<512>(m, wh, x, obs, B, L, out, st); break;
      default: e = launch_stream<1024>(m, wh, x, obs, B, L, out, st); break;
    }
  } else {
    const size_t lds = ((size_t)CDE_BT * (H + 8) + CDE_BT * 4) * sizeof(float);
    static unsigned long long attr_mask = 0;
    e = once_per_device(attr_mask, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(cde_last_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    });
    if (e == hipSuccess) hipLaunchKernelGGL(cde_last_generic_kernel, dim3(H), dim3(256), lds, st, wh, x, W, bias, obs, out, B, H, C, L);
  }
  return e == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

void cde_launch_combine(const CdeCtl* ctl, int only_on_jump, const float* y0, const float* y1, int y_sel, const float* kbase, const CdeCoefs& cf,
                        int nk, float scale, int scale_sel, float* out, int mirror_y1, int n, hipStream_t st) {
  hipLaunchKernelGGL(cde_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ctl, only_on_jump, y0, y1, y_sel, kbase, cf, nk, scale,
                     scale_sel, out, mirror_y1, n);
}
void cde_launch_emit_copy(const float* src, float* sol, int B, int H, int P, int p, hipStream_t st) {
  hipLaunchKernelGGL(cde_emit_copy_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, src, sol, B, H, P, p);
}
void cde_launch_ctl_init(CdeCtl* ctl, const double* t_out, int n_out, int n_knots, int max_steps, hipStream_t st) {
  hipLaunchKernelGGL(cde_ctl_init_kernel, dim3(1), dim3(1), 0, st, ctl, t_out, n_out, n_knots, max_steps);
}
void cde_launch_init_step(CdeCtl* ctl, int phase, const float* y0, const float* f0, const float* f1, float atol, float rtol, int n, hipStream_t st) {
  hipLaunchKernelGGL(cde_init_step_kernel, dim3(1), dim3(1024), 0, st, ctl, phase, y0, f0, f1, atol, rtol, n);
}
void cde_launch_ctl_begin(CdeCtl* ctl, hipStream_t st) { hipLaunchKernelGGL(cde_ctl_begin_kernel, dim3(1), dim3(1), 0, st, ctl); }
void cde_launch_err_ratio(CdeCtl* ctl, const float* ya, const float* yb, const float* kbase, const CdeCoefs& e, float atol, float rtol, int n, hipStream_t st) {
  hipLaunchKernelGGL(cde_err_ratio_kernel, dim3(1), dim3(1024), 0, st, ctl, ya, yb, kbase, e, atol, rtol, n);
}
void cde_launch_step_finish(const CdeCtl* ctl, const double* t_out, const float* ya, const float* yb, float* kbase, const CdeCoefs& mid,
                            float* interp, float* sol, int B, int H, int n_out, hipStream_t st) {
  hipLaunchKernelGGL(cde_step_finish_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, ctl, t_out, ya, yb, kbase, mid, interp, sol, B, H, n_out);
}
void cde_launch_ctl_update(CdeCtl* ctl, const double* t_out, hipStream_t st) { hipLaunchKernelGGL(cde_ctl_update_kernel, dim3(1), dim3(1), 0, st, ctl, t_out); }
