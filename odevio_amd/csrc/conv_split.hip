// Implicit-GEMM convolution with fp32 operands carried as three bf16 planes ("split" operands), gfx950.
//
// Replaces the same reference blocks as conv_igemm.hip (Conv2d + BatchNorm2d(eval) + LeakyReLU(0.1) of the
// ImageEncoder, src/models/Encoder.py:8-22,116-122) for conv2..conv6 - the 2.3 TFLOP that dominate DeepVIO.forward.
//
// Why: the fp32-input MFMA runs at the vector rate (1/16 of the bf16 MFMA).  An fp32 number is EXACTLY the sum of
// three bf16 numbers, x = h + m + l (8 + 8 + 8 significand bits; each piece is the bf16 rounding of what is left),
// and a bf16 x bf16 product is exact in the MFMA's fp32 accumulator.  So
//     x*w = hh + (hm + mh) + (hl + mm + lh) + [ml + lm + ll]
// where the bracket is below 2^-24 of the product: six bf16 MFMAs give the fp32 product to fp32 accuracy, accumulated
// in fp32 exactly like the fp32 MFMA does, at 6/16 of its cost.  Nothing is stored at reduced precision: the three
// planes together hold every bit of the fp32 activation.
//
// Layout ("P3"): an activation tensor [pixel][C] is stored as [pixel][C/16][3 planes][16 channels] bf16 (96 B per
// 16-channel group, 6 B per element); the producing kernel's epilogue (conv1, this kernel, the split-K combine) splits
// its fp32 result once, so the main loop only moves bytes.  Weights are split on the host at plan creation into
// [Cout][K-tile][3][16] with K-tile = (channel group, tap) in the order the loop walks them.
//
// Tiling: 128 pixels x 128 output channels x 16 input channels per 256-thread workgroup (4 waves, 64x64 each =
// 2x2 MFMA 32x32x16 tiles x 6 plane pairs = 24 MFMAs per K-tile), two workgroups per CU.
//
// Staging is LDS-DMA (global_load_lds_dwordx4): no staging registers and no ds_write pass (ds_write_b128 moves only
// ~79 B/clk/CU and was the busiest LDS client of the register-staged version).  One wave instruction deposits
// 64 lanes x 16 B = 1 KB contiguously, so the LDS image of a tile is [plane][row][32 B]: lane 2r+h of a wave brings
// half h of row r, and the MFMA fragment read (lane = row + 32*half) sweeps the same 1 KB block - contiguous, hence
// conflict-free without padding.  Out-of-image taps read a zero page instead of the activation (the DMA cannot
// select).  Three LDS buffers (24 KB each): the DMA of tile j+2 is issued while tile j is multiplied and tile j+1 is
// still in flight; barriers are raw s_barrier with counted vmcnt (a __syncthreads() would drain the DMAs).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SBM 128
#define SBN 128
#define SBLK 96            // bytes of one (row, K-tile) block in global memory: 3 planes x 16 channels x 2 B
#define SPLANE 4096        // LDS bytes of one plane of one operand tile: 128 rows x 32 B
#define STILE (6 * SPLANE) // one stage: A planes 0..2, then B planes 0..2
#define SSTAGES 3

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int TERMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_split_kernel(ConvSplitArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[SSTAGES * STILE];  // 72 KB
  constexpr int NPL = TERMS == 3 ? 2 : 3;   // planes in use

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: see conv_igemm_kernel
  const int NT = gridDim.y;
  int mt_idx = blockIdx.x, nt_idx = blockIdx.y;
  if (a.xcd_map) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int chunk = gridDim.x >> 3;
    mt_idx = xcd * chunk + slot / NT;
    nt_idx = slot - (slot / NT) * NT;
  }
  if (mt_idx * SBM >= a.M) return;
  const int m0 = mt_idx * SBM;
  const int n0 = nt_idx * SBN;

  // ---- loader role: thread t brings row t>>1 (rows 32w..32w+31 belong to wave w), 16-byte half t&1 of each plane
  const int lrow = tid >> 1;
  const int lhalf = (tid & 1) * 16;
  const int groups = a.Cin >> 4;            // 16-channel groups per pixel
  const int taps = a.KH * a.KW;
  const int nk = taps * groups;             // K-tiles: (channel group, tap), tap minor
  const unsigned char* in_b = reinterpret_cast<const unsigned char*>(a.in);
  const unsigned char* w_b = reinterpret_cast<const unsigned char*>(a.w);
  const unsigned char* zero_b = reinterpret_cast<const unsigned char*>(a.zeros) + lhalf;
  const unsigned char* a_row;
  int a_hi0, a_wi0;
  {
    const int m = m0 + lrow;
    if (m < a.M) {
      const int HoWo = a.Ho * a.Wo;
      const int img = m / HoWo;
      const int rem = m - img * HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      a_hi0 = ho * a.stride - a.pad;
      a_wi0 = wo * a.stride - a.pad;
      a_row = in_b + ((ptrdiff_t)img * a.Hi * a.Wi + (ptrdiff_t)a_hi0 * a.Wi + a_wi0) * groups * SBLK + lhalf;
    } else {
      a_row = in_b;
      a_hi0 = -(1 << 28);
      a_wi0 = -(1 << 28);
    }
  }
  const int nb = n0 + lrow;
  const unsigned char* b_row = (nb < a.Cout) ? w_b + (size_t)nb * nk * SBLK + lhalf : nullptr;

  int kt_begin = 0, kt_end = nk;
  if (a.splitk > 1) {
    kt_begin = blockIdx.z * a.ktiles_per_split;
    kt_end = min(nk, kt_begin + a.ktiles_per_split);
  }

  // K-tile walk (workgroup-uniform): channel group MAJOR, tap MINOR (the taps of one group re-read the same pixels
  // shifted by one, back to back: L1/L2 hits), without divisions.
  int t_kh = 0, t_kw = 0, t_g = 0, t_aoff = 0, t_boff = 0;
  const int px_bytes = groups * SBLK;
  {
    const int g = kt_begin / taps;
    const int tap = kt_begin - g * taps;
    t_g = g;
    t_kh = tap / a.KW;
    t_kw = tap - t_kh * a.KW;
    t_aoff = (t_kh * a.Wi + t_kw) * px_bytes + g * SBLK;
    t_boff = kt_begin * SBLK;
  }
  auto next_tile = [&]() {
    ++t_kw;
    t_aoff += px_bytes;
    t_boff += SBLK;
    if (t_kw == a.KW) {
      t_kw = 0;
      ++t_kh;
      t_aoff += (a.Wi - a.KW) * px_bytes;
      if (t_kh == a.KH) {
        t_kh = 0;
        ++t_g;
        t_aoff = t_g * SBLK;
      }
    }
  };
  // 2 * NPL DMAs per wave and tile; the LDS destination is wave-uniform (+ lane * 16 by the hardware)
  auto issue_tile = [&](int stage) __attribute__((always_inline)) {
    const bool ok = (unsigned)(a_hi0 + t_kh) < (unsigned)a.Hi && (unsigned)(a_wi0 + t_kw) < (unsigned)a.Wi;
    const unsigned char* pa = ok ? a_row + t_aoff : zero_b;
    const unsigned char* pb = b_row ? b_row + t_boff : zero_b;
#ifdef EXP_ZERO_A   // timing experiments only (results are wrong): take the operand from the zero page
    pa = zero_b;
#endif
#ifdef EXP_ZERO_B
    pb = zero_b;
#endif
    unsigned char* dst = lds + stage * STILE + wave * 1024;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      __builtin_amdgcn_global_load_lds((gptr_t)(pa + 32 * p), (lptr_t)(dst + p * SPLANE), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(pb + 32 * p), (lptr_t)(dst + (3 + p) * SPLANE), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fi = lane & 31, fh = lane >> 5;
  const int a_off = (wm * 64 + fi) * 32 + fh * 16;
  const int b_off = 3 * SPLANE + (wn * 64 + fi) * 32 + fh * 16;

  // D rows = output channels (weights are the MFMA's A operand), D columns = pixels: a lane ends up with 4
  // consecutive channels of one pixel per register group (one vector store each in the epilogue).
  // Plane pairs from the smallest to the largest contribution.
  auto multiply = [&](int stage) __attribute__((always_inline)) {
    const unsigned char* Ab = lds + stage * STILE + a_off;
    const unsigned char* Bb = lds + stage * STILE + b_off;
    bf16x8 xf[2][3], wf[2][3];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xf[i][p] = *reinterpret_cast<const bf16x8*>(Ab + p * SPLANE + i * 1024);
        wf[i][p] = *reinterpret_cast<const bf16x8*>(Bb + p * SPLANE + i * 1024);
      }
    }
    constexpr int PW6[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PX6[6] = {0, 1, 2, 0, 1, 0};
    constexpr int PW3[3] = {1, 0, 0};
    constexpr int PX3[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      const int pw = TERMS == 6 ? PW6[t] : PW3[t];
      const int px = TERMS == 6 ? PX6[t] : PX3[t];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][pw], xf[0][px], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][pw], xf[0][px], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][pw], xf[1][px], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][pw], xf[1][px], acc[1][1], 0, 0, 0);
    }
  };

  const int ntile = kt_end - kt_begin;
  // Prologue: tiles 0 and 1 in flight.  Past the end the walk stops and the same tile is fetched again (unused), so
  // every wave always has exactly 2*NPL DMAs per stage outstanding and the counted waits below stay exact.
  issue_tile(0);
  if (ntile > 1) next_tile();
  issue_tile(1);
  if (NPL == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int st_cur = 0, st_nxt = 2;   // stage holding tile j / stage to refill with tile j+2
  for (int j = 0; j < ntile; ++j) {
    if (j + 2 < ntile) next_tile();
    issue_tile(st_nxt);                    // tile j+2 -> the stage tile j-1 was read from (everyone is past that barrier)
    multiply(st_cur);
    // tile j+1 has landed once all but the newest tile's DMAs are done; only then may anyone read it
    if (NPL == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    st_nxt = st_cur;
    st_cur = st_cur == 2 ? 0 : st_cur + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unused look-ahead DMAs must not outlive the workgroup's LDS

  // ---- epilogue.  C/D map of the 32x32 MFMA: column (= pixel) = lane&31, row (= channel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + wm * 64 + mt * 32 + fi;
    if (m >= a.M) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * fh;  // first of 4 consecutive channels; Cout % 16 == 0
        if (n >= a.Cout) continue;
        f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        if (a.splitk > 1) {
          *reinterpret_cast<f32x4*>(a.partial + ((size_t)blockIdx.z * a.M + m) * a.Cout + n) = v;
        } else {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = v[e] * sc[e] + sh[e];
            v[e] = x > 0.f ? x : x * a.slope;
          }
          if (a.out_split) store_split4(reinterpret_cast<unsigned char*>(a.out), (size_t)m, n, a.Cout, v);
          else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.Cout + n) = v;
        }
      }
    }
  }
}

// Deterministic split-K combine: sums the slabs in slab order, then the same epilogue; 4 channels per thread.
__global__ __launch_bounds__(256) void splitk_reduce_split_kernel(ConvSplitArgs a) {
  const size_t total4 = (size_t)a.M * a.Cout / 4;
  const size_t slab = (size_t)a.M * a.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = idx * 4;
    const size_t m = e0 / a.Cout;
    const int n = (int)(e0 - m * a.Cout);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < a.splitk; ++z) v += *reinterpret_cast<const f32x4*>(a.partial + (size_t)z * slab + e0);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e] * sc[e] + sh[e];
      v[e] = x > 0.f ? x : x * a.slope;
    }
    if (a.out_split) store_split4(reinterpret_cast<unsigned char*>(a.out), m, n, a.Cout, v);
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + e0) = v;
  }
}

void launch_conv_split(const ConvSplitArgs& a_in, hipStream_t st) {
  ConvSplitArgs a = a_in;
  const int mt = (a.M + SBM - 1) / SBM;
  a.xcd_map = mt >= 16;
  dim3 grid(a.xcd_map ? (mt + 7) / 8 * 8 : mt, (a.Cout + SBN - 1) / SBN, a.splitk > 1 ? a.splitk : 1);
  if (a.terms == 3) hipLaunchKernelGGL(conv_split_kernel<3>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv_split_kernel<6>, grid, dim3(256), 0, st, a);
  if (a.splitk > 1) {
    const size_t total4 = (size_t)a.M * a.Cout / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_split_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
}

// fp32 [pixel][C] <-> P3, 4 channels per thread (API boundary of odevio_conv_block_fwd and tests; the encoder itself
// never converts: every producer writes P3 directly).
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, size_t pixels, int C) {
  const size_t total4 = pixels * C / 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = idx * 4;
    const size_t m = e0 / C;
    const int n = (int)(e0 - m * C);
    store_split4(out, m, n, C, *reinterpret_cast<const f32x4*>(in + e0));
  }
}

__global__ __launch_bounds__(256) void split_unpack_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, size_t pixels, int C) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const size_t total4 = pixels * C / 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = idx * 4;
    const size_t m = e0 / C;
    const int n = (int)(e0 - m * C);
    const unsigned char* p = in + (m * (C >> 4) + (n >> 4)) * SBLK + (n & 15) * 2;
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(p), mid = *reinterpret_cast<const bf16x4*>(p + 32), l = *reinterpret_cast<const bf16x4*>(p + 64);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = ((float)l[e] + (float)mid[e]) + (float)h[e];
    *reinterpret_cast<f32x4*>(out + e0) = v;
  }
}

void launch_split_pack(const float* in, void* out, size_t pixels, int C, hipStream_t st) {
  int blocks = (int)std::min<size_t>((pixels * C / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(split_pack_kernel, dim3(blocks), dim3(256), 0, st, in, reinterpret_cast<unsigned char*>(out), pixels, C);
}
void launch_split_unpack(const void* in, float* out, size_t pixels, int C, hipStream_t st) {
  int blocks = (int)std::min<size_t>((pixels * C / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL(split_unpack_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(in), out, pixels, C);
}
