// Persistent ODE-RNN integrator for gfx950: the whole `for i in range(seq_len)` loop of the
// reference's PoseODERNN.forward (src/models/PoseODERNN.py:97-123) - per interval an ODE solve of
// every RNN layer's hidden state (torchode in the reference, :70-75) followed by one nn.RNN/nn.GRU
// step (:114) - in ONE kernel launch.
//
// Why persistent: one RK4 step of the [R = L*B, 768] state is 16 dependent skinny GEMMs; at R = 32
// the arithmetic is microseconds of work and a launch boundary costs ~1.5 us each, so the loop is
// latency-bound by construction.
//
// Decomposition (MI355X-first):
//  * rows (batch elements x RNN layers) are independent in the ODE solve, so they are dealt to 8 ROW
//    GROUPS; a group is 32 workgroups = one XCD under the observed round-robin dispatch
//    (blockIdx & 7; speed only - correctness never depends on placement, see hand-off below);
//  * inside a group the MLP is COLUMN-sharded: member c owns N/32 output columns of every layer
//    and keeps its slice of the weights resident in LDS for the whole launch (a slice that does
//    not fit streams from L2), so the 5.25 MB of ODEFunc parameters are read from HBM once per
//    launch instead of once per stage;
//  * between layers the members all-gather the [rows, N] activations through global memory with
//    8-byte {tag, value} granules: written with relaxed agent-scope atomic stores (write-through,
//    sc1) and polled with relaxed agent-scope atomic loads - the data is its own flag, so there is
//    no fence, no separate flag and no dependence on XCD placement.  Two buffers alternate by
//    exchange parity: a member can only write exchange e+2 after it has read all of e+1, which
//    needs every member to have finished reading e, so a buffer is never overwritten while in use.
//    Every poll is bounded (2 s wall clock) and gives up with ODEVIO_ERR_TIMEOUT in the status word.
//    Groups that PROVE at run time (HW_REG_XCC_ID census through the safe protocol) that all 32
//    members share one XCD switch to plain stores that stay in that XCD's L2.
//  * per-row solver state (t, dt, accept, ...) is computed redundantly by every member from the
//    same gathered numbers in the same order, so all members take identical control flow.
//
// Thread map: 1024 threads = 16 waves = 4 waves per SIMD.  With one wave per SIMD a layer product cost
// ~2300 cycles for ~60 instructions of arithmetic (every dependent instruction and LDS read paid its
// full latency); four co-resident waves hide it.
//  * layer product:  wave w = column slot (local columns w and w+16), lane = K-slice: lane l multiplies
//    inputs 256j+4l .. 256j+4l+3 of every 256-wide chunk j, so a wave's 16-byte loads are one contiguous
//    1 KB run for weights and activations alike; a 64-lane transposing reduction leaves row r's total on
//    lane r, which drops it into LDS;
//  * everything per element (bias, activation, publish, Runge-Kutta arithmetic, controller) runs on the
//    OWNER thread of that element only: element (row r, local column cl) lives on thread cl*RT + r
//    (at most 256 threads = waves 0..3), so the other twelve waves skip it instead of repeating it.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "integrator.h"

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT
#define ST_TIMEOUT (-6)
#define ST_MAX_STEPS (-7)
#define SPIN_TIMEOUT_TICKS 200000000ull  // 2 s of the 100 MHz s_memrealtime clock
#define NT INTEG_THREADS

#ifdef ODEVIO_STAMPS
#define STAMP_NOW() __builtin_amdgcn_s_memtime()
#define STAMP_ADD(acc, t0) (acc) += __builtin_amdgcn_s_memtime() - (t0)
#else
#define STAMP_NOW() 0ull
#define STAMP_ADD(acc, t0) (void)(t0)
#endif

struct Ctx {
  unsigned long long t_gather, t_layer, t_rnn, n_gather, t_bar, t_epi;
  int tid, lane, wave, cu;
  unsigned epoch;
  bool local;   // all members of this group share one XCD (verified, not assumed)
  u64* xb;      // granule buffers: parity 0 at xb, parity 1 at xb + xstride.  Keep ONE pointer: LLVM turns a
  int xstride;  // select between two adjacent pointer fields into a runtime-indexed load, which pushes the whole
                // struct to scratch (and every poll becomes a flat_load).
  int* status;
  bool failed;
};

__device__ __forceinline__ u64* buf_of(const Ctx& c, unsigned epoch) { return c.xb + ((epoch & 1u) ? c.xstride : 0); }

// One granule = one naturally aligned 8-byte {tag, value} store: the data is its own flag.
//  * safe form (any placement): relaxed AGENT-scope store = write-through `sc1`, polled with `sc1` loads;
//  * local form (only after the group has PROVED at run time that all 32 members sit on one XCD):
//    a plain store that stays in that XCD's L2, where the members' L1-bypassing polls read it.
__device__ __forceinline__ void put(u64* p, float v, unsigned tag, bool local) {
  const u64 g = ((u64)tag << 32) | (u64)__float_as_uint(v);
  if (local) __hip_atomic_store(p, g, RLX, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, g, RLX, AGENT);
}

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}

// Collect the n = rows*width granules of exchange `tag` into LDS: value (row, col) goes to dst[row*ld + col];
// columns width..ld-1 of every row are zero-filled (K is padded to the 256-wide chunks of the layer product).
// PRECONDITION: a workgroup barrier has passed since the last read of dst (every caller has one right after the
// layer product that consumed it), so waves may start polling while the owner threads of THIS member are still
// publishing - their epilogue hides under the wait for the other members.  Ends with a barrier; uniform outcome.
template <int MAXG>
__device__ __forceinline__ void gather(Ctx& c, const u64* buf, unsigned tag, int rows, int width, int ld, float* dst) {
  const unsigned long long st0 = STAMP_NOW();
  bool fail = false;
  const int n = rows * width;
  if (ld > width) {
    const int pad = ld - width;
    for (int i = c.tid; i < rows * pad; i += NT) {
      const int r = i / pad;
      dst[r * ld + width + (i - r * pad)] = 0.f;
    }
  }
  if (!c.failed) {
    u64 g[MAXG];
    unsigned pend = 0;
#pragma unroll
    for (int j = 0; j < MAXG; ++j)
      if (c.tid + NT * j < n) pend |= 1u << j;
    unsigned spins = 0;
    u64 t_start = 0;
    while (pend) {
#pragma unroll
      for (int j = 0; j < MAXG; ++j)
        if ((pend >> j) & 1u) g[j] = __hip_atomic_load(buf + c.tid + NT * j, RLX, AGENT);
#pragma unroll
      for (int j = 0; j < MAXG; ++j)
        if (((pend >> j) & 1u) && (unsigned)(g[j] >> 32) == tag) {
          const int idx = c.tid + NT * j;
          int o = idx;
          if (ld != width) {
            const int r = idx / width;
            o = r * ld + (idx - r * width);
          }
          dst[o] = __uint_as_float((unsigned)g[j]);
          pend &= ~(1u << j);
        }
      if (pend) {
        if ((++spins & 127u) == 0) {
          const u64 now = __builtin_amdgcn_s_memrealtime();
          if (t_start == 0) t_start = now;
          if (now - t_start > SPIN_TIMEOUT_TICKS || __hip_atomic_load(c.status, RLX, AGENT) != 0) {
            atomicCAS(c.status, 0, ST_TIMEOUT);
            fail = true;
            break;
          }
        }
        __builtin_amdgcn_s_sleep(1);   // polling without the sleep measured 0.5 % faster only; keep the fabric quiet
      }
    }
  }
  if (__syncthreads_or(fail ? 1 : 0)) c.failed = true;
  STAMP_ADD(c.t_gather, st0);
  c.n_gather += 1;
}

// ---- transposing reduction over the 64 lanes of a wave ----------------------------------------------
// Every lane enters with RT partial sums (one per row) and leaves with the TOTAL of row (lane mod RT): at each of
// the first log2(RT) levels a lane keeps the half of its rows selected by one bit of its lane id and adds the
// partner's partials for those rows, so no lane ever holds (or selects from) all rows' totals.  lane^1 / lane^2
// partners are DPP quad_perms, lane^4 a ds_swizzle (crossbar only); the remaining lanes that hold the same row are
// folded with row_ror (rotate by 8 then 4 visits lanes i, i+4, i+8, i+12 of a 16-lane row), a swizzle (lane^16)
// and one ds_bpermute (lane^32).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int RT>
__device__ __forceinline__ float reduce_rows64(const float (&v)[RT], int lane) {
  static_assert(RT == 2 || RT == 4 || RT == 8, "rows per group");
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  float a[RT / 2];
#pragma unroll
  for (int i = 0; i < RT / 2; ++i) {
    const float keep = b0 ? v[2 * i + 1] : v[2 * i];
    const float send = b0 ? v[2 * i] : v[2 * i + 1];
    a[i] = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]: lane ^ 1
  }
  float d;
  if (RT == 2) {
    d = a[0];
    d += dpp_mov<0x4E>(d);   // lane ^ 2
    d += dpp_mov<0x128>(d);  // row_ror:8
    d += dpp_mov<0x124>(d);  // row_ror:4
  } else {
    float c[RT / 4 > 0 ? RT / 4 : 1];
#pragma unroll
    for (int i = 0; i < RT / 4; ++i) {
      const float keep = b1 ? a[2 * i + 1] : a[2 * i];
      const float send = b1 ? a[2 * i] : a[2 * i + 1];
      c[i] = keep + dpp_mov<0x4E>(send);  // quad_perm [2,3,0,1]: lane ^ 2
    }
    if (RT == 8) {
      const float keep = b2 ? c[RT / 4 - 1] : c[0];
      const float send = b2 ? c[0] : c[RT / 4 - 1];
      d = keep + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(send), 0x101F));  // lane ^ 4
      d += dpp_mov<0x128>(d);  // row_ror:8
    } else {
      d = c[0];
      d += dpp_mov<0x128>(d);  // row_ror:8
      d += dpp_mov<0x124>(d);  // row_ror:4
    }
  }
  d += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(d), 0x401F));  // lane ^ 16 (and 0x1f, xor 0x10)
  d += __shfl_xor(d, 32, 64);                                                    // lane ^ 32
  return d;
}

// One wave's share of a layer product: acc[c][r] += sum over this lane's K-slice of W[col_c][k] * x[r][k] for
// nseg 256-wide chunks starting at weight chunk jbase; row r's inputs start at xs + roff[r].
// Partial sums are (even k, odd k) pairs so that each multiply-add is one v_pk_fma_f32 on register pairs that the
// 16-byte loads deliver adjacent.
template <int RT, bool TWO>
__device__ __forceinline__ void layer_seg(const float* __restrict__ wbase, int NC, int jbase, int nseg, const float* xs,
                                          const int (&roff)[RT], int c0, int c1, int lane, f32x2 (&acc)[2][RT]) {
  const float* w0p = wbase + (((size_t)jbase * NC + c0) * 64 + lane) * 4;
  const float* w1p = wbase + (((size_t)jbase * NC + c1) * 64 + lane) * 4;
  const size_t wstep = (size_t)NC * 256;
  xs += 4 * lane;
#pragma unroll 2
  for (int j = 0; j < nseg; ++j) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w0p + j * wstep);
    f32x4 w1 = w0;
    if (TWO) w1 = *reinterpret_cast<const f32x4*>(w1p + j * wstep);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(xs + j * 256 + roff[r]);
      acc[0][r] = __builtin_elementwise_fma(w0.lo, x.lo, acc[0][r]);
      acc[0][r] = __builtin_elementwise_fma(w0.hi, x.hi, acc[0][r]);
      if (TWO) {
        acc[1][r] = __builtin_elementwise_fma(w1.lo, x.lo, acc[1][r]);
        acc[1][r] = __builtin_elementwise_fma(w1.hi, x.hi, acc[1][r]);
      }
    }
  }
}

// Layer product for local columns col0 = pass*32 + wave and col0 + 16 (where < NC): row r reads its first K1p
// inputs at xa + offa[r] and the remaining K2p at xb + offb[r] (both padded to multiples of 256).  The totals
// of rows 0..RT-1 land on lanes 0..RT-1, which store them to out[col * RT + row].
template <int RT>
__device__ __forceinline__ void layer(const float* __restrict__ wbase, int NC, int K1p, const float* xa,
                                      const int (&offa)[RT], int K2p, const float* xb, const int (&offb)[RT],
                                      int col0, int lane, float* out) {
  if (col0 >= NC) return;  // wave-uniform: this wave owns no column of this layer / pass
  f32x2 acc[2][RT];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[c][r] = (f32x2){0.f, 0.f};
  const int c0 = col0;
  const bool two = col0 + 16 < NC;  // wave-uniform
  const int c1 = two ? col0 + 16 : c0;
  const int nj1 = K1p >> 8, nj2 = K2p >> 8;
  float s0[RT], s1[RT];
  if (two) {
    layer_seg<RT, true>(wbase, NC, 0, nj1, xa, offa, c0, c1, lane, acc);
    if (nj2) layer_seg<RT, true>(wbase, NC, nj1, nj2, xb, offb, c0, c1, lane, acc);
  } else {
    layer_seg<RT, false>(wbase, NC, 0, nj1, xa, offa, c0, c1, lane, acc);
    if (nj2) layer_seg<RT, false>(wbase, NC, nj1, nj2, xb, offb, c0, c1, lane, acc);
  }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    s0[r] = acc[0][r].x + acc[0][r].y;
    s1[r] = acc[1][r].x + acc[1][r].y;
  }
  const float t0 = reduce_rows64<RT>(s0, lane);
  if (lane < RT) out[c0 * RT + lane] = t0;
  if (two) {
    const float t1 = reduce_rows64<RT>(s1, lane);
    if (lane < RT) out[c1 * RT + lane] = t1;
  }
}

// The same product for ONE layer whose slice did not fit in LDS, with this wave's weights held in registers for the whole launch
// (a 16-column x 512 slice is exactly 2 float4 per lane in the column-major map: wave = column, lane = K slice): the layer then
// reads nothing but its gathered input.  Same arithmetic and summation order as layer<RT>.
template <int RT>
__device__ __forceinline__ void layer_reg(const f32x4 (&wreg)[2], int NC, const float* xs, const int (&roff)[RT], int col0, int lane,
                                          float* out) {
  if (col0 >= NC) return;
  f32x2 acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = (f32x2){0.f, 0.f};
  xs += 4 * lane;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(xs + j * 256 + roff[r]);
      acc[r] = __builtin_elementwise_fma(wreg[j].lo, x.lo, acc[r]);
      acc[r] = __builtin_elementwise_fma(wreg[j].hi, x.hi, acc[r]);
    }
  float s0[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) s0[r] = acc[r].x + acc[r].y;
  const float t0 = reduce_rows64<RT>(s0, lane);
  if (lane < RT) out[col0 * RT + lane] = t0;
}

// tanh on the owner threads sits on the critical path of every exchange (one or two waves run it while the
// rest of the workgroup waits), and ocml's tanhf is ~150 dependent instructions.  This form is ~20:
// |x| < 0.25: odd Taylor polynomial to x^11 (truncation < 2e-9 relative); otherwise 1 - 2/(e^{2|x|} + 1) with the
// exponent's rounding error carried in a compensation term.  Absolute error < 2e-7 everywhere (fp32 eps 1.2e-7),
// far inside the 1e-4 parity bar; tests compare against torch's tanh.
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  float p = fmaf(x2, -0.00886323552990220f, 0.0218694885361552f);  // -1382/155925, 62/2835
  p = fmaf(x2, p, -0.0539682539682540f);                            // -17/315
  p = fmaf(x2, p, 0.133333333333333f);                              // 2/15
  p = fmaf(x2, p, -0.333333333333333f);                             // -1/3
  const float small = fmaf(x * x2, p, x);
  // e^{2|x|} = 2^(a + b): a = fl(2|x| * log2e), b = the rounding error of that product plus the low part of log2e
  const float a = ax * 2.885390043f;                                 // 2*log2(e) (hi)
  const float b = fmaf(ax, 2.885390043f, -a) + ax * 3.851925e-8f;    // 2*log2(e) (lo) = 2*1.9259629e-8
  const float e = __builtin_amdgcn_exp2f(a) * fmaf(b, 0.693147181f, 1.0f);
  const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float r = ax < 0.25f ? small : copysignf(big, x);
  return ax > 20.0f ? copysignf(1.0f, x) : r;
}

__device__ __forceinline__ float hidden_act(float v, int act) {
  switch (act) {
    case 0: return fast_tanh(v);
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.01f * v;
    default: return v > 20.f ? v : log1pf(expf(v));  // nn.Softplus(beta=1, threshold=20)
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 0.5f * fast_tanh(0.5f * v) + 0.5f; }

__device__ __forceinline__ int pad256(int k) { return (k + 255) & ~255; }

template <int RT>
__global__ __launch_bounds__(INTEG_THREADS) void integrator_kernel(const IntegArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MAXG = (RT * INTEG_KMAX + NT - 1) / NT;
  Ctx c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave = c.tid >> 6;
  const int g = blockIdx.x & (INTEG_GROUPS - 1);
  c.cu = blockIdx.x >> 3;
  if (g >= a.G) return;
  c.epoch = 0;
  c.status = a.status;
  c.failed = false;
  c.local = false;
  c.xb = a.xbuf + (size_t)(2 * g) * a.xstride;
  c.xstride = a.xstride;
  c.t_gather = c.t_layer = c.t_rnn = c.n_gather = c.t_bar = c.t_epi = 0;
  const unsigned long long t_begin = STAMP_NOW();
  const int tid = c.tid, lane = c.lane, wave = c.wave, cu = c.cu;

  float* xin = smem + a.lds_xin;    // [RT][KMAXp]   gathered layer input
  float* hst = smem + a.lds_hst;    // [RT][Fp]      gathered evolved state (RNN phase)
  float* lay = smem + a.lds_misc;   // [128][RT]     layer-product totals, out[col*RT + row]
  float* nrm = lay + 128 * RT;      // [RT][32]      per-member error-norm partials
  float* mv = nrm + RT * 32;        // [RT][32]      new hidden state on its way to the owner threads
  float* qb = mv + RT * 32;         // [256]         per-element error quotients
  float* bia = qb + 256;            // [INTEG_MAX_LIN][32] this member's ODEFunc biases
  float* wl = smem + a.lds_w;

  const bool seq_mode = (a.mode == MODE_ODE_RNN || a.mode == MODE_RNN_ONLY);
  const int F = a.F, Fp = pad256(F);
  const int Fio = a.Fio;            // columns Fio .. F-1 are padding: zero weights in and out, their state stays exactly 0
  const int NCF = F / INTEG_MEMBERS;
  const int R = a.rows_per_group;
  const int BPG = a.BPG;

  // ---- owner role: thread t < NCF*RT owns state element (row = t % RT, local column = t / RT)
  const int orow = tid % RT, ocl = tid / RT;
  const bool owner = ocl < NCF && orow < R;
  const int ocg = cu * NCF + (ocl < NCF ? ocl : 0);  // global column of the owned element
  int row_l = 0, row_b = 0, grow = 0;
  bool row_valid = false;
  if (orow < R) {
    if (seq_mode) {
      row_l = orow / BPG;
      row_b = a.b_begin + g * BPG + (orow - row_l * BPG);
      row_valid = row_b < a.b_end;
      grow = row_l * a.B + row_b;
    } else {
      grow = a.b_begin + g * BPG + orow;
      row_valid = grow < a.b_end;
    }
  }
  float y = 0.f;
  if (owner && row_valid && ocg < Fio) {
    if (seq_mode) y = a.hc ? a.hc[(size_t)grow * Fio + ocg] : 0.f;
    else y = a.y0[(size_t)grow * Fio + ocg];
  }

  // ---- placement census: the members tell each other their XCD through the SAFE protocol; only if all 32
  //      agree does the group switch to the L2-local hand-off (a pure speed choice made on observed facts)
  if (a.allow_local) {
    ++c.epoch;
    const unsigned mine = xcc_id();
    if (tid == 0) put(buf_of(c, c.epoch) + cu, __uint_as_float(mine + 1u), c.epoch, false);
    gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, 1, INTEG_MEMBERS, INTEG_MEMBERS, nrm);
    bool same = true;
    for (int m = 0; m < INTEG_MEMBERS; ++m) same = same && (__float_as_uint(nrm[m]) == mine + 1u);
    c.local = same && !c.failed;
    __syncthreads();
  }
  if (a.dbg && cu == 0 && tid == 0) reinterpret_cast<unsigned char*>(a.dbg + 5)[g] = c.local ? 1 : 0;  // byte per group

  for (int i = tid; i < a.nlin * 32; i += NT) {
    const int l = i >> 5, cl = i & 31;
    const int NC = a.dims[l + 1] / INTEG_MEMBERS;
    bia[i] = cl < NC ? a.b[l][cu * NC + cl] : 0.f;
  }
  // ---- resident weight slices -> LDS (read from HBM once per launch)
  for (int l = 0; l < a.nlin; ++l) {
    if (a.w_lds_off[l] < 0) continue;
    const int n = (a.dims[l + 1] / INTEG_MEMBERS) * pad256(a.dims[l]);
    const float* src = a.w[l] + (size_t)cu * n;
    float* dstw = wl + a.w_lds_off[l];
    for (int i = tid * 4; i < n; i += NT * 4)
      *reinterpret_cast<f32x4*>(dstw + i) = *reinterpret_cast<const f32x4*>(src + i);
  }
  __syncthreads();

  // one slice that did not fit in LDS (the greedy carve leaves out the smallest) lives in registers if it has the shape the
  // column-major map gives one wave in two loads: <= 16 local columns x 512 inputs
  int lreg = -1;
  f32x4 wreg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int l = 0; l < a.nlin; ++l)
    if (lreg < 0 && a.w_lds_off[l] < 0 && a.dims[l + 1] / INTEG_MEMBERS <= 16 && pad256(a.dims[l]) == 512) lreg = l;
  if (lreg >= 0) {
    const int NC = a.dims[lreg + 1] / INTEG_MEMBERS;
    if (wave < NC) {
      const float* src = a.w[lreg] + (size_t)cu * NC * 512 + ((size_t)wave * 64 + lane) * 4;
      wreg[0] = *reinterpret_cast<const f32x4*>(src);
      wreg[1] = *reinterpret_cast<const f32x4*>(src + (size_t)NC * 256);
    }
  }

  int off_ode[RT];

  // vector field: the owner threads hand in their stage value sv and get k = f(sv) back
  auto feval = [&](float sv) __attribute__((always_inline)) -> float {
    float ko = 0.f;
    ++c.epoch;
    if (owner) put(buf_of(c, c.epoch) + orow * F + ocg, sv, c.epoch, c.local);
    for (int l = 0; l < a.nlin; ++l) {
      const int K = a.dims[l], N = a.dims[l + 1];
      const int Kp = pad256(K);
      const int NC = N / INTEG_MEMBERS;
      const bool more = l + 1 < a.nlin;
      gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, R, K, Kp, xin);
      const unsigned long long sl0 = STAMP_NOW();
#pragma unroll
      for (int r = 0; r < RT; ++r) off_ode[r] = (r < R ? r : R - 1) * Kp;
      if (a.w_lds_off[l] >= 0)
        layer<RT>(wl + a.w_lds_off[l], NC, Kp, xin, off_ode, 0, xin, off_ode, wave, lane, lay);
      else if (l == lreg)
        layer_reg<RT>(wreg, NC, xin, off_ode, wave, lane, lay);
      else
        layer<RT>(a.w[l] + (size_t)cu * NC * Kp, NC, Kp, xin, off_ode, 0, xin, off_ode, wave, lane, lay);
      STAMP_ADD(c.t_layer, sl0);
      const unsigned long long sb0 = STAMP_NOW();
      __syncthreads();
      STAMP_ADD(c.t_bar, sb0);
      const unsigned long long se0 = STAMP_NOW();
      // owners of this layer's outputs: thread t < NC*RT -> (row t % RT, column t / RT)
      if (ocl < NC && orow < R) {
        const float v = lay[tid] + bia[l * 32 + ocl];
        if (more) put(buf_of(c, c.epoch + 1) + orow * N + cu * NC + ocl, hidden_act(v, a.act), c.epoch + 1, c.local);
        else ko = fast_tanh(v);
      }
      STAMP_ADD(c.t_epi, se0);
      if (more) ++c.epoch;
    }
    return ko;
  };

  if (a.mode == MODE_FEVAL) {
    const float kk = feval(y);
    if (owner && row_valid && ocg < Fio) a.y_out[(size_t)grow * Fio + ocg] = kk;
    return;
  }

  unsigned long long t_pro = 0, t_feval = 0, t_norm = 0, t_rnnphase = 0;   // (diagnostic build only)
  STAMP_ADD(t_pro, t_begin);
  const int S = a.tab.stages;
  const float inv_order = -1.f / (float)a.tab.order;
  const int n_int = seq_mode ? a.P : 1;
  const bool fixed = (a.tab.has_err == 0 && a.nsub > 0);
  int n_steps = 0, n_acc = 0;

  for (int it = 0; it < n_int && !c.failed; ++it) {
    // ======================= ODE phase =======================
    if (a.mode != MODE_RNN_ONLY) {
      float t = 0.f, t1 = 0.f;
      if (row_valid) {
        if (seq_mode) {
          const float* tr = a.ts + (size_t)row_b * (a.P + 1);
          const float base = a.ts_relative ? tr[0] : 0.f;
          t = tr[it] - base;
          t1 = tr[it + 1] - base;
        } else {
          t = a.t0[grow];
          t1 = a.t1[grow];
        }
      }
      float dt, dtn = a.dt0;
      bool last = false, running;
      int sub_left = a.nsub;
      if (fixed) {
        dt = (t1 - t) / (float)a.nsub;
        running = row_valid;
      } else {
        const float span = t1 - t;
        last = dtn >= span;
        dt = last ? span : dtn;
        running = row_valid && (t < t1);
      }
      float k[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) k[j] = 0.f;
      bool have_k1 = false;
      int guard = 0;
      int acc_it = 0;   // accepted steps of this row in this interval (logged for the backward)
      const bool logger = a.dtlog && seq_mode && row_valid && owner && ocl == 0 && cu == 0;   // one thread per row
      while (__syncthreads_or((running && owner) ? 1 : 0)) {
        if (c.failed) break;
        if (++guard > a.max_steps) {
          if (tid == 0) atomicCAS(c.status, 0, ST_MAX_STEPS);
          break;
        }
        float sv = y;
        for (int s = 0; s < S; ++s) {
          if (s == 0 && have_k1) continue;
          if (s > 0) {
            float a0 = 0.f;
            bool first = true;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
              if (j < s) {
                const float co = a.tab.a[s][j];
                if (co != 0.f) {
                  a0 = first ? k[j] * co : a0 + k[j] * co;  // same association as the oracle
                  first = false;
                }
              }
            }
            sv = y + dt * a0;
          }
          const unsigned long long sf0 = STAMP_NOW();
          const float ko = feval(sv);
          STAMP_ADD(t_feval, sf0);
#pragma unroll
          for (int j = 0; j < 7; ++j)
            if (j == s) k[j] = ko;
        }
        // y1 = y + dt * sum b_j k_j   (FSAL: b_last = 0 and the sum equals the last stage's argument)
        float s0 = 0.f, e0 = 0.f;
        {
          bool fb = true, fe = true;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            if (j < S) {
              const float bj = a.tab.b[j];
              if (bj != 0.f) {
                s0 = fb ? k[j] * bj : s0 + k[j] * bj;
                fb = false;
              }
              const float ej = a.tab.e[j];
              if (a.tab.has_err && ej != 0.f) {
                e0 = fe ? k[j] * ej : e0 + k[j] * ej;
                fe = false;
              }
            }
          }
        }
        const float y1 = y + dt * s0;
        const float er = dt * e0;
        bool accept = true;
        const unsigned long long sn0 = STAMP_NOW();
        if (a.tab.has_err) {
          // per-row RMS of err / (atol + rtol*max(|y0|,|y1|)) over all F columns (torchode rms_norm)
          __syncthreads();  // qb free
          if (tid < 256) {
            float q = 0.f;
            if (owner) {
              const float bound = a.atol + a.rtol * fmaxf(fabsf(y), fabsf(y1));
              const float z = er / bound;
              q = z * z;
            }
            qb[tid] = q;
          }
          __syncthreads();
          ++c.epoch;
          if (tid < R) {
            float s = 0.f;
            for (int cl = 0; cl < NCF; ++cl) s += qb[cl * RT + tid];
            put(buf_of(c, c.epoch) + tid * INTEG_MEMBERS + cu, s, c.epoch, c.local);
          }
          gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, R, INTEG_MEMBERS, INTEG_MEMBERS, nrm);
          float tot = 0.f;
          const int rr = orow < R ? orow : 0;
          for (int m = 0; m < INTEG_MEMBERS; ++m) tot += nrm[rr * INTEG_MEMBERS + m];
          const float ratio = sqrtf(tot / (float)Fio);   // (padded columns contribute exact zeros to the sum)
          accept = ratio < 1.0f;
          float factor = 0.9f * powf(ratio, inv_order);
          factor = fminf(fmaxf(factor, 0.2f), 10.0f);
          dtn = dt * factor;
        } else {
          dtn = dt;
        }
        STAMP_ADD(t_norm, sn0);
        const bool upd = accept && running;
        if (running) ++n_steps;
        if (upd) {
          ++n_acc;
          // the log of the backward: this step's size (one thread per row) and the state it starts from (every owner its element).
          // A log too short for the interval is not an error of the forward: the count keeps running and the host sees it.
          if (acc_it < a.dtlog_cap) {
            if (logger) a.dtlog[((size_t)grow * a.P + it) * a.dtlog_cap + acc_it] = dt;
            if (a.ylog && seq_mode && row_valid && owner && ocg < Fio)
              a.ylog[(((size_t)grow * a.P + it) * a.dtlog_cap + acc_it) * Fio + ocg] = y;
          }
          ++acc_it;
          y = y1;
          if (a.tab.fsal) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
              if (j == S - 1) k[0] = k[j];
          }
        }
        if (fixed) {
          if (--sub_left <= 0) running = false;
        } else {
          if (upd) t = last ? t1 : t + dt;
          running = row_valid && (t < t1);
          const float span = t1 - t;
          last = dtn >= span;
          dt = last ? span : dtn;
        }
        have_k1 = a.tab.fsal != 0;
      }
      if (logger) a.dtcnt[(size_t)grow * a.P + it] = acc_it;   // may exceed dtlog_cap: then the log is incomplete and the host asks again
      if (a.ylog && seq_mode && row_valid && owner && ocg < Fio) a.yend[((size_t)grow * a.P + it) * Fio + ocg] = y;   // the evolved state the RNN sees
      if (!seq_mode) break;
    }
    if (!seq_mode || c.failed) break;

    // ======================= RNN phase =======================
    // 1. all-gather the evolved states h~ [R][F] -> hst
    const unsigned long long sp0 = STAMP_NOW();
    ++c.epoch;
    if (owner) put(buf_of(c, c.epoch) + orow * F + ocg, y, c.epoch, c.local);
    gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, R, F, Fp, hst);
    const int NCV = a.rnn_vcols * NCF;
    for (int l = 0; l < a.L; ++l) {
      const bool more = l + 1 < a.L;
      const float* wsl = a.rw[l] + (size_t)cu * NCV * 2 * Fp;
      const float* rb = a.rb[l];
      // ---- inputs of this layer: the fused features (l = 0) or the gathered h' of the layer below
      if (l == 0) {
        __syncthreads();
        for (int i = tid; i < BPG * Fp; i += NT) {
          const int bi = i / Fp, col = i - bi * Fp;
          const int b = a.b_begin + g * BPG + bi;
          xin[i] = (b < a.b_end && col < Fio) ? a.fused[((size_t)b * a.P + it) * Fio + col] : 0.f;
        }
        __syncthreads();
      } else {
        gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, BPG, F, Fp, xin);
      }
      // ---- pre-activations for every virtual column of this member
      const unsigned long long sr0 = STAMP_NOW();
      int offa[RT], offb[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int bi = r < BPG ? r : BPG - 1;
        offa[r] = bi * Fp;
        offb[r] = (l * BPG + bi) * Fp;
      }
      for (int pass = 0; pass * 32 < NCV; ++pass)
        layer<RT>(wsl, NCV, Fp, xin, offa, Fp, hst, offb, pass * 32 + wave, lane, lay);
      STAMP_ADD(c.t_rnn, sr0);
      __syncthreads();
      // ---- gates / tanh on the owner threads (row index = sequence within the group), new hidden state
      if (ocl < NCF && orow < BPG) {
        const int b = a.b_begin + g * BPG + orow;
        const int ug = cu * NCF + ocl;
        float hv;
        if (a.rnn_type == 0) {
          hv = fast_tanh(lay[ocl * RT + orow] + rb[ug]);
        } else {
          const float rg = sigmoidf_(lay[ocl * RT + orow] + rb[ug]);
          const float zg = sigmoidf_(lay[(NCF + ocl) * RT + orow] + rb[F + ug]);
          const float ng = fast_tanh(lay[(2 * NCF + ocl) * RT + orow] + rb[2 * F + ug] +
                                 rg * (lay[(3 * NCF + ocl) * RT + orow] + rb[3 * F + ug]));
          const float hp = hst[(l * BPG + orow) * Fp + ug];
          hv = (1.f - zg) * ng + zg * hp;
        }
        mv[(l * BPG + orow) * 32 + ocl] = hv;
        if (!more && b < a.b_end && ug < Fio) a.out_seq[((size_t)b * a.P + it) * Fio + ug] = hv;
        if (more) put(buf_of(c, c.epoch + 1) + orow * F + ug, hv, c.epoch + 1, c.local);
      }
      if (more) ++c.epoch;
    }
    __syncthreads();
    if (owner) y = mv[orow * 32 + ocl];
    __syncthreads();
    STAMP_ADD(t_rnnphase, sp0);
  }

#ifdef ODEVIO_STAMPS
  if (a.dbg && g == 0 && cu == 0 && tid == 0) {
    a.dbg[0] = __builtin_amdgcn_s_memtime() - t_begin;
    a.dbg[1] = c.t_gather;
    a.dbg[2] = c.t_layer;
    a.dbg[3] = c.t_rnn;
    a.dbg[4] = c.n_gather;
    a.dbg[6] = c.t_bar;
    a.dbg[7] = c.t_epi;
    a.dbg[8] = t_pro;         // launch -> first interval (census, weights -> LDS)
    a.dbg[9] = t_feval;       // vector-field evaluations (their gathers, products, epilogues)
    a.dbg[10] = t_norm;       // error norm + controller (its gather included)
    a.dbg[11] = t_rnnphase;   // RNN phases (their gathers and products included)
  }
#else
  (void)t_begin; (void)t_pro; (void)t_feval; (void)t_norm; (void)t_rnnphase;
#endif
  // ---- outputs
  if (owner && row_valid && !c.failed) {
    if (ocg < Fio) {
      if (seq_mode) a.hT[(size_t)grow * Fio + ocg] = y;
      else a.y_out[(size_t)grow * Fio + ocg] = y;
    }
    if (a.stats && cu == 0 && ocl == 0) {
      a.stats[2 * grow] = n_steps;
      a.stats[2 * grow + 1] = n_acc;
    }
  }
}

// The all-gathers need all 256 workgroups resident at once.  A plain launch gives that on an idle device, but nothing
// checks it: if the grid does not fit (fewer CUs, a register / LDS change that drops the occupancy to zero) the members
// spin until the 2 s timeout.  So the launcher asks the occupancy API for exactly this kernel / block / LDS size (once
// per device and size) and REFUSES the launch when grid > blocks-per-CU x CUs - the check hipLaunchCooperativeKernel
// makes, without its +20 us per launch (measured: integrator stage 0.696 -> 0.716 ms with the cooperative launch;
// ODEVIO_COOP_LAUNCH=1 still selects it).
template <int RT>
static int launch_rt(const IntegArgs& a, size_t lds_bytes, hipStream_t st) {
  static unsigned long long attr_mask = 0;   // per device
  static size_t checked_lds[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    const hipError_t e = once_per_device(attr_mask, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(integrator_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    });
    if (e != hipSuccess) return (int)e;
  }
  const int grid = INTEG_GROUPS * INTEG_MEMBERS;
  std::lock_guard<std::mutex> guard(launch_once_mutex());   // checked_lds is per process: plans of two host threads may launch at once
  if (checked_lds[dev & 63] != lds_bytes) {
    int per_cu = 0, n_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(integrator_kernel<RT>),
                                                                INTEG_THREADS, lds_bytes);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if ((long)per_cu * n_cu < grid) return (int)hipErrorCooperativeLaunchTooLarge;
    checked_lds[dev & 63] = lds_bytes;
  }
  static const bool coop = getenv("ODEVIO_COOP_LAUNCH") != nullptr;
  if (!coop) {
    hipLaunchKernelGGL(integrator_kernel<RT>, dim3(grid), dim3(INTEG_THREADS), lds_bytes, st, a);
    return 0;
  }
  IntegArgs args = a;
  void* params[] = {&args};
  return (int)hipLaunchCooperativeKernel(reinterpret_cast<const void*>(integrator_kernel<RT>), dim3(grid), dim3(INTEG_THREADS), params,
                                         (unsigned)lds_bytes, st);
}

int launch_integrator(const IntegArgs& a, int rt, size_t lds_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();  // do not inherit a stale error from an unrelated call
  int e;
  if (rt <= 2) e = launch_rt<2>(a, lds_bytes, st);
  else if (rt <= 4) e = launch_rt<4>(a, lds_bytes, st);
  else e = launch_rt<8>(a, lds_bytes, st);
  if (e) return e;
  return (int)hipGetLastError();
}

// =====================================================================================================================
// The adjoint twin: reverse sweep of one interval's accepted steps (IntegAdjArgs, integrator.h).  Same groups, members, owner
// map, exchange protocol and layer products as integrator_kernel; the chain runs the ODEFunc backwards through W^T.
// =====================================================================================================================
// derivative of ODEFunc's hidden activation through its saved OUTPUT a (train.hip's tr_act_grad; ODEFunc.py:23-36)
__device__ __forceinline__ float hidden_act_grad(float a, int act) {
  switch (act) {
    case 0: return 1.f - a * a;
    case 1: return a > 0.f ? 1.f : 0.f;
    case 2: return a > 0.f ? 1.f : 0.01f;
    default: return a > 20.f ? 1.f : -expm1f(-a);
  }
}

template <int RT>
__global__ __launch_bounds__(INTEG_THREADS) void integrator_adj_kernel(const IntegAdjArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int MAXG = (RT * INTEG_KMAX + NT - 1) / NT;
  Ctx c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave = c.tid >> 6;
  const int g = blockIdx.x & (INTEG_GROUPS - 1);
  c.cu = blockIdx.x >> 3;
  if (g >= a.G) return;
  c.epoch = 0;
  c.status = a.status;
  c.failed = false;
  c.local = false;
  c.xb = a.xbuf + (size_t)(2 * g) * a.xstride;
  c.xstride = a.xstride;
  c.t_gather = c.t_layer = c.t_rnn = c.n_gather = c.t_bar = c.t_epi = 0;
  const int tid = c.tid, lane = c.lane, wave = c.wave, cu = c.cu;

  float* xin = smem + a.lds_xin;    // [RT][KMAXp]   gathered layer input
  float* lay = smem + a.lds_misc;   // [128][RT]     layer-product totals, out[col*RT + row]
  float* nrm = lay + 128 * RT;      // [RT][32]      census scratch
  float* wl = smem + a.lds_w;

  const int F = a.F, Fio = a.Fio;
  const int NCF = F / INTEG_MEMBERS;
  const int R = a.rows_per_group;
  const int BPG = a.BPG;
  const int nlin = a.nlin, S = a.S;

  // ---- owner role, as in the forward: thread t < NCF*RT owns state element (row = t % RT, local column = t / RT)
  const int orow = tid % RT, ocl = tid / RT;
  const bool owner = ocl < NCF && orow < R;
  const int ocg = cu * NCF + (ocl < NCF ? ocl : 0);
  int grow = 0;
  bool row_valid = false;
  if (orow < R) {
    const int row_l = orow / BPG;
    const int row_b = a.b_begin + g * BPG + (orow - row_l * BPG);
    row_valid = row_b < a.b_end;
    grow = row_l * a.B + row_b;
  }
  const bool mine_io = owner && row_valid && ocg < Fio;   // this thread's state element exists in the caller's tensors
  float lam = mine_io ? a.lam[(size_t)grow * Fio + ocg] : 0.f;

  if (a.allow_local) {   // placement census (see integrator_kernel)
    ++c.epoch;
    const unsigned mine = xcc_id();
    if (tid == 0) put(buf_of(c, c.epoch) + cu, __uint_as_float(mine + 1u), c.epoch, false);
    gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, 1, INTEG_MEMBERS, INTEG_MEMBERS, nrm);
    bool same = true;
    for (int m = 0; m < INTEG_MEMBERS; ++m) same = same && (__float_as_uint(nrm[m]) == mine + 1u);
    c.local = same && !c.failed;
    __syncthreads();
  }

  // ---- resident slices of the transposed weights -> LDS; one that does not fit lives in registers (layer_reg's shape)
  for (int l = 0; l < nlin; ++l) {
    if (a.w_lds_off[l] < 0) continue;
    const int n = (a.dims[l] / INTEG_MEMBERS) * pad256(a.dims[l + 1]);
    const float* src = a.wT[l] + (size_t)cu * n;
    float* dstw = wl + a.w_lds_off[l];
    for (int i = tid * 4; i < n; i += NT * 4)
      *reinterpret_cast<f32x4*>(dstw + i) = *reinterpret_cast<const f32x4*>(src + i);
  }
  __syncthreads();
  int lreg = -1;
  f32x4 wreg[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int l = 0; l < nlin; ++l)
    if (lreg < 0 && a.w_lds_off[l] < 0 && a.dims[l] / INTEG_MEMBERS <= 16 && pad256(a.dims[l + 1]) == 512) lreg = l;
  if (lreg >= 0) {
    const int NC = a.dims[lreg] / INTEG_MEMBERS;
    if (wave < NC) {
      const float* src = a.wT[lreg] + (size_t)cu * NC * 512 + ((size_t)wave * 64 + lane) * 4;
      wreg[0] = *reinterpret_cast<const f32x4*>(src);
      wreg[1] = *reinterpret_cast<const f32x4*>(src + (size_t)NC * 256);
    }
  }

  int off_ode[RT];

  // the ODEFunc backwards: the owners hand in the gradient dl at the last Linear's output (tape row rb of their row) and get the
  // gradient at the stage's input back; every layer's pre-activation gradient goes to the tape on the way
  auto feval_T = [&](float dl, size_t rb) __attribute__((always_inline)) -> float {
    float gx = 0.f;
    ++c.epoch;
    if (owner) put(buf_of(c, c.epoch) + orow * F + ocg, dl, c.epoch, c.local);
    for (int l = nlin - 1; l >= 0; --l) {
      const int K = a.dims[l + 1], N = a.dims[l];   // W_l^T: K inputs (the layer's outputs), N outputs (its inputs)
      const int Kp = pad256(K);
      const int NC = N / INTEG_MEMBERS;
      const bool more = l > 0;
      const bool mine = ocl < NC && orow < R;
      const int col = cu * NC + (ocl < NC ? ocl : 0);
      const bool mine_tape = mine && row_valid && col < a.dims_io[l];
      // the saved activation this thread's product is scaled with: asked for before the gather, used after the product
      float av = 0.f;
      if (more && mine_tape) av = a.tape_act[l][rb * a.dims_io[l] + col];
      gather<MAXG>(c, buf_of(c, c.epoch), c.epoch, R, K, Kp, xin);
#pragma unroll
      for (int r = 0; r < RT; ++r) off_ode[r] = (r < R ? r : R - 1) * Kp;
      if (a.w_lds_off[l] >= 0)
        layer<RT>(wl + a.w_lds_off[l], NC, Kp, xin, off_ode, 0, xin, off_ode, wave, lane, lay);
      else if (l == lreg)
        layer_reg<RT>(wreg, NC, xin, off_ode, wave, lane, lay);
      else
        layer<RT>(a.wT[l] + (size_t)cu * NC * Kp, NC, Kp, xin, off_ode, 0, xin, off_ode, wave, lane, lay);
      __syncthreads();
      if (mine) {
        float v = lay[tid];
        if (more) {
          v *= hidden_act_grad(av, a.act);
          if (mine_tape) a.tape_delta[l - 1][rb * a.dims_io[l] + col] = v;
          put(buf_of(c, c.epoch + 1) + orow * N + col, v, c.epoch + 1, c.local);
        } else {
          gx = v;
        }
      }
      if (more) ++c.epoch;
    }
    return gx;
  };

  for (int j = a.Jrun - 1; j >= 0 && !c.failed; --j) {   // (steps Jrun .. J - 1 are zero-length for every row: nothing to sweep)
    const size_t step_row = ((size_t)a.it * a.J + j) * a.Rtot + grow;
    const float dtr = (orow < R && row_valid) ? a.dt[step_row] : 0.f;
    float lk[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) lk[s] = s < S ? dtr * a.tb[s] * lam : 0.f;   // lamK_s = dt b_s lam
    for (int s = S - 1; s >= 0; --s) {
      if (c.failed) break;
      const size_t rb = (size_t)s * a.stage_rows + step_row;
      float dl = 0.f;
      if (mine_io) {
        const float ks = a.tape_act[nlin][rb * Fio + ocg];   // K_s = tanh(.)
        float lks = 0.f;
#pragma unroll
        for (int q = 0; q < 7; ++q)
          if (q == s) lks = lk[q];
        dl = lks * (1.f - ks * ks);
        a.tape_delta[nlin - 1][rb * Fio + ocg] = dl;
      }
      const float gx = feval_T(dl, rb);
      if (owner) {
        lam += gx;
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (q < s) {
            const float co = a.ta[s][q];
            if (co != 0.f) lk[q] += dtr * co * gx;
          }
      }
    }
  }
  if (mine_io && !c.failed) a.lam[(size_t)grow * Fio + ocg] = lam;
}

template <int RT>
static int launch_adj_rt(const IntegAdjArgs& a, size_t lds_bytes, hipStream_t st) {
  static unsigned long long attr_mask = 0;   // per device
  static size_t checked_lds[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    const hipError_t e = once_per_device(attr_mask, [] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(integrator_adj_kernel<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    });
    if (e != hipSuccess) return (int)e;
  }
  const int grid = INTEG_GROUPS * INTEG_MEMBERS;
  std::lock_guard<std::mutex> guard(launch_once_mutex());
  if (checked_lds[dev & 63] != lds_bytes) {   // every member resident at once, or no launch (see launch_rt)
    int per_cu = 0, n_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(integrator_adj_kernel<RT>), INTEG_THREADS, lds_bytes);
    if (e != hipSuccess) return (int)e;
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return (int)e;
    if ((long)per_cu * n_cu < grid) return (int)hipErrorCooperativeLaunchTooLarge;
    checked_lds[dev & 63] = lds_bytes;
  }
  hipLaunchKernelGGL(integrator_adj_kernel<RT>, dim3(grid), dim3(INTEG_THREADS), lds_bytes, st, a);
  return 0;
}

int launch_integrator_adj(const IntegAdjArgs& base, int L, int B, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();
  const int bpg_max = 8 / L;  // rows per group <= 8 (the forward's chunks: api.hip run_sequence)
  const int chunk = INTEG_GROUPS * bpg_max;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = std::min(chunk, B - b0);
    const int BPG = (nb + INTEG_GROUPS - 1) / INTEG_GROUPS;
    const int R = L * BPG;
    const int rt = R <= 2 ? 2 : (R <= 4 ? 4 : 8);
    IntegAdjArgs a = base;
    a.B = B; a.b_begin = b0; a.b_end = b0 + nb;
    a.BPG = BPG; a.G = (nb + BPG - 1) / BPG; a.rows_per_group = R;
    // LDS carve (floats), as the forward's: gathered input, product totals + census scratch, then the largest slices that fit
    int maxdim = 256;
    for (int l = 0; l <= a.nlin; ++l) maxdim = std::max(maxdim, (a.dims[l] + 255) & ~255);
    int off = 0;
    a.lds_xin = off; off += rt * maxdim;
    a.lds_misc = off; off += 128 * rt + rt * 32;
    off = (off + 3) & ~3;
    a.lds_w = off;
    int budget = (160 * 1024 - 1024) / 4 - off;
    int order[INTEG_MAX_LIN];
    auto slice = [&](int l) { return (a.dims[l] / INTEG_MEMBERS) * ((a.dims[l + 1] + 255) & ~255); };
    for (int l = 0; l < a.nlin; ++l) { order[l] = l; a.w_lds_off[l] = -1; }
    std::stable_sort(order, order + a.nlin, [&](int x, int y) { return slice(x) > slice(y); });
    int woff = 0;
    for (int i = 0; i < a.nlin; ++i) {
      const int l = order[i], n = slice(l);
      if (n <= budget) { a.w_lds_off[l] = woff; woff += n; budget -= n; }
    }
    const size_t lds = (size_t)(off + woff) * sizeof(float);
    if (hipMemsetAsync(a.xbuf, 0, (size_t)INTEG_GROUPS * 2 * a.xstride * sizeof(unsigned long long), st) != hipSuccess) return (int)hipGetLastError();
    int e;
    if (rt <= 2) e = launch_adj_rt<2>(a, lds, st);
    else if (rt <= 4) e = launch_adj_rt<4>(a, lds, st);
    else e = launch_adj_rt<8>(a, lds, st);
    if (e) return e;
  }
  return (int)hipGetLastError();
}
