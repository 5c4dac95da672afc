// Backward of the ODE-RNN pose path: gradients of a loss on the poses through the regressor, the
// nn.RNN / nn.GRU stack and the Runge-Kutta solve of every interval, down to the fused features, the carried hidden state and
// every weight of ODEFunc / RNN / regressor - what `loss.backward()` reaches below the encoders in the reference's training
// step (scripts/train_model.py:69-78: poses -> 100 * MSE(angles) + MSE(trans) -> autograd through torchode's AutoDiffAdjoint
// = backpropagation through the solver's own operations).  Also here: the skinny fp32-MFMA products every backward of the
// library uses, the fusion / inertial-encoder backwards, the loss, gradient clipping and the optimizer step.
//
// Discretise-then-optimise: the backward differentiates exactly the arithmetic the forward performs (same tableau, same
// steps), so the gradients are the gradients of the computed poses, not of the continuous ODE.
//
// Structure:
//   1. TAPE: every layer input / output of every stage of every step in row-stacked matrices act[l] [M, dims[l]],
//      M = stages x intervals x steps x rows, and the RNN cell inputs / outputs per layer.  With the forward's log (the state every
//      accepted step starts from, IntegArgs::ylog / yend) all steps are rebuilt in ONE batch per stage and layer; without it
//      the steps are walked in order from the recomputed states;
//   2. REVERSE SWEEP interval by interval: RNN cell backward (launches), then the adjoint of the interval's RK steps - one launch
//      of integrator_adj_kernel (integrator.hip: the persistent forward kernel's twin on W^T) or, where that is unavailable, one
//      launch per product - writing the pre-activation gradients into delta[l] [M, dims[l+1]];
//   3. WEIGHT GRADIENTS as ONE product per weight: dW_l = delta[l]^T act[l] (contraction over all M rows at once), the bias
//      gradient from the same launch.
// Solvers: fixed-step (rk4 = 3/8 rule, rk4_classic; any ode_substeps) and adaptive (dopri5, tsit5, heun, euler under torchode's
// controller: the forward's ACCEPTED steps are replayed from the log, their sizes treated as constants of the
// differentiation); tanh nn.RNN and nn.GRU; every ODEFunc activation; the gradient is returned w.r.t. the FUSED features.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/odevio.h"
#include "common.h"
#include "train.h"
#include "bn_train.h"

// element-wise launches: a grid-stride loop over n elements
#define EW_GRID(n) dim3((unsigned)std::min<size_t>(((size_t)(n) + 255) / 256, 4096)), dim3(256)
#define EW_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); i += (size_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------------------------------------------------------------
// Skinny GEMMs on the fp32 MFMA (v_mfma_f32_16x16x4_f32: an exact fmaf chain).  One 4-wave workgroup per 16 x 16 output
// tile; the waves take the 16-wide k-steps round robin and combine through LDS in wave order (deterministic).
//   NT: out[m][n] (+)= sum_k A[m * lda + k] * W[n * ldw + k] (+ bias[n])          K % 4 == 0, lda % 4 == 0, ldw % 4 == 0
//   TN: out[n][k] (+)= sum_m D[m * ldd + n] * A[m * lda + k]                      (weight gradients: contraction over rows)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tr_act(float v, int act) {   // ODEFunc.py:23-36 (LeakyReLU default slope 0.01)
  switch (act) {
    case 0: return tanhf(v);
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.01f * v;
    default: return v > 20.f ? v : log1pf(expf(v));
  }
}
// derivative expressed through the activation's OUTPUT a (what the tape keeps)
__device__ __forceinline__ float tr_act_grad(float a, int act) {
  switch (act) {
    case 0: return 1.f - a * a;
    case 1: return a > 0.f ? 1.f : 0.f;
    case 2: return a > 0.f ? 1.f : 0.01f;
    default: return a > 20.f ? 1.f : -expm1f(-a);   // softplus: sigmoid(z) = 1 - exp(-a)
  }
}

// Epilogues of the NT product (fused so that a layer of the tape / of the reverse sweep is ONE launch, not two):
//   EPI_ACT: out = act(sum + bias)  - the ODEFunc layer itself;   EPI_DACT: out = sum * act'(aux[m][n]) - the layer's adjoint, aux = the
//   activation's saved OUTPUT (what the tape keeps).  Same arithmetic, in the same order, as the separate element-wise kernels.
enum GemmEpi { GEPI_NONE = 0, GEPI_ACT = 1, GEPI_DACT = 2 };

// The 16 x 16 tile of A W^T both NT kernels compute: returns element (n0 + tid / 16, m0 + tid % 16).
// A wave's k-steps (16 * wave, += 64) U at a time: the 2 U 16-byte loads of a batch are issued before the first MFMA waits on one,
// so a row of K <= 64 U costs ONE round trip to memory, not U in a chain (U = 8 for K <= 512, 16 above); and FOUR accumulators (one
// per k of a lane's float4), because these products are latency, not bandwidth: the fp32 MFMA's result is ready ~40 ns after issue,
// and 32 of them chained on one accumulator were 1.2 us of a 4 us workgroup (tools/probes/skinny_gemm.hip).
// K % 4 == 0: a lane's four consecutive k are inside K or all outside (a tail adds zeros); the ADDRESS is clamped and the VALUE
// selected, so that no load sits behind a branch.
// NW waves share the k-steps of a row round robin (4, or 16 for the long rows of the RNN's 3F-wide products: 36 steps of one wave
// were three batches in a chain).
template <int U, int NW>
__device__ __forceinline__ float nt_tile(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int M, int N, int K, int m0, int n0,
                                         float (*red)[16][17]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // MFMA A operand = W rows (D rows = output columns n), B operand = A rows (D columns = m)
  const float* wrow = W + (size_t)min(n0 + r, N - 1) * ldw + 4 * q;
  const float* arow = A + (size_t)min(m0 + r, M - 1) * lda + 4 * q;
  f32x4 acc4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int k0 = 16 * wave; k0 < K; k0 += 16 * NW * U) {
    f32x4 wv[U], av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 16 * NW * u + 4 * q;
      const bool in = k < K;
      const int kc = in ? k - 4 * q : 0;
      wv[u] = *reinterpret_cast<const f32x4*>(wrow + kc);
      av[u] = *reinterpret_cast<const f32x4*>(arow + kc);
      if (!in) { wv[u] = f32x4{0.f, 0.f, 0.f, 0.f}; av[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j], av[u][j], acc4[j], 0, 0, 0);
  }
  const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][4 * q + e][r] = acc[e];   // [n local][m local]
  __syncthreads();
  const int nl = (tid >> 4) & 15, ml = tid & 15;
  float g[NW / 4];
#pragma unroll
  for (int i = 0; i < NW / 4; ++i) g[i] = (red[4 * i][nl][ml] + red[4 * i + 1][nl][ml]) + (red[4 * i + 2][nl][ml] + red[4 * i + 3][nl][ml]);
  if (NW == 4) return g[0];
  return (g[0] + g[1 % (NW / 4)]) + (g[2 % (NW / 4)] + g[3 % (NW / 4)]);
}

// What the epilogue reads besides the sum (bias, the saved activation, the old value when accumulating) is LOADED FIRST, under the
// operand loads: these kernels run in chains where every operand was written by the launch before and comes from beyond the L2 - a
// second round trip behind the product was a quarter of the launch.
template <int U, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo, int M, int N, int K,
                                                          int accumulate, int epi, int act, const float* __restrict__ aux, int ldaux) {
  __shared__ float red[NW][16][17];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int n = n0 + ((tid >> 4) & 15), m = m0 + (tid & 15);
  const bool mine = tid < 256 && n < N && m < M;
  float* o = out + (size_t)(mine ? m : 0) * ldo + (mine ? n : 0);
  const float bv = bias ? bias[mine ? n : 0] : 0.f;
  const float xv = epi == GEPI_DACT ? aux[(size_t)(mine ? m : 0) * ldaux + (mine ? n : 0)] : 0.f;
  const float ov = accumulate ? *o : 0.f;
  float v = nt_tile<U, NW>(A, lda, W, ldw, M, N, K, m0, n0, red);
  if (mine) {
    if (bias) v += bv;
    if (epi == GEPI_ACT) v = tr_act(v, act);
    else if (epi == GEPI_DACT) v *= tr_act_grad(xv, act);
    *o = accumulate ? ov + v : v;
  }
}

// bias_out (optional): the column sums of D (out_b[n] = sum_m D[m][n], the bias gradient that goes with a weight gradient) from the
// same launch: one more column of workgroups (blockIdx.x == tiles of K) contracts D with a column of ones.
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ A, int lda,
                                                      float* __restrict__ out, int ldo, int M, int N, int K, int accumulate,
                                                      float* __restrict__ bias_out) {
  __shared__ float red[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const bool ones = (int)blockIdx.x * 16 >= K;   // the bias column
  const int k0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int nn = min(n0 + r, N - 1), kk = min(k0 + r, K - 1);
  f32x4 acc4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // four chains (see gemm_nt_kernel)
  // contraction over rows m: MFMA j of a 16-row step uses rows m = 16 s + 4 q + j; four steps' loads are issued together
  for (int mb = 16 * wave; mb < M; mb += 256) {
    float dv[4][4], av[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mb + 64 * u + 4 * q + j;
        const int mc = m < M ? m : 0;
        const float d = D[(size_t)mc * ldd + nn];
        const float a = ones ? (r == 0 ? 1.f : 0.f) : A[(size_t)mc * lda + kk];
        dv[u][j] = m < M ? d : 0.f;
        av[u][j] = m < M ? a : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][j], av[u][j], acc4[j], 0, 0, 0);   // D rows = n, D columns = k
  }
  const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][4 * q + e][r] = acc[e];   // [n local][k local]
  __syncthreads();
  const int nl = tid >> 4, kl = tid & 15;
  const int n = n0 + nl, k = k0 + kl;
  if (ones) {
    if (kl == 0 && n < N) {
      const float v = (red[0][nl][0] + red[1][nl][0]) + (red[2][nl][0] + red[3][nl][0]);
      bias_out[n] = accumulate ? bias_out[n] + v : v;
    }
    return;
  }
  if (n < N && k < K) {
    const float v = (red[0][nl][kl] + red[1][nl][kl]) + (red[2][nl][kl] + red[3][nl][kl]);
    float* o = out + (size_t)n * ldo + k;
    *o = accumulate ? *o + v : v;
  }
}

// The TN product for weight gradients with many rows (the tape's M = intervals x steps x stages x rows): a 64 x 64 output tile per
// workgroup instead of 16 x 16 - the same two 16-byte loads per lane now feed SIXTEEN MFMAs (n = n0 + 4 r + e from the float4 of D,
// k = k0 + 4 r + f from the float4 of A), where gemm_tn_kernel reads two floats per MFMA: the 512 x 512 gradient over 11,520 rows
// took 310 us there, all of it operand traffic (1,024 workgroups each walking every row).  Rows are split over the four waves and
// over `splits` workgroup layers; the waves combine in a fixed tree through LDS, the layers through slabs added in order
// (deterministic).  Column sums of D (the bias gradient) come from one more column of workgroups, as in gemm_tn_kernel.
// N, K, ldd, lda multiples of 4.
__global__ __launch_bounds__(256) void gemm_tn64_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ A, int lda,
                                                        float* __restrict__ out, int ldo, float* __restrict__ bias_out, int M, int N, int K,
                                                        int rows_per_split, int ktiles) {
  __shared__ float slot[2][64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const bool ones = (int)blockIdx.x >= ktiles;
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int m_begin = blockIdx.z * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  const bool n_in = n0 + 4 * r < N, k_in = !ones && k0 + 4 * r < K;
  const float* dcol = D + (n_in ? n0 + 4 * r : 0);
  const float* acol = A + (k_in ? k0 + 4 * r : 0);
  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // a wave's 4-row steps: m_begin + 4 * wave, += 16; four steps' loads in flight
  for (int mb = m_begin + 4 * wave; mb < m_end; mb += 64) {
    f32x4 dv[4], av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = mb + 16 * u + q;
      const int mc = m < m_end ? m : m_begin;
      dv[u] = *reinterpret_cast<const f32x4*>(dcol + (size_t)mc * ldd);
      av[u] = ones ? (r == 0 ? f32x4{1.f, 0.f, 0.f, 0.f} : zero) : *reinterpret_cast<const f32x4*>(acol + (size_t)mc * lda);
      if (m >= m_end || !n_in) dv[u] = zero;
      if (m >= m_end || (!ones && !k_in)) av[u] = zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[u][e], av[u][f], acc[e][f], 0, 0, 0);
  }
  // ((wave 0 + wave 1) + (wave 2 + wave 3)): element (e, f, reg) of lane l sits at [((e * 4 + f) * 4 + reg) * 64 + l]
  auto put = [&](float* dst) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[((e * 4 + f) * 4 + g) * 64 + lane] = acc[e][f][g];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[e][f][g] += src[((e * 4 + f) * 4 + g) * 64 + lane];
  };
  if (wave == 1) put(slot[0]);
  if (wave == 3) put(slot[1]);
  __syncthreads();
  if (wave == 0) add(slot[0]);
  if (wave == 2) add(slot[1]);
  __syncthreads();
  if (wave == 2) put(slot[0]);
  __syncthreads();
  if (wave != 0) return;
  add(slot[0]);
  // lane l holds output rows i = 4 (l >> 4) + g and column j = l & 15 of every 16 x 16 block: n = n0 + 4 i + e, k = k0 + 4 j + f
  const int j = lane & 15;
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + 4 * (4 * (lane >> 4) + g) + e;
      if (n >= N) continue;
      if (ones) {
        if (j == 0) bias_out[(size_t)blockIdx.z * N + n] = acc[e][0][g];
      } else if (k0 + 4 * j < K) {
        *reinterpret_cast<f32x4*>(out + ((size_t)blockIdx.z * N + n) * ldo + k0 + 4 * j) = f32x4{acc[e][0][g], acc[e][1][g], acc[e][2][g], acc[e][3][g]};
      }
    }
}
// sum of `splits` slabs [N][K] (ld K) in slab order -> out [N][K] (ldo); the same for the bias rows [splits][N]
__global__ void tn64_combine_kernel(const float* __restrict__ partial, const float* __restrict__ bias_partial, float* __restrict__ out, int ldo,
                                    float* __restrict__ bias_out, int N, int K, int splits, int accumulate) {
  EW_LOOP(i, (size_t)N * (K + 1)) {
    const int n = (int)(i / (K + 1)), k = (int)(i - (size_t)n * (K + 1));
    if (k == K) {
      if (!bias_out) continue;
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += bias_partial[(size_t)z * N + n];
      bias_out[n] = accumulate ? bias_out[n] + s : s;
    } else {
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += partial[((size_t)z * N + n) * K + k];
      float* o = out + (size_t)n * ldo + k;
      *o = accumulate ? *o + s : s;
    }
  }
}
// scratch of the split form: one buffer per device, grown on demand (allocation is rare and synchronous; its users are ordered by their stream)
static float* tn64_scratch(size_t floats) {
  static std::mutex mu;
  static float* buf[16] = {};
  static size_t cap[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (cap[dev] < floats) {
    if (buf[dev]) { (void)hipDeviceSynchronize(); (void)hipFree(buf[dev]); buf[dev] = nullptr; cap[dev] = 0; }
    if (hipMalloc((void**)&buf[dev], floats * sizeof(float)) != hipSuccess) return nullptr;
    cap[dev] = floats;
  }
  return buf[dev];
}

// The NT product for MANY rows (the batched tape: every step of the window at once; the encoders' row matrices): a 64 x 64 output tile
// per workgroup - a k-step's eight 16-byte loads per lane feed 64 MFMAs, where the 16 x 16 tile re-reads its W rows once per 16 rows
// of A (120 times for the tape's 1,920 rows).  The four waves take the k-steps round robin and combine in a fixed tree through LDS.
// Epilogues: bias, EPI_ACT, accumulate.  N, ldo multiples of 4.
__global__ __launch_bounds__(256) void gemm_nt64_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ bias, float* __restrict__ out, int ldo, int M, int N, int K,
                                                        int accumulate, int epi, int act) {
  __shared__ float slot[2][64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const float* wrow[4];
  const float* arow[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    wrow[b] = W + (size_t)min(n0 + 16 * b + r, N - 1) * ldw + 4 * q;
    arow[b] = A + (size_t)min(m0 + 16 * b + r, M - 1) * lda + 4 * q;
  }
  f32x4 acc[4][4];   // [m block][n block]
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 16 * wave; k0 < K; k0 += 128) {   // two k-steps of this wave per batch
    f32x4 wv[2][4], av[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + 64 * u + 4 * q;
      const bool in = k < K;
      const int kc = in ? k - 4 * q : 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        wv[u][b] = *reinterpret_cast<const f32x4*>(wrow[b] + kc);
        av[u][b] = *reinterpret_cast<const f32x4*>(arow[b] + kc);
        if (!in) { wv[u][b] = f32x4{0.f, 0.f, 0.f, 0.f}; av[u][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][nb][j], av[u][mb][j], acc[mb][nb], 0, 0, 0);
  }
  auto put = [&](float* dst) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) dst[((mb * 4 + nb) * 4 + g) * 64 + lane] = acc[mb][nb][g];
  };
  auto add = [&](const float* src) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[mb][nb][g] += src[((mb * 4 + nb) * 4 + g) * 64 + lane];
  };
  if (wave == 1) put(slot[0]);
  if (wave == 3) put(slot[1]);
  __syncthreads();
  if (wave == 0) add(slot[0]);
  if (wave == 2) add(slot[1]);
  __syncthreads();
  if (wave == 2) put(slot[0]);
  __syncthreads();
  if (wave != 0) return;
  add(slot[0]);
  // lane l of block (mb, nb): output column n = n0 + 16 nb + 4 (l >> 4) + g (g = the accumulator's register), row m = m0 + 16 mb + (l & 15)
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = m0 + 16 * mb + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = n0 + 16 * nb + 4 * (lane >> 4);
      if (n >= N) continue;   // N % 4 == 0: the four columns are inside or outside together
      f32x4 v = acc[mb][nb];
      if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
      if (epi == GEPI_ACT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = tr_act(v[g], act);
      }
      f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)m * ldo + n);
      *o = accumulate ? *o + v : v;
    }
  }
}

static void gemm_nt(hipStream_t st, const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N,
                    int K, bool accumulate = false, int epi = GEPI_NONE, int act = 0, const float* aux = nullptr, int ldaux = 0) {
  const int acc = accumulate ? 1 : 0;
  // (only where the 64 x 64 tiles still fill the chip twice over: at the tape's 320 - 1,920 rows the small tiles' 10 x as many
  //  workgroups win - 2.07 against 2.46 ms for the rk4 training step)
  if ((long)((M + 63) / 64) * ((N + 63) / 64) >= 512 && N % 4 == 0 && ldo % 4 == 0 && epi != GEPI_DACT && getenv("ODEVIO_NT_NARROW") == nullptr &&
      (((uintptr_t)out | (uintptr_t)bias) & 15) == 0) {
    hipLaunchKernelGGL(gemm_nt64_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldo, M, N, K, acc, epi, act);
    return;
  }
  const dim3 grid((N + 15) / 16, (M + 15) / 16);
  if (K <= 512)
    hipLaunchKernelGGL((gemm_nt_kernel<8, 4>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, out, ldo, M, N, K, acc, epi, act, aux, ldaux);
  else if (K <= 1024 || grid.x * grid.y >= 512)   // (a grid that fills the chip anyway keeps the small workgroups)
    hipLaunchKernelGGL((gemm_nt_kernel<16, 4>), grid, dim3(256), 0, st, A, lda, W, ldw, bias, out, ldo, M, N, K, acc, epi, act, aux, ldaux);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<16, 16>), grid, dim3(1024), 0, st, A, lda, W, ldw, bias, out, ldo, M, N, K, acc, epi, act, aux, ldaux);
}
static void gemm_tn(hipStream_t st, const float* D, int ldd, const float* A, int lda, float* out, int ldo, int M, int N, int K,
                    bool accumulate = false, float* bias_out = nullptr) {
  // many rows and an output of at least a few 64 x 64 tiles: the wide-tile kernel (split over rows to fill the chip)
  if (M >= 512 && N >= 64 && K >= 64 && N % 4 == 0 && K % 4 == 0 && ldd % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && getenv("ODEVIO_TN_NARROW") == nullptr) {
    const int ktiles = (K + 63) / 64, ntiles = (N + 63) / 64;
    int splits = std::max(1, std::min({1024 / (ktiles * ntiles), M / 256, 64}));
    const int rows = ((M + splits - 1) / splits + 15) / 16 * 16;
    splits = (M + rows - 1) / rows;
    if (splits == 1 && !accumulate) {
      hipLaunchKernelGGL(gemm_tn64_kernel, dim3(ktiles + (bias_out ? 1 : 0), ntiles, 1), dim3(256), 0, st, D, ldd, A, lda, out, ldo, bias_out, M, N, K, rows, ktiles);
      return;
    }
    float* scratch = tn64_scratch((size_t)splits * N * (K + 1));
    if (scratch) {
      float* bias_partial = scratch + (size_t)splits * N * K;
      hipLaunchKernelGGL(gemm_tn64_kernel, dim3(ktiles + (bias_out ? 1 : 0), ntiles, splits), dim3(256), 0, st, D, ldd, A, lda, scratch, K, bias_partial, M, N, K,
                         rows, ktiles);
      hipLaunchKernelGGL(tn64_combine_kernel, EW_GRID((size_t)N * (K + 1)), 0, st, scratch, bias_partial, out, ldo, bias_out, N, K, splits, accumulate ? 1 : 0);
      return;
    }
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3((K + 15) / 16 + (bias_out ? 1 : 0), (N + 15) / 16), dim3(256), 0, st, D, ldd, A, lda, out, ldo, M, N, K,
                     accumulate ? 1 : 0, bias_out);
}

// ---------------------------------------------------------------------------------------------------------------------
// element-wise kernels
// ---------------------------------------------------------------------------------------------------------------------

__global__ void act_kernel(float* x, size_t n, int act) { EW_LOOP(i, n) x[i] = tr_act(x[i], act); }
__global__ void copy_kernel(const float* __restrict__ a, float* __restrict__ b, size_t n) { EW_LOOP(i, n) b[i] = a[i]; }

// dt[it][j][r]: fixed-step solvers: (ts[b][it+1] - ts[b][it]) / J with b = r % B (the relative shift of PoseODERNN.py:100
// cancels in the difference); adaptive solvers: the j-th ACCEPTED step of row r in interval it from the forward's log, or
// 0 once the row has reached the end of the interval
__global__ void dt_rows_kernel(const float* __restrict__ ts, float* __restrict__ dt, int B, int P, int R, int J,
                               const float* __restrict__ dtlog, const int* __restrict__ dtcnt, int cap) {
  EW_LOOP(i, (size_t)P * J * R) {
    const int r = (int)(i % R), j = (int)((i / R) % J), it = (int)(i / ((size_t)R * J));
    if (dtlog) {
      dt[i] = j < dtcnt[(size_t)r * P + it] ? dtlog[((size_t)r * P + it) * cap + j] : 0.f;
    } else {
      const float* tr = ts + (size_t)(r % B) * (P + 1);
      dt[i] = (tr[it + 1] - tr[it]) / (float)J;
    }
  }
}

// X_s = Y + dt[r] * sum_j a[j] K_j   (K_j = kbase + j * kstride; same association as the forward kernels and the oracle)
struct StageCoef { float c[8]; int n; };
__global__ void stage_input_kernel(float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ kbase, size_t kstride,
                                   StageCoef a, const float* __restrict__ dt, int R, int F) {
  EW_LOOP(i, (size_t)R * F) {
    const int r = (int)(i / F);
    float acc = 0.f;
    bool first = true;
    for (int j = 0; j < a.n; ++j) {
      if (a.c[j] == 0.f) continue;
      const float term = kbase[j * kstride + i] * a.c[j];
      acc = first ? term : acc + term;
      first = false;
    }
    X[i] = first ? Y[i] : Y[i] + dt[r] * acc;
  }
}

// rows of a [B][P][F] tensor <-> rows of an interval-major [P*B][F] matrix
__global__ void gather_interval_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int P, int F, int it) {
  EW_LOOP(i, (size_t)B * F) {
    const int b = (int)(i / F), c = (int)(i % F);
    dst[i] = src[((size_t)b * P + it) * F + c];
  }
}
__global__ void scatter_interval_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int P, int F, int it) {
  EW_LOOP(i, (size_t)B * F) {
    const int b = (int)(i / F), c = (int)(i % F);
    dst[((size_t)b * P + it) * F + c] = src[i];
  }
}

// tanh RNN cell: hn = tanh(gi + gh)
__global__ void rnn_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ hn, float* __restrict__ ynew, size_t n) {
  EW_LOOP(i, n) {
    const float v = tanhf(gi[i] + gh[i]);
    hn[i] = v;
    ynew[i] = v;
  }
}
// delta = (g [+ g2]) * (1 - hn^2)
__global__ void rnn_cell_bwd_kernel(const float* __restrict__ g, const float* __restrict__ g2, const float* __restrict__ hn,
                                    float* __restrict__ delta, size_t n) {
  EW_LOOP(i, n) {
    const float gg = g[i] + (g2 ? g2[i] : 0.f);
    delta[i] = gg * (1.f - hn[i] * hn[i]);
  }
}
// nn.GRU cell (gate order r, z, n): gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh, both [B][3F];
//   r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h
// gates [B][4F] keeps (r, z, n, gh_n) for the backward
__device__ __forceinline__ float tr_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
__global__ void gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ hp,
                                float* __restrict__ gates, float* __restrict__ hn, float* __restrict__ ynew, int B, int F) {
  EW_LOOP(i, (size_t)B * F) {
    const size_t b = i / F, c = i % F;
    const float* gib = gi + b * 3 * F;
    const float* ghb = gh + b * 3 * F;
    const float r = tr_sigmoid(gib[c] + ghb[c]);
    const float z = tr_sigmoid(gib[F + c] + ghb[F + c]);
    const float ghn = ghb[2 * F + c];
    const float n = tanhf(gib[2 * F + c] + r * ghn);
    const float v = (1.f - z) * n + z * hp[i];
    float* gt = gates + b * 4 * F;
    gt[c] = r; gt[F + c] = z; gt[2 * F + c] = n; gt[3 * F + c] = ghn;
    hn[i] = v;
    ynew[i] = v;
  }
}
// g = dL/dh' -> delta_i [B][3F] (w.r.t. gi), delta_h [B][3F] (w.r.t. gh), dhp_direct [B][F] = g * z
// (g and dhp_direct may be the same buffer: element i is read, then written, by the same thread)
__global__ void gru_cell_bwd_kernel(const float* g, const float* __restrict__ g2, const float* __restrict__ gates,
                                    const float* __restrict__ hp, float* __restrict__ di, float* __restrict__ dh,
                                    float* dhp_direct, int B, int F) {
  EW_LOOP(i, (size_t)B * F) {
    const size_t b = i / F, c = i % F;
    const float* gt = gates + b * 4 * F;
    const float r = gt[c], z = gt[F + c], n = gt[2 * F + c], ghn = gt[3 * F + c];
    const float gg = g[i] + (g2 ? g2[i] : 0.f);
    const float dn_pre = gg * (1.f - z) * (1.f - n * n);
    const float dz_pre = gg * (hp[i] - n) * z * (1.f - z);
    const float dr_pre = dn_pre * ghn * r * (1.f - r);
    float* dib = di + b * 3 * F;
    float* dhb = dh + b * 3 * F;
    dib[c] = dr_pre; dib[F + c] = dz_pre; dib[2 * F + c] = dn_pre;
    dhb[c] = dr_pre; dhb[F + c] = dz_pre; dhb[2 * F + c] = dn_pre * r;
    dhp_direct[i] = gg * z;
  }
}
__global__ void add_kernel(float* __restrict__ x, const float* __restrict__ y, size_t n) { EW_LOOP(i, n) x[i] += y[i]; }
// delta = g * act'(a)
__global__ void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ a, float* __restrict__ delta, size_t n, int act) {
  EW_LOOP(i, n) delta[i] = g[i] * tr_act_grad(a[i], act);
}
// column sums of a [M][N] matrix (bias gradients); one thread per column, rows in order (deterministic)
// out[n] = sum over the M rows of x[m][n].  One 256-thread block per 32 columns: 8 row groups stride the rows (each thread adds its
// rows in order), then the 8 partial sums are combined in group order - deterministic, and M (up to intervals x steps x stages x
// rows of the tape) is walked by 8 x as many threads, 128 bytes per row segment, instead of one thread per column.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float s = 0.f;
  if (n < N) {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};   // four independent chains (rows m, m + 8, m + 16, m + 24 of a 32-row step): loads in flight, fixed order
    int m = g;
    for (; m + 24 < M; m += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += x[(size_t)(m + 8 * u) * N + n];
    }
    for (int u = 0; m < M; m += 8, ++u) s4[u] += x[(size_t)m * N + n];
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  red[g][c] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    float t = red[0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][c];
    out[n] = t;
  }
}
static void launch_colsum(hipStream_t st, const float* x, float* out, int M, int N) {
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32), dim3(256), 0, st, x, out, M, N);
}
// weight gradient gw = D^T A and bias gradient gb = column sums of D; both wanted: one launch
static void wgrad_bias(hipStream_t st, const float* D, int ldd, const float* A, int lda, float* gw, int ldo, float* gb, int M, int N, int K) {
  if (gw) gemm_tn(st, D, ldd, A, lda, gw, ldo, M, N, K, false, ldd == N ? gb : nullptr);
  if (gb && !(gw && ldd == N)) launch_colsum(st, D, gb, M, N);
}
// regressor.2 backward: dhid[m][k] = (sum_n dp[m][n] W2[n][k]) * leaky'(hid[m][k])   (6 outputs: no GEMM needed)
__global__ void reg2_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ W2, const float* __restrict__ hid,
                                float* __restrict__ dhid, int M) {
  EW_LOOP(i, (size_t)M * 128) {
    const int m = (int)(i / 128), k = (int)(i % 128);
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) s += dp[(size_t)m * 6 + n] * W2[n * 128 + k];
    dhid[i] = s * (hid[i] > 0.f ? 1.f : 0.1f);
  }
}
__global__ void leaky_kernel(float* x, size_t n, float slope) { EW_LOOP(i, n) x[i] = x[i] > 0.f ? x[i] : slope * x[i]; }

// loss = 100 * mean((p - g)^2 over the 3 angle columns) + mean((p - g)^2 over the 3 translation columns)
// (scripts/train_model.py:72-77) and its gradient w.r.t. the poses; one workgroup, deterministic order.
__global__ __launch_bounds__(256) void pose_loss_kernel(const float* __restrict__ poses, const float* __restrict__ gts, int M,
                                                        float* __restrict__ loss3, float* __restrict__ grad) {
  __shared__ double sa[256], st[256];
  double a = 0.0, t = 0.0;
  const double inv = 1.0 / (3.0 * (double)M);
  for (int i = threadIdx.x; i < M * 6; i += 256) {
    const int c = i % 6;
    const float d = poses[i] - gts[i];
    if (c < 3) a += (double)d * d; else t += (double)d * d;
    if (grad) grad[i] = (float)((c < 3 ? 200.0 : 2.0) * inv * (double)d);
  }
  sa[threadIdx.x] = a; st[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; st[threadIdx.x] += st[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float al = (float)(sa[0] * inv), tl = (float)(st[0] * inv);
    loss3[0] = 100.f * al + tl; loss3[1] = al; loss3[2] = tl;
  }
}

int train_pose_loss(const float* poses, const float* gts, int M, float* loss3, float* grad, hipStream_t st) {
  hipLaunchKernelGGL(pose_loss_kernel, dim3(1), dim3(256), 0, st, poses, gts, M, loss3, grad);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels of the batched tape and of the fused reverse sweep
// ---------------------------------------------------------------------------------------------------------------------
// the state every accepted step starts from, for all intervals and steps at once: Y[(it * J + j) * R + r] = the forward's log
// (ylog [r][P][cap][F]) for j < cnt[r][it], the interval's evolved state (yend [r][P][F]) for the zero-length steps behind them
__global__ void ylog_rows_kernel(const float* __restrict__ ylog, const float* __restrict__ yend, const int* __restrict__ cnt, float* __restrict__ Y,
                                 int P, int J, int R, int cap, int F) {
  EW_LOOP(i, (size_t)P * J * R * F) {
    const int f = (int)(i % F);
    const size_t row = i / F;
    const int r = (int)(row % R), j = (int)((row / R) % J), it = (int)(row / ((size_t)R * J));
    const size_t ri = (size_t)r * P + it;
    Y[i] = j < cnt[ri] ? ylog[(ri * cap + j) * F + f] : yend[ri * F + f];
  }
}
// [B][P][C] -> interval-major [P][B][C] (all intervals in one launch) and the RNN's hidden inputs of layer l from yend
__global__ void to_interval_major_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int P, int C) {
  EW_LOOP(i, (size_t)P * B * C) {
    const int c = (int)(i % C), b = (int)((i / C) % B), it = (int)(i / ((size_t)C * B));
    dst[i] = src[((size_t)b * P + it) * C + c];
  }
}
__global__ void yend_rows_kernel(const float* __restrict__ yend, float* __restrict__ hp, int l, int B, int P, int F) {
  EW_LOOP(i, (size_t)P * B * F) {
    const int f = (int)(i % F), b = (int)((i / F) % B), it = (int)(i / ((size_t)F * B));
    hp[i] = yend[(((size_t)l * B + b) * P + it) * F + f];
  }
}
// start of a step's adjoint: lamK_s = dt[r] * b_s * lam for every stage, and the pre-activation gradient of the LAST stage's final
// Linear (K = tanh(.)): delta = lamK_{S-1} * (1 - K_{S-1}^2)
__global__ void step_adjoint_init_kernel(const float* __restrict__ lam, float* __restrict__ lamK, size_t kstride, StageCoef b,
                                         const float* __restrict__ dt, const float* __restrict__ Klast, float* __restrict__ delta_last, int R, int F) {
  EW_LOOP(i, (size_t)R * F) {
    const int r = (int)(i / F);
    float last = 0.f;
    for (int s = 0; s < b.n; ++s) {
      last = dt[r] * b.c[s] * lam[i];
      lamK[s * kstride + i] = last;
    }
    const float k = Klast[i];
    delta_last[i] = last * (1.f - k * k);
  }
}
// The product that ends a stage's adjoint, gX = delta_0 W_0 (gemm_nt_kernel's tile), with the stage bookkeeping as its epilogue:
//   lam += gX;  lamK_j += dt[r] * a_sj * gX for j < s;  and, for the stage below, delta = lamK_{s-1} * (1 - K_{s-1}^2).
struct AdjEpi {
  float* lam;
  float* lamK;
  size_t kstride;
  StageCoef a;
  const float* dt;
  const float* Kprev;    // K_{s-1} rows of this step (null at s = 0)
  float* delta_prev;     // where the stage below expects its last-Linear gradient
};
template <int U>
__global__ __launch_bounds__(256) void gemm_nt_adj_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int M, int N, int K,
                                                          AdjEpi e) {
  __shared__ float red[4][16][17];
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const int n = n0 + (tid >> 4), m = m0 + (tid & 15);
  const bool mine = n < N && m < M;
  const size_t i = mine ? (size_t)m * N + n : 0;
  // everything the bookkeeping reads, loaded under the product's operand loads (see gemm_nt_kernel)
  const float lam0 = e.lam[i], dtm = e.dt[mine ? m : 0], kp = e.Kprev ? e.Kprev[i] : 0.f;
  float lk[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) lk[j] = j < e.a.n ? e.lamK[j * e.kstride + i] : 0.f;
  const float g = nt_tile<U, 4>(A, lda, W, ldw, M, N, K, m0, n0, red);
  if (mine) {
    e.lam[i] = lam0 + g;
    float below = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j < e.a.n) {
        if (e.a.c[j] != 0.f) { lk[j] += dtm * e.a.c[j] * g; e.lamK[j * e.kstride + i] = lk[j]; }
        if (j == e.a.n - 1) below = lk[j];
      }
    if (e.Kprev) e.delta_prev[i] = below * (1.f - kp * kp);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
size_t train_workspace_floats(const TrainModel& m, int B, int P) {
  const int R = m.L * B, S = m.stages;
  const size_t M = (size_t)P * m.jmax * S * R, MB = (size_t)P * B;
  size_t n = 0;
  for (int l = 0; l <= m.nlin; ++l) n += M * m.dims[l];          // act
  for (int l = 0; l < m.nlin; ++l) n += M * m.dims[l + 1];       // delta
  n += ((size_t)P * m.jmax * R + 3) / 4 * 4;                     // dt (kept 16-byte aligned: the GEMM operands behind it are read as float4)
  n += 2 * (size_t)R * m.F;                                      // Y, lam
  n += 8 * (size_t)R * m.F;                                      // lamK
  const size_t G = m.gru ? 3 : 1;
  n += (size_t)m.L * MB * m.F * (3 + G);                         // rnn_in, rnn_hp, rnn_out, rnn_delta (GRU: [MB][3F])
  if (m.gru) n += (size_t)m.L * MB * m.F * (3 + 4);              // GRU: delta w.r.t. gh [MB][3F], gates [MB][4F]
  n += 2 * G * MB * m.F;                                         // gi, gh ([P*B][G*F] each: the batched tape does every interval at once)
  n += (size_t)B * m.F;                                          // dinp
  n += MB * 128 * 2 + (MB * 6 + 3) / 4 * 4 + MB * m.F;           // hid, dhid, dposes (interval-major), dout
  return n;
}

// Tape rows: stage s of step j of interval it, row r  ->  ((s * P + it) * J + j) * R + r  (a STAGE's rows of all steps are contiguous,
// which is what lets the batched tape run one product per stage and layer over every step of the window)
int train_ode_rnn_bwd(const TrainModel& m, float* ws, const float* fused, const float* ts, const float* hc, int B, int P,
                      const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc, const TrainGrads& g,
                      hipStream_t st) {
  const int L = m.L, F = m.F, R = L * B, S = m.stages, nl = m.nlin;
  const int J = m.jmax;
  const size_t PJR = (size_t)P * J * R;
  const size_t M = PJR * S, MB = (size_t)P * B, RF = (size_t)R * F, BF = (size_t)B * F;
  const size_t kstride = PJR * F;                                  // from K_s to K_{s+1} of the same step
  auto row0 = [&](int it, int j, int s) { return (size_t)s * PJR + ((size_t)it * J + j) * R; };
  // ---- carve the workspace
  float* q = ws;
  float* act[TRAIN_MAX_LIN + 1];
  float* delta[TRAIN_MAX_LIN];
  for (int l = 0; l <= nl; ++l) { act[l] = q; q += M * m.dims[l]; }
  for (int l = 0; l < nl; ++l) { delta[l] = q; q += M * m.dims[l + 1]; }
  float* dt = q; q += ((size_t)P * J * R + 3) / 4 * 4;
  float* Y = q; q += RF;
  float* lam = q; q += RF;
  float* lamK = q; q += 8 * RF;
  const int G = m.gru ? 3 : 1, GF = G * F;     // gate rows per hidden unit
  float *rnn_in[TRAIN_MAX_L], *rnn_hp[TRAIN_MAX_L], *rnn_out[TRAIN_MAX_L], *rnn_delta[TRAIN_MAX_L];
  float *rnn_delta_h[TRAIN_MAX_L] = {}, *rnn_gates[TRAIN_MAX_L] = {};
  for (int l = 0; l < L; ++l) {
    rnn_in[l] = q; q += MB * F; rnn_hp[l] = q; q += MB * F; rnn_out[l] = q; q += MB * F; rnn_delta[l] = q; q += MB * GF;
    if (m.gru) { rnn_delta_h[l] = q; q += MB * GF; rnn_gates[l] = q; q += MB * 4 * F; }
    else rnn_delta_h[l] = rnn_delta[l];      // tanh RNN: one delta serves W_ih and W_hh
  }
  float* gi = q; q += MB * GF;
  float* gh = q; q += MB * GF;
  float* dinp = q; q += BF;
  float* hid = q; q += MB * 128;
  float* dhid = q; q += MB * 128;
  float* dpo = q; q += (MB * 6 + 3) / 4 * 4;
  float* dout = q; q += MB * F;

  StageCoef arow[8], brow;
  for (int s = 0; s < S; ++s) {
    arow[s].n = s;
    for (int j = 0; j < 8; ++j) arow[s].c[j] = j < s ? m.a[s][j] : 0.f;
  }
  brow.n = S;
  for (int j = 0; j < 8; ++j) brow.c[j] = j < S ? m.b[j] : 0.f;

  // =============================== 1. tape ===============================
  hipLaunchKernelGGL(dt_rows_kernel, EW_GRID((size_t)P * J * R), 0, st, ts, dt, B, P, R, J, m.adaptive ? m.dtlog : nullptr, m.dtcnt, m.dtlog_cap);
  const bool batched = m.with_ode && m.ylog && m.yend && m.dtcnt;
  if (batched) {
    // Every step's starting state is in the forward's log: the stages of ALL steps of the window are rebuilt together, stage by
    // stage (S x (1 + layers) launches with P*J*R rows each, instead of that many per step), and the RNN cells of all intervals
    // layer by layer (their hidden inputs are the logged evolved states).
    hipLaunchKernelGGL(ylog_rows_kernel, EW_GRID(PJR * F), 0, st, m.ylog, m.yend, m.dtcnt, act[0], P, J, R, m.dtlog_cap, F);
    for (int s = 0; s < S; ++s) {
      if (s > 0)
        hipLaunchKernelGGL(stage_input_kernel, EW_GRID(PJR * F), 0, st, act[0] + (size_t)s * PJR * F, act[0], act[nl], kstride, arow[s], dt, (int)PJR, F);
      for (int l = 0; l < nl; ++l)
        gemm_nt(st, act[l] + (size_t)s * PJR * m.dims[l], m.dims[l], m.ode_w[l], m.dims[l], m.ode_b[l], act[l + 1] + (size_t)s * PJR * m.dims[l + 1],
                m.dims[l + 1], (int)PJR, m.dims[l + 1], m.dims[l], false, GEPI_ACT, l + 1 < nl ? m.act : 0);
    }
    hipLaunchKernelGGL(to_interval_major_kernel, EW_GRID(MB * F), 0, st, fused, rnn_in[0], B, P, F);
    for (int l = 0; l < L; ++l) {
      if (l > 0) rnn_in[l] = rnn_out[l - 1];     // (the layer below's outputs ARE this layer's inputs: no copy)
      hipLaunchKernelGGL(yend_rows_kernel, EW_GRID(MB * F), 0, st, m.yend, rnn_hp[l], l, B, P, F);
      gemm_nt(st, rnn_in[l], F, m.rnn_wih[l], F, m.rnn_bih[l], gi, GF, (int)MB, GF, F);
      gemm_nt(st, rnn_hp[l], F, m.rnn_whh[l], F, m.rnn_bhh[l], gh, GF, (int)MB, GF, F);
      if (m.gru) hipLaunchKernelGGL(gru_cell_kernel, EW_GRID(MB * F), 0, st, gi, gh, rnn_hp[l], rnn_gates[l], rnn_out[l], dout, (int)MB, F);
      else hipLaunchKernelGGL(rnn_cell_kernel, EW_GRID(MB * F), 0, st, gi, gh, rnn_out[l], dout, MB * F);   // (dout: scratch until the sweep)
    }
  } else {
    if (hc) hipLaunchKernelGGL(copy_kernel, EW_GRID(RF), 0, st, hc, Y, RF);
    else (void)hipMemsetAsync(Y, 0, RF * sizeof(float), st);
    for (int it = 0; it < P; ++it) {
      if (m.with_ode) {
        for (int j = 0; j < J; ++j) {
          const float* dtp = dt + ((size_t)it * J + j) * R;                   // this step's size, per row
          const float* kbase = act[nl] + row0(it, j, 0) * F;                  // K_0 of this step; K_s is kstride further
          for (int s = 0; s < S; ++s) {
            const size_t m0 = row0(it, j, s);
            hipLaunchKernelGGL(stage_input_kernel, EW_GRID(RF), 0, st, act[0] + m0 * F, Y, kbase, kstride, arow[s], dtp, R, F);
            for (int l = 0; l < nl; ++l) {
              float* o = act[l + 1] + m0 * m.dims[l + 1];
              gemm_nt(st, act[l] + m0 * m.dims[l], m.dims[l], m.ode_w[l], m.dims[l], m.ode_b[l], o, m.dims[l + 1], R, m.dims[l + 1], m.dims[l], false,
                      GEPI_ACT, l + 1 < nl ? m.act : 0);   // Linear + activation (the last one: Tanh, ODEFunc.py:13-14) in one launch
            }
          }
          // Y <- Y + dt * sum_s b_s K_s: the stage-input formula with the b row
          hipLaunchKernelGGL(stage_input_kernel, EW_GRID(RF), 0, st, Y, Y, kbase, kstride, brow, dtp, R, F);
        }
      }
      for (int l = 0; l < L; ++l) {
        float* in_l = rnn_in[l] + (size_t)it * BF;
        float* hp_l = rnn_hp[l] + (size_t)it * BF;
        float* out_l = rnn_out[l] + (size_t)it * BF;
        if (l == 0) hipLaunchKernelGGL(gather_interval_kernel, EW_GRID(BF), 0, st, fused, in_l, B, P, F, it);
        else hipLaunchKernelGGL(copy_kernel, EW_GRID(BF), 0, st, rnn_out[l - 1] + (size_t)it * BF, in_l, BF);
        hipLaunchKernelGGL(copy_kernel, EW_GRID(BF), 0, st, Y + (size_t)l * BF, hp_l, BF);
        gemm_nt(st, in_l, F, m.rnn_wih[l], F, m.rnn_bih[l], gi, GF, B, GF, F);
        gemm_nt(st, hp_l, F, m.rnn_whh[l], F, m.rnn_bhh[l], gh, GF, B, GF, F);
        if (m.gru)
          hipLaunchKernelGGL(gru_cell_kernel, EW_GRID(BF), 0, st, gi, gh, hp_l, rnn_gates[l] + (size_t)it * B * 4 * F, out_l, Y + (size_t)l * BF, B, F);
        else
          hipLaunchKernelGGL(rnn_cell_kernel, EW_GRID(BF), 0, st, gi, gh, out_l, Y + (size_t)l * BF, BF);
      }
    }
  }
  // regressor hidden layer on the top-layer outputs (interval-major rows)
  gemm_nt(st, rnn_out[L - 1], F, m.reg_w0, F, m.reg_b0, hid, 128, (int)MB, 128, F);
  hipLaunchKernelGGL(leaky_kernel, EW_GRID(MB * 128), 0, st, hid, MB * 128, 0.1f);

  // =============================== 2. reverse sweep ===============================
  hipLaunchKernelGGL(to_interval_major_kernel, EW_GRID(MB * 6), 0, st, grad_poses, dpo, B, P, 6);
  hipLaunchKernelGGL(reg2_bwd_kernel, EW_GRID(MB * 128), 0, st, dpo, m.reg_w2, hid, dhid, (int)MB);
  gemm_nt(st, dhid, 128, m.reg_w0_t, 128, nullptr, dout, F, (int)MB, F, 128);          // d out = dhid W0
  if (grad_hT) hipLaunchKernelGGL(copy_kernel, EW_GRID(RF), 0, st, grad_hT, lam, RF);
  else (void)hipMemsetAsync(lam, 0, RF * sizeof(float), st);
  // (steps the sweep skips - zero-length for every row of their interval - still have rows in the tape, and the weight gradients sum
  //  over ALL rows: their pre-activation gradients must read as the zeros the sweep would have written)
  if (m.with_ode && m.adj && m.steps_per_interval)
    for (int l = 0; l < nl; ++l) (void)hipMemsetAsync(delta[l], 0, M * m.dims[l + 1] * sizeof(float), st);
  for (int it = P - 1; it >= 0; --it) {
    // lam = dL/d(state after the RNN of interval it); the top layer's output also feeds the regressor (dout) and, below the top,
    // the layer above sent a gradient down its input (dinp): both enter the cell's backward as its second gradient
    for (int l = L - 1; l >= 0; --l) {
      float* d_l = rnn_delta[l] + (size_t)it * B * GF;
      float* dh_l = rnn_delta_h[l] + (size_t)it * B * GF;
      float* lam_l = lam + (size_t)l * BF;
      const float* g2 = l + 1 < L ? dinp : dout + (size_t)it * BF;
      // the cell's backward; the GRU leaves its direct path g * z in lam_l (in place), the product with W_hh is added to it
      if (m.gru)
        hipLaunchKernelGGL(gru_cell_bwd_kernel, EW_GRID(BF), 0, st, lam_l, g2, rnn_gates[l] + (size_t)it * B * 4 * F, rnn_hp[l] + (size_t)it * BF, d_l, dh_l,
                           lam_l, B, F);
      else
        hipLaunchKernelGGL(rnn_cell_bwd_kernel, EW_GRID(BF), 0, st, lam_l, g2, rnn_out[l] + (size_t)it * BF, d_l, BF);
      // d input = delta_i W_ih: the layer below's second gradient, or (bottom layer) the fused features' gradient, written in place
      if (l > 0) gemm_nt(st, d_l, GF, m.rnn_wih_t[l], GF, nullptr, dinp, F, B, F, GF);
      else if (grad_fused) gemm_nt(st, d_l, GF, m.rnn_wih_t[l], GF, nullptr, grad_fused + (size_t)it * F, P * F, B, F, GF);
      gemm_nt(st, dh_l, GF, m.rnn_whh_t[l], GF, nullptr, lam_l, F, B, F, GF, m.gru != 0);   // d hidden = delta_h W_hh (+ g * z) -> d evolved state
    }
    if (m.with_ode && m.adj) {
      // all J steps of the interval, every stage and layer, in one launch per chunk of rows (integrator_adj_kernel)
      IntegAdjArgs a = *m.adj;
      a.J = J; a.it = it; a.Rtot = R; a.stage_rows = PJR;
      a.Jrun = m.steps_per_interval ? std::max(0, std::min(J, m.steps_per_interval[it])) : J;
      if (a.Jrun == 0) continue;
      for (int l = 1; l <= nl; ++l) a.tape_act[l] = act[l];
      for (int l = 0; l < nl; ++l) a.tape_delta[l] = delta[l];
      a.dt = dt; a.lam = lam;
      const int e = launch_integrator_adj(a, L, B, st);
      if (e) return ODEVIO_ERR_HIP;
    } else if (m.with_ode) {
      for (int j = J - 1; j >= 0; --j) {
        const float* dtp = dt + ((size_t)it * J + j) * R;
        hipLaunchKernelGGL(step_adjoint_init_kernel, EW_GRID(RF), 0, st, lam, lamK, RF, brow, dtp, act[nl] + row0(it, j, S - 1) * F,
                           delta[nl - 1] + row0(it, j, S - 1) * F, R, F);
        for (int s = S - 1; s >= 0; --s) {
          const size_t m0 = row0(it, j, s);
          for (int l = nl - 1; l >= 1; --l) {
            // delta_{l-1} = (delta_l W_l) * act'(saved activation): one launch
            gemm_nt(st, delta[l] + m0 * m.dims[l + 1], m.dims[l + 1], m.ode_w_t[l], m.dims[l + 1], nullptr, delta[l - 1] + m0 * m.dims[l], m.dims[l], R,
                    m.dims[l], m.dims[l + 1], false, GEPI_DACT, m.act, act[l] + m0 * m.dims[l], m.dims[l]);
          }
          // gX = delta_0 W_0 with the stage's bookkeeping (and the next stage's first gradient) as the product's epilogue
          AdjEpi e;
          e.lam = lam; e.lamK = lamK; e.kstride = RF; e.a = arow[s]; e.dt = dtp;
          e.Kprev = s > 0 ? act[nl] + row0(it, j, s - 1) * F : nullptr;
          e.delta_prev = s > 0 ? delta[nl - 1] + row0(it, j, s - 1) * F : nullptr;
          if (m.dims[1] <= 512)
            hipLaunchKernelGGL(gemm_nt_adj_kernel<8>, dim3((F + 15) / 16, (R + 15) / 16), dim3(256), 0, st, delta[0] + m0 * m.dims[1], m.dims[1], m.ode_w_t[0],
                               m.dims[1], R, F, m.dims[1], e);
          else
            hipLaunchKernelGGL(gemm_nt_adj_kernel<16>, dim3((F + 15) / 16, (R + 15) / 16), dim3(256), 0, st, delta[0] + m0 * m.dims[1], m.dims[1], m.ode_w_t[0],
                               m.dims[1], R, F, m.dims[1], e);
        }
      }
    }
  }
  if (grad_hc) hipLaunchKernelGGL(copy_kernel, EW_GRID(RF), 0, st, lam, grad_hc, RF);

  // =============================== 3. weight gradients ===============================
  if (m.with_ode) {
    for (int l = 0; l < nl; ++l)
      wgrad_bias(st, delta[l], m.dims[l + 1], act[l], m.dims[l], g.ode_w[l], m.dims[l], g.ode_b[l], (int)M, m.dims[l + 1], m.dims[l]);
  }
  for (int l = 0; l < L; ++l) {
    wgrad_bias(st, rnn_delta[l], GF, rnn_in[l], F, g.rnn_wih[l], F, g.rnn_bih[l], (int)MB, GF, F);
    wgrad_bias(st, rnn_delta_h[l], GF, rnn_hp[l], F, g.rnn_whh[l], F, g.rnn_bhh[l], (int)MB, GF, F);
  }
  wgrad_bias(st, dhid, 128, rnn_out[L - 1], F, g.reg_w0, F, g.reg_b0, (int)MB, 128, F);
  wgrad_bias(st, dpo, 6, hid, 128, g.reg_w2, 128, g.reg_b2, (int)MB, 6, 128);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// FusionModule backward (FusionModule.py:17-23).  cat: the gradient is split.  soft: fused = c * (c W^T + b) with
// c = cat(fv, fi):  g_w = g * c;  g_c = g * w + g_w W;  g_W = g_w^T c;  g_b = column sums of g_w.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void cat_rows_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb, float* __restrict__ out, size_t rows) {
  const int F = na + nb;
  EW_LOOP(i, rows * F) {
    const size_t r = i / F;
    const int c = (int)(i - r * F);
    out[i] = c < na ? a[r * na + c] : b[r * nb + (c - na)];
  }
}
__global__ void split_rows_kernel(const float* __restrict__ in, float* __restrict__ a, int na, float* __restrict__ b, int nb, size_t rows) {
  const int F = na + nb;
  EW_LOOP(i, rows * F) {
    const size_t r = i / F;
    const int c = (int)(i - r * F);
    if (c < na) { if (a) a[r * na + c] = in[i]; }
    else if (b) b[r * nb + (c - na)] = in[i];
  }
}
// gw = g * c (in place over w's partner buffer), gc = g * w
__global__ void soft_gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ c, const float* __restrict__ w, float* __restrict__ gw,
                                     float* __restrict__ gc, size_t n) {
  EW_LOOP(i, n) {
    const float gi = g[i];
    gw[i] = gi * c[i];
    gc[i] = gi * w[i];
  }
}

size_t train_fuse_workspace_floats(int P, int F) { return (size_t)4 * P * F; }

int train_fuse_bwd(int soft, const float* W, const float* W_t, const float* bias, float* ws, const float* fv, int nv, const float* fi, int ni,
                   int P, const float* g_fused, float* g_fv, float* g_fi, float* g_W, float* g_b, hipStream_t st) {
  const int F = nv + ni;
  const size_t n = (size_t)P * F;
  if (!soft) {
    hipLaunchKernelGGL(split_rows_kernel, EW_GRID(n), 0, st, g_fused, g_fv, nv, g_fi, ni, (size_t)P);
    return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
  }
  float *c = ws, *w = ws + n, *gw = ws + 2 * n, *gc = ws + 3 * n;
  hipLaunchKernelGGL(cat_rows_kernel, EW_GRID(n), 0, st, fv, nv, fi, ni, c, (size_t)P);
  gemm_nt(st, c, F, W, F, bias, w, F, P, F, F);                       // w = c W^T + b
  hipLaunchKernelGGL(soft_gate_bwd_kernel, EW_GRID(n), 0, st, g_fused, c, w, gw, gc, n);
  gemm_nt(st, gw, F, W_t, F, nullptr, gc, F, P, F, F, true);           // gc += gw W   (W_t rows = columns of W)
  wgrad_bias(st, gw, F, c, F, g_W, F, g_b, P, F, F);                   // g_W[n][k] = sum_m gw[m][n] c[m][k];  g_b = column sums of gw
  hipLaunchKernelGGL(split_rows_kernel, EW_GRID(n), 0, st, gc, g_fv, nv, g_fi, ni, (size_t)P);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// Optimizer step of the reference's training loop (scripts/train_model.py:84-86, utils/utils.py:115-130):
// clip_grad_norm_(max_norm) over the gradients, then torch.optim.Adam(betas, eps, weight_decay = L2 added to the gradient).
// The clip coefficient stays on the device: no host synchronisation inside a training step.
// ---------------------------------------------------------------------------------------------------------------------
#define NORM_BLOCKS 64
// partial[t * NORM_BLOCKS + b] = sum of squares of block b's share of tensor t (fixed shares: deterministic)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)NORM_BLOCKS * 256) s += (double)g[i] * (double)g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// out[0] = total L2 norm, out[1] = min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int n_partial, float max_norm, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n_partial; i += 256) s += partial[i];   // fixed shares, fixed tree: deterministic
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    out[0] = norm;
    out[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
  }
}
// the same partial sums for up to OPT_TABLE_MAX tensors in one launch (blockIdx.y = tensor; entries' p = the gradient)
__global__ __launch_bounds__(256) void sumsq_multi_kernel(OptTable t, double* __restrict__ partial) {
  __shared__ double red[256];
  const OptEntry e = t.e[blockIdx.y];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < e.n; i += (size_t)NORM_BLOCKS * 256) s += (double)e.g[i] * (double)e.g[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)blockIdx.y * NORM_BLOCKS + blockIdx.x] = red[0];
}
int train_grad_clip(const float* const* grads, const size_t* numel, int n, float max_norm, double* partial_ws, float* out2, hipStream_t st) {
  for (int t0 = 0; t0 < n; t0 += OPT_TABLE_MAX) {
    OptTable t;
    t.n = std::min(OPT_TABLE_MAX, n - t0);
    for (int i = 0; i < t.n; ++i) { t.e[i].g = grads[t0 + i]; t.e[i].n = numel[t0 + i]; t.e[i].p = nullptr; t.e[i].s1 = t.e[i].s2 = nullptr; t.e[i].lr = 0.f; }
    hipLaunchKernelGGL(sumsq_multi_kernel, dim3(NORM_BLOCKS, t.n), dim3(256), 0, st, t, partial_ws + (size_t)t0 * NORM_BLOCKS);
  }
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, partial_ws, n * NORM_BLOCKS, max_norm, out2);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
size_t train_grad_clip_workspace_doubles(int n) { return (size_t)n * NORM_BLOCKS; }

// torch.optim.Adam, single tensor, the arithmetic of torch/optim/adam.py (_single_tensor_adam, amsgrad off, maximize off):
//   g = clip * grad + weight_decay * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),  bc_k = 1 - beta_k^step
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                            float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2, const float* __restrict__ clip2) {
  const float clip = clip2 ? clip2[1] : 1.0f;
  const float step_size = lr / bc1;
  EW_LOOP(i, n) {
    const float pi = p[i];
    float gi = g[i] * clip;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float m0 = m[i];
    const float mi = fmaf(gi - m0, 1.f - b1, m0);            // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}
int train_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, int step,
                    const float* clip2, hipStream_t st) {
  const float bc1 = 1.f - powf(b1, (float)step);
  const float sqrt_bc2 = sqrtf(1.f - powf(b2, (float)step));
  hipLaunchKernelGGL(adam_kernel, EW_GRID(n), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, sqrt_bc2, clip2);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// The same two updates over MANY tensors in one launch (blockIdx.y = tensor): a training step's twenty parameter tensors were twenty
// host calls and twenty launches of a few microseconds of work each.  Same arithmetic per element as adam_kernel / sgd_kernel.
__global__ void optimizer_multi_kernel(OptTable t, int kind, float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2, int first,
                                       const float* __restrict__ clip2) {
  const OptEntry e = t.e[blockIdx.y];
  const float clip = clip2 ? clip2[1] : 1.0f;
  if (kind == 0) {
    const float step_size = e.lr / bc1;
    EW_LOOP(i, e.n) {
      const float pi = e.p[i];
      float gi = e.g[i] * clip;
      if (wd != 0.f) gi = fmaf(wd, pi, gi);
      const float m0 = e.s1[i];
      const float mi = fmaf(gi - m0, 1.f - b1, m0);
      const float vi = fmaf(b2, e.s2[i], (1.f - b2) * gi * gi);
      e.s1[i] = mi;
      e.s2[i] = vi;
      const float denom = sqrtf(vi) / sqrt_bc2 + eps;
      e.p[i] = pi - step_size * (mi / denom);
    }
  } else {
    EW_LOOP(i, e.n) {
      const float pi = e.p[i];
      float gi = e.g[i] * clip;
      if (wd != 0.f) gi = fmaf(wd, pi, gi);
      float bi = gi;
      if (b1 != 0.f) {   // b1 = momentum
        bi = first ? gi : fmaf(b1, e.s1[i], gi);
        e.s1[i] = bi;
      }
      e.p[i] = pi - e.lr * bi;
    }
  }
}
int train_optimizer_multi(const OptTable& t, int kind, float b1, float b2, float eps, float wd, int step, const float* clip2, hipStream_t st) {
  const float bc1 = kind == 0 ? 1.f - powf(b1, (float)step) : 1.f;
  const float sqrt_bc2 = kind == 0 ? sqrtf(1.f - powf(b2, (float)step)) : 1.f;
  size_t most = 0;
  for (int i = 0; i < t.n; ++i) most = std::max(most, t.e[i].n);
  const unsigned bx = (unsigned)std::min<size_t>((most + 255) / 256, 256);
  hipLaunchKernelGGL(optimizer_multi_kernel, dim3(bx, t.n), dim3(256), 0, st, t, kind, b1, b2, eps, wd, bc1, sqrt_bc2, step <= 1 ? 1 : 0, clip2);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// torch.optim.SGD(momentum, dampening 0, no Nesterov, weight_decay wd), single tensor (torch/optim/sgd.py _single_tensor_sgd): the
// reference's other optimizer (utils/utils.py:120-121: SGD(param_groups, lr=1e-4, momentum=0.9); the groups' own lr overrides it)
//   g = clip * grad + wd * p;  buf = g (first step) | momentum buf + g;  p -= lr * buf
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n, float lr, float momentum,
                           float wd, int first, const float* __restrict__ clip2) {
  const float clip = clip2 ? clip2[1] : 1.0f;
  EW_LOOP(i, n) {
    const float pi = p[i];
    float gi = g[i] * clip;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    float bi = gi;
    if (momentum != 0.f) {
      bi = first ? gi : fmaf(momentum, buf[i], gi);
      buf[i] = bi;
    }
    p[i] = pi - lr * bi;
  }
}
int train_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int step, const float* clip2,
                   hipStream_t st) {
  hipLaunchKernelGGL(sgd_kernel, EW_GRID(n), 0, st, p, g, buf, n, lr, momentum, wd, step <= 1 ? 1 : 0, clip2);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// Parameter re-layout on the device (odevio_plan_update): the same index maps as the host code of odevio_plan_create
// (api.hip: transposed(), shard_columns(), the RNN's virtual matrix), so that an optimizer step needs no host round trip.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int K) {   // dst[k][n] = src[n][k]
  EW_LOOP(i, (size_t)N * K) {
    const int k = (int)(i / N), n = (int)(i - (size_t)k * N);
    dst[i] = src[(size_t)n * K + k];
  }
}
// [N][K] row-major -> member-major column shards with K padded to 256-wide chunks: member m, chunk j, column c, lane, 4
__global__ void shard_kernel(const float* __restrict__ W, float* __restrict__ out, int N, int K, int members) {
  const int NC = N / members, Kp = (K + 255) & ~255;
  EW_LOOP(i, (size_t)N * Kp) {
    const size_t per = (size_t)NC * Kp;
    const int m = (int)(i / per);
    const size_t r = i - (size_t)m * per;
    const int j = (int)(r / ((size_t)NC * 256));
    const int r2 = (int)(r - (size_t)j * NC * 256);
    const int col = r2 >> 8, k = j * 256 + (r2 & 255);
    out[i] = k < K ? W[(size_t)(m * NC + col) * K + k] : 0.f;
  }
}
// the RNN stack's virtual columns over K = [input | hidden] (tanh RNN: one per unit; GRU: r, z, n-input, n-hidden), sharded
__global__ void rnn_shard_kernel(const float* __restrict__ wih, const float* __restrict__ whh, float* __restrict__ out, int F, int gru,
                                 int members) {
  const int V = gru ? 4 : 1, NCF = F / members, NC = V * NCF, Fp = (F + 255) & ~255, Kp = 2 * Fp;
  EW_LOOP(i, (size_t)V * F * Kp) {
    const size_t per = (size_t)NC * Kp;
    const int m = (int)(i / per);
    const size_t r = i - (size_t)m * per;
    const int j = (int)(r / ((size_t)NC * 256));
    const int r2 = (int)(r - (size_t)j * NC * 256);
    const int col = r2 >> 8, kp = j * 256 + (r2 & 255);
    const int seg = kp / Fp, k = kp - seg * Fp;
    const int v = col / NCF, u = m * NCF + (col - v * NCF);
    float val = 0.f;
    if (k < F) {
      if (!gru) val = seg == 0 ? wih[(size_t)u * F + k] : whh[(size_t)u * F + k];
      else if (v < 2) val = seg == 0 ? wih[((size_t)v * F + u) * F + k] : whh[((size_t)v * F + u) * F + k];
      else if (v == 2) val = seg == 0 ? wih[((size_t)2 * F + u) * F + k] : 0.f;
      else val = seg == 0 ? 0.f : whh[((size_t)2 * F + u) * F + k];
    }
    out[i] = val;
  }
}
__global__ void rnn_bias_kernel(const float* __restrict__ bih, const float* __restrict__ bhh, float* __restrict__ vb, int F, int gru) {
  const int V = gru ? 4 : 1;
  EW_LOOP(i, (size_t)V * F) {
    const int v = (int)(i / F), u = (int)(i - (size_t)v * F);
    float x;
    if (!gru) x = bih[u] + bhh[u];
    else if (v < 2) x = bih[(size_t)v * F + u] + bhh[(size_t)v * F + u];
    else if (v == 2) x = bih[(size_t)2 * F + u];
    else x = bhh[(size_t)2 * F + u];
    vb[i] = x;
  }
}
// a device-to-device copy as a kernel: hipMemcpyAsync's copy-engine path costs ~14 us of stream time per call, whatever the size
void device_copy_f32(float* dst, const float* src, size_t n, hipStream_t st) { hipLaunchKernelGGL(copy_kernel, EW_GRID(n), 0, st, src, dst, n); }
void relayout_transpose(const float* src, float* dst, int N, int K, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, EW_GRID((size_t)N * K), 0, st, src, dst, N, K);
}
void relayout_shard(const float* W, float* out, int N, int K, int members, hipStream_t st) {
  hipLaunchKernelGGL(shard_kernel, EW_GRID((size_t)N * ((K + 255) & ~255)), 0, st, W, out, N, K, members);
}
void relayout_rnn(const float* wih, const float* whh, const float* bih, const float* bhh, float* out_w, float* out_b, int F, int gru, int members,
                  hipStream_t st) {
  const int V = gru ? 4 : 1;
  hipLaunchKernelGGL(rnn_shard_kernel, EW_GRID((size_t)V * F * 2 * ((F + 255) & ~255)), 0, st, wih, whh, out_w, F, gru, members);
  hipLaunchKernelGGL(rnn_bias_kernel, EW_GRID((size_t)V * F), 0, st, bih, bhh, out_b, F, gru);
}

// ---------------------------------------------------------------------------------------------------------------------
// InertialEncoder backward (Encoder.py:41-74; eval-mode BatchNorm like the forward kernel, imu_dropout = 0).
// Everything is a row matrix over rows = (frame pair, time step): a Conv1d(k3, p1) is X_col [rows][3 cin] x W^T, so the
// recomputed forward, the weight gradients (D^T X_col) and the input gradients (D W, then col2im) are the skinny GEMMs
// above; BatchNorm(eval) + conv bias are the folded pair (s, h) of the forward: y = leaky(s c + h), c = the bare convolution.
//   dz = g_y * leaky'(y);  dS = sum dz c;  dH = sum dz;  D = dz s
//   g_gamma = (dS + dH (b - mean)) rstd;  g_beta = dH;  g_bias = dH s;  g_W = D^T X_col;  g_Xcol = D W
// ---------------------------------------------------------------------------------------------------------------------
#define IMU_T 11
// x0 [rows][6] from the 100 Hz stream imu [B][T][6]: pair (b, p) covers samples 10p .. 10p+10
__global__ void imu_window_kernel(const float* __restrict__ imu, float* __restrict__ x0, int B, int T, int pps) {
  EW_LOOP(i, (size_t)B * pps * IMU_T * 6) {
    const int c = (int)(i % 6);
    const size_t r = i / 6;
    const int t = (int)(r % IMU_T);
    const size_t pair = r / IMU_T;
    const int b = (int)(pair / pps), p = (int)(pair - (size_t)b * pps);
    x0[i] = imu[((size_t)b * T + 10 * p + t) * 6 + c];
  }
}
// X_col[(pair,t)][ci*3 + k] = x[(pair, t + k - 1)][ci] (zero outside 0..10)
// (row stride ld >= 3 C, a multiple of 16 for the GEMM's k-steps: columns 3 C .. ld - 1 are zero)
__global__ void im2col3_kernel(const float* __restrict__ x, float* __restrict__ xcol, size_t rows, int C, int ld) {
  EW_LOOP(i, rows * ld) {
    const size_t r = i / ld;
    const int q = (int)(i - r * ld), ci = q / 3, k = q - 3 * ci;
    const int t = (int)(r % IMU_T) + k - 1;
    xcol[i] = (q < 3 * C && t >= 0 && t < IMU_T) ? x[(r + k - 1) * C + ci] : 0.f;
  }
}
// g_x[(pair,t')][ci] = sum_k g_xcol[(pair, t' - k + 1)][ci*3 + k]
__global__ void col2im3_kernel(const float* __restrict__ gcol, float* __restrict__ gx, size_t rows, int C) {
  EW_LOOP(i, rows * C) {
    const size_t r = i / C;
    const int ci = (int)(i - r * C), t = (int)(r % IMU_T);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ts = t - k + 1;
      if (ts >= 0 && ts < IMU_T) s += gcol[(r - k + 1) * 3 * C + ci * 3 + k];
    }
    gx[i] = s;
  }
}
__global__ void bn_leaky_fwd_kernel(const float* __restrict__ c, const float* __restrict__ s, const float* __restrict__ h, float* __restrict__ y,
                                    size_t rows, int C) {
  EW_LOOP(i, rows * C) {
    const int ch = (int)(i % C);
    const float v = fmaf(c[i], s[ch], h[ch]);
    y[i] = v > 0.f ? v : 0.1f * v;
  }
}
// dz = g * leaky'(y) -> dzc = dz * c (for dS), D = dz * s (over g in place allowed: outputs are separate buffers)
__global__ void bn_leaky_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ c, const float* __restrict__ s,
                                    float* __restrict__ dz, float* __restrict__ dzc, float* __restrict__ D, size_t rows, int C) {
  EW_LOOP(i, rows * C) {
    const int ch = (int)(i % C);
    const float d = g[i] * (y[i] > 0.f ? 1.f : 0.1f);
    dz[i] = d;
    dzc[i] = d * c[i];
    D[i] = d * s[ch];
  }
}
// BatchNorm / conv-bias gradients from the column sums dS, dH (eval mode: running statistics are constants)
__global__ void bn_param_grad_kernel(const float* __restrict__ dS, const float* __restrict__ dH, const float* __restrict__ s,
                                     const float* __restrict__ var, const float* __restrict__ mean, const float* __restrict__ bias, float eps,
                                     float* __restrict__ g_gamma, float* __restrict__ g_beta, float* __restrict__ g_bias, int C) {
  EW_LOOP(i, C) {
    const float rstd = 1.0f / sqrtf(var[i] + eps);
    if (g_gamma) g_gamma[i] = (dS[i] + dH[i] * (bias[i] - mean[i])) * rstd;
    if (g_beta) g_beta[i] = dH[i];
    if (g_bias) g_bias[i] = dH[i] * s[i];
  }
}
// [P][C][T] (the order proj sees) <-> rows [(P,t)][C]
__global__ void ct_to_rows_kernel(const float* __restrict__ in, float* __restrict__ out, size_t P, int C) {
  EW_LOOP(i, P * IMU_T * C) {
    const int c = (int)(i % C);
    const size_t r = i / C;
    const int t = (int)(r % IMU_T);
    const size_t p = r / IMU_T;
    out[i] = in[(p * C + c) * IMU_T + t];
  }
}
__global__ void rows_to_ct_kernel(const float* __restrict__ in, float* __restrict__ out, size_t P, int C) {
  EW_LOOP(i, P * IMU_T * C) {
    const int c = (int)(i % C);
    const size_t r = i / C;
    const int t = (int)(r % IMU_T);
    const size_t p = r / IMU_T;
    out[(p * C + c) * IMU_T + t] = in[i];
  }
}

size_t train_imu_workspace_floats(int P) {
  const size_t rows = (size_t)P * IMU_T;
  // x0..x3 and c1..c3 (6 + 2 * 448 per row), X_col (384), dz / dzc / D / g_y (4 * 256), g_xcol (384), flat + g_flat (2 * 256),
  // column sums, the transposed projection
  return rows * (6 + 2 * 448 + 384 + 4 * 256 + 384 + 2 * 256) + 1024 + (size_t)2816 * 256;
}

int train_imu_bwd(const ImuTrain& m, float* ws, const float* imu, int B, int T, const float* g_fi, float* g_imu_rows, const ImuGrads& g,
                  hipStream_t st) {
  const int pps = (T - 1) / 10, P = B * pps;
  const size_t rows = (size_t)P * IMU_T;
  const int C[4] = {6, 64, 128, 256};
  float* x[4];
  float* c[4] = {nullptr, nullptr, nullptr, nullptr};
  float* w = ws;
  for (int l = 0; l < 4; ++l) { x[l] = w; w += rows * C[l]; }
  for (int l = 1; l < 4; ++l) { c[l] = w; w += rows * C[l]; }
  float* xcol = w; w += rows * 384;
  float* dz = w; w += rows * 256;
  float* dzc = w; w += rows * 256;
  float* D = w; w += rows * 256;
  float* gy = w; w += rows * 256;
  float* gcol = w; w += rows * 384;
  float* flat = w; w += rows * 256;     // x3 per pair in (C,T) order: [P][2816], what proj sees
  float* gflat = w; w += rows * 256;
  float* dS = w; w += 256;
  float* dH = w; w += 256;
  w += 512;
  float* projT = w;                      // [2816][i_f_len]
  const int NF = m.i_f_len;
  // ---- forward, recomputed as GEMMs (X_col rows are padded to ldk = a multiple of 16 columns; the pad is zero on both sides)
  hipLaunchKernelGGL(imu_window_kernel, EW_GRID(rows * 6), 0, st, imu, x[0], B, T, pps);
  for (int l = 1; l < 4; ++l) {
    hipLaunchKernelGGL(im2col3_kernel, EW_GRID(rows * m.ldk[l - 1]), 0, st, x[l - 1], xcol, rows, C[l - 1], m.ldk[l - 1]);
    gemm_nt(st, xcol, m.ldk[l - 1], m.w[l - 1], m.ldk[l - 1], nullptr, c[l], C[l], (int)rows, C[l], m.ldk[l - 1]);
    hipLaunchKernelGGL(bn_leaky_fwd_kernel, EW_GRID(rows * C[l]), 0, st, c[l], m.s[l - 1], m.h[l - 1], x[l], rows, C[l]);
  }
  hipLaunchKernelGGL(rows_to_ct_kernel, EW_GRID(rows * 256), 0, st, x[3], flat, (size_t)P, 256);
  // ---- proj backward
  wgrad_bias(st, g_fi, NF, flat, 2816, g.proj_w, 2816, g.proj_b, P, NF, 2816);
  relayout_transpose(m.proj_w, projT, NF, 2816, st);                                   // [NF][2816] -> [2816][NF]
  gemm_nt(st, g_fi, NF, projT, NF, nullptr, gflat, 2816, P, 2816, NF);                  // g_flat = g_fi W
  hipLaunchKernelGGL(ct_to_rows_kernel, EW_GRID(rows * 256), 0, st, gflat, gy, (size_t)P, 256);
  // ---- the three conv blocks, last first
  for (int l = 3; l >= 1; --l) {
    const int Co = C[l], Ci = C[l - 1], ldk = m.ldk[l - 1];
    hipLaunchKernelGGL(bn_leaky_bwd_kernel, EW_GRID(rows * Co), 0, st, gy, x[l], c[l], m.s[l - 1], dz, dzc, D, rows, Co);
    launch_colsum(st, dzc, dS, (int)rows, Co);
    launch_colsum(st, dz, dH, (int)rows, Co);
    hipLaunchKernelGGL(bn_param_grad_kernel, EW_GRID(Co), 0, st, dS, dH, m.s[l - 1], m.var[l - 1], m.mean[l - 1], m.bias[l - 1], m.eps, g.gamma[l - 1],
                       g.beta[l - 1], g.b[l - 1], Co);
    hipLaunchKernelGGL(im2col3_kernel, EW_GRID(rows * ldk), 0, st, x[l - 1], xcol, rows, Ci, ldk);
    if (g.w[l - 1]) gemm_tn(st, D, Co, xcol, ldk, g.w[l - 1], 3 * Ci, (int)rows, Co, 3 * Ci);   // [Co][Ci][3]
    if (l > 1 || g_imu_rows) {
      gemm_nt(st, D, Co, m.wt[l - 1], Co, nullptr, gcol, 3 * Ci, (int)rows, 3 * Ci, Co);         // g_xcol = D W
      hipLaunchKernelGGL(col2im3_kernel, EW_GRID(rows * Ci), 0, st, gcol, l > 1 ? gy : g_imu_rows, rows, Ci);
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// InertialEncoder under model.train() (scripts/train_model.py:219; Encoder.py:43-57): BatchNorm1d with BATCH statistics over
// every (pair, time step) row of the batch, running statistics updated, Dropout(opt.imu_dropout) after every block.  Same row
// formulation as above; c = Conv1d(x) + bias this time (the bias cancels in the normalised value but not in running_mean).
//   xhat = (c - mean) invstd;  y = leaky(gamma xhat + beta);  x_next = y * mask / (1 - p)
// backward:  dz = g * mask / (1 - p) * leaky'(y);  g_beta = sum dz;  g_gamma = sum dz xhat;
//            D = gamma invstd (dz - g_beta / N - xhat g_gamma / N)   (torch's batch_norm backward with training=True)
// The dropout element index is the reference tensor's own order [pair][channel][time]: e = (pair * C + c) * 11 + t.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void bn_train_rows_fwd_kernel(const float* __restrict__ c, const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, DropoutSpec drop, float* __restrict__ x,
                                         size_t rows, int C) {
  EW_LOOP(i, rows * C) {
    const int ch = (int)(i % C);
    const size_t r = i / C;
    const float xh = (c[i] - mean[ch]) * invstd[ch];
    float y = fmaf(gamma[ch], xh, beta[ch]);
    y = y > 0.f ? y : 0.1f * y;
    const unsigned long long e = ((unsigned long long)(r / IMU_T) * C + ch) * IMU_T + (unsigned long long)(r % IMU_T);
    x[i] = y * dropout_factor(drop, e);
  }
}
__global__ void bn_train_rows_bwd1_kernel(const float* __restrict__ g, const float* __restrict__ c, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          DropoutSpec drop, float* __restrict__ dz, float* __restrict__ dzx, size_t rows, int C) {
  EW_LOOP(i, rows * C) {
    const int ch = (int)(i % C);
    const size_t r = i / C;
    const float xh = (c[i] - mean[ch]) * invstd[ch];
    const float y = fmaf(gamma[ch], xh, beta[ch]);
    const unsigned long long e = ((unsigned long long)(r / IMU_T) * C + ch) * IMU_T + (unsigned long long)(r % IMU_T);
    const float d = g[i] * dropout_factor(drop, e) * (y > 0.f ? 1.f : 0.1f);
    dz[i] = d;
    dzx[i] = d * xh;
  }
}
__global__ void bn_train_rows_bwd2_kernel(const float* __restrict__ dz, const float* __restrict__ c, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ dH,
                                          const float* __restrict__ dS, float* __restrict__ D, size_t rows, int C) {
  const float invn = 1.0f / (float)rows;
  EW_LOOP(i, rows * C) {
    const int ch = (int)(i % C);
    const float xh = (c[i] - mean[ch]) * invstd[ch];
    D[i] = gamma[ch] * invstd[ch] * (dz[i] - dH[ch] * invn - xh * dS[ch] * invn);
  }
}

size_t train_imu_train_workspace_floats(int P) { return train_imu_workspace_floats(P) + 2048; }

// the recomputed train-mode forward shared by both entry points; stats[l] = {mean, invstd} of block l in ws
static int imu_train_forward(const ImuTrain& m, const ImuTrainMode& tm, bool update_running, float* const (&x)[4], float* const (&c)[4],
                             float* xcol, float* const (&mean)[3], float* const (&invstd)[3], const float* imu, int B, int T, hipStream_t st) {
  const int pps = (T - 1) / 10, P = B * pps;
  const size_t rows = (size_t)P * IMU_T;
  const int C[4] = {6, 64, 128, 256};
  hipLaunchKernelGGL(imu_window_kernel, EW_GRID(rows * 6), 0, st, imu, x[0], B, T, pps);
  for (int l = 1; l < 4; ++l) {
    hipLaunchKernelGGL(im2col3_kernel, EW_GRID(rows * m.ldk[l - 1]), 0, st, x[l - 1], xcol, rows, C[l - 1], m.ldk[l - 1]);
    gemm_nt(st, xcol, m.ldk[l - 1], m.w[l - 1], m.ldk[l - 1], m.bias[l - 1], c[l], C[l], (int)rows, C[l], m.ldk[l - 1]);
    if (bn_stats_rows(c[l], rows, C[l], m.eps, tm.momentum, update_running ? tm.run_mean[l - 1] : nullptr,
                      update_running ? tm.run_var[l - 1] : nullptr, mean[l - 1], invstd[l - 1], st) != hipSuccess)
      return ODEVIO_ERR_HIP;
    hipLaunchKernelGGL(bn_train_rows_fwd_kernel, EW_GRID(rows * C[l]), 0, st, c[l], mean[l - 1], invstd[l - 1], tm.gamma[l - 1], tm.beta[l - 1],
                       tm.drop[l - 1], x[l], rows, C[l]);
  }
  return 0;
}

#define IMU_TRAIN_CARVE()                                                   \
  const int pps = (T - 1) / 10, P = B * pps;                                \
  const size_t rows = (size_t)P * IMU_T;                                    \
  const int C[4] = {6, 64, 128, 256};                                       \
  float* x[4];                                                              \
  float* c[4] = {nullptr, nullptr, nullptr, nullptr};                       \
  float* w = ws;                                                            \
  for (int l = 0; l < 4; ++l) { x[l] = w; w += rows * C[l]; }               \
  for (int l = 1; l < 4; ++l) { c[l] = w; w += rows * C[l]; }               \
  float* xcol = w; w += rows * 384;                                         \
  float* dz = w; w += rows * 256;                                           \
  float* dzc = w; w += rows * 256;                                          \
  float* D = w; w += rows * 256;                                            \
  float* gy = w; w += rows * 256;                                           \
  float* gcol = w; w += rows * 384;                                         \
  float* flat = w; w += rows * 256;                                         \
  float* gflat = w; w += rows * 256;                                        \
  float* dS = w; w += 256;                                                  \
  float* dH = w; w += 256;                                                  \
  w += 512;                                                                 \
  float* projT = w; w += (size_t)2816 * 256;                                \
  float* mean[3]; float* invstd[3];                                         \
  for (int l = 0; l < 3; ++l) { mean[l] = w; w += 256; invstd[l] = w; w += 256; }

int train_imu_fwd_train(const ImuTrain& m, const ImuTrainMode& tm, float* ws, const float* imu, int B, int T, const float* proj_b, float* fi,
                        int ld_fi, hipStream_t st) {
  IMU_TRAIN_CARVE();
  (void)dz; (void)dzc; (void)D; (void)gy; (void)gcol; (void)gflat; (void)dS; (void)dH; (void)projT; (void)C;
  const int rc = imu_train_forward(m, tm, true, x, c, xcol, mean, invstd, imu, B, T, st);
  if (rc) return rc;
  hipLaunchKernelGGL(rows_to_ct_kernel, EW_GRID(rows * 256), 0, st, x[3], flat, (size_t)P, 256);
  gemm_nt(st, flat, 2816, m.proj_w, 2816, proj_b, fi, ld_fi, P, m.i_f_len, 2816);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

int train_imu_bwd_train(const ImuTrain& m, const ImuTrainMode& tm, float* ws, const float* imu, int B, int T, const float* g_fi, const ImuGrads& g,
                        hipStream_t st) {
  IMU_TRAIN_CARVE();
  const int NF = m.i_f_len;
  int rc = imu_train_forward(m, tm, false, x, c, xcol, mean, invstd, imu, B, T, st);   // same masks, same statistics; running stats untouched
  if (rc) return rc;
  hipLaunchKernelGGL(rows_to_ct_kernel, EW_GRID(rows * 256), 0, st, x[3], flat, (size_t)P, 256);
  wgrad_bias(st, g_fi, NF, flat, 2816, g.proj_w, 2816, g.proj_b, P, NF, 2816);
  relayout_transpose(m.proj_w, projT, NF, 2816, st);
  gemm_nt(st, g_fi, NF, projT, NF, nullptr, gflat, 2816, P, 2816, NF);
  hipLaunchKernelGGL(ct_to_rows_kernel, EW_GRID(rows * 256), 0, st, gflat, gy, (size_t)P, 256);
  for (int l = 3; l >= 1; --l) {
    const int Co = C[l], Ci = C[l - 1], ldk = m.ldk[l - 1];
    hipLaunchKernelGGL(bn_train_rows_bwd1_kernel, EW_GRID(rows * Co), 0, st, gy, c[l], mean[l - 1], invstd[l - 1], tm.gamma[l - 1], tm.beta[l - 1],
                       tm.drop[l - 1], dz, dzc, rows, Co);
    launch_colsum(st, dz, dH, (int)rows, Co);
    launch_colsum(st, dzc, dS, (int)rows, Co);
    hipLaunchKernelGGL(bn_train_rows_bwd2_kernel, EW_GRID(rows * Co), 0, st, dz, c[l], mean[l - 1], invstd[l - 1], tm.gamma[l - 1], dH, dS, D, rows, Co);
    if (g.gamma[l - 1]) hipLaunchKernelGGL(copy_kernel, EW_GRID(Co), 0, st, dS, g.gamma[l - 1], (size_t)Co);
    if (g.beta[l - 1]) hipLaunchKernelGGL(copy_kernel, EW_GRID(Co), 0, st, dH, g.beta[l - 1], (size_t)Co);
    if (g.b[l - 1]) launch_colsum(st, D, g.b[l - 1], (int)rows, Co);
    hipLaunchKernelGGL(im2col3_kernel, EW_GRID(rows * ldk), 0, st, x[l - 1], xcol, rows, Ci, ldk);
    if (g.w[l - 1]) gemm_tn(st, D, Co, xcol, ldk, g.w[l - 1], 3 * Ci, (int)rows, Co, 3 * Ci);
    if (l > 1) {
      gemm_nt(st, D, Co, m.wt[l - 1], Co, nullptr, gcol, 3 * Ci, (int)rows, 3 * Ci, Co);
      hipLaunchKernelGGL(col2im3_kernel, EW_GRID(rows * Ci), 0, st, gcol, gy, rows, Ci);
    }
  }
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}

// out [M][ldo] = A [M][lda] W^T + bias on the skinny fp32-MFMA GEMM above (K % 16 == 0): for forward-path projections whose
// M is a few hundred rows (the InertialEncoder's proj: 160 x 2816 -> 256)
void skinny_linear(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N, int K, hipStream_t st) {
  gemm_nt(st, A, lda, W, ldw, bias, out, ldo, M, N, K);
}
void skinny_tn(const float* D, int ldd, const float* A, int lda, float* out, int ldo, int M, int N, int K, hipStream_t st, int accumulate) {
  gemm_tn(st, D, ldd, A, lda, out, ldo, M, N, K, accumulate != 0);
}
void skinny_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N, int K, int accumulate,
               int epi, int act, const float* aux, int ldaux, hipStream_t st) {
  gemm_nt(st, A, lda, W, ldw, bias, out, ldo, M, N, K, accumulate != 0, epi, act, aux, ldaux);
}
// The TN product with the contraction (M rows) split over `splits` workgroup layers: partial [splits][N][K] slabs, combined in slab
// order (deterministic).  For contractions over millions of rows with a 16-wide output (the Neural-CDE last layer's adjoint).
__global__ __launch_bounds__(256) void gemm_tn_split_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ A, int lda,
                                                            float* __restrict__ partial, int M, int N, int K, int rows_per_split) {
  __shared__ float red[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int k0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int nn = min(n0 + r, N - 1), kk = min(k0 + r, K - 1);
  const int m_begin = blockIdx.z * rows_per_split, m_end = min(M, m_begin + rows_per_split);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int mb = m_begin + 16 * wave; mb < m_end; mb += 64) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mb + 4 * q + j;
      const float dv = m < m_end ? D[(size_t)m * ldd + nn] : 0.f;
      const float av = m < m_end ? A[(size_t)m * lda + kk] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv, av, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][4 * q + e][r] = acc[e];
  __syncthreads();
  const int nl = tid >> 4, kl = tid & 15;
  const int n = n0 + nl, k = k0 + kl;
  if (n < N && k < K)
    partial[((size_t)blockIdx.z * N + n) * K + k] = (red[0][nl][kl] + red[1][nl][kl]) + (red[2][nl][kl] + red[3][nl][kl]);
}
__global__ void split_combine_kernel(const float* __restrict__ partial, float* __restrict__ out, int ldo, int N, int K, int splits, int accumulate) {
  EW_LOOP(i, (size_t)N * K) {
    const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += partial[(size_t)z * N * K + i];
    float* o = out + (size_t)n * ldo + k;
    *o = accumulate ? *o + s : s;
  }
}
void skinny_tn_split(const float* D, int ldd, const float* A, int lda, float* out, int ldo, int M, int N, int K, float* partial, int splits,
                     int accumulate, hipStream_t st) {
  const int rows = ((M + splits - 1) / splits + 15) / 16 * 16;
  splits = (M + rows - 1) / rows;
  hipLaunchKernelGGL(gemm_tn_split_kernel, dim3((K + 15) / 16, (N + 15) / 16, splits), dim3(256), 0, st, D, ldd, A, lda, partial, M, N, K, rows);
  hipLaunchKernelGGL(split_combine_kernel, EW_GRID((size_t)N * K), 0, st, partial, out, ldo, N, K, splits, accumulate);
}
void colsum_rows(const float* x, float* out, int M, int N, hipStream_t st) { launch_colsum(st, x, out, M, N); }
// regressor Linear(F,128) + LeakyReLU(0.1) + Linear(128,6) (PoseODERNN.py:64-68 / PoseCDE.py:68-72) backward from g_poses [M][6]:
// g_seq [M][F] and the four parameter gradients (null = not wanted).  ws: 2 * M * 128 floats.
int train_regressor_bwd(const float* seq, int F, const float* w0, const float* w0_t, const float* b0, const float* w2, const float* g_poses, int M,
                        float* ws, float* g_seq, float* g_w0, float* g_b0, float* g_w2, float* g_b2, hipStream_t st) {
  float *hid = ws, *dhid = ws + (size_t)M * 128;
  gemm_nt(st, seq, F, w0, F, b0, hid, 128, M, 128, F);
  hipLaunchKernelGGL(leaky_kernel, EW_GRID((size_t)M * 128), 0, st, hid, (size_t)M * 128, 0.1f);
  hipLaunchKernelGGL(reg2_bwd_kernel, EW_GRID((size_t)M * 128), 0, st, g_poses, w2, hid, dhid, M);
  if (g_seq) gemm_nt(st, dhid, 128, w0_t, 128, nullptr, g_seq, F, M, F, 128);
  wgrad_bias(st, dhid, 128, seq, F, g_w0, F, g_b0, M, 128, F);
  wgrad_bias(st, g_poses, 6, hid, 128, g_w2, 128, g_b2, M, 6, 128);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
void leaky_inplace(float* x, size_t n, float slope, hipStream_t st) { hipLaunchKernelGGL(leaky_kernel, EW_GRID(n), 0, st, x, n, slope); }
__global__ void mul_kernel(float* x, const float* y, size_t n) { EW_LOOP(i, n) x[i] *= y[i]; }
void mul_inplace(float* x, const float* y, size_t n, hipStream_t st) { hipLaunchKernelGGL(mul_kernel, EW_GRID(n), 0, st, x, y, n); }

// FusionModule "hard" backward: logits = cat W^T + b recomputed, the element-wise part in pose.hip (same Philox block as the forward),
// then g_W = g_logits^T cat, g_b = column sums, g_cat += g_logits W.
int train_fuse_hard_bwd(const float* W, const float* W_t, const float* bias, float* cat, float* logits, float* g_logits, float* g_cat,
                        const float* fv, int nv, const float* fi, int ni, int P, unsigned long long seed, unsigned long long call,
                        const float* g_fused, float* g_fv, float* g_fi, float* g_W, float* g_b, hipStream_t st) {
  const int F = nv + ni;
  const size_t n = (size_t)P * F;
  hipLaunchKernelGGL(cat_rows_kernel, EW_GRID(n), 0, st, fv, nv, fi, ni, cat, (size_t)P);
  gemm_nt(st, cat, F, W, F, bias, logits, 2 * F, P, 2 * F, F);
  launch_hard_mask_bwd(g_fused, cat, logits, g_cat, g_logits, n, seed, call, st);
  gemm_nt(st, g_logits, 2 * F, W_t, 2 * F, nullptr, g_cat, F, P, F, 2 * F, true);   // g_cat += g_logits W
  wgrad_bias(st, g_logits, 2 * F, cat, F, g_W, F, g_b, P, 2 * F, F);
  hipLaunchKernelGGL(split_rows_kernel, EW_GRID(n), 0, st, g_cat, g_fv, nv, g_fi, ni, (size_t)P);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
