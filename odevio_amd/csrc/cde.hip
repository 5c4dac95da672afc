// Neural-CDE pose path (reference src/models/PoseCDE.py:76-103, CDEFunc in src/models/ODEFunc.py:44-83;
// torchcde 0.2.5 -> torchdiffeq 0.2.3 in the reference).  First, correctness-oriented version:
// the solver loop is host-driven (as torchdiffeq's own Python loop is) and every piece of arithmetic is
// a small HIP kernel; the shared-step controller reads one scalar back per step.
//
//   f(t, z) = reshape(tanh(W_L act(... act(W_1 z + b_1)) + b_L), [B, H, C]) . dX/dt(t),   C = H + 1
//
// The last Linear has H*C outputs (2.1 M parameters at H = 128, 1.08 G at H = 1024) and is the only
// part with real traffic: `cde_last_kernel` streams each group of C weight rows once for the whole
// batch and fuses bias + tanh + the contraction with dX/dt, so the [B, H, C] tensor never exists.
#include "common.h"
#include "cde.h"

__device__ __forceinline__ float cde_act(float v, int act) {
  switch (act) {
    case 0: return tanhf(v);
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.01f * v;
    case 3: return v > 20.f ? v : log1pf(expf(v));
    default: return v;
  }
}

// out[b][n] = act(sum_k x[b][k] W[n][k] + bias[n]);  one wave per output column, lanes stride K.
__global__ __launch_bounds__(256) void cde_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int B, int K, int ldx, int N, int act) {
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

// dX/dt on linear piece `seg` of the rectilinear path built from obs [B][L][C] (channel 0 = time):
//   even piece 2i:  time moves from tau_i to tau_{i+1}, the other channels rest;
//   odd piece 2i+1: time rests, the channels jump from x_i to x_{i+1}.
__global__ void cde_control_grad_kernel(const float* __restrict__ obs, float* __restrict__ g, int B, int L, int C, int seg) {
  const int i = seg >> 1;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < B * C; idx += gridDim.x * blockDim.x) {
    const int b = idx / C, c = idx - b * C;
    const float* o0 = obs + ((size_t)b * L + i) * C;
    const float* o1 = o0 + C;
    float v;
    if ((seg & 1) == 0) v = (c == 0) ? o1[0] - o0[0] : 0.f;
    else v = (c == 0) ? 0.f : o1[c] - o0[c];
    g[idx] = v;
  }
}

// out[b][h] = sum_c tanh(sum_k W[h*C + c][k] x[b][k] + bias[h*C + c]) * g[b][c]
// One workgroup per h streams that h's C weight rows (C*H floats, contiguous) ONCE per 16 batch rows - the layer is a
// pure weight stream (4.3 GB at H = 1024 against 34 GFLOP for B = 16).  The products run on the fp32 MFMA
// (v_mfma_f32_16x16x4_f32, an exact fmaf chain): a wave takes 16 weight rows (MFMA rows) x 16 batch rows (MFMA
// columns).  Lane (r = lane&15, q = lane>>4) loads W[row r][16 i + 4 q .. +3] as one float4 - 64 contiguous bytes per
// row and instruction, eight of them in flight per lane - and feeds element j to MFMA j of the group, whose k-set is
// {4 q' + j}: any bijection of k works as long as the x fragment (x[b][16 i + 4 q + j], from LDS, row stride H + 8
// floats = conflict-free b128 reads) uses the same one.  Bias + tanh + the contraction with dX/dt are fused, so the
// [B, H, C] tensor never exists.
__global__ __launch_bounds__(256) void cde_last_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ g,
                                                       float* __restrict__ out, int B, int H, int C) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [CDE_BT][H + 8] then [4 waves][CDE_BT] reduction scratch
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
    float part = 0.f;                       // this lane's share of out[b0 + r][h]
    const float* xrow = xs + r * ldx + 4 * q;
    for (int blk = wave; blk < nblk; blk += 4) {
      const int c_ld = min(blk * 16 + r, C - 1);           // rows past C re-read the last row; masked below
      const float* wrow = Wh + (size_t)c_ld * H + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
      for (int k = 0; k < H; k += 16) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + k);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xrow + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], xv[j], acc, 0, 0, 0);
      }
      // D: column (= batch row) = lane&15, row (= weight row inside the block) = 4*(lane>>4) + reg
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = blk * 16 + 4 * q + e;
        if (c < C && r < nb) part = fmaf(tanhf(acc[e] + bh[c]), g[(size_t)(b0 + r) * C + c], part);
      }
    }
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lane < 16) red[wave * CDE_BT + lane] = part;
    __syncthreads();
    if (tid < nb) out[(size_t)(b0 + tid) * H + h] = (red[tid] + red[CDE_BT + tid]) + (red[2 * CDE_BT + tid] + red[3 * CDE_BT + tid]);
  }
}

// out = y + sum_j coef[j] * k_j   (k_j = kbase + j*n)
__global__ void cde_combine_kernel(const float* __restrict__ y, const float* __restrict__ kbase, CdeCoefs cf, int nk,
                                   float* __restrict__ out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float acc = 0.f;
    bool first = true;
    for (int j = 0; j < nk; ++j) {
      if (cf.c[j] == 0.f) continue;
      const float term = kbase[(size_t)j * n + i] * cf.c[j];
      acc = first ? term : acc + term;
      first = false;
    }
    out[i] = (y ? y[i] : 0.f) + acc;
  }
}

// scalar[slot] = sqrt(mean((num / (atol + rtol * ref))^2)) with
//   mode 0: num = a,       ref = |y0|                  (initial step: d0, d1)
//   mode 1: num = a - b,   ref = |y0|                  (initial step: d2)
//   mode 2: num = a,       ref = max(|y0|, |y1|)       (error ratio; a = error estimate)
__global__ __launch_bounds__(1024) void cde_rms_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ y0, const float* __restrict__ y1,
                                                       float atol, float rtol, int mode, int n, float* __restrict__ scalar,
                                                       int slot) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float num = mode == 1 ? a[i] - b[i] : a[i];
    const float ref = mode == 2 ? fmaxf(fabsf(y0[i]), fabsf(y1[i])) : fabsf(y0[i]);
    const float z = num / (atol + rtol * ref);
    s = fmaf(z, z, s);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    scalar[slot] = sqrtf(t / (float)n);
  }
}

// dense-output polynomial of an accepted step (torchdiffeq _interp_fit): coeffs [5][n] = e, d, c, b, a
__global__ void cde_interp_fit_kernel(const float* __restrict__ y0, const float* __restrict__ y1, const float* __restrict__ ymid,
                                      const float* __restrict__ f0, const float* __restrict__ f1, float dt,
                                      float* __restrict__ co, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float a0 = y0[i], a1 = y1[i], am = ymid[i], g0 = f0[i], g1 = f1[i];
    co[i] = a0;
    co[n + i] = dt * g0;
    co[2 * n + i] = dt * (g1 - 4.f * g0) - 11.f * a0 - 5.f * a1 + 16.f * am;
    co[3 * n + i] = dt * (5.f * g0 - 3.f * g1) + 18.f * a0 + 14.f * a1 - 32.f * am;
    co[4 * n + i] = 2.f * dt * (g1 - g0) - 8.f * (a1 + a0) + 16.f * am;
  }
}

// sol[b][p][:] = polynomial at x (or a plain copy of `src` when co == nullptr); n = B*H elements, row = b
__global__ void cde_emit_kernel(const float* __restrict__ co, const float* __restrict__ src, float x, float* __restrict__ sol,
                                int B, int H, int P, int p) {
  const int n = B * H;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float v;
    if (co) {
      v = co[i] + x * co[n + i];
      float xp = x;
#pragma unroll
      for (int j = 2; j < 5; ++j) {
        xp = xp * x;
        v = v + xp * co[j * n + i];
      }
    } else {
      v = src[i];
    }
    const int b = i / H, hh = i - b * H;
    sol[((size_t)b * P + p) * H + hh] = v;
  }
}

void cde_launch_linear(const float* x, int ldx, const float* W, const float* bias, float* out, int B, int K, int N, int act, hipStream_t st) {
  hipLaunchKernelGGL(cde_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, st, x, W, bias, out, B, K, ldx, N, act);
}
void cde_launch_control_grad(const float* obs, float* g, int B, int L, int C, int seg, hipStream_t st) {
  hipLaunchKernelGGL(cde_control_grad_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, obs, g, B, L, C, seg);
}
void cde_launch_last(const float* x, const float* W, const float* bias, const float* g, float* out, int B, int H, int C, hipStream_t st) {
  const size_t lds = ((size_t)CDE_BT * (H + 8) + CDE_BT * 4) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cde_last_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(cde_last_kernel, dim3(H), dim3(256), lds, st, x, W, bias, g, out, B, H, C);
}
void cde_launch_combine(const float* y, const float* kbase, const CdeCoefs& cf, int nk, float* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(cde_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, y, kbase, cf, nk, out, n);
}
void cde_launch_rms(const float* a, const float* b, const float* y0, const float* y1, float atol, float rtol, int mode, int n,
                    float* scalar, int slot, hipStream_t st) {
  hipLaunchKernelGGL(cde_rms_kernel, dim3(1), dim3(1024), 0, st, a, b, y0, y1, atol, rtol, mode, n, scalar, slot);
}
void cde_launch_interp_fit(const float* y0, const float* y1, const float* ymid, const float* f0, const float* f1, float dt,
                           float* co, int n, hipStream_t st) {
  hipLaunchKernelGGL(cde_interp_fit_kernel, dim3((n + 255) / 256), dim3(256), 0, st, y0, y1, ymid, f0, f1, dt, co, n);
}
void cde_launch_emit(const float* co, const float* src, float x, float* sol, int B, int H, int P, int p, hipStream_t st) {
  hipLaunchKernelGGL(cde_emit_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, co, src, x, sol, B, H, P, p);
}
