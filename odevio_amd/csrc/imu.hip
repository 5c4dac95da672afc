// InertialEncoder convolution stack (reference src/models/Encoder.py:43-56,60-72), one workgroup
// per frame pair: window the 100 Hz IMU stream (11 samples, stride 10), then three
// Conv1d(k3,p1)+BatchNorm1d(eval)+LeakyReLU(0.1) layers 6->64->128->256 entirely in LDS, and write
// the [256][11] result flattened in (C,T) order, the layout the reference's `x.view(N, -1)` feeds to
// `proj`.  Weights are pre-transposed to [(ci,k)][co] so a wave reads them coalesced; activations are
// LDS broadcasts.  4.2 MFLOP per pair: latency, not bandwidth.
#include "common.h"

#define IMU_T 11
#define IMU_TP 13  // padded time axis (one zero on each side)

template <int CIN, int COUT, int NT>
__device__ __forceinline__ void conv1d_layer(const float* __restrict__ xin, float* __restrict__ xout,
                                             const float* __restrict__ wt, const float* __restrict__ sc,
                                             const float* __restrict__ sh, int co, int t0, int tstep,
                                             float* gout) {
  // this thread: output channel co, time steps t0, t0+tstep, ... (NT of them, those < 11 are real)
  float acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = 0.f;
  for (int ci = 0; ci < CIN; ++ci) {
    const float w0 = wt[(ci * 3 + 0) * COUT + co];
    const float w1 = wt[(ci * 3 + 1) * COUT + co];
    const float w2 = wt[(ci * 3 + 2) * COUT + co];
    const float* xr = xin + ci * IMU_TP;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = t0 + j * tstep;  // padded index t..t+2 covers taps t-1..t+1
      if (t < IMU_T) acc[j] = fmaf(w2, xr[t + 2], fmaf(w1, xr[t + 1], fmaf(w0, xr[t], acc[j])));
    }
  }
  const float s = sc[co], h = sh[co];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + j * tstep;
    if (t < IMU_T) {
      float v = acc[j] * s + h;
      v = v > 0.f ? v : 0.1f * v;
      if (gout) gout[co * IMU_T + t] = v;
      else xout[co * IMU_TP + t + 1] = v;
    }
  }
}

__global__ __launch_bounds__(256) void imu_convs_kernel(ImuArgs a) {
  __shared__ float x0[6 * IMU_TP];
  __shared__ float x1[64 * IMU_TP];
  __shared__ float x2[128 * IMU_TP];
  const int tid = threadIdx.x;
  const int pair = blockIdx.x;
  const int b = pair / a.pairs_per_seq, p = pair - b * a.pairs_per_seq;
  const float* src = a.imu + ((size_t)b * a.T + 10 * p) * 6;  // window rows 10p .. 10p+10, [t][6]
  for (int i = tid; i < 6 * IMU_TP; i += 256) x0[i] = 0.f;
  for (int i = tid; i < 64 * IMU_TP; i += 256) x1[i] = 0.f;
  for (int i = tid; i < 128 * IMU_TP; i += 256) x2[i] = 0.f;
  __syncthreads();
  if (tid < 66) {
    const int t = tid / 6, c = tid - 6 * t;
    x0[c * IMU_TP + t + 1] = src[t * 6 + c];  // permute(0,2,1): channels-first
  }
  __syncthreads();
  conv1d_layer<6, 64, 3>(x0, x1, a.w1t, a.s1, a.h1, tid & 63, tid >> 6, 4, nullptr);
  __syncthreads();
  conv1d_layer<64, 128, 6>(x1, x2, a.w2t, a.s2, a.h2, tid & 127, tid >> 7, 2, nullptr);
  __syncthreads();
  conv1d_layer<128, 256, 11>(x2, nullptr, a.w3t, a.s3, a.h3, tid, 0, 1, a.out + (size_t)pair * 256 * IMU_T);
}

void launch_imu_convs(const ImuArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(imu_convs_kernel, dim3(a.B * a.pairs_per_seq), dim3(256), 0, st, a);
}
