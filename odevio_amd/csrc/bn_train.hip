// Train-mode BatchNorm2d / BatchNorm1d statistics and the BatchNorm + LeakyReLU + Dropout pass of the image encoder, gfx950.
//
// Replaces, for model.train() (scripts/train_model.py:219), what conv()'s nn.BatchNorm2d + nn.LeakyReLU(0.1) + nn.Dropout(p) do
// after every convolution of the reference's ImageEncoder (src/models/Encoder.py:8-22,82-90): the convolution kernels run with an
// identity epilogue (z = conv(x), stored in the P2 two-piece layout), then
//   1. bn_stats_p2: per-channel sum and sum of squares over every pixel of the batch, in double, fixed order -> mean, biased
//      variance -> (scale, shift); running_mean / running_var updated in place exactly as torch's batch_norm does;
//   2. bn_apply_p2: z <- leaky(z * scale + shift) * mask / (1 - p) in place, the mask drawn from Philox (philox.h).
// Both passes are HBM streams (8 bytes per element read, resp. read + written).
#include <algorithm>

#include "bn_train.h"
#include "common.h"

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

// thread -> (pixel lane pl, channel quad cq); a block walks pixels blockIdx.x * PL + pl, += gridDim.x * PL
__global__ __launch_bounds__(256) void bn_stats_p2_kernel(const unsigned char* __restrict__ z, size_t M, int C, double* __restrict__ partial) {
  __shared__ double red[256][8];
  const int tid = threadIdx.x;
  const int Q = C >> 2;           // channel quads per pixel: 16 .. 256
  const int PL = 256 / Q;         // pixels per block iteration
  const int cq = tid % Q, pl = tid / Q;
  const int c = 4 * cq;
  const size_t gpp = (size_t)(C >> 5);
  const size_t within = (size_t)(c >> 5) * 128 + (size_t)(c & 31) * 2;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  for (size_t m = (size_t)blockIdx.x * PL + pl; m < M; m += (size_t)gridDim.x * PL) {
    const unsigned char* p = z + m * gpp * 128 + within;
    const f16x4_t h = *reinterpret_cast<const f16x4_t*>(p), l = *reinterpret_cast<const f16x4_t*>(p + 64);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const double x = (double)((float)h[e] + (float)l[e]);   // the fp32 value the two pieces carry
      s[e] += x;
      q[e] += x * x;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid][e] = s[e]; red[tid][4 + e] = q[e]; }
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double ss = 0.0, qq = 0.0;
      for (int j = 0; j < PL; ++j) { ss += red[j * Q + cq][e]; qq += red[j * Q + cq][4 + e]; }
      double* o = partial + ((size_t)blockIdx.x * C + c + e) * 2;
      o[0] = ss;
      o[1] = qq;
    }
  }
}

// one workgroup per channel: thread t sums block partials t, t + 256, ... in order, then a fixed tree -> statistics -> affine pair;
// running statistics like torch  (a single thread per channel walking 1024 strided partials cost 0.26 ms per layer)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int nblk, int C, double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var, float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  __shared__ double rs[256], rq[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = tid; b < nblk; b += 256) {
    s += partial[((size_t)b * C + c) * 2];
    q += partial[((size_t)b * C + c) * 2 + 1];
  }
  rs[tid] = s;
  rq[tid] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { rs[tid] += rs[tid + w]; rq[tid] += rq[tid + w]; }
    __syncthreads();
  }
  if (tid != 0) return;
  const double mean = rs[0] / count;
  double var = rq[0] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (mean_out) { mean_out[c] = (float)mean; invstd_out[c] = invstd; }   // kept for the backward
  if (run_mean) run_mean[c] = momentum * (float)mean + (1.0f - momentum) * run_mean[c];
  if (run_var) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    run_var[c] = momentum * (float)unbiased + (1.0f - momentum) * run_var[c];
  }
}

__global__ __launch_bounds__(256) void bn_apply_p2_kernel(const unsigned char* z, unsigned char* out, size_t M, int C, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float slope, DropoutSpec drop, int* status) {
  const int Q = C >> 2;
  const size_t total = M * (size_t)Q;
  const size_t gpp = (size_t)(C >> 5);
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / Q;
    const int c = 4 * (int)(i - m * Q);
    const size_t off = (m * gpp + (size_t)(c >> 5)) * 128 + (size_t)(c & 31) * 2;
    const unsigned char* p = z + off;
    const f16x4_t h = *reinterpret_cast<const f16x4_t*>(p), l = *reinterpret_cast<const f16x4_t*>(p + 64);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
    unsigned bits[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (drop.thr) dropout_bits4(drop, i, bits);   // element index pixel * C + channel: block = i, word = channel % 4
    f16x4_t ho, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = ((float)h[e] + (float)l[e]) * sc[e] + sh[e];
      x = x > 0.f ? x : x * slope;
      if (drop.thr) x = bits[e] >= drop.thr ? x * drop.scale : 0.f;
      bad |= !(fabsf(x) <= 65504.f);
      ho[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)ho[e]);
    }
    *reinterpret_cast<f16x4_t*>(out + off) = ho;
    *reinterpret_cast<f16x4_t*>(out + off + 64) = lo;
  }
  if (bad) status[ODEVIO_STATUS_RANGE] = 1;
}

// x [rows][C] fp32: one workgroup per channel, double sums combined in a fixed tree
__global__ __launch_bounds__(256) void bn_stats_rows_kernel(const float* __restrict__ x, size_t rows, int C, float eps, float momentum,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ mean_out,
                                                            float* __restrict__ invstd_out) {
  __shared__ double rs[256], rq[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (size_t r = tid; r < rows; r += 256) {
    const double v = (double)x[r * C + c];
    s += v;
    q += v * v;
  }
  rs[tid] = s;
  rq[tid] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { rs[tid] += rs[tid + w]; rq[tid] += rq[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double count = (double)rows;
    const double mean = rs[0] / count;
    double var = rq[0] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) run_mean[c] = momentum * (float)mean + (1.0f - momentum) * run_mean[c];
    if (run_var) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      run_var[c] = momentum * (float)unbiased + (1.0f - momentum) * run_var[c];
    }
  }
}

__global__ void dropout_dump_kernel(float* __restrict__ out, size_t n, DropoutSpec d) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) out[e] = dropout_factor(d, e);
}

hipError_t bn_stats_p2(const void* z, size_t M, int C, double* partial, const float* gamma, const float* beta, float eps, float momentum,
                       float* run_mean, float* run_var, float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t st) {
  if (!z || !partial || !gamma || !beta || !scale || !shift || M == 0 || C < 64 || C > 1024 || (C & (C - 1))) return hipErrorInvalidValue;
  const int PL = 256 / (C >> 2);
  const int nblk = (int)std::min<size_t>(BN_MAX_BLOCKS, (M + PL - 1) / PL);
  (void)hipGetLastError();
  hipLaunchKernelGGL(bn_stats_p2_kernel, dim3(nblk), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(z), M, C, partial);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, st, partial, nblk, C, (double)M, gamma, beta, eps, momentum, run_mean,
                     run_var, scale, shift, mean_out, invstd_out);
  return hipGetLastError();
}

hipError_t bn_apply_p2(const void* z, void* out, size_t M, int C, const float* scale, const float* shift, float slope, const DropoutSpec& drop,
                       int* status, hipStream_t st) {
  if (!z || !out || !scale || !shift || !status || M == 0 || C % 32) return hipErrorInvalidValue;
  const size_t total = M * (size_t)(C >> 2);
  const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 16384);
  (void)hipGetLastError();
  hipLaunchKernelGGL(bn_apply_p2_kernel, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(z),
                     reinterpret_cast<unsigned char*>(out), M, C, scale, shift, slope, drop, status);
  return hipGetLastError();
}

hipError_t bn_stats_rows(const float* x, size_t rows, int C, float eps, float momentum, float* run_mean, float* run_var, float* mean,
                         float* invstd, hipStream_t st) {
  if (!x || !mean || !invstd || rows == 0 || C <= 0) return hipErrorInvalidValue;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bn_stats_rows_kernel, dim3(C), dim3(256), 0, st, x, rows, C, eps, momentum, run_mean, run_var, mean, invstd);
  return hipGetLastError();
}

void launch_dropout_dump(float* out, size_t n, const DropoutSpec& d, hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(dropout_dump_kernel, dim3(blocks), dim3(256), 0, st, out, n, d);
}
