// Frame resize on the device: what the reference's loader does on the host before the encoder sees a frame
// (src/data/KITTI_eval.py:101 / src/data/utils.py:366-371: torchvision TF.resize of a PIL image = PIL.Image.resize with
// BILINEAR resampling; KITTI frames are 1241 x 376, the network takes 512 x 256).
//
// PIL's resampling (Pillow src/libImaging/Resample.c, 8 bits per channel) is integer arithmetic and is reproduced BIT FOR
// BIT: a triangle filter whose support is stretched by the scale factor when shrinking (antialiasing), coefficients
// normalised in double and rounded to 22-bit fixed point, a horizontal pass into an 8-bit intermediate image followed
// by a vertical pass, each accumulating in int32 from 2^21 and clipping (sum >> 22) to [0, 255].  The coefficient
// tables depend only on (input size, output size): they are computed on the host exactly as Pillow does (the double
// arithmetic is part of the contract) and cached per size; the kernels do the integer passes.
// oracle/pil_resize.py restates the same algorithm in numpy and is pinned against Pillow itself (tests/golden/resize.npz).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/odevio.h"

#define RS_PRECISION_BITS (32 - 8 - 2)

struct ResizeTable {
  int ksize = 0;
  std::vector<int> bounds;   // [out][2] = first input index, count
  std::vector<int> kk;       // [out][ksize] fixed-point coefficients
  int *d_bounds = nullptr, *d_kk = nullptr;
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter, whole-image box.
static void build_table(int in_size, int out_size, ResizeTable& t) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;            // bilinear support = 1
  t.ksize = (int)std::ceil(support) * 2 + 1;
  t.bounds.assign((size_t)out_size * 2, 0);
  t.kk.assign((size_t)out_size * t.ksize, 0);
  std::vector<double> pre((size_t)t.ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double w = a < 1.0 ? 1.0 - a : 0.0;
      pre[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) pre[x] /= ww;
    for (int x = xmax; x < t.ksize; ++x) pre[x] = 0.0;
    for (int x = 0; x < t.ksize; ++x) {
      const double v = pre[x] * (double)(1 << RS_PRECISION_BITS);
      t.kk[(size_t)xx * t.ksize + x] = pre[x] < 0 ? (int)(-0.5 + v) : (int)(0.5 + v);
    }
    t.bounds[2 * xx] = xmin;
    t.bounds[2 * xx + 1] = xmax;
  }
}

// Host-only view of the table (tests: compared with the oracle's on the CPU; exercised by the ASan driver)
int resize_table_host(int in_size, int out_size, int* ksize, int* bounds, int* kk, int kk_capacity) {
  if (in_size <= 0 || out_size <= 0 || !ksize || !bounds || !kk) return ODEVIO_ERR_BAD_ARG;
  ResizeTable t;
  build_table(in_size, out_size, t);
  *ksize = t.ksize;
  if ((size_t)kk_capacity < t.kk.size()) return ODEVIO_ERR_BAD_ARG;
  std::copy(t.bounds.begin(), t.bounds.end(), bounds);
  std::copy(t.kk.begin(), t.kk.end(), kk);
  return 0;
}

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= RS_PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [n][Hin][Win][3] -> tmp [n][Hin][Wout][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, size_t rows,
                                                       int Win, int Wout) {
  const size_t total = rows * Wout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / Wout;
    const int xx = (int)(i - row * Wout);
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const unsigned char* p = src + (row * Win + xmin) * 3;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) {
      const int c = k[x];
      s0 += p[3 * x] * c;
      s1 += p[3 * x + 1] * c;
      s2 += p[3 * x + 2] * c;
    }
    unsigned char* o = tmp + i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
}

// vertical pass: tmp [n][Hin][Wout][3] -> dst [n][Hout][Wout][3]
__global__ __launch_bounds__(256) void resize_v_kernel(const unsigned char* __restrict__ tmp, unsigned char* __restrict__ dst,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int n, int Hin,
                                                       int Hout, int Wout) {
  const size_t total = (size_t)n * Hout * Wout;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % Wout);
    const size_t r = i / Wout;
    const int yy = (int)(r % Hout);
    const size_t img = r / Hout;
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    const unsigned char* p = tmp + ((img * Hin + ymin) * Wout + xx) * 3;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) {
      const int c = k[y];
      const unsigned char* q = p + (size_t)y * Wout * 3;
      s0 += q[0] * c;
      s1 += q[1] * c;
      s2 += q[2] * c;
    }
    unsigned char* o = dst + i * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
  }
}

static std::mutex g_mu;
static std::map<long long, ResizeTable> g_tables;   // (device, in, out) -> table (device copies live for the process)

static const ResizeTable* table_for(int in_size, int out_size, hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const long long key = ((long long)dev << 48) | ((long long)in_size << 24) | out_size;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_tables.find(key);
  if (it != g_tables.end()) return &it->second;
  ResizeTable& t = g_tables[key];
  build_table(in_size, out_size, t);
  if (hipMalloc((void**)&t.d_bounds, t.bounds.size() * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&t.d_kk, t.kk.size() * sizeof(int)) != hipSuccess ||
      hipMemcpyAsync(t.d_bounds, t.bounds.data(), t.bounds.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(t.d_kk, t.kk.data(), t.kk.size() * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {
    g_tables.erase(key);
    return nullptr;
  }
  return &t;
}

// tmp: device scratch of n * Hin * Wout * 3 bytes (unused when the width does not change)
int resize_u8_launch(const unsigned char* src, int n, int Hin, int Win, unsigned char* dst, int Hout, int Wout, unsigned char* tmp,
                     hipStream_t st) {
  // Pillow skips a pass whose size does not change (need_horizontal / need_vertical)
  const bool need_h = Win != Wout, need_v = Hin != Hout;
  if (!need_h && !need_v)
    return hipMemcpyAsync(dst, src, (size_t)n * Hin * Win * 3, hipMemcpyDeviceToDevice, st) == hipSuccess ? 0 : ODEVIO_ERR_HIP;
  const unsigned char* vin = src;
  if (need_h) {
    const ResizeTable* th = table_for(Win, Wout, st);
    if (!th) return ODEVIO_ERR_HIP;
    unsigned char* hout = need_v ? tmp : dst;
    const size_t rows = (size_t)n * Hin;
    const size_t total = rows * Wout;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, st, src, hout, th->d_bounds,
                       th->d_kk, th->ksize, rows, Win, Wout);
    vin = hout;
  }
  if (need_v) {
    const ResizeTable* tv = table_for(Hin, Hout, st);
    if (!tv) return ODEVIO_ERR_HIP;
    const size_t total = (size_t)n * Hout * Wout;
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, st, vin, dst, tv->d_bounds,
                       tv->d_kk, tv->ksize, n, Hin, Hout, Wout);
  }
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
