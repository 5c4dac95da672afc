// Trajectory accumulation on the device: relative 6-DoF poses -> global 4x4 pose matrices.
//
// Replaces path_accu / pose_accu / pose_6DoF_to_matrix / eulerAnglesToRotationMatrix of the reference
// (src/data/utils.py:93-161) for the streaming evaluator (src/data/KITTI_eval.py:124-160, 231-232): out[0] = carry
// (identity for a new drive), out[i+1] = out[i] * [Rz(theta2) Ry(theta1) Rx(theta0) | t_i].
//
// The product of rigid transforms is associative, so the sequential loop of the reference becomes a scan: every
// thread multiplies its own run of consecutive poses, the 256 run products are scanned in LDS (Hillis-Steele, order
// preserving), and every thread replays its run from its exclusive prefix.  One workgroup handles one drive
// (a KITTI drive is <= ~4.6k frames); several drives go to several workgroups.
//
// Numerics follow the reference's dtypes: a float32 input (the network output) has sin/cos taken in float32 (numpy
// keeps float32 for np.cos(float32)); the 3x3 factors, their products and the running product are float64 (numpy
// promotes the literal 3x3 lists to float64).  A float64 input (ground truth) is float64 throughout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "philox.h"

namespace {

constexpr int PA_THREADS = 256;

struct Rigid {  // 3x4 [R | t], the bottom row is implicit
  double m[12];
};

__device__ inline Rigid rigid_identity() {
  Rigid r;
#pragma unroll
  for (int i = 0; i < 12; ++i) r.m[i] = 0.0;
  r.m[0] = r.m[5] = r.m[10] = 1.0;
  return r;
}

__device__ inline Rigid rigid_mul(const Rigid& a, const Rigid& b) {
  Rigid c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = a.m[i * 4 + 0] * b.m[0 * 4 + j];
      s += a.m[i * 4 + 1] * b.m[1 * 4 + j];
      s += a.m[i * 4 + 2] * b.m[2 * 4 + j];
      if (j == 3) s += a.m[i * 4 + 3];
      c.m[i * 4 + j] = s;
    }
  }
  return c;
}

// R = Rz(th[2]) * (Ry(th[1]) * Rx(th[0])) in precision T (reference utils.py:93-117: np.dot(R_z, np.dot(R_y, R_x)))
template <typename T>
__device__ inline Rigid rigid_from_pose(const T* p) {
  const double cx = (double)(T)cos(p[0]), sx = (double)(T)sin(p[0]);
  const double cy = (double)(T)cos(p[1]), sy = (double)(T)sin(p[1]);
  const double cz = (double)(T)cos(p[2]), sz = (double)(T)sin(p[2]);
  // Ry * Rx
  const double a00 = cy, a01 = sy * sx, a02 = sy * cx;
  const double a11 = cx, a12 = -sx;
  const double a20 = -sy, a21 = cy * sx, a22 = cy * cx;
  Rigid r;
  r.m[0] = cz * a00;
  r.m[1] = cz * a01 - sz * a11;
  r.m[2] = cz * a02 - sz * a12;
  r.m[4] = sz * a00;
  r.m[5] = sz * a01 + cz * a11;
  r.m[6] = sz * a02 + cz * a12;
  r.m[8] = a20;
  r.m[9] = a21;
  r.m[10] = a22;
  r.m[3] = (double)p[3];
  r.m[7] = (double)p[4];
  r.m[11] = (double)p[5];
  return r;
}

__device__ inline void rigid_store44(double* o, const Rigid& r) {
#pragma unroll
  for (int i = 0; i < 12; ++i) o[i] = r.m[i];
  o[12] = 0.0;
  o[13] = 0.0;
  o[14] = 0.0;
  o[15] = 1.0;
}

template <typename T>
__global__ __launch_bounds__(PA_THREADS) void path_accu_kernel(const T* __restrict__ poses, const int64_t* __restrict__ offsets,
                                                               const double* __restrict__ carry, double* __restrict__ out) {
  __shared__ Rigid scan[2][PA_THREADS];
  const int drive = blockIdx.x;
  const int64_t begin = offsets[drive], end = offsets[drive + 1];
  const int64_t n = end - begin;
  const T* p = poses + begin * 6;
  double* o = out + (begin + drive) * 16;  // every drive has n + 1 matrices
  const int tid = threadIdx.x;
  const int64_t run = (n + PA_THREADS - 1) / PA_THREADS;
  const int64_t lo = min((int64_t)tid * run, n), hi = min(lo + run, n);

  Rigid acc = rigid_identity();
  for (int64_t i = lo; i < hi; ++i) acc = rigid_mul(acc, rigid_from_pose<T>(p + i * 6));
  scan[0][tid] = acc;
  __syncthreads();
  int cur = 0;
  for (int d = 1; d < PA_THREADS; d <<= 1) {
    Rigid v = scan[cur][tid];
    if (tid >= d) v = rigid_mul(scan[cur][tid - d], v);
    scan[cur ^ 1][tid] = v;
    cur ^= 1;
    __syncthreads();
  }
  Rigid pre = rigid_identity();
  if (carry) {
#pragma unroll
    for (int i = 0; i < 12; ++i) pre.m[i] = carry[drive * 16 + i];
  }
  if (tid == 0) rigid_store44(o, pre);
  if (tid > 0) pre = rigid_mul(pre, scan[cur][tid - 1]);
  for (int64_t i = lo; i < hi; ++i) {
    pre = rigid_mul(pre, rigid_from_pose<T>(p + i * 6));
    rigid_store44(o + (i + 1) * 16, pre);
  }
}

}  // namespace

hipError_t launch_path_accu(const void* poses, int is_f64, const int64_t* offsets_dev, int n_drives, const double* carry,
                            double* out, hipStream_t stream) {
  (void)hipGetLastError();
  if (is_f64)
    hipLaunchKernelGGL(path_accu_kernel<double>, dim3(n_drives), dim3(PA_THREADS), 0, stream, (const double*)poses, offsets_dev,
                       carry, out);
  else
    hipLaunchKernelGGL(path_accu_kernel<float>, dim3(n_drives), dim3(PA_THREADS), 0, stream, (const float*)poses, offsets_dev,
                       carry, out);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// FusionModule "hard" (reference src/models/FusionModule.py:24-29): mask = F.gumbel_softmax(logits.view(..., F, 2), tau=1,
// hard=True)[..., 0], fused = cat * mask.  The forward value of the straight-through estimator is the one-hot arg-max of
// logits + Gumbel noise, so feature j is kept iff l0 + g0 >= l1 + g1.  torch draws g = -log(Exponential(1)) from its own
// generator; here the two uniforms of element j come from Philox 4x32-10 keyed by the plan's seed, counter = (element / 2,
// call): reproducible per seed, never bit-equal to torch.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gumbel_from_bits(unsigned bits) {
  const float u = ((float)(bits >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1), 24 bits
  return -logf(-logf(u));
}
__global__ void hard_mask_kernel(const float* __restrict__ cat, const float* __restrict__ logits, float* __restrict__ fused, size_t n,
                                 unsigned long long seed, unsigned long long call) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n + 1) / 2; i += (size_t)gridDim.x * blockDim.x) {
    unsigned c[4] = {(unsigned)i, (unsigned)(i >> 32), (unsigned)call, (unsigned)(call >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const size_t j = 2 * i + e;
      if (j < n) {
        const float keep = logits[2 * j] + gumbel_from_bits(c[2 * e]), drop = logits[2 * j + 1] + gumbel_from_bits(c[2 * e + 1]);
        fused[j] = keep >= drop ? cat[j] : 0.f;
      }
    }
  }
}
void launch_hard_mask(const float* cat, const float* logits, float* fused, size_t n, unsigned long long seed, unsigned long long call,
                      hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<size_t>(((n + 1) / 2 + 255) / 256, 2048);
  hipLaunchKernelGGL(hard_mask_kernel, dim3(blocks), dim3(256), 0, st, cat, logits, fused, n, seed, call);
}

// Straight-through backward of the hard mask (F.gumbel_softmax(..., hard=True): ret = y_hard - y_soft.detach() + y_soft, so the
// gradient is y_soft's).  With the same Philox block as the forward: s = sigmoid((l0 + g0) - (l1 + g1)) = y_soft[..., 0];
//   g_cat (direct part) = g * mask;   g_l0 = g * cat * s (1 - s);   g_l1 = -g_l0      (logits interleaved [P][2F])
__global__ void hard_mask_bwd_kernel(const float* __restrict__ g, const float* __restrict__ cat, const float* __restrict__ logits,
                                     float* __restrict__ g_cat, float* __restrict__ g_logits, size_t n, unsigned long long seed,
                                     unsigned long long call) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n + 1) / 2; i += (size_t)gridDim.x * blockDim.x) {
    unsigned c[4] = {(unsigned)i, (unsigned)(i >> 32), (unsigned)call, (unsigned)(call >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const size_t j = 2 * i + e;
      if (j < n) {
        const float keep = logits[2 * j] + gumbel_from_bits(c[2 * e]), drop = logits[2 * j + 1] + gumbel_from_bits(c[2 * e + 1]);
        const float sg = 1.0f / (1.0f + expf(drop - keep));
        const float gj = g[j];
        g_cat[j] = keep >= drop ? gj : 0.f;
        const float gl = gj * cat[j] * sg * (1.0f - sg);
        g_logits[2 * j] = gl;
        g_logits[2 * j + 1] = -gl;
      }
    }
  }
}
void launch_hard_mask_bwd(const float* g, const float* cat, const float* logits, float* g_cat, float* g_logits, size_t n,
                          unsigned long long seed, unsigned long long call, hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<size_t>(((n + 1) / 2 + 255) / 256, 2048);
  hipLaunchKernelGGL(hard_mask_bwd_kernel, dim3(blocks), dim3(256), 0, st, g, cat, logits, g_cat, g_logits, n, seed, call);
}
// diagnostic / test hook: the Gumbel pair of every element of call `call` (out [n][2]), what both kernels above draw
__global__ void gumbel_dump_kernel(float* __restrict__ out, size_t n, unsigned long long seed, unsigned long long call) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n + 1) / 2; i += (size_t)gridDim.x * blockDim.x) {
    unsigned c[4] = {(unsigned)i, (unsigned)(i >> 32), (unsigned)call, (unsigned)(call >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const size_t j = 2 * i + e;
      if (j < n) {
        out[2 * j] = gumbel_from_bits(c[2 * e]);
        out[2 * j + 1] = gumbel_from_bits(c[2 * e + 1]);
      }
    }
  }
}
void launch_gumbel_dump(float* out, size_t n, unsigned long long seed, unsigned long long call, hipStream_t st) {
  const unsigned blocks = (unsigned)std::min<size_t>(((n + 1) / 2 + 255) / 256, 2048);
  hipLaunchKernelGGL(gumbel_dump_kernel, dim3(blocks), dim3(256), 0, st, out, n, seed, call);
}
