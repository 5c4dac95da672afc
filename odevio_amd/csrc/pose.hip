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
