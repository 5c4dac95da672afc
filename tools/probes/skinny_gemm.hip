// Probe: where do the 8 us of a [32 x 512] x [512 x 512] skinny product go?  hipcc --offload-arch=gfx950 -O3 -o skinny_gemm skinny_gemm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
template <int MODE, int CLONE = 0>   // 0 full, 1 no MFMA (sum the loads), 2 no loads (constants); CLONE: the same code as another kernel
__global__ __launch_bounds__(256) void gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, float* __restrict__ out, int ldo,
                                               int M, int N, int K, u64* stamps) {
  __shared__ float red[4][16][17];
  const u64 c0 = __builtin_amdgcn_s_memtime() + CLONE, r0 = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  const float* wrow = W + (size_t)min(n0 + r, N - 1) * ldw + 4 * q;
  const float* arow = A + (size_t)min(m0 + r, M - 1) * lda + 4 * q;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 16 * wave; k0 < K; k0 += 512) {
    f32x4 wv[8], av[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + 64 * u + 4 * q;
      const bool in = k < K;
      const int kc = in ? k - 4 * q : 0;
      if (MODE != 2) { wv[u] = *reinterpret_cast<const f32x4*>(wrow + kc); av[u] = *reinterpret_cast<const f32x4*>(arow + kc); }
      else { wv[u] = f32x4{1.f, 2.f, 3.f, (float)kc}; av[u] = f32x4{1.f, 2.f, 3.f, (float)k}; }
      if (!in) { wv[u] = f32x4{0.f, 0.f, 0.f, 0.f}; av[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE != 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][j], av[u][j], acc, 0, 0, 0);
        else acc[j] += wv[u][j] * av[u][j];
      }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][4 * q + e][r] = acc[e];
  __syncthreads();
  const int nl = tid >> 4, ml = tid & 15;
  const int n = n0 + nl, m = m0 + ml;
  if (n < N && m < M) out[(size_t)m * ldo + n] = (red[0][nl][ml] + red[1][nl][ml]) + (red[2][nl][ml] + red[3][nl][ml]);
  if (stamps && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) { stamps[0] = __builtin_amdgcn_s_memtime() - c0; stamps[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int MODE>
void run(const char* what, const float* A, const float* W, float* out, int M, int N, int K, u64* stamps, int chain) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(gemm_nt<MODE>, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, 0, A, K, W, K, out, N, M, N, K, stamps);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) {
    // chain: each product reads the previous one's output (N == K), like the sweep
    const float* a = chain && (i & 1) ? out : A;
    float* o = chain && (i & 1) ? const_cast<float*>(A) : out;
    hipLaunchKernelGGL(gemm_nt<MODE>, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, 0, a, K, W + (size_t)(i % 8) * N * K, K, o, N, M, N, K, stamps);
  }
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  u64 s[2]; hipMemcpy(s, stamps, 16, hipMemcpyDeviceToHost);
  printf("%-28s M %d N %d K %d: %.2f us per launch back to back; inside workgroup 0: %.2f us at %.2f GHz\n", what, M, N, K, ms * 1e3 / reps, s[1] * 0.01,
         s[1] ? (double)s[0] / ((double)s[1] * 10.0) : 0.0);
}
void run_alt(const float* A, const float* W, float* out, int M, int N, int K, u64* stamps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 2000;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) {
    const float* a = (i & 1) ? out : A;
    float* o = (i & 1) ? const_cast<float*>(A) : out;
    const dim3 g((N + 15) / 16, (M + 15) / 16);
    const float* w = W + (size_t)(i % 8) * N * K;
    switch (i % 4) {
      case 0: hipLaunchKernelGGL((gemm_nt<0, 1>), g, dim3(256), 0, 0, a, K, w, K, o, N, M, N, K, stamps); break;
      case 1: hipLaunchKernelGGL((gemm_nt<0, 2>), g, dim3(256), 0, 0, a, K, w, K, o, N, M, N, K, stamps); break;
      case 2: hipLaunchKernelGGL((gemm_nt<0, 3>), g, dim3(256), 0, 0, a, K, w, K, o, N, M, N, K, stamps); break;
      default: hipLaunchKernelGGL((gemm_nt<0, 4>), g, dim3(256), 0, 0, a, K, w, K, o, N, M, N, K, stamps); break;
    }
  }
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("four different kernels in turn, chained: %.2f us per launch\n", ms * 1e3 / reps);
}
int main() {
  const int M = 32, N = 512, K = 512;
  float *A, *W, *out; u64* stamps;
  hipMalloc(&A, (size_t)M * 1024 * 4); hipMalloc(&W, (size_t)8 * 1024 * 1024 * 4); hipMalloc(&out, (size_t)M * 1024 * 4); hipMalloc(&stamps, 16);
  hipMemset(A, 0, (size_t)M * 1024 * 4); hipMemset(W, 0, (size_t)8 * 1024 * 1024 * 4); hipMemset(out, 0, (size_t)M * 1024 * 4);
  run<0>("full", A, W, out, M, N, K, stamps, 1);
  run<1>("no MFMA", A, W, out, M, N, K, stamps, 1);
  run<2>("no loads", A, W, out, M, N, K, stamps, 1);
  run<0>("full, independent launches", A, W, out, M, N, K, stamps, 0);
  run_alt(A, W, out, M, N, K, stamps);
  hipMemset(A, 0x3c, (size_t)M * 1024 * 4); hipMemset(W, 0x3c, (size_t)8 * 1024 * 1024 * 4);
  run<0>("full, nonzero data", A, W, out, M, N, K, stamps, 1);
  run<0>("full 768", A, W, out, M, 768, 512, stamps, 0);
  run<0>("full K 1024", A, W, out, M, 512, 1024, stamps, 0);
  return 0;
}
