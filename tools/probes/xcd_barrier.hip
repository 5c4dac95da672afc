// Probe: cost of a phase barrier among the workgroups of ONE XCD (slots in L2, no cache maintenance) - census, member count,
// time per phase with a little work in it.  hipcc --offload-arch=gfx950 -O3 -o xcd_barrier xcd_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xfu; }
__device__ __forceinline__ bool wait_ge(const u64* p, u64 want) {
  unsigned spins = 0; u64 t0 = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    if ((++spins & 63u) == 0) { const u64 now = __builtin_amdgcn_s_memrealtime(); if (!t0) t0 = now; if (now - t0 > 100000000ull) return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
struct Args { u64* census; u64* slots; float* data; u64* out; int phases; int work; int sleep; };
__global__ __launch_bounds__(256) void probe(Args a) {
  __shared__ int s_rank, s_members;
  const int tid = threadIdx.x;
  const u64 t_begin = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) {
    const unsigned mine = xcc_id();
    if (blockIdx.x == 0) __hip_atomic_store(a.census, (u64)mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int rank = -1, members = 0;
    if (wait_ge(a.census, 1)) {
      const unsigned target = (unsigned)__hip_atomic_load(a.census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
      if (mine == target) rank = (int)__hip_atomic_fetch_add(a.census + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(a.census + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (rank >= 0) {
        if (wait_ge(a.census + 2, gridDim.x)) members = (int)__hip_atomic_fetch_add(a.census + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else rank = -1;
      }
    }
    s_rank = rank; s_members = members;
  }
  __syncthreads();
  const int rank = s_rank, members = s_members;
  if (rank < 0) return;
  const u64 t_census = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  u64 t_wait = 0;
  for (int ph = 1; ph <= a.phases; ++ph) {
    // work: write my 1 KB, (barrier), read the next member's 1 KB past the L1
    float* mine = a.data + (size_t)((ph & 1) * 1024 + rank) * 256;
    for (int w = 0; w < a.work; ++w) mine[tid] = acc + (float)(ph + w);
    const u64 w0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.slots + rank, (u64)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool ok = true;
    for (int t = tid; t < members; t += 256) ok = ok && wait_ge(a.slots + t, (u64)ph);
    if (__syncthreads_or(ok ? 0 : 1)) return;
    t_wait += __builtin_amdgcn_s_memrealtime() - w0;
    const float* other = a.data + (size_t)((ph & 1) * 1024 + (rank + 1) % members) * 256;
    const u64 v = __hip_atomic_load(reinterpret_cast<const u64*>(other) + (tid >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += __uint_as_float((unsigned)v) * 1e-9f;
  }
  const u64 t_end = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && rank == 0) { a.out[0] = members; a.out[1] = t_census - t_begin; a.out[2] = t_end - t_census; a.out[3] = t_wait; a.out[4] = (u64)(acc * 0.f); }
}
// the same phases with an agent-scope fence barrier over the whole grid (monotonic counter)
__global__ __launch_bounds__(256) void probe_fence(Args a) {
  __shared__ int ok;
  const int tid = threadIdx.x;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int ph = 1; ph <= a.phases; ++ph) {
    float* mine = a.data + (size_t)((ph & 1) * 1024 + blockIdx.x) * 256;
    for (int w = 0; w < a.work; ++w) mine[tid] = acc + (float)(ph + w);
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(a.census, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = wait_ge(a.census, (u64)ph * gridDim.x) ? 1 : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!ok) return;
    const float* other = a.data + (size_t)((ph & 1) * 1024 + (blockIdx.x + 1) % gridDim.x) * 256;
    acc += other[tid] * 1e-9f;
  }
  if (tid == 0 && blockIdx.x == 0) { a.out[0] = gridDim.x; a.out[1] = 0; a.out[2] = __builtin_amdgcn_s_memrealtime() - t0; a.out[4] = (u64)(acc * 0.f); }
}
int main() {
  Args a;
  hipMalloc(&a.census, 64); hipMalloc(&a.slots, 1024 * 8); hipMalloc(&a.data, 2 * 1024 * 256 * 4); hipMalloc(&a.out, 64);
  a.phases = 200; a.work = 1; a.sleep = 1;
  for (int grid : {64 * 8, 96 * 8, 32 * 8, 16 * 8}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(a.census, 0, 64); hipMemset(a.slots, 0, 1024 * 8); hipMemset(a.out, 0, 64);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, a);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      u64 o[5]; hipMemcpy(o, a.out, 40, hipMemcpyDeviceToHost);
      printf("xcd-local grid %4d: members %llu, census %.2f us, %d phases %.2f us each (of which barrier %.2f us), launch %.1f us total\n", grid, o[0], o[1] * 0.01,
             a.phases, o[2] * 0.01 / a.phases, o[3] * 0.01 / a.phases, ms * 1e3);
    }
  }
  for (int grid : {64, 96}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(a.census, 0, 64); hipMemset(a.out, 0, 64);
      hipLaunchKernelGGL(probe_fence, dim3(grid), dim3(256), 0, 0, a);
      hipDeviceSynchronize();
      u64 o[5]; hipMemcpy(o, a.out, 40, hipMemcpyDeviceToHost);
      printf("agent-fence grid %4d: %d phases %.2f us each\n", grid, a.phases, o[2] * 0.01 / a.phases);
    }
  }
  return 0;
}
