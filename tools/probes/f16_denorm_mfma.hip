// Probe: does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (decides whether the fp16x2 operand split is usable)
// Build: hipcc -O2 --offload-arch=gfx950 f16_denorm_mfma.hip -o f16_denorm_mfma   (result on MI355X: subnormal inputs are honoured, every case exact)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float a_val, float b_val, float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)b_val; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float cases[][2] = {{1.0f, 1.0f}, {9.5367431640625e-07f /*2^-20, subnormal*/, 1024.0f}, {5.9604644775390625e-08f /*2^-24, smallest*/, 16384.0f},
                            {3.0517578125e-05f /*2^-15 subnormal*/, 3.0517578125e-05f}};
  for (auto& c : cases) {
    probe<<<1, 64>>>(c[0], c[1], d);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g  mfma sum over k=16: %.9g   expected %.9g\n", c[0], c[1], h, 16.0 * (double)c[0] * (double)c[1]);
  }
  return 0;
}
