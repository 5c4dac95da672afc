// Philox 4x32-10 (Salmon et al., SC'11) on the device and what the kernels draw from it.  One counter block per (element group,
// draw): counter = (group lo, group hi, call lo, call hi), key = the plan's seed.  The numpy restatement pinned by the
// published known-answer vectors is oracle/philox.py; tests hold these functions to it bit for bit.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// nn.Dropout(p) in train mode (reference src/models/Encoder.py:21,46-56: Dropout(0.2 / 0.5) after every block of the image encoder,
// Dropout(opt.imu_dropout) in the inertial encoder): element e of the tensor (in the order its kernel documents) is KEPT iff word
// e % 4 of block e / 4 of draw `call` is >= thr = floor(p * 2^32); kept values are multiplied by 1 / (1 - p) like torch does.
// torch draws its Bernoulli mask from its own generator, so there is no bit-level parity with the reference: parity tests hand
// THIS mask to the oracle (odevio_debug_dropout).
struct DropoutSpec {
  unsigned long long seed, call;
  unsigned thr;     // 0: keep everything (p = 0)
  float scale;      // 1 / (1 - p)
};
__device__ __forceinline__ void dropout_bits4(const DropoutSpec& d, unsigned long long block, unsigned (&c)[4]) {
  c[0] = (unsigned)block; c[1] = (unsigned)(block >> 32); c[2] = (unsigned)d.call; c[3] = (unsigned)(d.call >> 32);
  philox4x32_10(c, (unsigned)d.seed, (unsigned)(d.seed >> 32));
}
__device__ __forceinline__ float dropout_factor(const DropoutSpec& d, unsigned long long e) {
  if (d.thr == 0) return 1.0f;
  unsigned c[4];
  dropout_bits4(d, e >> 2, c);
  const unsigned w = (unsigned)(e & 3);
  const unsigned bits = w == 0 ? c[0] : (w == 1 ? c[1] : (w == 2 ? c[2] : c[3]));
  return bits >= d.thr ? d.scale : 0.0f;
}
inline DropoutSpec make_dropout(unsigned long long seed, unsigned long long call, float p) {
  DropoutSpec d;
  d.seed = seed; d.call = call;
  const double t = (double)p * 4294967296.0;
  d.thr = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xffffffffu : (unsigned)t);
  d.scale = 1.0f / (1.0f - p);
  return d;
}
