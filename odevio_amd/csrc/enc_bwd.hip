// Backward of the image encoder's conv blocks under model.train(), gfx950 (see enc_bwd.h for what each entry computes and the
// reference lines it stands for).  Everything here is fp32: gradients span many orders of magnitude, which the two-fp16-piece
// layout of the forward activations cannot carry without per-tensor exponents; the fp32-input MFMA (v_mfma_f32_16x16x4_f32, an
// exact fmaf chain) does the contractions.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "enc_bwd.h"

typedef _Float16 h16x4_t __attribute__((ext_vector_type(4)));

namespace {

// value (m, c .. c+3) of a P2 tensor as fp32
__device__ __forceinline__ f32x4 p2_load4(const unsigned char* z, size_t m, int c, int C) {
  const unsigned char* p = z + (m * (size_t)(C >> 5) + (size_t)(c >> 5)) * 128 + (size_t)(c & 31) * 2;
  const h16x4_t h = *reinterpret_cast<const h16x4_t*>(p), l = *reinterpret_cast<const h16x4_t*>(p + 64);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (float)h[e] + (float)l[e];
  return v;
}

// dz and xhat of four consecutive channels of pixel m; i = m * (C / 4) + c / 4 is also the dropout block index
__device__ __forceinline__ void bn_bwd_point(const float* __restrict__ g_a, const unsigned char* __restrict__ z, size_t i, int Q, int C,
                                             const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, const DropoutSpec& drop, f32x4& dz, f32x4& xh, int& c_out) {
  const size_t m = i / Q;
  const int c = 4 * (int)(i - m * Q);
  c_out = c;
  const f32x4 zv = p2_load4(z, m, c, C);
  const f32x4 g = *reinterpret_cast<const f32x4*>(g_a + m * (size_t)C + c);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  unsigned bits[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  if (drop.thr) dropout_bits4(drop, i, bits);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    xh[e] = (zv[e] - mu[e]) * is[e];
    const float y = fmaf(ga[e], xh[e], be[e]);
    float f = y > 0.f ? 1.0f : 0.1f;
    if (drop.thr) f = bits[e] >= drop.thr ? f * drop.scale : 0.f;
    dz[e] = g[e] * f;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g_a, const unsigned char* __restrict__ z, size_t M, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, DropoutSpec drop,
                                                            double* __restrict__ partial) {
  __shared__ double red[256][8];
  const int tid = threadIdx.x;
  const int Q = C >> 2, PL = 256 / Q;
  const int cq = tid % Q, pl = tid / Q;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
  for (size_t m = (size_t)blockIdx.x * PL + pl; m < M; m += (size_t)gridDim.x * PL) {
    f32x4 dz, xh;
    int c;
    bn_bwd_point(g_a, z, m * Q + cq, Q, C, mean, invstd, gamma, beta, drop, dz, xh, c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[e] += (double)dz[e];
      q[e] += (double)dz[e] * (double)xh[e];
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
      double* o = partial + ((size_t)blockIdx.x * C + 4 * cq + e) * 2;
      o[0] = ss;
      o[1] = qq;
    }
  }
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, int C, float* __restrict__ sums) {
  __shared__ double rs[256], rq[256];
  const int c = blockIdx.x, tid = threadIdx.x;   // one workgroup per channel, fixed summation tree
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
  if (tid == 0) {
    sums[c] = (float)rs[0];
    sums[C + c] = (float)rq[0];
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g_a, const unsigned char* __restrict__ z, size_t M, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, DropoutSpec drop,
                                                           const float* __restrict__ sums, float* __restrict__ D) {
  const int Q = C >> 2;
  const size_t total = M * (size_t)Q;
  const float invm = 1.0f / (float)M;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 dz, xh, d;
    int c;
    bn_bwd_point(g_a, z, i, Q, C, mean, invstd, gamma, beta, drop, dz, xh, c);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = gamma[c + e] * invstd[c + e] * (dz[e] - sums[c + e] * invm - xh[e] * sums[C + c + e] * invm);
    *reinterpret_cast<f32x4*>(D + (i / Q) * (size_t)C + c) = d;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient.  One workgroup = one 64 (co) x 64 (ci) tile of one filter tap over one range of pixels; 4 waves as 2 x 2, each
// 32 x 32 = 2 x 2 MFMA tiles.  Both operands arrive PIXEL-major (D [pixel][co], x [pixel][ci]), which is what the 16x16x4 fp32
// MFMA wants when the contraction runs over pixels: lane (r, q) supplies A[row r][k q] = D[pixel q][co r] and B[k q][col r] =
// x[pixel q][ci r] - no transpose anywhere.  64-pixel chunks are staged in LDS (row stride 80 floats: the two 32-lane halves of a
// ds_read_b32 then touch 32 different banks).
// ---------------------------------------------------------------------------------------------------------------------
// pixels per chunk: measured 16 / 32 / 64 / 128 -> Image_net backward 64.2 / 64.2 / 69.6 / 85.4 ms at the bench batch: the kernel lives on
// the number of workgroups a CU holds (its loads are latency, 40 KB of LDS per workgroup at 64 pixels allowed four)
#ifndef WG_PX
#define WG_PX 32
#endif
#define WG_ROWS (WG_PX / 16)   // loader rows per thread and chunk
#define WG_LD 80
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  __shared__ float Dt[WG_PX * WG_LD], Xt[WG_PX * WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int taps = a.KH * a.KW;
  // conv1 (6 channels in 8 slots): a 64-wide tile of input channels would multiply 56 columns of zeros per tap - and did, as many
  // MFMAs as all the other layers together.  Folded form: the tile's 64 columns are (kw, channel slot) of ONE filter row, a
  // workgroup per kh; column j of the loader / of the result = (kw = j / 8, slot = j % 8).
  const bool fold = a.fold_kw != 0;
  const int gtaps = fold ? a.KH : taps;
  const int tap = blockIdx.x % gtaps, split = blockIdx.x / gtaps;
  const int kh = fold ? tap : tap / a.KW;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lrow = tid >> 4, lcol = 4 * (tid & 15);     // loader: rows lrow + 16 j, four consecutive channels at lcol
  const int HoWo = a.Ho * a.Wo;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
  const float* xf = reinterpret_cast<const float*>(a.x);
  const int G = a.Cin >> 5;
  const int chunk_begin = split * a.chunks_per_split;
  const int chunk_end = min((split + 1) * a.chunks_per_split, (a.M + WG_PX - 1) / WG_PX);
  const int kw = fold ? (lcol >> 3) : tap - kh * a.KW;          // (folded: this loader thread's own filter column)
  const bool co_ok = co0 + lcol < a.Cout, ci_ok = fold ? kw < a.KW : ci0 + lcol < a.Cin;
  // (image, row, column) of this thread's four loader rows, advanced by carries from chunk to chunk (two divisions per row ONCE, not per chunk:
  // the address arithmetic of the first version cost as much issue time as the MFMAs)
  int pn[WG_ROWS], pho[WG_ROWS], pwo[WG_ROWS];
#pragma unroll
  for (int j = 0; j < WG_ROWS; ++j) {
    const int m = chunk_begin * WG_PX + lrow + 16 * j;
    pn[j] = m / HoWo;
    const int rem = m - pn[j] * HoWo;
    pho[j] = rem / a.Wo;
    pwo[j] = rem - pho[j] * a.Wo;
  }
  f32x4 dv[WG_ROWS], xv[WG_ROWS];
  auto fetch = [&](int ch) __attribute__((always_inline)) {   // chunk ch -> registers (the loads stay in flight under the previous chunk's MFMAs)
#pragma unroll
    for (int j = 0; j < WG_ROWS; ++j) {
      const int m = ch * WG_PX + lrow + 16 * j;
      dv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < a.M) {
        if (co_ok) dv[j] = *reinterpret_cast<const f32x4*>(a.D + (size_t)m * a.Cout + co0 + lcol);   // (Cout % 4 == 0)
        const int hi = pho[j] * a.stride + kh - a.pad, wi = pwo[j] * a.stride + kw - a.pad;
        if ((unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi && ci_ok) {
          const size_t pi = ((size_t)pn[j] * a.Hi + hi) * a.Wi + wi;
          const int ci = fold ? (lcol & 7) : ci0 + lcol;
          if (a.x_f32) {
            xv[j] = *reinterpret_cast<const f32x4*>(xf + pi * a.ldx + ci);
          } else {
            const unsigned char* p = xb + (pi * (size_t)G + (size_t)(ci >> 5)) * 128 + (size_t)(ci & 31) * 2;
            const h16x4_t h = *reinterpret_cast<const h16x4_t*>(p), l = *reinterpret_cast<const h16x4_t*>(p + 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[j][e] = (float)h[e] + (float)l[e];
          }
        }
      }
      pwo[j] += WG_PX;                                  // this row's pixel in the NEXT chunk
      while (pwo[j] >= a.Wo) {
        pwo[j] -= a.Wo;
        if (++pho[j] == a.Ho) { pho[j] = 0; ++pn[j]; }
      }
    }
  };
  if (chunk_begin < chunk_end) fetch(chunk_begin);
  for (int ch = chunk_begin; ch < chunk_end; ++ch) {
    __syncthreads();                                    // every wave is done reading the previous chunk
#pragma unroll
    for (int j = 0; j < WG_ROWS; ++j) {
      *reinterpret_cast<f32x4*>(&Dt[(lrow + 16 * j) * WG_LD + lcol]) = dv[j];
      *reinterpret_cast<f32x4*>(&Xt[(lrow + 16 * j) * WG_LD + lcol]) = xv[j];
    }
    __syncthreads();
    if (ch + 1 < chunk_end) fetch(ch + 1);
#pragma unroll 4
    for (int kk = 0; kk < WG_PX / 4; ++kk) {
      const int prow = (4 * kk + q) * WG_LD;
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = Dt[prow + wr * 32 + 16 * i + r];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Xt[prow + wc * 32 + 16 * j + r];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  // C/D map of the 16x16x4 MFMA: lane (r, q), register e = [row 4 q + e][column r] = [co][ci]
  const size_t per = (size_t)a.Cout * a.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + wr * 32 + 16 * i + 4 * q + e, col = wc * 32 + 16 * j + r;
        const int ci = fold ? (col & 7) : ci0 + col;
        const int t = fold ? kh * a.KW + (col >> 3) : tap;
        if (co < a.Cout && ci < a.Cin && (!fold || (col >> 3) < a.KW)) a.partial[((size_t)split * taps + t) * per + (size_t)co * a.Cin + ci] = acc[i][j][e];
      }
}
// The same gradient for the wide layers (Wo a multiple of 32: conv2 .. conv4_1, four fifths of the time): one workgroup = one 64 x 64
// tile of one filter ROW kh - all KW taps of it - over a range of 32-pixel runs of output rows.  A run's gradient tile D [32][64 co] is
// staged once and multiplied against KW shifted views of ONE input patch ((32 - 1) s + KW pixels of input row ho s + kh - pad): the
// per-tap kernel loaded D and a fresh x tile for every tap - 2 KW loads where this one makes 1 + (s + (KW - 1) / 32), and KW x the
// MFMAs between two barriers.  Accumulators: KW x (2 x 2) tiles per wave.  Same slabs, same reduce kernel.
#define WR_PX 32
template <int KW>
__global__ __launch_bounds__(256) void wgrad_row_kernel(WgradArgs a) {
  constexpr int NPC_MAX = (WR_PX - 1) * 2 + KW;           // patch pixels at stride 2
  __shared__ float Dt[WR_PX * WG_LD], Xp[NPC_MAX * WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int kh = blockIdx.x % a.KH, split = blockIdx.x / a.KH;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int wr = wave >> 1, wc = wave & 1;
  const int s = a.stride;
  const int npc = (WR_PX - 1) * s + KW;                   // patch pixels of this layer
  const int ldx = s == 1 ? WG_LD : 72;                    // patch row stride: lanes q = 0 .. 3 read rows s apart - (s ldx) mod 64 = 16 keeps them on four bank groups
  f32x4 acc[KW][2][2];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lrow = tid >> 4, lcol = 4 * (tid & 15);       // loader: pixel lrow + 16 j, four consecutive channels at lcol
  const bool co_ok = co0 + lcol < a.Cout, ci_ok = ci0 + lcol < a.Cin;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
  const int G = a.Cin >> 5;
  const int runs_per_row = a.Wo / WR_PX;
  const int run_begin = split * a.chunks_per_split, run_end = min((split + 1) * a.chunks_per_split, a.N * a.Ho * runs_per_row);
  f32x4 dv[WR_PX / 16], xv[(NPC_MAX + 15) / 16];
  auto fetch = [&](int run) __attribute__((always_inline)) {   // run -> registers (the loads stay in flight under the previous run's MFMAs)
    const int row = run / runs_per_row, wo0 = (run - row * runs_per_row) * WR_PX;     // row = n * Ho + ho
    const int n = row / a.Ho, ho = row - n * a.Ho;
    const int hi = ho * s + kh - a.pad;
    const int wi0 = wo0 * s - a.pad;
#pragma unroll
    for (int j = 0; j < WR_PX / 16; ++j) {
      const size_t m = (size_t)row * a.Wo + wo0 + lrow + 16 * j;
      dv[j] = co_ok ? *reinterpret_cast<const f32x4*>(a.D + m * a.Cout + co0 + lcol) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < (NPC_MAX + 15) / 16; ++j) {
      const int pp = lrow + 16 * j, wi = wi0 + pp;
      xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pp < npc && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi && ci_ok) {
        const size_t pi = ((size_t)n * a.Hi + hi) * a.Wi + wi;
        const int ci = ci0 + lcol;
        const unsigned char* p = xb + (pi * (size_t)G + (size_t)(ci >> 5)) * 128 + (size_t)(ci & 31) * 2;
        const h16x4_t h = *reinterpret_cast<const h16x4_t*>(p), l = *reinterpret_cast<const h16x4_t*>(p + 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[j][e] = (float)h[e] + (float)l[e];
      }
    }
  };
  if (run_begin < run_end) fetch(run_begin);
  for (int run = run_begin; run < run_end; ++run) {
    __syncthreads();                                      // every wave is done reading the previous run
#pragma unroll
    for (int j = 0; j < WR_PX / 16; ++j) *reinterpret_cast<f32x4*>(&Dt[(lrow + 16 * j) * WG_LD + lcol]) = dv[j];
#pragma unroll
    for (int j = 0; j < (NPC_MAX + 15) / 16; ++j)
      if (lrow + 16 * j < NPC_MAX) *reinterpret_cast<f32x4*>(&Xp[(lrow + 16 * j) * ldx + lcol]) = xv[j];
    __syncthreads();
    if (run + 1 < run_end) fetch(run + 1);
#pragma unroll 2
    for (int kk = 0; kk < WR_PX / 4; ++kk) {
      const int px = 4 * kk + q;                          // this lane's pixel of the run (the MFMA's k index)
      float av[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = Dt[px * WG_LD + wr * 32 + 16 * i + r];
#pragma unroll
      for (int t = 0; t < KW; ++t) {
        const int xrow = (px * s + t) * ldx;
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = Xp[xrow + wc * 32 + 16 * j + r];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[t][i][j], 0, 0, 0);
      }
    }
  }
  const int taps = a.KH * a.KW;
  const size_t per = (size_t)a.Cout * a.Cin;
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = co0 + wr * 32 + 16 * i + 4 * q + e, ci = ci0 + wc * 32 + 16 * j + r;
          if (co < a.Cout && ci < a.Cin) a.partial[((size_t)split * taps + kh * a.KW + t) * per + (size_t)co * a.Cin + ci] = acc[t][i][j][e];
        }
}

// slabs in slab order -> dW [Cout][Cin][KH][KW]
__global__ void wgrad_reduce_kernel(WgradArgs a) {
  const int taps = a.KH * a.KW;
  const size_t per = (size_t)a.Cout * a.Cin;
  const size_t total = per * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const size_t cc = i / taps;                         // co * Cin + ci
    const int ci = (int)(cc % a.Cin);
    if (ci >= a.cin_out) continue;                      // (conv1: 6 real channels in 8 slots)
    float s = 0.f;
    for (int z = 0; z < a.splits; ++z) s += a.partial[((size_t)z * taps + tap) * per + cc];
    a.dW[((cc / a.Cin) * a.cin_out + ci) * taps + tap] = s;
  }
}

__global__ void dilate_kernel(const float* __restrict__ D, float* __restrict__ Dd, int N, int Ho, int Wo, int Hd, int Wd, int C, int stride) {
  const size_t total4 = (size_t)N * Hd * Wd * (C >> 2);
  const int Q = C >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % Q);
    size_t p = i / Q;
    const int wd = (int)(p % Wd);
    p /= Wd;
    const int hd = (int)(p % Hd);
    const int n = (int)(p / Hd);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hd % stride == 0 && wd % stride == 0 && hd / stride < Ho && wd / stride < Wo)
      v = *reinterpret_cast<const f32x4*>(D + (((size_t)n * Ho + hd / stride) * Wo + wd / stride) * C + c);
    *reinterpret_cast<f32x4*>(Dd + i * 4) = v;
  }
}
// ---- input gradient on the forward's fp16x2 kernel: D (fp32) -> P2 pieces of D * 2^e, zero-dilated for a stride-s convolution.
// e is chosen per tensor from max|D| (device word `amax`, float bits) so that the largest element lands in [2^10, 2^11): gradients are
// many orders of magnitude smaller than activations and would otherwise sit in fp16's subnormals.  The factor is divided back out in the
// convolution's epilogue scale (enc_fill_scale), exactly (a power of two).
// (n % 4 == 0, x 16-byte aligned: four 16-byte loads in flight per thread - one 4-byte load at a time made this pass, pure latency, cost
//  0.75 ms per layer, 6 ms of the image encoder's backward)
__global__ void absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ amax) {
  unsigned m = 0;
  const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[u][e]) & 0x7fffffffu);   // |x| bit patterns order like the values (NaN / inf on top)
  }
  for (; i < n4; i += stride) {
    const f32x4 v = x4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  // one atomic per WORKGROUP (and at most 1,024 workgroups): 65 k same-address atomics, one per wave, serialised into milliseconds
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (t) atomicMax(amax, t);
  }
}
__device__ __forceinline__ int grad_exponent(unsigned amax_bits) {
  if (amax_bits == 0 || amax_bits >= 0x7f800000u) return 0;
  int ex;
  (void)frexpf(__uint_as_float(amax_bits), &ex);          // |max| = f * 2^ex, f in [0.5, 1)
  return max(-100, min(100, 11 - ex));                     // |max| * 2^e in [2^10, 2^11)
}
__global__ void pack_dilate_kernel(const float* __restrict__ D, unsigned char* __restrict__ out, int N, int Ho, int Wo, int Hd, int Wd, int C, int stride,
                                   const unsigned* __restrict__ amax) {
  const float sc = ldexpf(1.0f, grad_exponent(*amax));
  const int Q = C >> 2;
  const size_t total4 = (size_t)N * Hd * Wd * Q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % Q);
    const size_t pix = i / Q;
    size_t p = pix;
    const int wd = (int)(p % Wd);
    p /= Wd;
    const int hd = (int)(p % Hd);
    const int n = (int)(p / Hd);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hd % stride == 0 && wd % stride == 0 && hd / stride < Ho && wd / stride < Wo)
      v = *reinterpret_cast<const f32x4*>(D + (((size_t)n * Ho + hd / stride) * Wo + wd / stride) * C + c) * sc;
    h16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (_Float16)v[e];
      l[e] = (_Float16)(v[e] - (float)h[e]);
    }
    unsigned char* q = out + (pix * (size_t)(C >> 5) + (size_t)(c >> 5)) * 128 + (size_t)(c & 31) * 2;
    *reinterpret_cast<h16x4_t*>(q) = h;
    *reinterpret_cast<h16x4_t*>(q + 64) = l;
  }
}
__global__ void fill_scale_kernel(float* __restrict__ sc, int n, float inv_prescale, const unsigned* __restrict__ amax) {
  const float v = ldexpf(inv_prescale, -grad_exponent(*amax));
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) sc[i] = v;
}

__global__ void pairs_nhwc8_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int S, int H, int W) {
  const size_t HW = (size_t)H * W;
  const size_t total = (size_t)B * (S - 1) * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pair = i / HW, px = i - pair * HW;
    const int b = (int)(pair / (S - 1)), s = (int)(pair - (size_t)b * (S - 1));
    const float* f0 = img + ((size_t)b * S + s) * 3 * HW + px;   // frames s and s + 1: six consecutive planes (Encoder.py:101)
    float v[8];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = f0[(size_t)c * HW];
    v[6] = v[7] = 0.f;
    float* o = out + i * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}
__global__ void head_grad_permute_kernel(const float* __restrict__ in, float* __restrict__ out, int n_out, int C, int HW) {
  const size_t total = (size_t)n_out * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(i % HW);
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const size_t n = t / C;
    out[i] = in[(n * HW + s) * C + c];
  }
}

inline unsigned ew_blocks(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 16384); }

}  // namespace

hipError_t enc_bn_bwd_reduce(const float* g_a, const void* z, size_t M, int C, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const DropoutSpec& drop, double* partial, float* sums, hipStream_t st) {
  if (!g_a || !z || !mean || !invstd || !gamma || !beta || !partial || !sums || M == 0 || C < 64 || C > 1024 || (C & (C - 1))) return hipErrorInvalidValue;
  const int PL = 256 / (C >> 2);
  const int nblk = (int)std::min<size_t>(1024, (M + PL - 1) / PL);
  (void)hipGetLastError();
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, g_a, reinterpret_cast<const unsigned char*>(z), M, C, mean, invstd, gamma,
                     beta, drop, partial);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, st, partial, nblk, C, sums);
  return hipGetLastError();
}

hipError_t enc_bn_bwd_apply(const float* g_a, const void* z, size_t M, int C, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const DropoutSpec& drop, const float* sums, float* D, hipStream_t st) {
  if (!g_a || !z || !sums || !D || M == 0 || C % 32) return hipErrorInvalidValue;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(M * (size_t)(C >> 2))), dim3(256), 0, st, g_a, reinterpret_cast<const unsigned char*>(z), M, C,
                     mean, invstd, gamma, beta, drop, sums, D);
  return hipGetLastError();
}

size_t enc_wgrad_partial_floats(int Cout, int Cin, int taps, int splits) { return (size_t)splits * taps * Cout * Cin; }

// conv1's shape: the filter row folded into the tile's columns (wgrad_kernel)
static bool wgrad_folds(int Cin, int KW, int x_f32) { return x_f32 && Cin == 8 && KW * 8 <= 64; }

// the layers whose output rows hold whole 32-pixel runs (wgrad_row_kernel)
static bool wgrad_rowwise(int Wo, int KW, int x_f32, int stride) { return !x_f32 && Wo % WR_PX == 0 && (KW == 3 || KW == 5) && (stride == 1 || stride == 2); }

int enc_wgrad_pick_splits(int M, int Cout, int Cin, int taps, int Wo) {
  long tiles = (long)taps * ((Cout + 63) / 64) * ((Cin + 63) / 64);
  if (Cin == 8 && taps == 49) tiles = 7L * ((Cout + 63) / 64);     // (the folded form launches a workgroup per filter ROW)
  if (Cin % 32 == 0 && Wo % WR_PX == 0 && (taps == 9 || taps == 25)) tiles /= (taps == 9 ? 3 : 5);   // (so does the row-wise kernel)
  const int chunks = (M + WG_PX - 1) / WG_PX;
  long s = (2048 + tiles - 1) / tiles;            // ~8 workgroups per CU in flight over the launch
  s = std::max(1L, std::min<long>(s, chunks));
  s = std::min<long>(s, 256);
  return (int)s;
}

hipError_t enc_wgrad(const WgradArgs& a_in, hipStream_t st) {
  WgradArgs a = a_in;
  if (!a.D || !a.x || !a.partial || !a.dW || a.M <= 0 || a.Cout % 4 || a.Cin % 4 || a.splits < 1 || (!a.x_f32 && a.Cin % 32) ||
      (size_t)a.N * a.Ho * a.Wo != (size_t)a.M)
    return hipErrorInvalidValue;
  const int chunks = (a.M + WG_PX - 1) / WG_PX;
  if (a.cin_out <= 0 || a.cin_out > a.Cin) a.cin_out = a.Cin;
  a.chunks_per_split = (chunks + a.splits - 1) / a.splits;
  a.splits = (chunks + a.chunks_per_split - 1) / a.chunks_per_split;
  const int taps = a.KH * a.KW;
  a.fold_kw = wgrad_folds(a.Cin, a.KW, a.x_f32) && getenv("ODEVIO_WGRAD_NO_FOLD") == nullptr ? 1 : 0;
  (void)hipGetLastError();
  if (wgrad_rowwise(a.Wo, a.KW, a.x_f32, a.stride) && getenv("ODEVIO_WGRAD_PER_TAP") == nullptr) {
    // runs of 32 output pixels instead of chunks: the same split arithmetic over runs
    const int runs = a.N * a.Ho * (a.Wo / WR_PX);
    a.chunks_per_split = (runs + a_in.splits - 1) / a_in.splits;
    a.splits = (runs + a.chunks_per_split - 1) / a.chunks_per_split;
    const dim3 grid(a.KH * a.splits, (a.Cout + 63) / 64, (a.Cin + 63) / 64);
    if (a.KW == 3) hipLaunchKernelGGL(wgrad_row_kernel<3>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_row_kernel<5>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_blocks((size_t)a.Cout * a.Cin * taps)), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  if (a.fold_kw)
    hipLaunchKernelGGL(wgrad_kernel, dim3(a.KH * a.splits, (a.Cout + 63) / 64, 1), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(wgrad_kernel, dim3(taps * a.splits, (a.Cout + 63) / 64, (a.Cin + 63) / 64), dim3(256), 0, st, a);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ew_blocks((size_t)a.Cout * a.Cin * taps)), dim3(256), 0, st, a);
  return hipGetLastError();
}

void enc_dilate(const float* D, float* Dd, int N, int Ho, int Wo, int Hd, int Wd, int C, int stride, hipStream_t st) {
  hipLaunchKernelGGL(dilate_kernel, dim3(ew_blocks((size_t)N * Hd * Wd * (C >> 2))), dim3(256), 0, st, D, Dd, N, Ho, Wo, Hd, Wd, C, stride);
}
void enc_pack_dilate(const float* D, void* out, int N, int Ho, int Wo, int Hd, int Wd, int C, int stride, unsigned* amax, float* scale, int n_scale,
                     float inv_prescale, hipStream_t st) {
  (void)hipMemsetAsync(amax, 0, sizeof(unsigned), st);
  const size_t n = (size_t)N * Ho * Wo * C;
  hipLaunchKernelGGL(absmax_kernel, dim3(std::max(1u, std::min(1024u, ew_blocks(n / 16)))), dim3(256), 0, st, D, n, amax);
  hipLaunchKernelGGL(pack_dilate_kernel, dim3(ew_blocks((size_t)N * Hd * Wd * (C >> 2))), dim3(256), 0, st, D, reinterpret_cast<unsigned char*>(out), N, Ho,
                     Wo, Hd, Wd, C, stride, amax);
  hipLaunchKernelGGL(fill_scale_kernel, dim3((n_scale + 255) / 256), dim3(256), 0, st, scale, n_scale, inv_prescale, amax);
}
// out [N][2 Ho][2 Wo][C] <- src [4 classes][N][Ho][Wo][C]: class (py, px) = 2 py + px holds the pixels (2 i + py, 2 j + px)
__global__ void interleave_parity_kernel(const float* __restrict__ src, float* __restrict__ out, int N, int Ho, int Wo, int C) {
  const int Q = C >> 2;
  const size_t per = (size_t)N * Ho * Wo * Q, total = 4 * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % Q);
    size_t px = i / Q;                       // pixel of the OUTPUT grid, row-major
    const int x = (int)(px % (2 * Wo));
    px /= 2 * Wo;
    const int y = (int)(px % (2 * Ho));
    const int n = (int)(px / (2 * Ho));
    const int cl = 2 * (y & 1) + (x & 1);
    const size_t s = (size_t)cl * per + (((size_t)n * Ho + (y >> 1)) * Wo + (x >> 1)) * Q + c4;
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(src)[s];
  }
}
void enc_interleave_parity(const float* src, float* out, int N, int Ho, int Wo, int C, hipStream_t st) {
  hipLaunchKernelGGL(interleave_parity_kernel, dim3(ew_blocks((size_t)N * Ho * Wo * C)), dim3(256), 0, st, src, out, N, Ho, Wo, C);
}
void enc_pairs_nhwc8(const float* img, float* out, int B, int S, int H, int W, hipStream_t st) {
  hipLaunchKernelGGL(pairs_nhwc8_kernel, dim3(ew_blocks((size_t)B * (S - 1) * H * W)), dim3(256), 0, st, img, out, B, S, H, W);
}
void enc_head_grad_permute(const float* in, float* out, int n_out, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(head_grad_permute_kernel, dim3(ew_blocks((size_t)n_out * C * HW)), dim3(256), 0, st, in, out, n_out, C, HW);
}
