// conv1 of the FlowNetS stack on the fp16 MFMA: Conv2d(6 -> 64, k7, s2, p3) + BN + LeakyReLU(0.1), reading frame
// pairs IN PLACE from img [B][S][3][H][W] (reference src/models/Encoder.py:101,116: the torch.cat of consecutive
// frames is never materialised - frames i and i+1 are adjacent in memory, so a pair's six planes are one run).
//
// Same arithmetic as conv_f16x2.hip: every fp32 operand is carried as two fp16 pieces x = h + l (2^-22), a product is
// three fp16 MFMAs (h h + h l + l h) accumulated in fp32.
//
// Persistent workgroups (one per CU) keep the whole filter bank in LDS and walk 8 x 32-pixel output tiles.
//   K order : k = (c, kh) row x 8 kw slots; kw = 7 is a zero weight, so a (c, kh) filter row is exactly the 8
//             consecutive k-values one lane feeds to a 32x32x16 MFMA, and the matching activations are 8 CONSECUTIVE
//             input pixels (2*ox .. 2*ox + 7) of one patch row: one 16-byte run of the patch, no gather.
//             42 rows -> 21 k-steps (the MFMA's two lane halves take rows 2s and 2s + 1).
//   weights : LDS [piece][42 rows][64 channels][8 kw] fp16 (16 B per lane and fragment, contiguous over lanes)
//   patch   : LDS [piece][6][21][72] fp16, double-buffered; the next tile's pixels are loaded into registers before the
//             MFMA loop and split / stored after it.
//   wave w  : output rows 2w, 2w + 1 of the tile x 32 columns x 64 channels = 2 x 2 MFMA tiles; 12 MFMAs per k-step,
//             fragments of k-step s + 1 are read while k-step s is multiplied.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2), aligned(4)));   // 4-byte aligned: ds_read2_b32
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define H1_TH 8
#define H1_TW 32
#define H1_PH (2 * H1_TH + 5)             // 21 patch rows
#define H1_PWU 70                         // patch columns in use: 2*31 + 7 = 69 is the last one read
#define H1_PW 72                          // row stride (fp16 elements)
#define H1_ROWS (6 * H1_PH)               // 126 (channel, row) lines
#define H1_PAIRS (H1_ROWS * (H1_PWU / 2)) // 4410 column pairs per patch
#define H1_PAIRS_PER_THREAD ((H1_PAIRS + 255) / 256)   // 18
#define H1_PIECE_BYTES (H1_ROWS * H1_PW * 2)           // 18144
#define H1_PATCH_BYTES (2 * H1_PIECE_BYTES)            // 36288
#define H1_KROWS 42
#define H1_WPIECE_BYTES (H1_KROWS * 64 * 16)           // 43008
#define H1_W_BYTES (2 * H1_WPIECE_BYTES)               // 86016
#define H1_LDS (H1_W_BYTES + 2 * H1_PATCH_BYTES)       // 158592

__global__ __launch_bounds__(256) void conv1_f16x2_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem;                        // [2][42][64][8] fp16
  unsigned char* Ps0 = smem + H1_W_BYTES;          // patch buffers [2][126][72] fp16
  unsigned char* Ps1 = Ps0 + H1_PATCH_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;

  {  // filter bank: already split and laid out by the host (wt16), 86016 B
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wt16);
    u32x4* dst = reinterpret_cast<u32x4*>(Ws);
    for (int i = tid; i < H1_W_BYTES / 16; i += 256) dst[i] = src[i];
  }

  const int tiles_per_pair = a.tiles_y * a.tiles_x;
  const size_t plane = (size_t)a.H * a.W;
  float st0[H1_PAIRS_PER_THREAD], st1[H1_PAIRS_PER_THREAD];   // the next tile's pixels, two per slot

  // The next tile's patch is fetched and stored PIECEWISE inside the MFMA loop (one wave per SIMD: nothing else would
  // hide its ~2000 VALU/memory instructions): column pairs 2s, 2s+1 are loaded in k-step s (s < 9) and split / stored
  // in k-step s + 10, ~4000 cycles later.  Pair idx = tid + 256 j of the [126][35] pair grid; the walk state advances
  // without divisions: +256 = +7 lines +11 pairs.  (A per-thread offset table with a bounds-check-free branch for
  // interior tiles was measured 35 % SLOWER: the branch puts every pair's loads in their own basic block and hipcc
  // opens each with s_waitcnt vmcnt(0).)
  const float* pl_base = a.img;
  int pl_gy0 = 0, pl_gx0 = 0, pl_xp = 0, pl_yy = 0, ps_xp = 0, ps_yy = 0;
  auto patch_begin = [&](int tile) __attribute__((always_inline)) {
    const int pair = tile / tiles_per_pair;
    const int t = tile - pair * tiles_per_pair;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int b = pair / (a.S - 1), fr = pair - b * (a.S - 1);
    pl_base = a.img + ((size_t)b * a.S + fr) * 3 * plane;
    pl_gy0 = 2 * ty * H1_TH - 3;
    pl_gx0 = 2 * tx * H1_TW - 3;
    pl_xp = ps_xp = tid % (H1_PWU / 2);
    pl_yy = ps_yy = tid / (H1_PWU / 2);   // yy = c*21 + y
  };
  auto load_pair = [&](int j) __attribute__((always_inline)) {
    const int c = pl_yy / H1_PH, y = pl_yy - c * H1_PH;  // constant divisor: one mul-hi
    const int gy = pl_gy0 + y, gx = pl_gx0 + 2 * pl_xp;
    const bool oky = pl_yy < H1_ROWS && (unsigned)gy < (unsigned)a.H;
    const bool ok0 = oky && (unsigned)gx < (unsigned)a.W;
    const bool ok1 = oky && (unsigned)(gx + 1) < (unsigned)a.W;
    const float* row = pl_base + c * plane + (size_t)gy * a.W;
    // unconditional loads from clamped addresses, zero fill by select (see conv_igemm_kernel)
    const float v0 = *(ok0 ? row + gx : pl_base);
    const float v1 = *(ok1 ? row + gx + 1 : pl_base);
    st0[j] = ok0 ? v0 : 0.f;
    st1[j] = ok1 ? v1 : 0.f;
    pl_xp += 256 % (H1_PWU / 2);
    pl_yy += 256 / (H1_PWU / 2);
    if (pl_xp >= H1_PWU / 2) {
      pl_xp -= H1_PWU / 2;
      ++pl_yy;
    }
  };
  auto store_pair = [&](unsigned char* Ps, int j) __attribute__((always_inline)) {
    if (ps_yy < H1_ROWS) {
      f16x2 h, l;
      h[0] = (_Float16)st0[j];
      h[1] = (_Float16)st1[j];
      l[0] = (_Float16)(st0[j] - (float)h[0]);
      l[1] = (_Float16)(st1[j] - (float)h[1]);
      const int off = (ps_yy * H1_PW + 2 * ps_xp) * 2;
      *reinterpret_cast<f16x2*>(Ps + off) = h;
      *reinterpret_cast<f16x2*>(Ps + H1_PIECE_BYTES + off) = l;
    }
    ps_xp += 256 % (H1_PWU / 2);
    ps_yy += 256 / (H1_PWU / 2);
    if (ps_xp >= H1_PWU / 2) {
      ps_xp -= H1_PWU / 2;
      ++ps_yy;
    }
  };

  // lane bases.  Activations: output row 2*wave (+1), column fi -> patch row 4*wave (+2) + kh, columns 2*fi .. 2*fi+7.
  const int x_lane = ((4 * wave) * H1_PW + 2 * fi) * 2;
  // Weights: k-row (2s + fh), channel fi (+32): 16 bytes at ((2s + fh) * 64 + fi) * 16
  const int w_lane = (fh * 64 + fi) * 16;

  int tile = blockIdx.x;
  int buf = 0;
  if (tile < a.n_tiles) {
    patch_begin(tile);
#pragma unroll
    for (int j = 0; j < H1_PAIRS_PER_THREAD; ++j) load_pair(j);
#pragma unroll
    for (int j = 0; j < H1_PAIRS_PER_THREAD; ++j) store_pair(Ps0, j);
  }
  __syncthreads();
  for (; tile < a.n_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool more = next < a.n_tiles;
    patch_begin(more ? next : tile);     // the last tile re-stages itself (unused): branch-free loop body
    const unsigned char* Ps = buf ? Ps1 : Ps0;
    unsigned char* Pn = buf ? Ps0 : Ps1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments [set][tile][piece]; x fragments are 16 bytes at 4-byte alignment: two ds_read2_b32 each
    f16x8 xf[2][2][2], wf[2][2][2];
    auto krow_off = [](int kr) { return ((kr / 7) * H1_PH + (kr % 7)) * H1_PW * 2; };   // (c, kh) -> patch line offset
    auto read_frags = [&](int set, int s) __attribute__((always_inline)) {
      const int xo = x_lane + (fh ? krow_off(2 * s + 1) : krow_off(2 * s));
      const int wo = w_lane + 2 * s * 64 * 16;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned char* px = Ps + p * H1_PIECE_BYTES + xo + i * 2 * H1_PW * 2;   // output row +1 = patch row +2
          const u32x2 lo = *reinterpret_cast<const u32x2*>(px);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(px + 8);
          u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
          xf[set][i][p] = __builtin_bit_cast(f16x8, v);
          wf[set][i][p] = *reinterpret_cast<const f16x8*>(Ws + p * H1_WPIECE_BYTES + wo + i * 32 * 16);
        }
      }
    };
    read_frags(0, 0);
#pragma unroll
    for (int s = 0; s < 21; ++s) {
      const int cur = s & 1;
      if (s + 1 < 21) read_frags(cur ^ 1, s + 1);
#ifndef EXP_C1_NOSTAGE
      if (s < 9) {
        load_pair(2 * s);
        load_pair(2 * s + 1);
      } else if (s >= 10 && s < 19) {
        store_pair(Pn, 2 * (s - 10));
        store_pair(Pn, 2 * (s - 10) + 1);
      }
#endif
      constexpr int PW[3] = {1, 0, 0};   // l_w h_x, h_w l_x, h_w h_x
      constexpr int PX[3] = {0, 1, 0};
#ifdef EXP_C1_NOMFMA
      if (s == 0)
#endif
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        // weights as the MFMA's A operand: channels land on the register axis (vector stores below)
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][0][PW[t]], xf[cur][0][PX[t]], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][1][PW[t]], xf[cur][0][PX[t]], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][0][PW[t]], xf[cur][1][PX[t]], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][1][PW[t]], xf[cur][1][PX[t]], acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: MFMA columns (lanes) are the 32 pixels of one output row segment, rows (registers) the channels.
    // Stored straight from that layout a wave instruction would touch 64 different cache lines with 8 bytes each;
    // with P2 output the 32 pixels x 64 channels of a row segment are 8 KB CONTIGUOUS in memory, so the split values
    // take a detour through LDS (the patch just consumed; pixel stride 264 B keeps the 8-byte writes conflict-free)
    // and leave as 16 bytes per lane, 1 KB per wave instruction.
    {
      const int pair = tile / tiles_per_pair;
      const int t = tile - pair * tiles_per_pair;
      const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
      bool range_bad = false;
      if (a.out_split) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        __syncthreads();   // every wave is done reading this tile's patch
        unsigned char* stg = const_cast<unsigned char*>(Ps) + wave * (H1_PATCH_BYTES / 4);   // 9072 B >= 32 * 264
        unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int oy = ty * H1_TH + 2 * wave + mt;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = nt * 32 + 8 * g + 4 * fh;
              const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
              const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
              f16x4 h, l;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float x = acc[mt][nt][4 * g + e] * sc[e] + sh[e];
                x = x > 0.f ? x : x * a.slope;
                range_bad |= !(fabsf(x) <= 65504.f);
                h[e] = (_Float16)x;
                l[e] = (_Float16)(x - (float)h[e]);
              }
              unsigned char* q = stg + fi * 264 + nt * 128 + (8 * g + 4 * fh) * 2;
              *reinterpret_cast<f16x4*>(q) = h;
              *reinterpret_cast<f16x4*>(q + 64) = l;
            }
          }
          // read back in memory order: byte o of the row segment = pixel o / 256, offset o % 256
          if (oy < a.Ho) {
            const size_t opix0 = ((size_t)pair * a.Ho + oy) * a.Wo + tx * H1_TW;
            const int px_valid = min(H1_TW, a.Wo - tx * H1_TW);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
              const int o = (it * 64 + lane) * 16;
              const int px = o >> 8, within = o & 255;
              const unsigned char* q = stg + px * 264 + within;
              const u32x2 lo = *reinterpret_cast<const u32x2*>(q);
              const u32x2 hi = *reinterpret_cast<const u32x2*>(q + 8);
#ifdef EXP_C1_NOSTORE
              if (px < px_valid && lo[0] == 0x12345678u) {
#else
              if (px < px_valid) {
#endif
                u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                *reinterpret_cast<u32x4*>(outb + opix0 * 256 + o) = v;
              }
            }
          }
        }
      } else {
        const int ox = tx * H1_TW + fi;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int oy = ty * H1_TH + 2 * wave + mt;
          if (oy >= a.Ho || ox >= a.Wo) continue;
          const size_t opix = ((size_t)pair * a.Ho + oy) * a.Wo + ox;
          float* orow = reinterpret_cast<float*>(a.out) + opix * 64;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = nt * 32 + 8 * g + 4 * fh;
              const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
              const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = acc[mt][nt][4 * g + e] * sc[e] + sh[e];
                v[e] = x > 0.f ? x : x * a.slope;
              }
              *reinterpret_cast<f32x4*>(orow + n) = v;
            }
          }
        }
      }
      if (range_bad) a.status[ODEVIO_STATUS_RANGE] = 1;
    }
    buf ^= 1;
    __syncthreads();
  }
}

hipError_t launch_conv1_f16x2(const Conv1Args& a, int n_cu, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, H1_LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  (void)hipGetLastError();
  const int grid = n_cu < a.n_tiles ? n_cu : a.n_tiles;
  hipLaunchKernelGGL(conv1_f16x2_kernel, dim3(grid), dim3(256), H1_LDS, st, a);
  return hipGetLastError();
}
