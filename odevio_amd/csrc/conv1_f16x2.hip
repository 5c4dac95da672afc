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
//   patch   : LDS [piece][6][21][72] fp16, double-buffered, filled by LDS-DMA from the INGESTED frames: ingest_kernel
//             (below) turns the fp32 (or uint8) frames once per forward into zero-bordered fp16-piece planes
//             [frame][channel][piece][Hp][Wp], so a patch row is one aligned 144-byte run that needs no bounds check,
//             no conversion and no ds_write: 36 DMA instructions per tile (7 rows x 9 x 16 B each) instead of ~900
//             load/convert/store instructions - with one wave per SIMD those set the kernel's time, not the MFMAs.
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
#define H1_PIECE_BYTES (H1_ROWS * H1_PW * 2 + 16)      // 18144 + one 16-byte slot for the 64th lane of the last DMA
#define H1_PATCH_BYTES (2 * H1_PIECE_BYTES)            // 36320
#define H1_KROWS 42
#define H1_WPIECE_BYTES (H1_KROWS * 64 * 16)           // 43008
#define H1_W_BYTES (2 * H1_WPIECE_BYTES)               // 86016
#define H1_LDS (H1_W_BYTES + 2 * H1_PATCH_BYTES)       // 158656
#define H1_DMA_PER_WAVE 9                              // 2 pieces x 18 row groups of 7 rows, over 4 waves

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// fp32 [frame][3][H][W] or uint8 [frame][H][W][3] frames -> fp16 pieces [frame][3][2][Hp][Wp] with a zero border:
// padded row yp = y + 3, padded column xp = x + 3.  uint8 input is normalised like the reference's loader
// (ToTensor() - 0.5, src/data/utils.py:359; KITTI_eval.py:100-103): float(byte) / 255 - 0.5 in fp32.
// One thread = 8 consecutive padded columns of one (plane, row): two 16-byte stores.
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a) {
  const int chunks = a.Wp >> 3;
  const size_t total = (size_t)a.n_frames * 3 * a.Hp * chunks;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % chunks);
    size_t r = idx / chunks;
    const int yp = (int)(r % a.Hp);
    const size_t plane = r / a.Hp;           // frame * 3 + c
    const int y = yp - 3, x0 = 8 * k - 3;
    const bool oky = (unsigned)y < (unsigned)a.H;
    f16x8 h, l;
    float v8[8];
    if (!a.src_u8 && (a.W & 3) == 0) {
      // fp32 frames: the 8 pixels x0 .. x0+7 (x0 = 8k - 3) lie inside the three aligned float4s starting at 8k - 4;
      // W % 4 == 0 makes each float4 entirely inside or entirely outside the row
      const float* row = reinterpret_cast<const float*>(a.src) + (plane * a.H + (oky ? y : 0)) * (size_t)a.W;
      float w12[12];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int xq = 8 * k - 4 + 4 * q;
        const bool ok = oky && (unsigned)xq < (unsigned)a.W;
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + (ok ? xq : 0));   // unconditional load, clamped address
                                                                                 // (row 0 / column 0 of the same plane)
#pragma unroll
        for (int e = 0; e < 4; ++e) w12[4 * q + e] = ok ? t[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = w12[e + 1];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int x = x0 + e;
        const bool ok = oky && (unsigned)x < (unsigned)a.W;
        if (a.src_u8) {
          const size_t frame = plane / 3;
          const int c = (int)(plane - frame * 3);
          const unsigned char* sp = reinterpret_cast<const unsigned char*>(a.src);
          const unsigned char b = sp[ok ? ((frame * a.H + y) * a.W + x) * 3 + c : 0];
          v8[e] = ok ? (float)b / 255.0f - 0.5f : 0.f;
        } else {
          const float* sp = reinterpret_cast<const float*>(a.src);
          const float f = sp[ok ? (plane * a.H + y) * a.W + x : 0];
          v8[e] = ok ? f : 0.f;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h[e] = (_Float16)v8[e];
      l[e] = (_Float16)(v8[e] - (float)h[e]);
    }
    _Float16* dst = reinterpret_cast<_Float16*>(a.planes) + ((plane * 2) * a.Hp + yp) * (size_t)a.Wp + 8 * k;
    if (!AUDIT_DST_OK(dst, 16, a.planes, a.planes_bytes, a.status, AK_INGEST_DST) ||
        !AUDIT_DST_OK(dst + (size_t)a.Hp * a.Wp, 16, a.planes, a.planes_bytes, a.status, AK_INGEST_DST)) continue;
    *reinterpret_cast<f16x8*>(dst) = h;
    *reinterpret_cast<f16x8*>(dst + (size_t)a.Hp * a.Wp) = l;
  }
}

void launch_ingest(const IngestArgs& a, hipStream_t st) {
  const size_t total = (size_t)a.n_frames * 3 * a.Hp * (a.Wp >> 3);
  int blocks = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(ingest_kernel, dim3(blocks), dim3(256), 0, st, a);
}

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
  // Patch staging by LDS-DMA.  The 36 instructions of a tile (piece p, row group q: rows 7q .. 7q+6, 9 x 16 B each) are
  // dealt to the waves 9 apiece; which bytes a lane brings does not depend on the tile: its offset from the tile's
  // corner is computed once.  Lane 63 lands 16 bytes behind its group: it brings the first 16 bytes of the NEXT row
  // (what the next instruction puts there anyway), or zeros into the spare slot behind the piece.
  const size_t plane_bytes = (size_t)a.Hp * a.Wp * 2;          // one (channel, piece) plane
  const unsigned char* planes_b = reinterpret_cast<const unsigned char*>(a.planes);
  int d_off[H1_DMA_PER_WAVE];
#pragma unroll
  for (int k = 0; k < H1_DMA_PER_WAVE; ++k) {
    const int g = H1_DMA_PER_WAVE * wave + k, p = g / 18, q = g - 18 * p;
    const int r = 7 * q + (lane == 63 ? 7 : lane / 9), pc = lane == 63 ? 0 : lane % 9;
    const int c = r / H1_PH, y = r - c * H1_PH;
    d_off[k] = r < H1_ROWS ? (int)((c * 2 + p) * plane_bytes) + y * a.Wp * 2 + pc * 16 : -1;
  }
  auto issue_patch_dma = [&](const unsigned char* corner, unsigned char* Pdst, int k) __attribute__((always_inline)) {
    const int g = H1_DMA_PER_WAVE * wave + k, p = g / 18, q = g - 18 * p;
    const unsigned char* src = d_off[k] >= 0 ? corner + d_off[k] : reinterpret_cast<const unsigned char*>(a.zeros);
    src = AUDIT_SRC(src, 16, planes_b, a.planes_bytes, reinterpret_cast<const unsigned char*>(a.zeros), a.status, AK_CONV1_PATCH);
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Pdst + p * H1_PIECE_BYTES + 7 * q * H1_PW * 2), 16, 0, 0);
  };
  auto tile_corner = [&](int tile) __attribute__((always_inline)) {
    const int pair = tile / tiles_per_pair;
    const int t = tile - pair * tiles_per_pair;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int b = pair / (a.S - 1), fr = pair - b * (a.S - 1);
    // padded row of patch row 0 = 16 ty - 3 + 3, padded column of patch column 0 = 64 tx - 3 + 3
    return planes_b + ((size_t)b * a.S + fr) * 6 * plane_bytes + ((size_t)(2 * ty * H1_TH) * a.Wp + 2 * tx * H1_TW) * 2;
  };

  // lane bases.  Activations: output row 2*wave (+1), column fi -> patch row 4*wave (+2) + kh, columns 2*fi .. 2*fi+7.
  const int x_lane = ((4 * wave) * H1_PW + 2 * fi) * 2;
  // Weights: k-row (2s + fh), channel fi (+32): 16 bytes at ((2s + fh) * 64 + fi) * 16
  const int w_lane = (fh * 64 + fi) * 16;

  int tile = blockIdx.x;
  int buf = 0;
  if (tile < a.n_tiles) {
    const unsigned char* corner = tile_corner(tile);
#pragma unroll
    for (int k = 0; k < H1_DMA_PER_WAVE; ++k) issue_patch_dma(corner, Ps0, k);
  }
  __syncthreads();   // drains the DMAs (vmcnt(0)) and the filter-bank copy
  // BatchNorm scale / shift of this lane's 32 channels, once (the epilogue runs at one wave per SIMD: every load it
  // does not have to wait for counts)
  f32x4 e_sc[2][4], e_sh[2][4];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      e_sc[nt][g] = *reinterpret_cast<const f32x4*>(a.scale + nt * 32 + 8 * g + 4 * fh);
      e_sh[nt][g] = *reinterpret_cast<const f32x4*>(a.shift + nt * 32 + 8 * g + 4 * fh);
    }
  for (; tile < a.n_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool more = next < a.n_tiles;
    const unsigned char* corner = tile_corner(more ? next : tile);   // the last tile re-stages itself (unused): branch-free body
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
      if (s < H1_DMA_PER_WAVE) issue_patch_dma(corner, Pn, s);   // next tile's patch, one DMA per k-step
      constexpr int PW[3] = {1, 0, 0};   // l_w h_x, h_w l_x, h_w h_x
      constexpr int PX[3] = {0, 1, 0};
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
        // Every wave is done reading this tile's patch and has its share of the NEXT patch in LDS (own DMAs: vmcnt(0);
        // they were issued in k-steps 0..8).  Raw barriers from here on: a __syncthreads() would also wait for the
        // global stores below, i.e. put an HBM write round trip between every two tiles (1.34 GB leave this kernel).
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned char* stg = const_cast<unsigned char*>(Ps) + wave * (H1_PATCH_BYTES / 4);   // 9072 B >= 32 * 264
        unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int oy = ty * H1_TH + 2 * wave + mt;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 sc = e_sc[nt][g], sh = e_sh[nt][g];
              f16x4 h, l;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float x = acc[mt][nt][4 * g + e] * sc[e] + sh[e];
                x = fmaxf(x, x * a.slope);   // LeakyReLU for 0 < slope < 1
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
              if (px < px_valid) {
                u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                if (AUDIT_DST_OK(outb + opix0 * 256 + o, 16, a.out, a.out_bytes, a.status, AK_CONV1_OUT))
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
    // the staging area (this tile's patch buffer) is refilled by next iteration's DMAs: all LDS reads must be done.
    // The output stores stay in flight.
    if (a.out_split) {
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Two tiles in flight per CU.  The kernel above owns the CU's LDS with ONE 4-wave workgroup, i.e. one wave per SIMD:
// nothing hides the dependent-latency chains of its epilogue (~600 instructions per tile) or the LDS latency of the
// fragment reads.  Here a 512-thread workgroup holds the same filter bank once and runs two independent 4-wave GROUPS,
// each walking its own tiles with its own (single) patch buffer: load patch -> k-loop -> epilogue, unsynchronised
// with the other group, so every SIMD has two waves in different phases.  Groups synchronise internally with a
// monotonic LDS counter (s_barrier would couple the groups); every spin is bounded.
// P2 output only (the fp32-output form of the API uses the kernel above).
// ------------------------------------------------------------------------------------------------------------------
#define H1G_SCALE_BYTES 512
#define H1G_LDS (H1_W_BYTES + 2 * H1_PATCH_BYTES + H1G_SCALE_BYTES + 64)

__device__ __forceinline__ bool group_barrier(unsigned* ctr, unsigned& target, int lane) {
  target += 4;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  unsigned spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 24)) return false;   // ~seconds: never in a healthy run; keeps a broken one from hanging
  }
  asm volatile("" ::: "memory");
  return true;
}

template <int TERMS>   // 3: fp32-grade product; 1: h_w h_x only (ODEVIO_CONV_MATH=f16, see conv_f16x2.hip)
__global__ __launch_bounds__(512) void conv1_f16x2_g2_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem;                                   // [2][42][64][8] fp16
  float* sc_s = reinterpret_cast<float*>(smem + H1_W_BYTES + 2 * H1_PATCH_BYTES);   // scale[64], shift[64]
  unsigned* ctrs = reinterpret_cast<unsigned*>(smem + H1_W_BYTES + 2 * H1_PATCH_BYTES + H1G_SCALE_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int grp = tid >> 8;              // 0 / 1
  const int wave = (tid >> 6) & 3;       // wave inside the group
  const int fi = lane & 31, fh = lane >> 5;
  unsigned char* Pg = smem + H1_W_BYTES + grp * H1_PATCH_BYTES;   // this group's patch buffer
  unsigned* ctr = ctrs + 8 * grp;

  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wt16);
    u32x4* dst = reinterpret_cast<u32x4*>(Ws);
    for (int i = tid; i < H1_W_BYTES / 16; i += 512) dst[i] = src[i];
    if (tid < 64) sc_s[tid] = a.scale[tid];
    else if (tid < 128) sc_s[tid] = a.shift[tid - 64];
    if (tid < 16) ctrs[tid] = 0u;
  }
  __syncthreads();

  const int tiles_per_pair = a.tiles_y * a.tiles_x;
  const size_t plane_bytes = (size_t)a.Hp * a.Wp * 2;
  const unsigned char* planes_b = reinterpret_cast<const unsigned char*>(a.planes);
  int d_off[H1_DMA_PER_WAVE];
#pragma unroll
  for (int k = 0; k < H1_DMA_PER_WAVE; ++k) {
    const int g = H1_DMA_PER_WAVE * wave + k, p = g / 18, q = g - 18 * p;
    const int r = 7 * q + (lane == 63 ? 7 : lane / 9), pc = lane == 63 ? 0 : lane % 9;
    const int c = r / H1_PH, y = r - c * H1_PH;
    d_off[k] = r < H1_ROWS ? (int)((c * 2 + p) * plane_bytes) + y * a.Wp * 2 + pc * 16 : -1;
  }
  const int x_lane = ((4 * wave) * H1_PW + 2 * fi) * 2;
  const int w_lane = (fh * 64 + fi) * 16;
  unsigned target = 0;
  bool healthy = true;
  bool range_bad = false;
  unsigned absmax = 0;   // largest |x| bit pattern this lane stored (NaN / infinity compare above every finite value)
  // (A deliberate half-tile phase offset between the groups was measured 4 % slower than letting them drift.)

  for (int tile = blockIdx.x * 2 + grp; tile < a.n_tiles && healthy; tile += gridDim.x * 2) {
    const int pair = tile / tiles_per_pair;
    const int t = tile - pair * tiles_per_pair;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    {
      const int b = pair / (a.S - 1), fr = pair - b * (a.S - 1);
      const unsigned char* corner =
          planes_b + ((size_t)b * a.S + fr) * 6 * plane_bytes + ((size_t)(2 * ty * H1_TH) * a.Wp + 2 * tx * H1_TW) * 2;
#pragma unroll
      for (int k = 0; k < H1_DMA_PER_WAVE; ++k) {
        const int g = H1_DMA_PER_WAVE * wave + k, p = g / 18, q = g - 18 * p;
        const unsigned char* src = d_off[k] >= 0 ? corner + d_off[k] : reinterpret_cast<const unsigned char*>(a.zeros);
        src = AUDIT_SRC(src, 16, planes_b, a.planes_bytes, reinterpret_cast<const unsigned char*>(a.zeros), a.status, AK_CONV1_PATCH);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Pg + p * H1_PIECE_BYTES + 7 * q * H1_PW * 2), 16, 0, 0);
      }
    }
    healthy = group_barrier(ctr, target, lane);   // own DMAs landed (vmcnt(0) inside) -> the group's patch is complete

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 xf[2][2][2], wf[2][2][2];
    auto krow_off = [](int kr) { return ((kr / 7) * H1_PH + (kr % 7)) * H1_PW * 2; };
    auto read_frags = [&](int set, int s) __attribute__((always_inline)) {
      const int xo = x_lane + (fh ? krow_off(2 * s + 1) : krow_off(2 * s));
      const int wo = w_lane + 2 * s * 64 * 16;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned char* px = Pg + p * H1_PIECE_BYTES + xo + i * 2 * H1_PW * 2;
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
      constexpr int PW[3] = {1, 0, 0};
      constexpr int PX[3] = {0, 1, 0};
#pragma unroll
      for (int tt = 3 - TERMS; tt < 3; ++tt) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][0][PW[tt]], xf[cur][0][PX[tt]], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][1][PW[tt]], xf[cur][0][PX[tt]], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][0][PW[tt]], xf[cur][1][PX[tt]], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][1][PW[tt]], xf[cur][1][PX[tt]], acc[1][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    healthy = healthy && group_barrier(ctr, target, lane);   // every wave of the group is done reading the patch

    // epilogue through this group's patch buffer (see the kernel above), BN constants from LDS
    {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      unsigned char* stg = Pg + wave * (H1_PATCH_BYTES / 4);
      unsigned char* outb = reinterpret_cast<unsigned char*>(a.out);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int oy = ty * H1_TH + 2 * wave + mt;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = nt * 32 + 8 * g + 4 * fh;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(sc_s + n);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(sc_s + 64 + n);
            // vector BatchNorm + LeakyReLU (max(x, slope x), 0 < slope < 1), packed conversions, integer-maximum range check
            f32x4 x = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
            x = x * sc + sh;
            x = __builtin_elementwise_max(x, x * a.slope);
            const f16x4 h = __builtin_convertvector(x, f16x4);
            const f16x4 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), f16x4);
            {
              const u32x4 xb = __builtin_bit_cast(u32x4, x) & 0x7fffffffu;
              absmax = max(absmax, max(max(xb[0], xb[1]), max(xb[2], xb[3])));
            }
            unsigned char* q = stg + fi * 264 + nt * 128 + (8 * g + 4 * fh) * 2;
            *reinterpret_cast<f16x4*>(q) = h;
            *reinterpret_cast<f16x4*>(q + 64) = l;
          }
        }
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
            if (px < px_valid) {
              u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
              if (AUDIT_DST_OK(outb + opix0 * 256 + o, 16, a.out, a.out_bytes, a.status, AK_CONV1_OUT))
                *reinterpret_cast<u32x4*>(outb + opix0 * 256 + o) = v;
            }
          }
        }
      }
    }
    // the staging area is the patch buffer the next iteration's DMAs refill: every wave's staging reads must be done
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    healthy = healthy && group_barrier(ctr, target, lane);
  }
  range_bad |= absmax > 0x477fe000u;   // bits of 65504.0f
  if (range_bad) a.status[ODEVIO_STATUS_RANGE] = 1;
  if (!healthy) a.status[ODEVIO_STATUS_RANGE + 1] = 1;
}

hipError_t launch_conv1_f16x2(const Conv1Args& a, int n_cu, hipStream_t st) {
  // what the kernels assume (their patch DMAs are not bounds-checked): the ingested planes cover every tile's patch
  if (a.Hp != 16 * a.tiles_y + 8 || a.Wp != 64 * a.tiles_x + 8 || a.tiles_y * H1_TH < a.Ho || a.tiles_x * H1_TW < a.Wo ||
      a.planes_bytes < (size_t)a.B * a.S * 6 * a.Hp * a.Wp * 2 || a.n_tiles != a.B * (a.S - 1) * a.tiles_y * a.tiles_x ||
      a.out_bytes < (size_t)a.B * (a.S - 1) * a.Ho * a.Wo * 256 || !a.planes || !a.zeros || !a.wt16 || !a.out)
    return hipErrorInvalidValue;
  static unsigned long long attr_mask = 0;
  {
    const hipError_t e = once_per_device(attr_mask, [] {
      hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, H1_LDS);
      if (e2 != hipSuccess) return e2;
      e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x2_g2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, H1G_LDS);
      if (e2 != hipSuccess) return e2;
      return hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_f16x2_g2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, H1G_LDS);
    });
    if (e != hipSuccess) return e;
  }
  (void)hipGetLastError();
  static const bool single = getenv("ODEVIO_CONV1_SINGLE") != nullptr;   // diagnostic: the one-group kernel
  if (a.out_split && !single) {
    const int pairs_of_tiles = (a.n_tiles + 1) / 2;
    const int grid = n_cu < pairs_of_tiles ? n_cu : pairs_of_tiles;
    if (a.terms == 1) hipLaunchKernelGGL(conv1_f16x2_g2_kernel<1>, dim3(grid), dim3(512), H1G_LDS, st, a);
    else hipLaunchKernelGGL(conv1_f16x2_g2_kernel<3>, dim3(grid), dim3(512), H1G_LDS, st, a);
  } else {
    const int grid = n_cu < a.n_tiles ? n_cu : a.n_tiles;
    hipLaunchKernelGGL(conv1_f16x2_kernel, dim3(grid), dim3(256), H1_LDS, st, a);
  }
  return hipGetLastError();
}
