// Implicit-GEMM convolution / linear layer on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces the cuDNN/ATen Conv2d + BatchNorm2d(eval) + LeakyReLU(0.1) blocks of the reference's
// ImageEncoder (src/models/Encoder.py:8-22,116-122) for conv2..conv6, and every nn.Linear of the
// path that is not inside the integrator (visual_head, Inertial proj, fusion, regressor).
//
// Tiling: 128(M) x 128(N) x 32(K) per 256-thread workgroup; 4 waves, each a 64x64 sub-tile =
// 2x2 MFMA 32x32 accumulators (64 VGPRs).  K runs over (kh, kw, cin-chunk-of-32): with NHWC
// activations and [Cout][kh][kw][Cin] weights both operands are 128-byte contiguous runs, so the
// im2col matrix is never materialised.  Operands are register-staged (global_load_dwordx4 issued
// one K-tile ahead, ds_write_b128 after the MFMAs) into double-buffered LDS with a 36-float row
// stride, which makes the ds_read_b128 fragment reads conflict-free.  The MFMA's two k-lanes are
// mapped to k = h*16 + kk (h = lane>>5): any bijection works as long as A and B agree, and this
// one lets each lane fetch its 16 k-values of a row with four ds_read_b128.
// The f32 MFMA is an exact k-ordered fmaf chain, so results differ from the CPU oracle only by
// summation order.
#include "common.h"

#define BM 128
#define BN 128
#define BK 32
#define LDS_LD 36  // padded row stride (floats): 144 B = 9 x 16 B, co-prime slot walk

__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (speed only).  Workgroups are dealt round-robin to the 8 XCDs in launch order, so with the
  // plain (x, y) -> (M tile, N tile) map an XCD gets M tiles 8 apart (no shared halo rows) and the N tiles of one
  // M tile run far apart in time: the PMC pass showed ~10x the algorithmic bytes crossing the fabric.  Here every
  // XCD owns a CONTIGUOUS range of M tiles and walks it N-tile-fastest, so the workgroups that share an activation
  // panel (all N tiles of an M tile, and the neighbouring M tiles with their halo) are co-resident on one L2.
  // For >= 16 M tiles the launcher pads gridDim.x to a multiple of 8; the map is then a bijection onto
  // [0, chunk*8) x [0, NT).  Fewer M tiles (the linear layers: M = frame pairs) keep the plain map - the padded one
  // would leave most XCDs without work.
  const int NT = gridDim.y;
  int mt_idx = blockIdx.x, nt_idx = blockIdx.y;
  if (a.xcd_map) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int chunk = gridDim.x >> 3;                 // M tiles per XCD
    mt_idx = xcd * chunk + slot / NT;
    nt_idx = slot - (slot / NT) * NT;
  }
  if (mt_idx * BM >= a.M) return;                     // padding tiles
  const int m0 = mt_idx * BM;
  const int n0 = nt_idx * BN;

  // ---- loader role: rows lrow + 32q, one float4 at column lc4*4
  const int lrow = tid >> 3;
  const int lc4 = (tid & 7) * 4;
  const float* a_row[4];  // pixel (hi0, wi0) of the row's receptive field + this thread's 4 channels (may lie outside the image)
  int a_hi0[4], a_wi0[4];
  const float* b_row[4];
  const int HoWo = a.Ho * a.Wo;
  const int taps = a.KH * a.KW;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + lrow + 32 * q;
    if (m < a.M) {
      const int img = m / HoWo;
      const int rem = m - img * HoWo;
      const int ho = rem / a.Wo;
      const int wo = rem - ho * a.Wo;
      a_hi0[q] = ho * a.stride - a.pad;
      a_wi0[q] = wo * a.stride - a.pad;
      a_row[q] = a.in + ((size_t)img * a.Hi * a.Wi + (ptrdiff_t)a_hi0[q] * a.Wi + a_wi0[q]) * a.Cin + lc4;
    } else {
      a_row[q] = a.in;
      a_hi0[q] = -(1 << 28);
      a_wi0[q] = -(1 << 28);
    }
    const int n = n0 + lrow + 32 * q;
    b_row[q] = (n < a.Cout) ? a.w + (size_t)n * taps * a.Cin + lc4 : nullptr;
  }

  const int nk = taps * (a.Cin / BK);  // K-tiles: (cin chunk, tap)
  int kt_begin = 0, kt_end = nk;
  if (a.splitk > 1) {
    kt_begin = blockIdx.z * a.ktiles_per_split;
    kt_end = min(nk, kt_begin + a.ktiles_per_split);
  }

  f32x4 ra[4], rb[4];
  // K-tile coordinates (workgroup-uniform scalars): tap (kh, kw), offset of the tap+chunk inside a pixel row / a filter
  // K order: cin-chunk MAJOR, filter tap MINOR.  For one 32-channel chunk the kh*kw taps re-read the same few KB
  // of activations shifted by a pixel, back to back, so they hit in L1/L2.  (Tap-major order makes every tap sweep
  // the workgroup's whole 128-pixel x Cin slab - 256 KB per workgroup, 16 MB per XCD in flight.)
  int t_kh = 0, t_kw = 0, t_cc = 0, t_aoff = 0, t_boff = 0;
  auto set_tile = [&](int kt) {  // once, for the first tile of this workgroup
    const int chunk_i = kt / taps;
    const int tap = kt - chunk_i * taps;
    t_cc = chunk_i * BK;
    t_kh = tap / a.KW;
    t_kw = tap - t_kh * a.KW;
    t_aoff = (t_kh * a.Wi + t_kw) * a.Cin + t_cc;
    t_boff = tap * a.Cin + t_cc;
  };
  auto next_tile = [&]() {  // advance (cin-chunk, kh, kw) by one K-tile without divisions
    ++t_kw;
    t_aoff += a.Cin;  // next pixel to the right
    t_boff += a.Cin;  // next tap of the filter
    if (t_kw == a.KW) {
      t_kw = 0;
      ++t_kh;
      t_aoff += (a.Wi - a.KW) * a.Cin;  // first tap of the next filter row
      if (t_kh == a.KH) {
        t_kh = 0;
        t_cc += BK;
        t_aoff = t_cc;
        t_boff = t_cc;
      }
    }
  };
  // Loads are UNCONDITIONAL (out-of-image taps / out-of-range rows read a clamped, valid address) and the zero
  // fill is a select at LDS-store time.  A predicated load (`if (ok) v = *p; else v = 0`) makes hipcc zero the
  // destination first and guard that write with s_waitcnt vmcnt(0): every load then waits for all earlier ones,
  // the eight loads of a tile serialise on memory latency and the in-order wave cannot issue MFMAs meanwhile.
  bool oka[4];
  auto load_a = [&](int q) {
    oka[q] = (unsigned)(a_hi0[q] + t_kh) < (unsigned)a.Hi && (unsigned)(a_wi0[q] + t_kw) < (unsigned)a.Wi;
    const float* p = oka[q] ? a_row[q] + t_aoff : a.in + lc4;
    ra[q] = *reinterpret_cast<const f32x4*>(p);
  };
  auto load_b = [&](int q) {
    const float* p = b_row[q] ? b_row[q] + t_boff : a.w + lc4;
    rb[q] = *reinterpret_cast<const f32x4*>(p);
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto store_a = [&](int buf, int q) {
    *reinterpret_cast<f32x4*>(&As[buf][(lrow + 32 * q) * LDS_LD + lc4]) = oka[q] ? ra[q] : zero4;
  };
  auto store_b = [&](int buf, int q) {
    *reinterpret_cast<f32x4*>(&Bs[buf][(lrow + 32 * q) * LDS_LD + lc4]) = b_row[q] ? rb[q] : zero4;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fi = lane & 31, fh = lane >> 5;
  const int a_off = (wm * 64 + fi) * LDS_LD + fh * 16;
  const int b_off = (wn * 64 + fi) * LDS_LD + fh * 16;

  if (kt_begin < kt_end) {
    set_tile(kt_begin);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      load_a(q);
      load_b(q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      store_a(0, q);
      store_b(0, q);
    }
  }
  __syncthreads();
  // Main loop.  A wave issues in order, and an MFMA occupies the matrix pipe for 64 cycles but the issue port for
  // only a few: everything else this wave has to do for the NEXT tile (8 global loads with their address
  // arithmetic, later the 8 LDS stores, the fragment reads of the next k-group) is placed BETWEEN the MFMAs of the
  // current tile, one piece per slot of four MFMAs, and sched_barrier(0) keeps the compiler from clustering it
  // back in front of the MFMA stream.  Only the first fragment read after the barrier is exposed.
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    // Branch-free body: the last iteration simply re-stages one more (unused) tile.  With an `if (more)` around each
    // piece every piece becomes its own basic block and hipcc opens each with s_waitcnt vmcnt(0).
    if (kt + 1 < kt_end) next_tile();
    const float* Ab = &As[buf][a_off];
    const float* Bb = &Bs[buf][b_off];
    f32x4 fa[2][2], fb[2][2];
    fa[0][0] = *reinterpret_cast<const f32x4*>(Ab);
    fa[0][1] = *reinterpret_cast<const f32x4*>(Ab + 32 * LDS_LD);
    fb[0][0] = *reinterpret_cast<const f32x4*>(Bb);
    fb[0][1] = *reinterpret_cast<const f32x4*>(Bb + 32 * LDS_LD);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1, nxt = cur ^ 1;
      if (g < 3) {
        fa[nxt][0] = *reinterpret_cast<const f32x4*>(Ab + (g + 1) * 4);
        fa[nxt][1] = *reinterpret_cast<const f32x4*>(Ab + 32 * LDS_LD + (g + 1) * 4);
        fb[nxt][0] = *reinterpret_cast<const f32x4*>(Bb + (g + 1) * 4);
        fb[nxt][1] = *reinterpret_cast<const f32x4*>(Bb + 32 * LDS_LD + (g + 1) * 4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int slot = g * 4 + e;
        if (slot < 4) load_a(slot);
        else if (slot < 8) load_b(slot - 4);
        else if (slot < 12) store_a(buf ^ 1, slot - 8);
        else store_b(buf ^ 1, slot - 12);
        // D = W-fragment (MFMA rows = output channels) x pixel-fragment (MFMA columns = pixels): a lane then holds
        // 4 CONSECUTIVE channels of one pixel in registers 4g..4g+3, i.e. one 16-byte NHWC store per group.
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][0][e], fa[cur][0][e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][1][e], fa[cur][0][e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][0][e], fa[cur][1][e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][1][e], fa[cur][1][e], acc[1][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: column (= pixel) = lane&31, row (= channel) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool vec_ok = (a.Cout % 4 == 0) && (a.ld_out % 4 == 0) && (a.mul == nullptr || a.ld_mul % 4 == 0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + wm * 64 + mt * 32 + fi;
    if (m >= a.M) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + nt * 32 + 8 * g + 4 * fh;  // first of 4 consecutive channels
        if (n >= a.Cout) continue;
        f32x4 v = {acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]};
        if (vec_ok) {  // n % 4 == 0 and Cout % 4 == 0  =>  n + 3 < Cout
          if (a.splitk > 1) {
            *reinterpret_cast<f32x4*>(a.partial + ((size_t)blockIdx.z * a.M + m) * a.Cout + n) = v;
          } else {
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
            if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_epi(v[e] * sc[e] + sh[e], a.act, a.slope);
            if (a.mul) {
              const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mul + (size_t)m * a.ld_mul + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= mu[e];
            }
            *reinterpret_cast<f32x4*>(a.out + (size_t)m * a.ld_out + n) = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e >= a.Cout) continue;
            if (a.splitk > 1) {
              a.partial[((size_t)blockIdx.z * a.M + m) * a.Cout + n + e] = v[e];
            } else {
              float x = apply_epi(v[e] * (a.scale ? a.scale[n + e] : 1.f) + (a.shift ? a.shift[n + e] : 0.f), a.act, a.slope);
              if (a.mul) x *= a.mul[(size_t)m * a.ld_mul + n + e];
              a.out[(size_t)m * a.ld_out + n + e] = x;
            }
          }
        }
      }
    }
  }
}

// Deterministic split-K combine: sums the slabs in slab order, then the same epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs a) {
  const size_t total = (size_t)a.M * a.Cout;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / a.Cout);
    const int n = (int)(idx - (size_t)m * a.Cout);
    float v = 0.f;
    for (int z = 0; z < a.splitk; ++z) v += a.partial[(size_t)z * total + idx];
    const float sc = a.scale ? a.scale[n] : 1.f;
    const float sh = a.shift ? a.shift[n] : 0.f;
    v = apply_epi(v * sc + sh, a.act, a.slope);
    if (a.mul) v *= a.mul[(size_t)m * a.ld_mul + n];
    a.out[(size_t)m * a.ld_out + n] = v;
  }
}

void launch_conv_igemm(const ConvArgs& a_in, hipStream_t st) {
  ConvArgs a = a_in;
  const int mt = (a.M + BM - 1) / BM;
  a.xcd_map = mt >= 16;
  dim3 grid(a.xcd_map ? (mt + 7) / 8 * 8 : mt, (a.Cout + BN - 1) / BN, a.splitk > 1 ? a.splitk : 1);  // x padded for the XCD map
  hipLaunchKernelGGL(conv_igemm_kernel, grid, dim3(256), 0, st, a);
  if (a.splitk > 1) {
    const size_t total = (size_t)a.M * a.Cout;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, a);
  }
}

// ------------------------------------------------------------------------------------------------
// conv1: Conv2d(6 -> 64, k7, s2, p3) + BN + LeakyReLU(0.1), reading frame pairs IN PLACE from
// img [B][S][3][H][W] (the reference's torch.cat of consecutive frames, Encoder.py:101, is never
// materialised: frames i and i+1 are adjacent in memory, so the pair's six planes are one run).
//
// Persistent workgroups (one per CU) keep the whole 294 x 64 weight matrix resident in LDS and walk
// 8 x 32-pixel output tiles; the input patch (6 x 21 x 69 floats) is double-buffered in LDS and
// register-prefetched under the previous tile's MFMAs.  K = 294 is split between the MFMA's two
// k-lanes as frame i (k = 0..146) / frame i+1 (k = 147..293), so at k-step s both halves read the
// same (channel, kh, kw) offset from their own frame: every LDS address is lane_base + immediate.
// ------------------------------------------------------------------------------------------------
#define C1_TH 8
#define C1_TW 32
#define C1_PH (2 * C1_TH + 5)       // 21
#define C1_PW (2 * C1_TW + 5)       // 69
#define C1_PATCH (6 * C1_PH * C1_PW)  // 8694 floats
#define C1_PATCH_PER_THREAD ((C1_PATCH + 255) / 256)  // 34
#define C1_KHALF 147

__global__ __launch_bounds__(256) void conv1_kernel(Conv1Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                       // [294][64]
  float* Ps0 = smem + 294 * 64;           // patch buffers
  float* Ps1 = Ps0 + C1_PATCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int fi = lane & 31, fh = lane >> 5;

  for (int i = tid; i < 294 * 64; i += 256) Ws[i] = a.wt[i];

  const int tiles_per_pair = a.tiles_y * a.tiles_x;
  const size_t plane = (size_t)a.H * a.W;
  float stage[C1_PATCH_PER_THREAD];

  auto load_patch = [&](int tile) {
    const int pair = tile / tiles_per_pair;
    const int t = tile - pair * tiles_per_pair;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int b = pair / (a.S - 1), fr = pair - b * (a.S - 1);
    const float* base = a.img + ((size_t)b * a.S + fr) * 3 * plane;
    const int gy0 = 2 * ty * C1_TH - 3, gx0 = 2 * tx * C1_TW - 3;
    // element idx = tid + 256 j of the [6][21][69] patch, walked without divisions: +256 = +3 rows +49 columns
    int x = tid % C1_PW, yy = tid / C1_PW;  // yy = c*21 + y (< 126)
#pragma unroll
    for (int j = 0; j < C1_PATCH_PER_THREAD; ++j) {
      const int c = yy / C1_PH, y = yy - c * C1_PH;  // constant divisor: one mul-hi
      const int gy = gy0 + y, gx = gx0 + x;
      const bool ok = yy < 6 * C1_PH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const float* p = ok ? base + c * plane + (size_t)gy * a.W + gx : base;
      const float v = *p;  // unconditional load from a clamped address (see conv_igemm_kernel)
      stage[j] = ok ? v : 0.f;
      x += 256 % C1_PW;
      yy += 256 / C1_PW;
      if (x >= C1_PW) {
        x -= C1_PW;
        ++yy;
      }
    }
  };
  auto store_patch = [&](float* Ps) {
#pragma unroll
    for (int j = 0; j < C1_PATCH_PER_THREAD; ++j) {
      const int idx = tid + 256 * j;
      if (idx < C1_PATCH) Ps[idx] = stage[j];
    }
  };

  // lane bases: frame half fh owns channels 3*fh..3*fh+2; wave owns output rows 2*wave, 2*wave+1
  const int a_base = (fh * 3) * C1_PH * C1_PW + (4 * wave) * C1_PW + 2 * fi;
  const int b_base = (fh * C1_KHALF) * 64 + fi;

  int tile = blockIdx.x;
  int buf = 0;
  if (tile < a.n_tiles) {
    load_patch(tile);
    store_patch(Ps0);
  }
  __syncthreads();
  for (; tile < a.n_tiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < a.n_tiles) load_patch(next);
    const float* Ps = buf ? Ps1 : Ps0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* Ap = Ps + a_base;
    const float* Bp = Ws + b_base;
    // One wave per SIMD here (the workgroup owns the CU's LDS), so nothing hides an LDS round trip except this
    // wave's own MFMAs: the four operand reads of k-step s+2 are issued before the MFMAs of step s and a
    // sched_barrier per step keeps hipcc from sinking them back next to their use (it then waits lgkmcnt(0)
    // in front of every group of four MFMAs and the matrix pipe idles ~40 % of the time).
    constexpr int C1_PF = 2;  // prefetch distance in k-steps
    float fa0[C1_PF + 1], fa1[C1_PF + 1], fb0[C1_PF + 1], fb1[C1_PF + 1];
    auto frag_off = [](int s) { return ((s / 49) * C1_PH + (s % 49) / 7) * C1_PW + s % 7; };
#pragma unroll
    for (int s = 0; s < C1_PF; ++s) {
      fa0[s] = Ap[frag_off(s)];
      fa1[s] = Ap[frag_off(s) + 2 * C1_PW];
      fb0[s] = Bp[s * 64];
      fb1[s] = Bp[s * 64 + 32];
    }
#pragma unroll
    for (int s = 0; s < C1_KHALF; ++s) {
      const int cur = s % (C1_PF + 1), nxt = (s + C1_PF) % (C1_PF + 1);
      if (s + C1_PF < C1_KHALF) {
        const int off = frag_off(s + C1_PF);
        fa0[nxt] = Ap[off];
        fa1[nxt] = Ap[off + 2 * C1_PW];
        fb0[nxt] = Bp[(s + C1_PF) * 64];
        fb1[nxt] = Bp[(s + C1_PF) * 64 + 32];
      }
      // weights as the MFMA's A operand: channels land on the register axis (16-byte NHWC stores below)
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb0[cur], fa0[cur], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb1[cur], fa0[cur], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb0[cur], fa1[cur], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb1[cur], fa1[cur], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: MFMA columns (lanes) are the 32 pixels of one output row segment, rows (registers) the channels
    {
      const int pair = tile / tiles_per_pair;
      const int t = tile - pair * tiles_per_pair;
      const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
      const int ox = tx * C1_TW + fi;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int oy = ty * C1_TH + 2 * wave + mt;
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
            if (a.out_split) { if (store_pair4(reinterpret_cast<unsigned char*>(a.out), opix, n, 64, v)) a.status[ODEVIO_STATUS_RANGE] = 1; }
            else *reinterpret_cast<f32x4*>(orow + n) = v;
          }
        }
      }
    }
    if (next < a.n_tiles) store_patch(buf ? Ps0 : Ps1);
    buf ^= 1;
    __syncthreads();
  }
}

void launch_conv1(const Conv1Args& a, int n_cu, hipStream_t st) {
  const size_t lds = (size_t)(294 * 64 + 2 * C1_PATCH) * sizeof(float);  // 144,816 B
  static unsigned long long attr_mask = 0;   // per device
  (void)once_per_device(attr_mask, [&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  int grid = n_cu < a.n_tiles ? n_cu : a.n_tiles;
  hipLaunchKernelGGL(conv1_kernel, dim3(grid), dim3(256), lds, st, a);
}
