// Shared declarations of libodevio's HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum EpilogueAct { EPI_NONE = 0, EPI_LEAKY = 1, EPI_TANH = 2 };

// Implicit-GEMM convolution / linear layer:  out[m][n] = act(scale[n] * sum_k A[m][k] W[n][k] + shift[n]) (* mul[m][n])
//   A[m][k] is gathered on the fly from an NHWC activation tensor (m = (image, ho, wo), k = (kh, kw, cin)).
//   A linear layer is the 1x1 case (Hi = Wi = Ho = Wo = 1, N = rows).
struct ConvArgs {
  const float* in;      // NHWC [N][Hi][Wi][Cin]
  const float* w;       // [Cout][KH][KW][Cin]
  const float* scale;   // [Cout] or nullptr (=1)
  const float* shift;   // [Cout] or nullptr (=0)
  const float* mul;     // optional [M][ld_mul] elementwise multiplier applied last
  float* out;           // [M][ld_out]
  float* partial;       // split-K slabs [splitk][M][Cout] (raw sums) when splitk > 1
  int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int M;                // N*Ho*Wo
  int ld_out, ld_mul;
  int act;              // EpilogueAct
  float slope;
  int splitk, ktiles_per_split;
  int xcd_map;          // set by the launcher: XCD-aware tile order (>= 16 M tiles)
};

// Same convolution with fp32 operands carried as two fp16 pieces (conv_f16x2.hip).
//   in  : P2 layout [pixel][Cin/32][2][32] fp16;  w : [Cout][K-tile][2][32] fp16, K-tile = (channel group, tap), tap minor
//   out : P2 (out_split = 1) or fp32 [M][Cout];  Cin % 32 == 0, Cout % 32 == 0
#define ODEVIO_STATUS_RANGE 2   // word of the plan's status buffer raised when an activation leaves the fp16 range
#define ODEVIO_STATUS_AUDIT 4   // audit build (make AUDIT=1): a kernel computed an address outside its buffers; word 5 = which
#define ODEVIO_ZERO_PAGE_BYTES 256

// Audit build (make AUDIT=1 -> libodevio_audit.so): every address the LDS-DMA / unchecked-load kernels compute is compared
// with the extents of the buffer it must fall in BEFORE it is used; a violation raises the status word, records the
// kernel, and the access is redirected (loads to the zero page, stores dropped), so the run never faults and the GPU
// test-suite run against this library reports instead (tests/conftest.py).  The production
// library compiles the checks out.
#ifdef ODEVIO_AUDIT
__device__ __forceinline__ bool audit_in(const void* p, size_t n, const void* lo, size_t bytes) {
  const uintptr_t a = (uintptr_t)p, b = (uintptr_t)lo;
  return a >= b && a + n <= b + bytes;
}
__device__ __forceinline__ void audit_flag(int* status, int kernel_id) {
  atomicExch(status + ODEVIO_STATUS_AUDIT, 1);
  atomicCAS(status + ODEVIO_STATUS_AUDIT + 1, 0, kernel_id);
}
// load source: must lie in [lo, lo+bytes) or in the zero page
#define AUDIT_SRC(p, n, lo, bytes, zero, status, id) \
  ((audit_in(p, n, lo, bytes) || audit_in(p, n, zero, ODEVIO_ZERO_PAGE_BYTES)) ? (p) : (audit_flag(status, id), (decltype(p))(zero)))
#define AUDIT_DST_OK(p, n, lo, bytes, status, id) (audit_in(p, n, lo, bytes) ? true : (audit_flag(status, id), false))
#else
#define AUDIT_SRC(p, n, lo, bytes, zero, status, id) (p)
#define AUDIT_DST_OK(p, n, lo, bytes, status, id) true
#endif
enum AuditKernel { AK_CONV_A = 1, AK_CONV_B = 2, AK_CONV_OUT = 3, AK_CONV_SLAB = 4, AK_CONV1_PATCH = 5, AK_CONV1_OUT = 6,
                   AK_INGEST_SRC = 7, AK_INGEST_DST = 8, AK_REDUCE = 9 };

struct ConvSplitArgs {
  const void* in;
  const void* w;
  const void* zeros;    // ODEVIO_ZERO_PAGE_BYTES zero bytes: what the LDS-DMA reads for taps outside the image
  size_t in_bytes, w_bytes, out_bytes, partial_bytes;   // extents of in / w / out / partial: checked on the host at
                                                        // every launch, and per access by the audit build
  // 32-bit addressing of the DMA sources (off32 = 1): both `in` and `w` are followed, inside their own allocation, by
  // ODEVIO_ZERO_PAGE_BYTES zero bytes at byte offset in_zero_off / w_zero_off (< 2^32): a lane's source is base (scalar) +
  // a 32-bit offset, and "outside the image" is one more offset instead of a second 64-bit pointer to select from
  unsigned in_zero_off, w_zero_off;
  int off32;
  int stamp;            // diagnostic build (STAMPS=1): 0, or 1 + the linear index of the workgroup whose phase stamps this launch writes
  const float* scale;   // [Cout] (BatchNorm scale with the weights' power-of-two pre-scale folded in)
  const float* shift;   // [Cout]
  void* out;
  float* partial;       // split-K slabs [splitk][M][Cout] fp32 when splitk > 1
  int* status;
  int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int M;
  float slope;          // LeakyReLU slope (the only epilogue the encoder needs)
  int out_split;
  int terms;            // 3 (default): h h + h l + l h; 1: h h only (reduced-precision mode)
  int wide;             // 256 x 256 output tiles (Cout % 256 == 0) instead of 256 x 128
  int ld_out;           // row stride (floats) of an fp32 output (out_split = 0)
  int splitk, ktiles_per_split;
  int xcd_map;
  int bm;               // pixels per tile: 256 (0 = default) or 192
  int m_begin, m_end;   // pixels of the layer this launch covers (m_end = 0: all M); split-K slabs are laid out for all M
};

// conv1 of the FlowNetS stack, reading frame pairs in place from img [B][S][3][H][W]
// Frames -> zero-bordered fp16-piece planes [frame][3][2][Hp][Wp] for conv1_f16x2_kernel (conv1_f16x2.hip)
struct IngestArgs {
  const void* src;      // fp32 [frame][3][H][W], or uint8 [frame][H][W][3] (src_u8: normalised as byte / 255 - 0.5)
  int src_u8;
  void* planes;
  size_t planes_bytes;
  int* status;
  int n_frames, H, W, Hp, Wp;
};

struct Conv1Args {
  const float* img;     // fp32-input MFMA kernel only
  const void* planes;   // ingested frames (fp16x2 kernel): Hp = 16 tiles_y + 8 rows, Wp = 64 tiles_x + 8 columns
  const void* zeros;
  int Hp, Wp;
  const float* wt;      // [294][64]  (k = c*49 + kh*7 + kw, c in 0..5): fp32-input MFMA kernel
  const void* wt16;     // [2 pieces][42 (c,kh) rows][64][8 kw] fp16, pre-scaled: fp16x2 kernel (conv1_f16x2.hip)
  const float* scale;   // [64]
  const float* shift;   // [64]
  void* out;            // NHWC [P][Ho][Wo][64] fp32, or the same pixels in P2 layout (out_split)
  size_t planes_bytes, out_bytes;
  int out_split;
  int terms;            // 3 / 1 piece pairs per product (two-group kernel)
  int* status;
  int B, S, H, W, Ho, Wo;
  int tiles_y, tiles_x, n_tiles;  // per-pair tile grid and total tile count
  float slope;
};

struct ImuArgs {
  const float* imu;     // [B][T][6]
  const float* w1t;     // [6*3][64]    (ci,k) major, co minor
  const float* w2t;     // [64*3][128]
  const float* w3t;     // [128*3][256]
  const float* s1; const float* h1;   // folded conv-bias + BN: y = s*conv + h
  const float* s2; const float* h2;
  const float* s3; const float* h3;
  float* out;           // [P][256*11] in (C,T) order
  int B, T, pairs_per_seq;
};

// hipFuncSetAttribute (dynamic LDS beyond 64 KB) is per device.
// Runs `fn` (a hipFuncSetAttribute sequence) the first time the calling site is used on the current device; one mask per call
// site.  Two host threads may create plans / launch at once: the per-process masks are guarded by one mutex, held while `fn` runs,
// so a second thread never launches before the attribute is set.
inline std::mutex& launch_once_mutex() {
  static std::mutex m;
  return m;
}
template <class Fn>
inline hipError_t once_per_device(unsigned long long& mask, Fn fn) {
  std::lock_guard<std::mutex> guard(launch_once_mutex());
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (mask & bit) return hipSuccess;
  const hipError_t e = fn();
  if (e == hipSuccess) mask |= bit;
  return e;
}

void launch_conv_igemm(const ConvArgs& a, hipStream_t st);
void launch_conv1(const Conv1Args& a, int n_cu, hipStream_t st);
hipError_t launch_conv1_f16x2(const Conv1Args& a, int n_cu, hipStream_t st);
void launch_ingest(const IngestArgs& a, hipStream_t st);
void launch_imu_convs(const ImuArgs& a, hipStream_t st);
hipError_t launch_conv_f16x2(const ConvSplitArgs& a, hipStream_t st);
void launch_pair_pack(const float* in, void* out, size_t pixels, int C, int* status, hipStream_t st);   // fp32 [pixel][C] -> P2
void launch_pair_unpack(const void* in, float* out, size_t pixels, int C, hipStream_t st);              // P2 -> fp32 [pixel][C]
// pose.hip: relative 6-DoF poses -> global 4x4 matrices, one workgroup per drive (all pointers on the device)
hipError_t launch_path_accu(const void* poses, int is_f64, const int64_t* offsets_dev, int n_drives, const double* carry,
                            double* out, hipStream_t stream);

// resize.hip: PIL-exact BILINEAR resize of uint8 HWC frames (tmp: n * Hin * Wout * 3 bytes of scratch)
int resize_u8_launch(const unsigned char* src, int n, int Hin, int Win, unsigned char* dst, int Hout, int Wout, unsigned char* tmp,
                     hipStream_t st);

int resize_table_host(int in_size, int out_size, int* ksize, int* bounds, int* kk, int kk_capacity);

// x = h + l with fp16 pieces (h = fp16(x), l = fp16(x - h); the subtraction is exact), written to the two halves of a
// P2 block [pixel][C/32][2][32] for 4 consecutive channels n..n+3 (n % 4 == 0).  Returns true when a value is outside
// the fp16 range (the caller raises the plan's status word; such an activation cannot be represented).
__device__ __forceinline__ bool store_pair4(unsigned char* base, size_t pixel, int n, int C, f32x4 v) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 h, l;
  bool bad = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bad |= !(fabsf(v[e]) <= 65504.f);
    h[e] = (_Float16)v[e];
    l[e] = (_Float16)(v[e] - (float)h[e]);
  }
  unsigned char* p = base + (pixel * (size_t)(C >> 5) + (size_t)(n >> 5)) * 128 + (n & 31) * 2;
  *reinterpret_cast<f16x4*>(p) = h;
  *reinterpret_cast<f16x4*>(p + 64) = l;
  return bad;
}

__device__ __forceinline__ float apply_epi(float v, int act, float slope) {
  if (act == EPI_LEAKY) return v > 0.f ? v : v * slope;
  if (act == EPI_TANH) return tanhf(v);
  return v;
}

// fuse_method "hard" (api.hip fuse_from_cat): fused = cat * [logit(2j) + g0 >= logit(2j+1) + g1], Gumbel noise from Philox 4x32-10
void launch_hard_mask(const float* cat, const float* logits, float* fused, size_t n, unsigned long long seed, unsigned long long call,
                      hipStream_t st);
void launch_hard_mask_bwd(const float* g, const float* cat, const float* logits, float* g_cat, float* g_logits, size_t n,
                          unsigned long long seed, unsigned long long call, hipStream_t st);
void launch_gumbel_dump(float* out, size_t n, unsigned long long seed, unsigned long long call, hipStream_t st);
