// Shared declarations of libodevio's HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// conv1 of the FlowNetS stack, reading frame pairs in place from img [B][S][3][H][W]
struct Conv1Args {
  const float* img;
  const float* wt;      // [294][64]  (k = c*49 + kh*7 + kw, c in 0..5)
  const float* scale;   // [64]
  const float* shift;   // [64]
  float* out;           // NHWC [P][Ho][Wo][64]
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

void launch_conv_igemm(const ConvArgs& a, hipStream_t st);
void launch_conv1(const Conv1Args& a, int n_cu, hipStream_t st);
void launch_imu_convs(const ImuArgs& a, hipStream_t st);
// pose.hip: relative 6-DoF poses -> global 4x4 matrices, one workgroup per drive (all pointers on the device)
hipError_t launch_path_accu(const void* poses, int is_f64, const int64_t* offsets_dev, int n_drives, const double* carry,
                            double* out, hipStream_t stream);

__device__ __forceinline__ float apply_epi(float v, int act, float slope) {
  if (act == EPI_LEAKY) return v > 0.f ? v : v * slope;
  if (act == EPI_TANH) return tanhf(v);
  return v;
}
