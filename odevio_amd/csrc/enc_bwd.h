// Backward of the image encoder under model.train() (enc_bwd.hip): what loss.backward() does to the nine conv blocks of the
// reference's ImageEncoder (src/models/Encoder.py:8-22,82-90,116-122) when --freeze_encoder is off (scripts/train_model.py:78,84:
// Image_net's gradients then count in clip_grad_norm_):
//   per block, last first:  g_a (grad of the block's output) -> Dropout / LeakyReLU / batch-statistics BatchNorm backward -> D = grad
//   of the bare convolution z -> weight gradient (enc_wgrad: contraction over every pixel of the batch on the fp32 MFMA) and input
//   gradient (api.hip: D packed into the forward's two-fp16-piece layout with a per-tensor power-of-two scale and convolved on the
//   forward's fp16x2 kernel - stride-2 blocks as four parity classes of the undilated gradient, woven together by enc_interleave_parity).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "philox.h"

// ---- BatchNorm2d(training) + LeakyReLU(0.1) + Dropout backward over a P2 tensor z [M][C] (the saved bare convolution)
//   xhat = (z - mean) invstd;  y = gamma xhat + beta;  dz = g_a * dropout(e) * leaky'(y)
//   pass 1 (enc_bn_bwd_reduce): sums[0][c] = sum dz (= g_beta), sums[1][c] = sum dz xhat (= g_gamma); double partials, fixed order
//   pass 2 (enc_bn_bwd_apply):  D = gamma invstd (dz - sums[0] / M - xhat sums[1] / M)          (torch's batch_norm backward, training)
// g_a, D: fp32 [M][C]; partial: 2 * 1024 * C doubles of scratch; element index of the dropout mask = m * C + c.
hipError_t enc_bn_bwd_reduce(const float* g_a, const void* z, size_t M, int C, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, const DropoutSpec& drop, double* partial, float* sums, hipStream_t st);
hipError_t enc_bn_bwd_apply(const float* g_a, const void* z, size_t M, int C, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const DropoutSpec& drop, const float* sums, float* D, hipStream_t st);

// ---- weight gradient of one convolution:  dW[co][ci][kh][kw] = sum over (n, ho, wo) of D[n,ho,wo,co] * x[n, s ho + kh - pad, s wo + kw - pad, ci]
struct WgradArgs {
  const float* D;      // [M][Cout] fp32, M = N * Ho * Wo
  const void* x;       // the convolution's input: P2 [N*Hi*Wi][Cin/32][2][32] fp16 pieces, or (x_f32) fp32 [N*Hi*Wi][ldx]
  int x_f32, ldx;
  float* partial;      // [splits][KH*KW][Cout][Cin] fp32 (one slab per pixel range; combined in slab order)
  float* dW;           // [Cout][cin_out][KH][KW] fp32 - the reference parameter's layout
  int cin_out;         // real input channels (0 = Cin); conv1 presents its 6 channels in 8 slots
  int N, Hi, Wi, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
  int M, splits, chunks_per_split;
  int fold_kw;         // (set by enc_wgrad) conv1's form: the 64 tile columns are (kw, channel slot) of one filter row
};
size_t enc_wgrad_partial_floats(int Cout, int Cin, int taps, int splits);
int enc_wgrad_pick_splits(int M, int Cout, int Cin, int taps, int Wo);
hipError_t enc_wgrad(const WgradArgs& a, hipStream_t st);

// D [N][Ho][Wo][C] -> Dd [N][Hd][Wd][C] with Dd[n][s ho][s wo] = D[n][ho][wo] and zeros elsewhere (the input of the stride-1
// convolution that IS the input gradient of a stride-s convolution)
void enc_dilate(const float* D, float* Dd, int N, int Ho, int Wo, int Hd, int Wd, int C, int stride, hipStream_t st);
// The same dilation straight into the forward kernel's two-fp16-piece layout, scaled by a per-tensor power of two 2^e chosen from
// max|D| on the device (largest element in [2^10, 2^11)); scale[0 .. n_scale) <- inv_prescale * 2^-e = the epilogue scale that undoes
// both the filter's pre-scale and e.  amax: one device word of scratch.
void enc_pack_dilate(const float* D, void* out, int N, int Ho, int Wo, int Hd, int Wd, int C, int stride, unsigned* amax, float* scale, int n_scale,
                     float inv_prescale, hipStream_t st);
// The four parity classes of a stride-2 block's input gradient (each [N][Ho][Wo][C], class 2 py + px = the pixels (2 i + py, 2 j + px)),
// src [4][N][Ho][Wo][C], woven into out [N][2 Ho][2 Wo][C].  C % 4 == 0.
void enc_interleave_parity(const float* src, float* out, int N, int Ho, int Wo, int C, hipStream_t st);
// img [B][S][3][H][W] -> frame pairs as NHWC with 8 channel slots [B*(S-1)][H][W][8] (conv1's input for the weight gradient; slots 6, 7 zero)
void enc_pairs_nhwc8(const float* img, float* out, int B, int S, int H, int W, hipStream_t st);
// visual_head weight gradient from the kernels' (H, W, C) column order back to the reference's flatten order (C, H, W): out[n][c][s] = in[n][s][c]
void enc_head_grad_permute(const float* in, float* out, int n_out, int C, int HW, hipStream_t st);
