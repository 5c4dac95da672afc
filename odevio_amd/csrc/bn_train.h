// Train-mode BatchNorm + Dropout of the two encoders (bn_train.hip, train.hip): the reference trains under model.train()
// (scripts/train_model.py:219), which puts every BatchNorm2d / BatchNorm1d in batch-statistics mode and turns every Dropout on -
// in the frozen Image_net too (src/models/Encoder.py:8-22,43-57,82-90).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "philox.h"

#define BN_MAX_BLOCKS 1024

// Per-channel batch statistics of a P2 activation tensor z [M pixels][C/32][2 pieces][32] (fp32 values carried as two fp16
// pieces) -> scale / shift of y = z * scale + shift = gamma (z - mean) / sqrt(var + eps) + beta (biased variance, like
// torch.nn.functional.batch_norm(training=True)), and the running statistics updated in place the way torch does:
//   running_mean = momentum * mean + (1 - momentum) * running_mean;  running_var likewise with the UNBIASED variance.
// partial: 2 * BN_MAX_BLOCKS * C doubles of scratch.  run_mean / run_var may be null (no update); mean_out / invstd_out (both or
// neither): the batch statistics themselves, kept for the backward.  Sums run in double in a fixed order (deterministic).
hipError_t bn_stats_p2(const void* z, size_t M, int C, double* partial, const float* gamma, const float* beta, float eps, float momentum,
                       float* run_mean, float* run_var, float* scale, float* shift, float* mean_out, float* invstd_out, hipStream_t st);
// out <- split(leaky(z * scale[c] + shift[c], slope) * dropout) (out may be z: in place), element index of the mask = pixel * C + channel (NHWC).
hipError_t bn_apply_p2(const void* z, void* out, size_t M, int C, const float* scale, const float* shift, float slope, const DropoutSpec& drop,
                       int* status, hipStream_t st);
// The same statistics for a row matrix x [rows][C] fp32 (the inertial encoder's (pair, time step) rows): mean / invstd out,
// running statistics updated as above.  One workgroup per channel.
hipError_t bn_stats_rows(const float* x, size_t rows, int C, float eps, float momentum, float* run_mean, float* run_var, float* mean,
                         float* invstd, hipStream_t st);
// Test hook: the factor (0 or 1 / (1 - p)) dropout applies to each of n elements for draw `call` of `seed` -> out [n]
void launch_dropout_dump(float* out, size_t n, const DropoutSpec& d, hipStream_t st);
