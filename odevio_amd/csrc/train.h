// Backward of the ODE-RNN pose path (train.hip): model description and gradient sinks.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define TRAIN_MAX_LIN 6
#define TRAIN_MAX_L 4

// Plain row-major copies of the weights ([N][K]) and their transposes ([K][N]) - the persistent forward kernel keeps
// its own column-sharded layout.
struct TrainModel {
  int F, H, nlin, act, L, with_ode;
  int dims[TRAIN_MAX_LIN + 1];
  const float *ode_w[TRAIN_MAX_LIN], *ode_w_t[TRAIN_MAX_LIN], *ode_b[TRAIN_MAX_LIN];
  const float *rnn_wih[TRAIN_MAX_L], *rnn_wih_t[TRAIN_MAX_L], *rnn_whh[TRAIN_MAX_L], *rnn_whh_t[TRAIN_MAX_L];
  const float *rnn_bih[TRAIN_MAX_L], *rnn_bhh[TRAIN_MAX_L];
  const float *reg_w0, *reg_w0_t, *reg_b0, *reg_w2, *reg_b2;
  // fixed-step tableau
  int stages, nsub;
  float a[8][8], b[8];
};

// Where the weight gradients go (device pointers, same shapes as the reference's parameters; null = not wanted).
struct TrainGrads {
  float *ode_w[TRAIN_MAX_LIN], *ode_b[TRAIN_MAX_LIN];
  float *rnn_wih[TRAIN_MAX_L], *rnn_whh[TRAIN_MAX_L], *rnn_bih[TRAIN_MAX_L], *rnn_bhh[TRAIN_MAX_L];
  float *reg_w0, *reg_b0, *reg_w2, *reg_b2;
};

size_t train_workspace_floats(const TrainModel& m, int B, int P);
int train_ode_rnn_bwd(const TrainModel& m, float* ws, const float* fused, const float* ts, const float* hc, int B, int P,
                      const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc, const TrainGrads& g,
                      hipStream_t st);
int train_pose_loss(const float* poses, const float* gts, int M, float* loss3, float* grad, hipStream_t st);
