// Backward of the ODE-RNN pose path (train.hip): model description and gradient sinks.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "integrator.h"
#include "philox.h"

#define TRAIN_MAX_LIN 6
#define TRAIN_MAX_L 4

// Plain row-major copies of the weights ([N][K]) and their transposes ([K][N]) - the persistent forward kernel keeps
// its own column-sharded layout.
struct TrainModel {
  int F, H, nlin, act, L, with_ode;
  int gru;                // 0: tanh nn.RNN; 1: nn.GRU (gate order r, z, n: weights [3F][F], biases [3F])
  int dims[TRAIN_MAX_LIN + 1];
  const float *ode_w[TRAIN_MAX_LIN], *ode_w_t[TRAIN_MAX_LIN], *ode_b[TRAIN_MAX_LIN];
  const float *rnn_wih[TRAIN_MAX_L], *rnn_wih_t[TRAIN_MAX_L], *rnn_whh[TRAIN_MAX_L], *rnn_whh_t[TRAIN_MAX_L];
  const float *rnn_bih[TRAIN_MAX_L], *rnn_bhh[TRAIN_MAX_L];
  const float *reg_w0, *reg_w0_t, *reg_b0, *reg_w2, *reg_b2;
  // tableau (for FSAL pairs the last stage, which only feeds the error estimate, is left out) and steps per interval:
  // fixed-step solvers take `jmax` = ode_substeps equal steps; adaptive ones replay the forward's ACCEPTED steps from the
  // integrator's log (dtlog [rows][P][dtlog_cap], dtcnt [rows][P]); rows with fewer than `jmax` steps in an interval
  // take zero-length steps for the rest, which change nothing and carry no gradient
  int stages, jmax, adaptive;
  float a[8][8], b[8];
  const float* dtlog;
  const int* dtcnt;
  int dtlog_cap;
  // optional, from a logging forward (integrator.h): ylog [rows][P][dtlog_cap][F] the state each accepted step starts from, yend
  // [rows][P][F] the evolved state of each interval (+ dtcnt).  With them the tape is rebuilt in one batch over all steps.
  const float* ylog;
  const float* yend;
  // optional: the launch-independent arguments of the integrator's adjoint twin (integrator.h); with them the reverse sweep of an
  // interval's Runge-Kutta steps is ONE persistent launch instead of one launch per product
  const IntegAdjArgs* adj;
  // optional (adaptive solvers): per interval the largest number of accepted steps any row took (host array of P ints) - the reverse sweep
  // of an interval skips the zero-length steps behind it
  const int* steps_per_interval;
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

// FusionModule backward; ws: train_fuse_workspace_floats(P, F) floats (soft only).  W [F][F], W_t its transpose.
size_t train_fuse_workspace_floats(int P, int F);
int train_fuse_bwd(int soft, const float* W, const float* W_t, const float* bias, float* ws, const float* fv, int nv, const float* fi, int ni,
                   int P, const float* g_fused, float* g_fv, float* g_fi, float* g_W, float* g_b, hipStream_t st);
// clip_grad_norm_ over n gradient tensors: out2[0] = total norm, out2[1] = clip coefficient (device)
size_t train_grad_clip_workspace_doubles(int n);
int train_grad_clip(const float* const* grads, const size_t* numel, int n, float max_norm, double* partial_ws, float* out2, hipStream_t st);
// one torch.optim.Adam update of one tensor; clip2 = the device pair written by train_grad_clip (or null)
int train_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, int step,
                    const float* clip2, hipStream_t st);
// many tensors, one launch: kind 0 = Adam (s1 = exp_avg, s2 = exp_avg_sq, b1 / b2 the betas), 1 = SGD (s1 = momentum buffer, b1 = momentum)
#define OPT_TABLE_MAX 64
struct OptEntry {
  float* p;
  const float* g;
  float *s1, *s2;
  size_t n;
  float lr;
};
struct OptTable {
  OptEntry e[OPT_TABLE_MAX];
  int n;
};
int train_optimizer_multi(const OptTable& t, int kind, float b1, float b2, float eps, float wd, int step, const float* clip2, hipStream_t st);
// one torch.optim.SGD update (momentum buffer `buf`; step 1 initialises it with the gradient like torch does)
int train_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float wd, int step, const float* clip2,
                   hipStream_t st);
// parameter re-layout on the device (the index maps of odevio_plan_create's host code)
void device_copy_f32(float* dst, const float* src, size_t n, hipStream_t st);
void relayout_transpose(const float* src, float* dst, int N, int K, hipStream_t st);
void relayout_shard(const float* W, float* out, int N, int K, int members, hipStream_t st);
void relayout_rnn(const float* wih, const float* whh, const float* bih, const float* bhh, float* out_w, float* out_b, int F, int gru, int members,
                  hipStream_t st);

// InertialEncoder backward (eval-mode BatchNorm).  Conv weights twice: reference layout [cout][ldk] (rows zero-padded from
// 3 cin to ldk, a multiple of 16) and the forward kernel's [(ci,k)][cout]; s / h = the folded BatchNorm + conv bias.
struct ImuTrain {
  const float *w[3], *wt[3], *s[3], *h[3], *var[3], *mean[3], *bias[3];
  int ldk[3];
  float eps;
  const float* proj_w;   // [i_f_len][2816]
  int i_f_len;
};
struct ImuGrads {       // reference shapes; null = not wanted
  float *w[3], *b[3], *gamma[3], *beta[3], *proj_w, *proj_b;
};
size_t train_imu_workspace_floats(int P);
// The same encoder under model.train(): BatchNorm1d with batch statistics (+ running-statistics update in the forward), Dropout.
// gamma / beta: the BatchNorm affine parameters; run_mean / run_var: the module's buffers, updated in place by the forward (may be
// null); drop[l]: the mask of block l (philox.h).  The backward recomputes the forward with the same masks.
struct ImuTrainMode {
  const float *gamma[3], *beta[3];
  float *run_mean[3], *run_var[3];
  float momentum;
  DropoutSpec drop[3];
};
size_t train_imu_train_workspace_floats(int P);
int train_imu_fwd_train(const ImuTrain& m, const ImuTrainMode& tm, float* ws, const float* imu, int B, int T, const float* proj_b, float* fi,
                        int ld_fi, hipStream_t st);
int train_imu_bwd_train(const ImuTrain& m, const ImuTrainMode& tm, float* ws, const float* imu, int B, int T, const float* g_fi, const ImuGrads& g,
                        hipStream_t st);
// imu [B][T][6], g_fi [B * (T-1)/10][i_f_len] contiguous; g_imu_rows (optional) [(pair, t)][6]: gradient w.r.t. the windowed samples
int train_imu_bwd(const ImuTrain& m, float* ws, const float* imu, int B, int T, const float* g_fi, float* g_imu_rows, const ImuGrads& g,
                  hipStream_t st);
void skinny_linear(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N, int K, hipStream_t st);
// out [N][K] (ldo) = sum over M rows of D[m][n] * A[m][k] (weight gradients: contraction over rows); out [N] = column sums of x [M][N]
void skinny_tn(const float* D, int ldd, const float* A, int lda, float* out, int ldo, int M, int N, int K, hipStream_t st, int accumulate = 0);
// the same contraction split over `splits` row ranges (partial: splits * N * K floats), for M in the millions
void skinny_tn_split(const float* D, int ldd, const float* A, int lda, float* out, int ldo, int M, int N, int K, float* partial, int splits,
                     int accumulate, hipStream_t st);
// out [M][N] (+)= A W^T (+ bias) with an optional fused epilogue: epi 0 none, 1 out = act(.) (ODEFunc.py's activations: 0 tanh, 1 relu,
// 2 leaky_relu 0.01, 3 softplus), 2 out = (.) * act'(aux[m][n]) with aux = the activation's saved OUTPUT.  K % 4 == 0.
void skinny_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo, int M, int N, int K, int accumulate,
               int epi, int act, const float* aux, int ldaux, hipStream_t st);
void colsum_rows(const float* x, float* out, int M, int N, hipStream_t st);
int train_regressor_bwd(const float* seq, int F, const float* w0, const float* w0_t, const float* b0, const float* w2, const float* g_poses, int M,
                        float* ws, float* g_seq, float* g_w0, float* g_b0, float* g_w2, float* g_b2, hipStream_t st);
void leaky_inplace(float* x, size_t n, float slope, hipStream_t st);
void mul_inplace(float* x, const float* y, size_t n, hipStream_t st);   // x *= y
int train_fuse_hard_bwd(const float* W, const float* W_t, const float* bias, float* cat, float* logits, float* g_logits, float* g_cat,
                        const float* fv, int nv, const float* fi, int ni, int P, unsigned long long seed, unsigned long long call,
                        const float* g_fused, float* g_fv, float* g_fi, float* g_W, float* g_b, hipStream_t st);
