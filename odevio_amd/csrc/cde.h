// Neural-CDE path: kernel launchers (cde.hip) and the solver schedule (cde_solver.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CDE_BT 16         // batch rows per pass of the last-layer kernels (the N of their 16x16x4 MFMA)
#define CDE_MAX_LIN 6

struct CdeCoefs { double c[8]; };   // tableau coefficients in double: a stage uses fl(a_ij * dt) like the oracle

struct CdeModel {
  int H, C;               // hidden size, control channels (= H + 1)
  int n_hidden, act;      // CDEFunc: n_hidden x [Linear(H,H), act], Linear(H, H*C), Tanh  (ODEFunc.py:52-58)
  const float* w[CDE_MAX_LIN];
  const float* b[CDE_MAX_LIN];
  const void* w_last16;   // optional reduced-precision copy of the last layer (bf16, same [H*C][H] order) or null
  float atol, rtol;       // 1e-6, 1e-4 (PoseCDE.py:101)
  int solver;             // 0 dopri5 (adaptive), 1 rk4 (3/8 rule), 2 euler
  int max_steps;
  int n_cu;
};

// Controller state of the adaptive solver, in device memory.  Written only by the single-thread control kernels
// (cde_ctl_*), read by every kernel of a step attempt: a kernel whose attempt is not wanted any more (`done`) or whose
// stage is not needed (`need_jump_eval` = 0) returns at once, so the host can enqueue attempts ahead without knowing
// how many the controller will take.  Time is double like torchdiffeq's (host) time, the state fp32.
struct CdeCtl {
  double t_begin, tcur, tprev, dt, t1, step;
  double jump_next;         // next knot of the control path after tcur (a jump point of f), or +inf
  int done, status;         // status: 0 or a negative odevio_status
  int n_steps, n_acc, max_steps;
  int p_next, n_out;        // next output time to emit
  int n_knots;              // 2L - 1
  int on_jump, accept, need_jump_eval, have_interp;
  int yi;                   // which of the two state buffers holds y (the other receives y1)
  float dtf, ratio;
  float h0, d0, d1, d2;     // initial-step selection
  float t_stage[8];         // fp32 time handed to the vector field at stage i (slot 0: the re-evaluation after a jump; 7: initial step)
  int seg_stage[8];         // piece of the control path that time falls in
};

// device scratch, all sized for n = B*H floats unless noted
struct CdeWork {
  float *ha, *hb;           // hidden activations
  float *ytmp, *y, *y1;     // y / y1: the two state buffers (CdeCtl::yi says which is which in the adaptive solver)
  float *k;                 // [7][n] stages
  float *interp;            // [5][n] dense-output polynomial
  CdeCtl* ctl;
  double* t_out;            // [n_out] output times (device copy)
  CdeCtl* ctl_host;         // pinned mirror the host polls between batches of attempts
};

// x [B][ldx] -> out [B][N] = act(x W^T + bias); generic sizes (the initial layer: K = C)
void cde_launch_linear(const float* x, int ldx, const float* W, const float* bias, float* out, int B, int K, int N, int act, hipStream_t st);

// What a kernel of the vector field is told about "when": either host values (fixed-grid solvers: ctl = null) or a slot
// of the device controller.
struct CdeWhen {
  const CdeCtl* ctl;   // null: unconditional, seg given by the host
  int slot;            // stage slot of ctl->seg_stage / t_stage
  int seg;             // host-side piece index (ctl == null)
  int only_on_jump;    // run only when ctl->need_jump_eval is set
};

// CDEFunc hidden layer on the fp32 MFMA: out [B][H] = act(x W^T + b), H % 64 == 0
void cde_launch_hidden(const CdeWhen& wh, const float* x, const float* W, const float* bias, float* out, int B, int H, int act, hipStream_t st);
// f(t, z) last layer fused with bias + tanh + the contraction with dX/dt of piece `seg` of the rectilinear control path
// built from obs [B][L][C]:  out[b][h] = sum_c tanh(W[h*C + c] . x[b] + bias[h*C + c]) * dXdt[b][c].
// Even pieces move only the time channel (c = 0): 1 of C weight rows per h takes part (H rows in all);
// odd pieces move the C - 1 feature channels: the whole [H*C, H] matrix streams once.
int cde_launch_last(const CdeModel& m, const CdeWhen& wh, const float* x, const float* obs, int B, int L, float* out, hipStream_t st);
// out = y + scale * sum_j coef[j] * k_j  (k_j = kbase + j*n); scale = host value or ctl->dtf / ctl->h0 (scale_sel 1 / 2);
// y_sel: 0 = `y0` as given, 1 = the controller's current y (y0 / y1 by ctl->yi), -1 = no y term; mirror_y1: also write the
// result into the controller's y1 buffer
void cde_launch_combine(const CdeCtl* ctl, int only_on_jump, const float* y0, const float* y1, int y_sel, const float* kbase, const CdeCoefs& cf,
                        int nk, float scale, int scale_sel, float* out, int mirror_y1, int n, hipStream_t st);
void cde_launch_emit_copy(const float* src, float* sol, int B, int H, int P, int p, hipStream_t st);

// ---- adaptive controller (dopri5), device side
void cde_launch_ctl_init(CdeCtl* ctl, const double* t_out, int n_out, int n_knots, int max_steps, hipStream_t st);
// Hairer initial step (torchdiffeq _select_initial_step, order 4): phase 1 after f0 = f(t0, y0): d0, d1, h0 and the time /
// piece of the probe evaluation (slot 7); phase 2 after f1 = f(t0 + h0, y0 + h0 f0): d2, h1, dt
void cde_launch_init_step(CdeCtl* ctl, int phase, const float* y0, const float* f0, const float* f1, float atol, float rtol, int n, hipStream_t st);
void cde_launch_ctl_begin(CdeCtl* ctl, hipStream_t st);
// err = dtf * sum_j e_j k_j; ratio = rms(err / (atol + rtol * max(|y|, |y1|)))
void cde_launch_err_ratio(CdeCtl* ctl, const float* ya, const float* yb, const float* kbase, const CdeCoefs& e, float atol, float rtol, int n, hipStream_t st);
// accepted step: dense-output polynomial, every output time it covers -> sol, FSAL copy k0 = k6 (unless the step ended on a jump)
void cde_launch_step_finish(const CdeCtl* ctl, const double* t_out, const float* ya, const float* yb, float* kbase, const CdeCoefs& mid,
                            float* interp, float* sol, int B, int H, int n_out, hipStream_t st);
void cde_launch_ctl_update(CdeCtl* ctl, const double* t_out, hipStream_t st);

// Solves dz/dt = CDEFunc(z) . dX/dt from t_out[0], writing z(t_out[p]) to sol[b][p][:].  Fixed-grid solvers never
// synchronise; the adaptive one reads the controller's `done` word once per BATCH of enqueued attempts (not per step).
// Returns 0 or a negative odevio_status; stats = {steps, accepted}.  `hint_steps`: expected attempts (0 = unknown).
struct CdeTape;   // cde_bwd.h: when given, every accepted step is recorded for the backward
int cde_solve(const CdeModel& m, const CdeWork& w, const float* obs, int B, int L, const double* t_out, int n_out,
              const float* z0, float* sol, int* stats, int hint_steps, hipStream_t st, const CdeTape* tape = nullptr);
