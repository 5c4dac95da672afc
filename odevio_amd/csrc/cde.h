// Neural-CDE path: kernel launchers (cde.hip) and the host-driven solver (cde_solver.hip).
#pragma once
#include <hip/hip_runtime.h>

#define CDE_BT 16         // batch rows per pass of cde_last_kernel (the N of its 16x16x4 MFMA)
#define CDE_MAX_LIN 6

struct CdeCoefs { float c[8]; };

struct CdeModel {
  int H, C;               // hidden size, control channels (= H + 1)
  int n_hidden, act;      // CDEFunc: n_hidden x [Linear(H,H), act], Linear(H, H*C), Tanh  (ODEFunc.py:52-58)
  const float* w[CDE_MAX_LIN];
  const float* b[CDE_MAX_LIN];
  float atol, rtol;       // 1e-6, 1e-4 (PoseCDE.py:101)
  int solver;             // 0 dopri5 (adaptive), 1 rk4 (3/8 rule), 2 euler
  int max_steps;
};

// device scratch, all sized for n = B*H floats unless noted
struct CdeWork {
  float *g;               // [B][C] control derivative of the current piece
  float *ha, *hb;         // hidden activations
  float *ytmp, *y, *y1, *ymid, *err, *fnext;
  float *k;               // [7][n] stages
  float *interp;          // [5][n] dense-output polynomial
  float *scal;            // [8] device scalars
};

void cde_launch_linear(const float* x, int ldx, const float* W, const float* bias, float* out, int B, int K, int N, int act, hipStream_t st);
void cde_launch_control_grad(const float* obs, float* g, int B, int L, int C, int seg, hipStream_t st);
void cde_launch_last(const float* x, const float* W, const float* bias, const float* g, float* out, int B, int H, int C, hipStream_t st);
void cde_launch_combine(const float* y, const float* kbase, const CdeCoefs& cf, int nk, float* out, int n, hipStream_t st);
void cde_launch_rms(const float* a, const float* b, const float* y0, const float* y1, float atol, float rtol, int mode, int n,
                    float* scalar, int slot, hipStream_t st);
void cde_launch_interp_fit(const float* y0, const float* y1, const float* ymid, const float* f0, const float* f1, float dt,
                           float* co, int n, hipStream_t st);
void cde_launch_emit(const float* co, const float* src, float x, float* sol, int B, int H, int P, int p, hipStream_t st);

// Solves dz/dt = CDEFunc(z) . dX/dt from t_out[0], writing z(t_out[p]) to sol[b][p][:].  Host-driven: synchronises
// `st` (one scalar read-back per adaptive step).  Returns 0 or a negative odevio_status; stats = {steps, accepted}.
int cde_solve(const CdeModel& m, const CdeWork& w, const float* obs, int B, int L, const double* t_out, int n_out,
              const float* z0, float* sol, int* stats, hipStream_t st);
