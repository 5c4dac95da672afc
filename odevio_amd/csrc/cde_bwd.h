// Backward of the Neural-CDE path (cde_bwd.hip): the tape a forward solve leaves behind and the reverse sweep over it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "cde.h"

// per accepted step: what the host needs to drive the reverse sweep
struct CdeTapeMeta {
  float dtf;        // step size as the stages saw it
  int on_jump;      // the step ended on a knot of the control path: the next step's first derivative was re-evaluated behind it
  int seg[7];       // piece of the control path each stage's evaluation saw (seg[0]: this step's OWN first derivative, when it evaluated one)
  int p_lo, p_hi;   // outputs p_lo .. p_hi - 1 were emitted by this step
};
struct CdeTape {
  float* y;         // [cap][n]      state at the beginning of the step
  float* k;         // [cap][7][n]   stage derivatives
  CdeTapeMeta* meta;
  float* out_x;     // [n_out] interpolation parameter of each output inside its step
  int* out_step;    // [n_out]
  int* overflow;    // device flag: more accepted steps than cap
  int cap;
};
struct CdeBwdCoef { float c[8]; };
// parameter gradients (device pointers, reference shapes; null = not wanted).  w / b: CDEFunc's Linears, last one at index n_hidden;
// these ACCUMULATE over the vector-field evaluations and are zeroed by the caller.
struct CdeBwdGrads {
  float *w[CDE_MAX_LIN], *b[CDE_MAX_LIN];
  float *init_w, *init_b;
  float *reg_w0, *reg_b0, *reg_w2, *reg_b2;
};

void cde_launch_tape_record(const CdeCtl* ctl, const double* t_out, const float* ya, const float* yb, const float* kbase, const CdeTape& tp, int n,
                            int n_out, hipStream_t st);
size_t cde_bwd_workspace_floats(const CdeModel& m, int B, int n_out, int cap);
// obs [B][L][C]; g_poses [B][n_out][6]; g_z0_out (optional) = gradient of the returned z0; g_obs [B][L][C] out; g_z0_in out when z0_in is given.
int cde_backward(const CdeModel& m, const CdeWork& w, float* ws, const float* obs, int B, int L, const double* t_out, int n_out, const float* z0_in,
                 const float* init_w, const float* init_b, const float* reg_w0, const float* reg_w0_t, const float* reg_b0, const float* reg_w2,
                 const float* g_poses, const float* g_z0_out, float* g_obs, float* g_z0_in, const CdeBwdGrads& g, int cap, int* stats, hipStream_t st);
