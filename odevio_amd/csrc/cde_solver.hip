// Launch schedules of the Neural-CDE solvers: torchdiffeq 0.2.3's dopri5 (adaptive, ONE step size shared by the whole
// batch, Hairer initial step, 4th-order dense output at the requested times, `jump_t` at the knots of the control path)
// and its fixed-grid euler / rk4 (3/8 rule), restated from the library's published algorithm (DESIGN.md section 3.5;
// torchdiffeq is not installable offline, so parity with it is unpinned).
//
// The host only ENQUEUES.  Fixed-grid solvers know their whole schedule up front.  For the adaptive solver every
// decision (error norm, accept / reject, next step size, clipping at a knot, which outputs a step covers, FSAL or
// re-evaluation behind a jump) is taken by the device-side controller (cde.hip, struct CdeCtl): the host enqueues
// step ATTEMPTS ahead - each a fixed sequence of kernels that return at once when the controller says "done" - and
// looks at the controller's `done` word once per batch of attempts, never per step.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/odevio.h"
#include "cde.h"
#include "cde_bwd.h"

namespace {

const CdeCoefs DP_A[7] = {{{}},
                          {{1 / 5.}},
                          {{3 / 40., 9 / 40.}},
                          {{44 / 45., -56 / 15., 32 / 9.}},
                          {{19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.}},
                          {{9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.}},
                          {{35 / 384., 0., 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}}};
const CdeCoefs DP_E = {{35 / 384. - 5179 / 57600., 0., 500 / 1113. - 7571 / 16695., 125 / 192. - 393 / 640.,
                        -2187 / 6784. + 92097 / 339200., 11 / 84. - 187 / 2100., -1 / 40.}};
const CdeCoefs DP_MID = {{6025192743. / 30085553152. / 2, 0., 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
                          187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2}};

struct Field {
  const CdeModel& m;
  const CdeWork& w;
  const float* obs;
  int B, L;
  hipStream_t st;
  // f(t, z) -> out.  `wh` says which piece of the control path (host value or controller slot) and whether the
  // evaluation is wanted at all.
  int eval(const CdeWhen& wh, const float* z, float* out) const {
    const float* x = z;
    float* bufs[2] = {w.ha, w.hb};
    for (int l = 0; l < m.n_hidden; ++l) {
      cde_launch_hidden(wh, x, m.w[l], m.b[l], bufs[l & 1], B, m.H, m.act, st);
      x = bufs[l & 1];
    }
    return cde_launch_last(m, wh, x, obs, B, L, out, st);
  }
};

int host_seg(float t, int L) {
  const int n_knots = 2 * L - 1;
  int seg = (int)std::ceil((double)t) - 1;  // t on a knot belongs to the piece on its left (torch.bucketize)
  return std::max(0, std::min(seg, n_knots - 2));
}
float f32_prev(float t) { return std::nextafterf(t, t - 1.0f); }

}  // namespace

int cde_solve(const CdeModel& m, const CdeWork& w, const float* obs, int B, int L, const double* t_out, int n_out,
              const float* z0, float* sol, int* stats, int hint_steps, hipStream_t st, const CdeTape* tape) {
  const int n = B * m.H;
  Field F{m, w, obs, B, L, st};
  auto kj = [&](int j) { return w.k + (size_t)j * n; };
  for (int p = 1; p < n_out; ++p)
    if (!(t_out[p] > t_out[p - 1])) return ODEVIO_ERR_BAD_ARG;   // strictly ascending output times
  if (hipMemcpyAsync(w.y, z0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return ODEVIO_ERR_HIP;
  cde_launch_emit_copy(w.y, sol, B, m.H, n_out, 0, st);  // the first output is z0 itself
  int n_steps = 0, n_acc = 0;

  if (m.solver != 0) {  // fixed grid = the output times; the host knows every time and piece in advance
    float *y = w.y, *y1 = w.y1;
    for (int p = 1; p < n_out; ++p) {
      const double t0 = t_out[p - 1], t1 = t_out[p];
      const float dt = (float)(t1 - t0);
      auto when = [&](float t) { return CdeWhen{nullptr, 0, host_seg(t, L), 0}; };
      int rc = F.eval(when((float)t0), y, kj(0));
      if (m.solver == 2) {  // euler
        cde_launch_combine(nullptr, 0, y, nullptr, 0, w.k, CdeCoefs{{1.0}}, 1, dt, 0, y1, 0, n, st);
      } else {  // rk4, 3/8 rule (torchdiffeq rk4_alt_step_func)
        cde_launch_combine(nullptr, 0, y, nullptr, 0, w.k, CdeCoefs{{1 / 3.}}, 1, dt, 0, w.ytmp, 0, n, st);
        rc |= F.eval(when((float)(t0 + (double)dt / 3)), w.ytmp, kj(1));
        cde_launch_combine(nullptr, 0, y, nullptr, 0, w.k, CdeCoefs{{-1 / 3., 1.0}}, 2, dt, 0, w.ytmp, 0, n, st);
        rc |= F.eval(when((float)(t0 + (double)dt * 2 / 3)), w.ytmp, kj(2));
        cde_launch_combine(nullptr, 0, y, nullptr, 0, w.k, CdeCoefs{{1.0, -1.0, 1.0}}, 3, dt, 0, w.ytmp, 0, n, st);
        rc |= F.eval(when(f32_prev((float)t1)), w.ytmp, kj(3));
        cde_launch_combine(nullptr, 0, y, nullptr, 0, w.k, CdeCoefs{{0.125, 0.375, 0.375, 0.125}}, 4, dt, 0, y1, 0, n, st);
      }
      if (rc) return ODEVIO_ERR_HIP;
      if (tape) {   // the step's state and stage derivatives for the backward (the host knows the rest of a fixed-grid step)
        if (p - 1 >= tape->cap) return ODEVIO_ERR_MAX_STEPS;
        (void)hipMemcpyAsync(tape->y + (size_t)(p - 1) * n, y, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(tape->k + (size_t)(p - 1) * 7 * n, w.k, (size_t)(m.solver == 2 ? 1 : 4) * n * sizeof(float), hipMemcpyDeviceToDevice, st);
      }
      std::swap(y, y1);
      cde_launch_emit_copy(y, sol, B, m.H, n_out, p, st);
      ++n_steps;
      ++n_acc;
    }
    if (stats) { stats[0] = n_steps; stats[1] = n_acc; }
    return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
  }

  // ---------------- dopri5, adaptive, controller on the device
  if (hipMemcpyAsync(w.t_out, t_out, (size_t)n_out * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return ODEVIO_ERR_HIP;
  cde_launch_ctl_init(w.ctl, w.t_out, n_out, 2 * L - 1, m.max_steps, st);
  int rc = 0;
  // f0 = f(t_begin, y0) lives in stage slot 0 (FSAL); then _select_initial_step(order = 4)
  rc |= F.eval(CdeWhen{w.ctl, 0, 0, 0}, w.y, kj(0));
  cde_launch_init_step(w.ctl, 1, w.y, kj(0), nullptr, m.atol, m.rtol, n, st);
  cde_launch_combine(w.ctl, 0, w.y, nullptr, 0, w.k, CdeCoefs{{1.0}}, 1, 0.f, 2, w.ytmp, 0, n, st);   // y0 + h0 f0
  rc |= F.eval(CdeWhen{w.ctl, 7, 0, 0}, w.ytmp, kj(1));
  cde_launch_init_step(w.ctl, 2, w.y, kj(0), kj(1), m.atol, m.rtol, n, st);
  if (rc) return ODEVIO_ERR_HIP;

  auto enqueue_attempt = [&]() {
    cde_launch_ctl_begin(w.ctl, st);
    for (int i = 1; i < 7; ++i) {
      // stage argument: y + dt * sum_j a_ij k_j -> ytmp; the last stage is evaluated AT y1 (FSAL), which is kept in the
      // controller's y1 buffer as well
      cde_launch_combine(w.ctl, 0, w.y, w.y1, 1, w.k, DP_A[i], i, 0.f, 1, w.ytmp, i == 6 ? 1 : 0, n, st);
      rc |= F.eval(CdeWhen{w.ctl, i, 0, 0}, w.ytmp, kj(i));
    }
    cde_launch_err_ratio(w.ctl, w.y, w.y1, w.k, DP_E, m.atol, m.rtol, n, st);
    if (tape) cde_launch_tape_record(w.ctl, w.t_out, w.y, w.y1, w.k, *tape, n, n_out, st);   // before step_finish's FSAL copy overwrites k0
    cde_launch_step_finish(w.ctl, w.t_out, w.y, w.y1, w.k, DP_MID, w.interp, sol, B, m.H, n_out, st);
    cde_launch_ctl_update(w.ctl, w.t_out, st);
    // f on the far side of a jump (only if the accepted step ended on a knot): argument = the new y, result -> slot 0
    cde_launch_combine(w.ctl, 1, w.y, w.y1, 1, w.k, CdeCoefs{{}}, 0, 0.f, 0, w.ytmp, 0, n, st);
    rc |= F.eval(CdeWhen{w.ctl, 0, 0, 1}, w.ytmp, kj(0));
  };

  // Batches of attempts; one look at the controller per batch.  First batch: what the previous solve of this plan
  // needed (streaming windows repeat), else a guess; later batches are small.
  int enq = 0;
  int batch = hint_steps > 0 ? hint_steps : 8;
  for (;;) {
    for (int i = 0; i < batch; ++i) enqueue_attempt();
    enq += batch;
    if (rc) return ODEVIO_ERR_HIP;
    if (hipMemcpyAsync(w.ctl_host, w.ctl, sizeof(CdeCtl), hipMemcpyDeviceToHost, st) != hipSuccess) return ODEVIO_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return ODEVIO_ERR_HIP;
    if (w.ctl_host->done) break;
    if (enq > m.max_steps) return ODEVIO_ERR_MAX_STEPS;
    batch = 2;
  }
  if (stats) { stats[0] = w.ctl_host->n_steps; stats[1] = w.ctl_host->n_acc; }
  if (w.ctl_host->status) return w.ctl_host->status;
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
