// Host-driven ODE solvers of the Neural-CDE path: torchdiffeq 0.2.3's dopri5 (adaptive, ONE step size shared by
// the whole batch, Hairer initial step, 4th-order dense output at the requested times, `jump_t` at the knots of
// the control path) and its fixed-grid euler / rk4 (3/8 rule), restated from the library's published algorithm
// (torchdiffeq is not installable offline, so parity with it is unpinned).  Time lives in
// double on the host, the state in fp32 on the device; the vector field sees time as fp32, one ulp before the end
// of a step (Perturb.PREV) and one ulp after a jump (Perturb.NEXT), exactly as torchdiffeq hands it over.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/odevio.h"
#include "cde.h"

namespace {

const double DP_A[7][6] = {{},
                           {1 / 5.},
                           {3 / 40., 9 / 40.},
                           {44 / 45., -56 / 15., 32 / 9.},
                           {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.},
                           {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.},
                           {35 / 384., 0., 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}};
const double DP_C[7] = {0., 1 / 5., 3 / 10., 4 / 5., 8 / 9., 1., 1.};
const double DP_E[7] = {35 / 384. - 5179 / 57600., 0., 500 / 1113. - 7571 / 16695., 125 / 192. - 393 / 640.,
                        -2187 / 6784. + 92097 / 339200., 11 / 84. - 187 / 2100., -1 / 40.};
const double DP_MID[7] = {6025192743. / 30085553152. / 2, 0., 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
                          187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2};

struct Solver {
  const CdeModel& m;
  const CdeWork& w;
  const float* obs;
  int B, L, n;
  hipStream_t st;
  int cur_seg = -1;

  // f(t, z) -> out.  t is the fp32 time the vector field receives.
  void feval(float t, const float* z, float* out) {
    const int n_knots = 2 * L - 1;
    int seg = (int)std::ceil((double)t) - 1;  // t on a knot belongs to the piece on its left (torch.bucketize)
    seg = std::max(0, std::min(seg, n_knots - 2));
    if (seg != cur_seg) {
      cde_launch_control_grad(obs, w.g, B, L, m.C, seg, st);
      cur_seg = seg;
    }
    const float* x = z;
    float* bufs[2] = {w.ha, w.hb};
    for (int l = 0; l < m.n_hidden; ++l) {
      cde_launch_linear(x, m.H, m.w[l], m.b[l], bufs[l & 1], B, m.H, m.H, m.act, st);
      x = bufs[l & 1];
    }
    cde_launch_last(x, m.w[m.n_hidden], m.b[m.n_hidden], w.g, out, B, m.H, m.C, st);
  }
  float scalar(int slot) {
    float v = 0.f;
    (void)hipMemcpyAsync(&v, w.scal + slot, sizeof(float), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    return v;
  }
  void combine(const float* y, const double* coef, int nk, double scale, float* out) {
    CdeCoefs cf;
    for (int j = 0; j < 8; ++j) cf.c[j] = j < nk ? (float)(coef[j] * scale) : 0.f;
    cde_launch_combine(y, w.k, cf, nk, out, n, st);
  }
};

float f32_prev(float t) { return std::nextafterf(t, t - 1.0f); }
float f32_next(float t) { return std::nextafterf(t, t + 1.0f); }

}  // namespace

int cde_solve(const CdeModel& m, const CdeWork& w, const float* obs, int B, int L, const double* t_out, int n_out,
              const float* z0, float* sol, int* stats, hipStream_t st) {
  CdeWork ww = w;  // local copy: y / y1 are swapped when a step is accepted
  Solver A{m, ww, obs, B, L, B * m.H, st};
  Solver& S = A;
  const int n = A.n;
  auto kj = [&](int j) { return ww.k + (size_t)j * n; };
  (void)hipMemcpyAsync(ww.y, z0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
  cde_launch_emit(nullptr, ww.y, 0.f, sol, B, m.H, n_out, 0, st);  // the first output is z0 itself
  int n_steps = 0, n_acc = 0;

  if (m.solver != 0) {  // fixed grid = the output times
    for (int p = 1; p < n_out; ++p) {
      const double t0 = t_out[p - 1], t1 = t_out[p];
      const float dt = (float)(t1 - t0);
      S.feval((float)t0, ww.y, kj(0));
      if (m.solver == 2) {  // euler
        const double b[1] = {1.0};
        S.combine(ww.y, b, 1, dt, ww.y1);
      } else {  // rk4, 3/8 rule (torchdiffeq rk4_alt_step_func)
        const double a2[1] = {1 / 3.}, a3[2] = {-1 / 3., 1.0}, a4[3] = {1.0, -1.0, 1.0}, b[4] = {0.125, 0.375, 0.375, 0.125};
        S.combine(ww.y, a2, 1, dt, ww.ytmp);
        S.feval((float)(t0 + (double)dt / 3), ww.ytmp, kj(1));
        S.combine(ww.y, a3, 2, dt, ww.ytmp);
        S.feval((float)(t0 + (double)dt * 2 / 3), ww.ytmp, kj(2));
        S.combine(ww.y, a4, 3, dt, ww.ytmp);
        S.feval(f32_prev((float)t1), ww.ytmp, kj(3));
        S.combine(ww.y, b, 4, dt, ww.y1);
      }
      std::swap(ww.y, ww.y1);
      cde_launch_emit(nullptr, ww.y, 0.f, sol, B, m.H, n_out, p, st);
      ++n_steps;
      ++n_acc;
    }
    if (stats) { stats[0] = n_steps; stats[1] = n_acc; }
    (void)hipStreamSynchronize(st);
    return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
  }

  // ---------------- dopri5, adaptive
  const double t_begin = t_out[0];
  float* fy = kj(0);  // f(t, y) lives in stage slot 0 (FSAL)
  A.feval((float)t_begin, ww.y, fy);
  // _select_initial_step(order = 4)
  cde_launch_rms(ww.y, nullptr, ww.y, nullptr, m.atol, m.rtol, 0, n, ww.scal, 0, st);
  cde_launch_rms(fy, nullptr, ww.y, nullptr, m.atol, m.rtol, 0, n, ww.scal, 1, st);
  const float d0 = A.scalar(0), d1 = A.scalar(1);
  float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : (float)(0.01 * (double)d0 / (double)d1);
  {
    const double one[1] = {1.0};
    A.combine(ww.y, one, 1, h0, ww.ytmp);  // y + h0 * f0
    A.feval((float)((float)t_begin + h0), ww.ytmp, ww.fnext);
    cde_launch_rms(ww.fnext, fy, ww.y, nullptr, m.atol, m.rtol, 1, n, ww.scal, 2, st);
  }
  const float d2 = std::fabs(A.scalar(2) / h0);
  float h1;
  if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = std::max(1e-6f, h0 * 1e-3f);
  else h1 = std::pow(0.01f / std::max(d1, d2), 1.0f / 5.0f);
  double dt = std::min(100.0 * (double)h0, (double)h1);
  // jump points: the knots of the control path after t_begin
  std::vector<double> jumps;
  for (int kn = 0; kn < 2 * L - 1; ++kn)
    if ((double)kn > t_begin) jumps.push_back((double)kn);
  size_t ji = 0;
  double tcur = t_begin, tprev = t_begin;
  bool have_interp = false;
  for (int p = 1; p < n_out; ++p) {
    const double target = t_out[p];
    while (target > tcur) {
      if (++n_steps > m.max_steps) return ODEVIO_ERR_MAX_STEPS;
      double step = dt, t1 = tcur + step;
      bool on_jump = false;
      if (!jumps.empty() && tcur < jumps[ji] && jumps[ji] < tcur + step) {
        on_jump = true;
        t1 = jumps[ji];
        step = t1 - tcur;
      }
      const float dtf = (float)step;
      for (int i = 1; i < 7; ++i) {
        float* dst = (i == 6) ? ww.y1 : ww.ytmp;  // the last stage is evaluated at y1 (FSAL)
        A.combine(ww.y, DP_A[i], i, dtf, dst);
        const float ti = (i == 6) ? f32_prev((float)t1) : (float)((float)tcur + DP_C[i] * dtf);
        A.feval(ti, dst, kj(i));
      }
      A.combine(nullptr, DP_E, 7, dtf, ww.err);
      cde_launch_rms(ww.err, nullptr, ww.y, ww.y1, m.atol, m.rtol, 2, n, ww.scal, 3, st);
      const float ratio = A.scalar(3);
      const bool accept = ratio <= 1.0f;
      if (accept) {
        ++n_acc;
        A.combine(ww.y, DP_MID, 7, dtf, ww.ymid);
        cde_launch_interp_fit(ww.y, ww.y1, ww.ymid, kj(0), kj(6), dtf, ww.interp, n, st);
        have_interp = true;
        tprev = tcur;
        tcur = t1;
        std::swap(ww.y, ww.y1);
        if (on_jump) {
          if (ji + 1 != jumps.size()) ++ji;
          A.feval(f32_next((float)tcur), ww.y, kj(0));  // f on the far side of the discontinuity
        } else {
          (void)hipMemcpyAsync(kj(0), kj(6), (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
        }
      }
      // _optimal_step_size
      double factor;
      if (ratio == 0.f) factor = 10.0;
      else {
        const double dfac = ratio < 1.0f ? 1.0 : 0.2;
        factor = std::min(10.0, std::max(0.9 / std::pow((double)ratio, 0.2), dfac));
      }
      dt = step * factor;
    }
    if (!have_interp) return ODEVIO_ERR_BAD_ARG;
    const float x = (float)((target - tprev) / (tcur - tprev));
    cde_launch_emit(ww.interp, nullptr, x, sol, B, m.H, n_out, p, st);
  }
  if (stats) { stats[0] = n_steps; stats[1] = n_acc; }
  (void)hipStreamSynchronize(st);
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
