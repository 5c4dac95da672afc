// Backward of the Neural-CDE path: what loss.backward() does to PoseCDE.forward (reference src/models/PoseCDE.py:76-103 with
// adjoint = False, :98-101: plain autograd through torchcde's cdeint -> torchdiffeq's odeint), discretise-then-optimise.
//
// torchcde / torchdiffeq are not installable offline, so - like the forward - this follows the libraries' published algorithm
//  and is checked against torch.autograd through the oracle's restatement: parity with the real libraries is
// UNPINNED.  The step sizes are constants of the differentiation (the controller is detached), as for the ODE-RNN path.
//
// Structure.  The forward solve is run once more with a TAPE: every accepted step leaves its state y, its seven stage derivatives,
// its step size, the piece of the control path each stage saw, whether it ended on a knot, and which outputs it emitted at which
// interpolation parameter.  The reverse sweep then walks the accepted steps backwards:
//   * an output z(t_p) = dense-output polynomial(x_p; y, y1, ymid, k0, k6)  ->  gradients of y and of the stage derivatives;
//   * the next step's carried derivative (FSAL) is this step's k6; after a knot it was re-evaluated: one more vector-field adjoint;
//   * stage i (last first): v = J_f^T g_k[i] at the stage's argument y + dt sum_j a_ij k_j; g_y += v, g_k[j] += dt a_ij v;
// and every vector-field adjoint accumulates the parameter gradients of CDEFunc and the gradient of dX/dt (-> the observations).
// The vector field's adjoint is a handful of skinny fp32-MFMA GEMMs (train.hip) around three element-wise kernels; on an EVEN piece of
// the control path only the time channel moves, so only the H rows h * (H + 1) of the last layer take part (the reference's training
// windows, relative time <= 1 s, never leave piece 0); an odd piece streams the whole [H (H + 1), H] matrix three times.
// Plain launches, no fusion: 14 ms forward + backward for a hidden-1024 training window on piece 0, 1.3 s when it crosses knots.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/odevio.h"
#include "cde.h"
#include "cde_bwd.h"
#include "train.h"

namespace {

#define EW_BLOCKS(n) dim3((unsigned)std::min<size_t>(((size_t)(n) + 255) / 256, 8192)), dim3(256)
#define EW_FOR(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); i += (size_t)gridDim.x * blockDim.x)

// dX/dt of piece `seg` of the rectilinear control path (knots at the integers): coefficient 2i = observation i, coefficient 2i+1 =
// (time of observation i+1, features of observation i); channel 0 is time  (oracle rectilinear_coeffs / cde_field)
__device__ __forceinline__ float coef_at(const float* __restrict__ obs, int b, int L, int C, int knot, int c) {
  const int i = knot >> 1;
  if ((knot & 1) && c == 0) return obs[((size_t)b * L + i + 1) * C];
  return obs[((size_t)b * L + i) * C + c];
}
__device__ __forceinline__ float dxdt(const float* __restrict__ obs, int b, int L, int C, int seg, int c) {
  return coef_at(obs, b, L, C, seg + 1, c) - coef_at(obs, b, L, C, seg, c);
}

// accepted step of the adaptive solver -> tape (launched between the error norm and step_finish: k0 is still this step's own)
__global__ void tape_record_kernel(const CdeCtl* __restrict__ ctl, const double* __restrict__ t_out, const float* __restrict__ ya,
                                   const float* __restrict__ yb, const float* __restrict__ kbase, float* __restrict__ tape_y,
                                   float* __restrict__ tape_k, CdeTapeMeta* __restrict__ meta, float* __restrict__ out_x, int* __restrict__ out_step,
                                   int* __restrict__ overflow, int cap, int n, int n_out) {
  if (ctl->done || !ctl->accept) return;
  const int idx = ctl->n_acc;            // (incremented by ctl_update, which runs after this kernel)
  if (idx >= cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *overflow = 1;
    return;
  }
  const float* y = ctl->yi ? yb : ya;
  EW_FOR(i, n) {
    tape_y[(size_t)idx * n + i] = y[i];
    for (int j = 0; j < 7; ++j) tape_k[((size_t)idx * 7 + j) * n + i] = kbase[(size_t)j * n + i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    CdeTapeMeta m;
    m.dtf = ctl->dtf;
    m.on_jump = ctl->on_jump;
    for (int j = 0; j < 7; ++j) m.seg[j] = ctl->seg_stage[j];
    const double tprev = ctl->tcur, tcur = ctl->t1;
    int p = ctl->p_next;
    m.p_lo = p;
    for (; p < n_out && t_out[p] <= tcur; ++p) {
      out_x[p] = (float)((t_out[p] - tprev) / (tcur - tprev));
      out_step[p] = idx;
    }
    m.p_hi = p;
    meta[idx] = m;
  }
}

// dense-output adjoint: g_y += (wy + wy1 + wym) g;  g_k[j] += c[j] g   (c[j] precomputed on the host from x, dt, the tableau)
__global__ void output_adjoint_kernel(const float* __restrict__ g_sol, int B, int H, int n_out, int p, float wsum, CdeBwdCoef c, float* __restrict__ gy,
                                      float* __restrict__ gk, int n) {
  EW_FOR(i, n) {
    const int b = (int)(i / H), h = (int)(i - (size_t)b * H);
    const float g = g_sol[((size_t)b * n_out + p) * H + h];
    gy[i] += wsum * g;
    for (int j = 0; j < 7; ++j)
      if (c.c[j] != 0.f) gk[(size_t)j * n + i] += c.c[j] * g;
  }
}
// gy += v;  gk[j] += c[j] v  (j < nj)
__global__ void stage_adjoint_kernel(const float* __restrict__ v, CdeBwdCoef c, int nj, float* __restrict__ gy, float* __restrict__ gk, int n) {
  EW_FOR(i, n) {
    const float x = v[i];
    gy[i] += x;
    for (int j = 0; j < nj; ++j)
      if (c.c[j] != 0.f) gk[(size_t)j * n + i] += c.c[j] * x;
  }
}
__global__ void add_kernel(float* __restrict__ x, const float* __restrict__ y, size_t n) { EW_FOR(i, n) x[i] += y[i]; }
__global__ void add_strided_kernel(float* __restrict__ x, const float* __restrict__ g_sol, int B, int H, int n_out, int p) {
  EW_FOR(i, (size_t)B * H) {
    const int b = (int)(i / H), h = (int)(i - (size_t)b * H);
    x[i] += g_sol[((size_t)b * n_out + p) * H + h];
  }
}
// stage argument from the tape: out = y + sum_j c[j] k_j
__global__ void stage_input_kernel(const float* __restrict__ y, const float* __restrict__ k, CdeBwdCoef c, int nj, float* __restrict__ out, int n) {
  EW_FOR(i, n) {
    float acc = 0.f;
    bool first = true;
    for (int j = 0; j < nj; ++j) {
      if (c.c[j] == 0.f) continue;
      const float term = k[(size_t)j * n + i] * c.c[j];
      acc = first ? term : acc + term;
      first = false;
    }
    out[i] = y[i] + acc;
  }
}

// last layer, element-wise part.  Row set: r = 0 .. N-1 <-> matrix row gr = r * rs (rs = C on an even piece: the time-channel rows only;
// rs = 1 on an odd piece), (h, c) = (gr / C, gr % C).  a_raw [B][N] = x W^T (no bias).
//   u = tanh(a_raw + bias[gr]);  g_aT[r][b] = g_f[b][h] * dXdt[b][c] * (1 - u^2)          ([N][Bp] layout, Bp = B rounded up to 4, pad = 0)
__global__ void last_ew_kernel(const float* __restrict__ a_raw, const float* __restrict__ bias, const float* __restrict__ g_f,
                               const float* __restrict__ obs, int B, int Bp, int H, int C, int L, int seg, int N, int rs, float* __restrict__ g_aT) {
  EW_FOR(i, (size_t)N * Bp) {
    const int r = (int)(i / Bp), b = (int)(i - (size_t)r * Bp);
    float v = 0.f;
    if (b < B) {
      const int gr = r * rs, h = gr / C, c = gr - h * C;
      const float u = tanhf(a_raw[(size_t)b * N + r] + bias[gr]);
      v = g_f[(size_t)b * H + h] * dxdt(obs, b, L, C, seg, c) * (1.f - u * u);
    }
    g_aT[i] = v;
  }
}
// g_dX[b][c] = sum_h g_f[b][h] * u[b][h][c] for the channels of the row set, then into the observations: piece seg spans coefficients
// seg, seg + 1 (see coef_at): + to the later one, - to the earlier one.  One thread per (b, channel), h in order (deterministic).
__global__ void gdx_kernel(const float* __restrict__ a_raw, const float* __restrict__ bias, const float* __restrict__ g_f, int B, int H, int C, int L,
                           int seg, int N, int rs, float* __restrict__ g_obs) {
  const int nch = rs == 1 ? C : 1;   // odd piece: every channel (channel 0 has dX/dt = 0 but its gradient exists); even piece: channel 0
  EW_FOR(i, (size_t)B * nch) {
    const int b = (int)(i / nch), c = (int)(i - (size_t)b * nch);
    float s = 0.f;
    for (int h = 0; h < H; ++h) {
      const int gr = h * C + c, r = gr / rs;
      s += g_f[(size_t)b * H + h] * tanhf(a_raw[(size_t)b * N + r] + bias[gr]);
    }
    // d(dXdt[c]) -> coefficients (seg + 1, c) and (seg, c)
    for (int side = 0; side < 2; ++side) {
      const int knot = seg + 1 - side, o = knot >> 1;
      const float sg = side == 0 ? s : -s;
      const size_t idx = ((knot & 1) && c == 0) ? ((size_t)b * L + o + 1) * C : ((size_t)b * L + o) * C + c;
      g_obs[idx] += sg;   // (b, c) pairs are disjoint across threads; the two sides of one thread may hit the same element: sequential
    }
  }
}
// gb[gr] += sum_b g_aT[r][b]
__global__ void rowsum_acc_kernel(const float* __restrict__ g_aT, int N, int Bp, int rs, float* __restrict__ gb) {
  EW_FOR(r, (size_t)N) {
    float s = 0.f;
    for (int b = 0; b < Bp; ++b) s += g_aT[r * Bp + b];
    gb[r * rs] += s;
  }
}
// xT [H][Bp] = x [B][H]^T (zero pad)
__global__ void transpose_pad_kernel(const float* __restrict__ x, float* __restrict__ xT, int B, int Bp, int H) {
  EW_FOR(i, (size_t)H * Bp) {
    const int k = (int)(i / Bp), b = (int)(i - (size_t)k * Bp);
    xT[i] = b < B ? x[(size_t)b * H + k] : 0.f;
  }
}
// g *= act'(a) with a = the activation's output (ODEFunc.py:23-36)
__global__ void dact_kernel(float* __restrict__ g, const float* __restrict__ a, size_t n, int act) {
  EW_FOR(i, n) {
    const float v = a[i];
    float d;
    switch (act) {
      case 0: d = 1.f - v * v; break;
      case 1: d = v > 0.f ? 1.f : 0.f; break;
      case 2: d = v > 0.f ? 1.f : 0.01f; break;
      default: d = v > 20.f ? 1.f : -expm1f(-v); break;
    }
    g[i] *= d;
  }
}
__global__ void colsum_acc_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N) {   // out[n] += sum_m x[m][n]
  EW_FOR(n, (size_t)N) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += x[(size_t)m * N + n];
    out[n] += s;
  }
}
__global__ void tanh_bwd_kernel(float* __restrict__ g, const float* __restrict__ z, size_t n) { EW_FOR(i, n) g[i] *= 1.f - z[i] * z[i]; }
__global__ void obs0_add_kernel(const float* __restrict__ g0, float* __restrict__ g_obs, int B, int L, int C) {   // g_obs[b][0][c] += g0[b][c]
  EW_FOR(i, (size_t)B * C) {
    const int b = (int)(i / C), c = (int)(i - (size_t)b * C);
    g_obs[(size_t)b * L * C + c] += g0[i];
  }
}

struct Vjp {
  const CdeModel& m;
  const CdeBwdGrads& g;
  const float* obs;
  float* g_obs;
  int B, Bp, L;
  float* const* wT;        // transposed hidden weights [H][H]
  float *hbuf, *a_raw, *g_aT, *xT, *gtmp, *partial;   // hbuf: (n_hidden + 1) x [B][H]
  hipStream_t st;
  // v [B][H] = J^T g_f at (z_in, seg); parameter and observation gradients accumulated
  void run(const float* z_in, int seg, const float* g_f, float* v) const {
    const int H = m.H, C = m.C, nh = m.n_hidden, n = B * H;
    (void)hipMemcpyAsync(hbuf, z_in, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
    for (int l = 0; l < nh; ++l)
      skinny_nt(hbuf + (size_t)l * n, H, m.w[l], H, m.b[l], hbuf + (size_t)(l + 1) * n, H, B, H, H, 0, 1, m.act, nullptr, 0, st);
    const float* x = hbuf + (size_t)nh * n;
    const bool even = !(seg & 1);
    const int rs = even ? C : 1, N = even ? H : H * C;
    const float* WL = m.w[nh];
    skinny_nt(x, H, WL, rs * H, nullptr, a_raw, N, B, N, H, 0, 0, 0, nullptr, 0, st);
    hipLaunchKernelGGL(last_ew_kernel, EW_BLOCKS((size_t)N * Bp), 0, st, a_raw, m.b[nh], g_f, obs, B, Bp, H, C, L, seg, N, rs, g_aT);
    hipLaunchKernelGGL(gdx_kernel, EW_BLOCKS((size_t)B * (rs == 1 ? C : 1)), 0, st, a_raw, m.b[nh], g_f, B, H, C, L, seg, N, rs, g_obs);
    // g_x [Bp][H] = sum_r g_aT[r][:] W[r][:]
    const int splits = std::max(1, std::min(256, N / 2048));
    skinny_tn_split(g_aT, Bp, WL, rs * H, gtmp, H, N, Bp, H, partial, splits, 0, st);
    if (g.w[nh]) {
      hipLaunchKernelGGL(transpose_pad_kernel, EW_BLOCKS((size_t)H * Bp), 0, st, x, xT, B, Bp, H);
      skinny_nt(g_aT, Bp, xT, Bp, nullptr, g.w[nh], rs * H, N, H, Bp, 1, 0, 0, nullptr, 0, st);   // gW[gr][:] += g_aT[r][:] x
    }
    if (g.b[nh]) hipLaunchKernelGGL(rowsum_acc_kernel, EW_BLOCKS((size_t)N), 0, st, g_aT, N, Bp, rs, g.b[nh]);
    // hidden layers, last first: gtmp = gradient of the layer's OUTPUT (rows 0 .. B-1 are real)
    for (int l = nh - 1; l >= 0; --l) {
      hipLaunchKernelGGL(dact_kernel, EW_BLOCKS((size_t)n), 0, st, gtmp, hbuf + (size_t)(l + 1) * n, (size_t)n, m.act);
      if (g.w[l]) skinny_tn(gtmp, H, hbuf + (size_t)l * n, H, g.w[l], H, B, H, H, st, 1);
      if (g.b[l]) hipLaunchKernelGGL(colsum_acc_kernel, EW_BLOCKS((size_t)H), 0, st, gtmp, g.b[l], B, H);
      float* dst = l == 0 ? v : a_raw;   // (a_raw is free by now: reuse it as the next gradient buffer)
      skinny_nt(gtmp, H, wT[l], H, nullptr, dst, H, B, H, H, 0, 0, 0, nullptr, 0, st);
      if (l > 0) (void)hipMemcpyAsync(gtmp, dst, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
    if (nh == 0) (void)hipMemcpyAsync(v, gtmp, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
  }
};

const double kA[7][7] = {{},
                         {1 / 5.},
                         {3 / 40., 9 / 40.},
                         {44 / 45., -56 / 15., 32 / 9.},
                         {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.},
                         {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.},
                         {35 / 384., 0., 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.}};
const double kMID[7] = {6025192743. / 30085553152. / 2, 0., 51252292925. / 65400821598. / 2, -2691868925. / 45128329728. / 2,
                        187940372067. / 1594534317056. / 2, -1776094331. / 19743644256. / 2, 11237099. / 235043384. / 2};
// fixed-grid tableaux of cde_solver.hip: euler; rk4 = 3/8 rule
const double kRK4A[4][4] = {{}, {1 / 3.}, {-1 / 3., 1.0}, {1.0, -1.0, 1.0}};
const double kRK4B[4] = {0.125, 0.375, 0.375, 0.125};

int host_seg(float t, int L) {
  const int n_knots = 2 * L - 1;
  int seg = (int)std::ceil((double)t) - 1;
  return std::max(0, std::min(seg, n_knots - 2));
}
float f32_prev(float t) { return std::nextafterf(t, t - 1.0f); }

}  // namespace

void cde_launch_tape_record(const CdeCtl* ctl, const double* t_out, const float* ya, const float* yb, const float* kbase, const CdeTape& tp, int n,
                            int n_out, hipStream_t st) {
  hipLaunchKernelGGL(tape_record_kernel, EW_BLOCKS((size_t)n), 0, st, ctl, t_out, ya, yb, kbase, tp.y, tp.k, tp.meta, tp.out_x, tp.out_step,
                     tp.overflow, tp.cap, n, n_out);
}

size_t cde_bwd_workspace_floats(const CdeModel& m, int B, int n_out, int cap) {
  const size_t n = (size_t)B * m.H, Bp = (size_t)(B + 3) / 4 * 4, NL = (size_t)m.H * m.C;
  size_t f = 0;
  f += (size_t)cap * 8 * n;                         // tape: y + 7 stage derivatives per accepted step
  f += ((size_t)cap * sizeof(CdeTapeMeta) + 15) / 16 * 4 + 2 * (((size_t)n_out + 3) / 4 * 4) + 64;   // (every buffer stays 16-byte aligned)
  f += 12 * n;                                      // gy, gk[7], lam_k0, v, zin, z0 copy
  f += (size_t)(m.n_hidden + 1) * n;                // hbuf
  f += (size_t)B * NL + NL * Bp + (size_t)m.H * Bp + Bp * m.H + 256 * Bp * (size_t)m.H;   // a_raw, g_aT, xT, gtmp, split partials
  f += (size_t)m.n_hidden * m.H * m.H + (size_t)m.C * m.H + (size_t)B * m.C;                // transposed weights, g_obs0
  f += 2 * (size_t)B * n_out * m.H + (size_t)B * n_out * 128 * 2 + 4096;                    // sol, g_sol, regressor scratch
  return f;
}

// Runs the taped forward (cde_solve with tape) and the reverse sweep.  reg: the regressor's weights (W0 [128][H], its transpose, b0, W2).
int cde_backward(const CdeModel& m, const CdeWork& w, float* ws, const float* obs, int B, int L, const double* t_out, int n_out, const float* z0_in,
                 const float* init_w, const float* init_b, const float* reg_w0, const float* reg_w0_t, const float* reg_b0, const float* reg_w2,
                 const float* g_poses, const float* g_z0_out, float* g_obs, float* g_z0_in, const CdeBwdGrads& g, int cap, int* stats, hipStream_t st) {
  const int H = m.H, C = m.C, nh = m.n_hidden, n = B * H, Bp = (B + 3) / 4 * 4;
  const size_t NL = (size_t)H * C;
  // ---- carve
  float* q = ws;
  CdeTape tp;
  tp.cap = cap;
  tp.y = q; q += (size_t)cap * n;
  tp.k = q; q += (size_t)cap * 7 * n;
  tp.meta = reinterpret_cast<CdeTapeMeta*>(q); q += ((size_t)cap * sizeof(CdeTapeMeta) + 15) / 16 * 4;
  tp.out_x = q; q += ((size_t)n_out + 3) / 4 * 4;
  tp.out_step = reinterpret_cast<int*>(q); q += ((size_t)n_out + 3) / 4 * 4;
  tp.overflow = reinterpret_cast<int*>(q); q += 64;
  float* gy = q; q += n;
  float* gk = q; q += 7 * (size_t)n;
  float* lam_k0 = q; q += n;
  float* v = q; q += n;
  float* zin = q; q += n;
  float* z0 = q; q += n;
  float* hbuf = q; q += (size_t)(nh + 1) * n;
  float* a_raw = q; q += (size_t)B * NL;
  float* g_aT = q; q += NL * Bp;
  float* xT = q; q += (size_t)H * Bp;
  float* gtmp = q; q += (size_t)Bp * H;
  float* partial = q; q += 256 * (size_t)Bp * H;
  std::vector<float*> wT(nh);
  for (int l = 0; l < nh; ++l) { wT[l] = q; q += (size_t)H * H; }
  float* initT = q; q += (size_t)C * H;      // [C][H]
  float* g_obs0 = q; q += (size_t)B * C;
  float* sol = q; q += (size_t)B * n_out * H;
  float* g_sol = q; q += (size_t)B * n_out * H;
  float* reg_ws = q; q += (size_t)B * n_out * 128 * 2;

  (void)hipMemsetAsync(tp.overflow, 0, sizeof(int), st);
  (void)hipMemsetAsync(g_obs, 0, (size_t)B * L * C * sizeof(float), st);
  (void)hipMemsetAsync(gtmp, 0, (size_t)Bp * H * sizeof(float), st);
  // ---- z0 and the taped forward
  if (z0_in) (void)hipMemcpyAsync(z0, z0_in, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
  else cde_launch_linear(obs, L * C, init_w, init_b, z0, B, C, H, 0 /*tanh*/, st);
  int fstats[2] = {0, 0};
  int rc = cde_solve(m, w, obs, B, L, t_out, n_out, z0, sol, fstats, 0, st, &tp);
  if (rc) return rc;
  if (stats) { stats[0] = fstats[0]; stats[1] = fstats[1]; }
  // ---- what the host needs of the tape: per-step scalars (the fixed-grid solvers filled them on the host already)
  const int n_steps = fstats[1];
  if (n_steps > cap) return ODEVIO_ERR_MAX_STEPS;
  std::vector<CdeTapeMeta> meta(std::max(n_steps, 1));
  std::vector<float> out_x(n_out, 1.f);
  std::vector<int> out_step(n_out, -1);
  if (m.solver == 0) {
    int ovf = 0;
    if (hipMemcpyAsync(meta.data(), tp.meta, (size_t)n_steps * sizeof(CdeTapeMeta), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(out_x.data(), tp.out_x, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(out_step.data(), tp.out_step, (size_t)n_out * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&ovf, tp.overflow, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return ODEVIO_ERR_HIP;
    if (ovf) return ODEVIO_ERR_MAX_STEPS;
  } else {
    for (int s = 0; s < n_steps; ++s) {
      const double t0 = t_out[s], t1 = t_out[s + 1];
      const float dt = (float)(t1 - t0);
      CdeTapeMeta& mm = meta[s];
      memset(&mm, 0, sizeof(mm));
      mm.dtf = dt; mm.on_jump = 0; mm.p_lo = s + 1; mm.p_hi = s + 2;
      mm.seg[0] = host_seg((float)t0, L);
      if (m.solver == 1) {
        mm.seg[1] = host_seg((float)(t0 + (double)dt / 3), L);
        mm.seg[2] = host_seg((float)(t0 + (double)dt * 2 / 3), L);
        mm.seg[3] = host_seg(f32_prev((float)t1), L);
      }
      out_step[s + 1] = s;
    }
  }
  // ---- regressor backward: g_sol [B][n_out][H]
  rc = train_regressor_bwd(sol, H, reg_w0, reg_w0_t, reg_b0, reg_w2, g_poses, B * n_out, reg_ws, g_sol, g.reg_w0, g.reg_b0, g.reg_w2, g.reg_b2, st);
  if (rc) return rc;
  for (int l = 0; l < nh; ++l) relayout_transpose(m.w[l], wT[l], H, H, st);
  Vjp vjp{m, g, obs, g_obs, B, Bp, L, wT.data(), hbuf, a_raw, g_aT, xT, gtmp, partial, st};

  const bool adaptive = m.solver == 0;
  const int S = adaptive ? 7 : (m.solver == 1 ? 4 : 1);
  auto a_of = [&](int i, int j) { return adaptive ? kA[i][j] : (m.solver == 1 ? kRK4A[i][j] : 0.0); };
  auto b_of = [&](int j) { return adaptive ? kA[6][j] : (m.solver == 1 ? kRK4B[j] : 1.0); };
  (void)hipMemsetAsync(gy, 0, (size_t)n * sizeof(float), st);
  (void)hipMemsetAsync(lam_k0, 0, (size_t)n * sizeof(float), st);
  for (int s = n_steps - 1; s >= 0; --s) {
    const CdeTapeMeta& mm = meta[s];
    const float dtf = mm.dtf;
    const float* ty = tp.y + (size_t)s * n;
    const float* tk = tp.k + (size_t)s * 7 * n;
    // entering: gy = gradient of the state AFTER this step (y1), lam_k0 = gradient of the derivative carried into the next step
    (void)hipMemsetAsync(gk, 0, 7 * (size_t)n * sizeof(float), st);
    const bool next_fresh = adaptive && mm.on_jump;    // the next step's k0 was re-evaluated behind the knot, at y1
    if (adaptive && s + 1 < n_steps) {
      if (next_fresh) {
        // k0_next = f(seg0_next, y1): its adjoint lands on y1
        CdeBwdCoef cb{};
        for (int j = 0; j < 6; ++j) cb.c[j] = (float)(b_of(j) * (double)dtf);
        hipLaunchKernelGGL(stage_input_kernel, EW_BLOCKS((size_t)n), 0, st, ty, tk, cb, 6, zin, n);
        vjp.run(zin, meta[s + 1].seg[0], lam_k0, v);
        hipLaunchKernelGGL(add_kernel, EW_BLOCKS((size_t)n), 0, st, gy, v, (size_t)n);
      } else {
        hipLaunchKernelGGL(add_kernel, EW_BLOCKS((size_t)n), 0, st, gk + 6 * (size_t)n, lam_k0, (size_t)n);   // FSAL: k0_next = k6
      }
    }
    // gy currently = g_y1 (+ jump adjoint).  Outputs of this step and y1 itself are functions of (y, k): move to those variables.
    // start the y-gradient from zero and express everything through u1 = total gradient of y1
    float* gy1 = v;   // reuse v as the y1 gradient holder until the stage loop
    (void)hipMemcpyAsync(gy1, gy, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(gy, 0, (size_t)n * sizeof(float), st);
    for (int p = mm.p_hi - 1; p >= mm.p_lo; --p) {
      const double x = adaptive ? (double)out_x[p] : 1.0;
      const double x2 = x * x, x3 = x2 * x, x4 = x3 * x;
      const double wy = 1 - 11 * x2 + 18 * x3 - 8 * x4, wy1 = -5 * x2 + 14 * x3 - 8 * x4, wym = 16 * x2 - 32 * x3 + 16 * x4;
      const double wfa = x - 4 * x2 + 5 * x3 - 2 * x4, wfb = x2 - 3 * x3 + 2 * x4;
      CdeBwdCoef c{};
      if (adaptive) {
        for (int j = 0; j < 7; ++j) c.c[j] = (float)((double)dtf * (wy1 * b_of(j) + wym * kMID[j]));
        c.c[0] += (float)((double)dtf * wfa);
        c.c[6] += (float)((double)dtf * wfb);
        hipLaunchKernelGGL(output_adjoint_kernel, EW_BLOCKS((size_t)n), 0, st, g_sol, B, H, n_out, p, (float)(wy + wy1 + wym), c, gy, gk, n);
      } else {   // fixed grid: the output IS y1
        hipLaunchKernelGGL(add_strided_kernel, EW_BLOCKS((size_t)n), 0, st, gy1, g_sol, B, H, n_out, p);
      }
    }
    if (adaptive) {
      // stage 6 is evaluated AT y1: v6 = J^T g_k[6]; then y1 = y + dt sum_j b_j k_j carries (gy1 + v6)
      CdeBwdCoef cb{};
      for (int j = 0; j < 6; ++j) cb.c[j] = (float)(b_of(j) * (double)dtf);
      hipLaunchKernelGGL(stage_input_kernel, EW_BLOCKS((size_t)n), 0, st, ty, tk, cb, 6, zin, n);
      vjp.run(zin, mm.seg[6], gk + 6 * (size_t)n, lam_k0 /*scratch*/);
      hipLaunchKernelGGL(add_kernel, EW_BLOCKS((size_t)n), 0, st, gy1, lam_k0, (size_t)n);
      hipLaunchKernelGGL(stage_adjoint_kernel, EW_BLOCKS((size_t)n), 0, st, gy1, cb, 6, gy, gk, n);
    } else {
      CdeBwdCoef cb{};
      for (int j = 0; j < S; ++j) cb.c[j] = (float)(b_of(j) * (double)dtf);
      hipLaunchKernelGGL(stage_adjoint_kernel, EW_BLOCKS((size_t)n), 0, st, gy1, cb, S, gy, gk, n);
    }
    // stages (adaptive: 5 .. 1; rk4: 3 .. 1): k_i = f(seg_i, y + dt sum_{j<i} a_ij k_j)
    for (int i = (adaptive ? 5 : S - 1); i >= 1; --i) {
      CdeBwdCoef ca{};
      for (int j = 0; j < i; ++j) ca.c[j] = (float)(a_of(i, j) * (double)dtf);
      hipLaunchKernelGGL(stage_input_kernel, EW_BLOCKS((size_t)n), 0, st, ty, tk, ca, i, zin, n);
      vjp.run(zin, mm.seg[i], gk + (size_t)i * n, v);
      hipLaunchKernelGGL(stage_adjoint_kernel, EW_BLOCKS((size_t)n), 0, st, v, ca, i, gy, gk, n);
    }
    // k0: evaluated in THIS step at y (fixed grid; the first step; the step after a knot) or carried from the previous step (FSAL)
    const bool k0_here = !adaptive || s == 0 || meta[s - 1].on_jump;
    if (k0_here) {
      if (adaptive && s > 0) {
        // the evaluation belongs to the previous step's "behind the knot" branch: hand the gradient over
        (void)hipMemcpyAsync(lam_k0, gk, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
      } else {
        vjp.run(ty, mm.seg[0], gk, v);
        hipLaunchKernelGGL(add_kernel, EW_BLOCKS((size_t)n), 0, st, gy, v, (size_t)n);
      }
    } else {
      (void)hipMemcpyAsync(lam_k0, gk, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
  }
  // ---- gy = gradient of z0 through the solve; the first output is z0 itself; the caller may add its own (z0 is a return value)
  hipLaunchKernelGGL(add_strided_kernel, EW_BLOCKS((size_t)n), 0, st, gy, g_sol, B, H, n_out, 0);
  if (g_z0_out) hipLaunchKernelGGL(add_kernel, EW_BLOCKS((size_t)n), 0, st, gy, g_z0_out, (size_t)n);
  if (z0_in) {
    if (g_z0_in) (void)hipMemcpyAsync(g_z0_in, gy, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st);
  } else {
    // z0 = tanh(W obs[:, 0] + b)  (PoseCDE.py:96)
    hipLaunchKernelGGL(tanh_bwd_kernel, EW_BLOCKS((size_t)n), 0, st, gy, z0, (size_t)n);
    if (g.init_w) skinny_tn(gy, H, obs, L * C, g.init_w, C, B, H, C, st, 0);
    if (g.init_b) colsum_rows(gy, g.init_b, B, H, st);
    relayout_transpose(init_w, initT, H, C, st);                                  // [H][C] -> [C][H]
    skinny_nt(gy, H, initT, H, nullptr, g_obs0, C, B, C, H, 0, 0, 0, nullptr, 0, st);
    hipLaunchKernelGGL(obs0_add_kernel, EW_BLOCKS((size_t)B * C), 0, st, g_obs0, g_obs, B, L, C);
  }
  return hipGetLastError() == hipSuccess ? 0 : ODEVIO_ERR_HIP;
}
