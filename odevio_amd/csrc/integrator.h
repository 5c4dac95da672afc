// Argument block of the persistent ODE-RNN integrator kernel (integrator.hip).
#pragma once
#include <stdint.h>

#define INTEG_MAX_LIN 6    // Linears in ODEFunc (ode_fn_num_layers + 1)
#define INTEG_MAX_L 4      // RNN layers
#define INTEG_MEMBERS 32   // workgroups (CUs) per row group = one XCD under round-robin dispatch
#define INTEG_GROUPS 8
#define INTEG_KMAX 1024    // widest vector exchanged between layers
#define INTEG_THREADS 1024 // 16 waves per workgroup = 4 per SIMD: latency hiding for the dependent layer chain

enum IntegMode { MODE_ODE_RNN = 0, MODE_RNN_ONLY = 1, MODE_ODE_STEPS = 2, MODE_FEVAL = 3 };

// Butcher tableau handed to the kernel in the argument block (scalar loads).
struct IntegTableau {
  int stages;
  int fsal;       // last stage is f(y1): reuse it as the next step's first stage
  int has_err;    // embedded error estimate -> adaptive I-controller
  int order;      // exponent -1/order of the controller
  float a[7][7];
  float b[7];
  float e[7];     // b - b_hat
};

struct IntegArgs {
  // ---- vector field  f(y) = tanh(W_n act(... act(W_1 y + b_1)) + b_n)
  int F, H, nlin, act;                    // F: INTERNAL state width, a multiple of INTEG_MEMBERS (zero-padded weights, exact)
  int Fio;                                // width AND row stride of every [.., F] tensor in global memory (<= F): the caller's F
  int dims[INTEG_MAX_LIN + 1];            // F, H, ..., H, F (internal, padded)
  const float* w[INTEG_MAX_LIN];          // per-member slices [member][chunk j][col][lane 0..63][4], K padded to 256
  const float* b[INTEG_MAX_LIN];          // full bias vectors
  int w_lds_off[INTEG_MAX_LIN];           // float offset of the LDS-resident copy, or -1 = stream from L2
  // ---- RNN stack
  int rnn_type, L, rnn_vcols;             // virtual columns per hidden unit: 1 (tanh RNN) or 4 (GRU: r, z, n_i, n_h)
  const float* rw[INTEG_MAX_L];           // same layout, K = [input (F padded) | hidden (F padded)]
  const float* rb[INTEG_MAX_L];           // [vcols][F] folded biases
  // ---- solver
  IntegTableau tab;
  int nsub;
  float atol, rtol, dt0;
  int max_steps;
  // ---- problem
  int mode;
  int B, P;                               // batch (tensor stride), intervals (timestamps per row = P + 1)
  int b_begin, b_end;                     // batch elements (sequence modes) or rows (row modes) of THIS launch
  int G, BPG;                             // active groups; sequences (sequence modes) or rows (row modes) per group
  int rows_per_group;                     // L*BPG (sequence modes) or BPG (row modes)
  const float* fused;                     // [B][P][F]
  const float* ts;                        // [B][P+1]
  int ts_relative;                        // 1: subtract ts[:,0] (hc == NULL), reference PoseODERNN.py:100
  const float* hc;                        // [L][B][F] or null
  float* out_seq;                         // [B][P][F] top-layer RNN outputs
  float* hT;                              // [L][B][F]
  const float* y0;                        // row modes: [rows][F]
  const float* t0;                        // [rows]
  const float* t1;                        // [rows]
  float* y_out;                           // [rows][F]
  int* stats;                             // [rows][2] or null
  // accepted-step log for the backward (train.hip replays the forward's accepted steps): dtlog [rows][P][dtlog_cap] step
  // sizes in order, dtcnt [rows][P] how many ACCEPTED steps the interval took (a count above dtlog_cap says the log is
  // incomplete; the forward itself is not affected); null = no log.  ylog [rows][P][dtlog_cap][Fio] (optional, with dtlog): the
  // state each accepted step starts from, yend [rows][P][Fio]: the evolved state at the end of the interval - with these the
  // backward rebuilds every stage of every step in ONE batch instead of walking the steps in order.
  float* dtlog;
  int* dtcnt;
  int dtlog_cap;
  float* ylog;
  float* yend;
  // ---- infrastructure
  unsigned long long* xbuf;               // [G][2][xstride] 8-byte {tag, value} granules
  int xstride;
  int* status;                            // device status word (0 = ok)
  int allow_local;                        // 1: groups that prove to sit on one XCD use the L2-local hand-off
  unsigned long long* dbg;                // phase stamps (only written by the ODEVIO_STAMPS diagnostic build)
  // ---- LDS carve (float offsets)
  int lds_xin, lds_hst, lds_misc, lds_w;
};

int launch_integrator(const IntegArgs& a, int rt, size_t lds_bytes, void* stream);

// ---- the adjoint twin (integrator_adj_kernel): the reverse sweep of ONE interval's Runge-Kutta steps for every row, with the
// TRANSPOSED ODEFunc weights resident in LDS / registers and the same granule exchange between layers - what the backward
// (train.hip) otherwise does with one launch per product (S x layers per accepted step).  Per stage, last first:
//   dl = lamK_s * (1 - K_s^2) -> [delta_l = (delta_{l+1} W_{l+1}) * act'(a_l)] for the layers, last first -> gX;  lam += gX;  lamK_j += dt a_sj gX
// reading the saved activations from the backward's tape and writing every layer's pre-activation gradient there (the weight
// gradients are products over the whole tape afterwards).  Rows whose interval took fewer steps carry dt = 0 for the rest:
// zeros through the same arithmetic, no special case.
struct IntegAdjArgs {
  int F, Fio, nlin, act;                  // F internal (padded) state width; Fio the tape's / caller's
  int dims[INTEG_MAX_LIN + 1];            // internal widths F, H, .., H, F
  int dims_io[INTEG_MAX_LIN + 1];         // the tape's widths (row strides of act[l] / delta[l-1])
  const float* wT[INTEG_MAX_LIN];         // W_l^T as a layer (inputs dims[l+1], outputs dims[l]) in the members' slice layout
  int w_lds_off[INTEG_MAX_LIN];
  int S;                                  // stages that carry gradient (FSAL: without the last)
  float ta[7][7], tb[7];
  // tape geometry: row of (stage s, interval it, step j, row r) = s * stage_rows + (it * J + j) * Rtot + r
  int J, it, Rtot;
  int Jrun;                               // steps of THIS interval that any row took (<= J, the tape's stride): the sweep starts there, not at J
  size_t stage_rows;
  const float* tape_act[INTEG_MAX_LIN + 1];   // saved activation OUTPUTS, [l] for l = 1 .. nlin ([nlin] = K, the stage derivative)
  float* tape_delta[INTEG_MAX_LIN];        // [l]: the gradient at the output of Linear l (pre-activation), width dims_io[l+1]
  const float* dt;                        // [(it * J + j) * Rtot + r]
  float* lam;                             // [Rtot][Fio] in / out: dL/d(state at the interval's end) -> dL/d(state at its start)
  // rows of this launch (as IntegArgs)
  int B, b_begin, b_end, G, BPG, rows_per_group;
  unsigned long long* xbuf;
  int xstride;
  int* status;
  int allow_local;
  int lds_xin, lds_misc, lds_w;
};
// every chunk of rows of one interval (L RNN layers x B sequences, as run in the forward); base: everything but the rows and the LDS carve
int launch_integrator_adj(const IntegAdjArgs& base, int L, int B, void* stream);
