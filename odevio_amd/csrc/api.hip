// C ABI of libodevio.so (include/odevio.h): plan construction (weight re-layout, BatchNorm
// folding, column-sharding for the persistent integrator), activation workspace, and the launch
// sequences that replace the reference's DeepVIO.forward (src/models/DeepVIO.py:61-68).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/odevio.h"
#include "cde.h"
#include "common.h"
#include "integrator.h"
#include "train.h"
#include "bn_train.h"
#include "enc_bwd.h"
#include "cde_bwd.h"

// roctx ranges at the two sites the reference marks with NVTX (src/models/PoseODERNN.py:103-104 "ODE", :118-119 "RNN"),
// plus the encoders: visible in rocprofv3 --marker-trace.  libroctx64 is looked up at run time (no link dependency);
// without it the ranges are no-ops.
#include <dlfcn.h>
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (void* h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL)) {
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (!push || !pop) push = nullptr, pop = nullptr;
    }
  }
};
struct RoctxRange {
  static Roctx& api() { static Roctx r; return r; }
  explicit RoctxRange(const char* name) { if (api().push) api().push(name); }
  ~RoctxRange() { if (api().pop) api().pop(); }
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) return fail(ODEVIO_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct ConvSpec {
  const char* name;
  int cin, cout, k, stride;
};
static const ConvSpec kConvs[9] = {
    {"conv1", 6, 64, 7, 2},     {"conv2", 64, 128, 5, 2},    {"conv3", 128, 256, 5, 2},
    {"conv3_1", 256, 256, 3, 1}, {"conv4", 256, 512, 3, 2},   {"conv4_1", 512, 512, 3, 1},
    {"conv5", 512, 512, 3, 2},   {"conv5_1", 512, 512, 3, 1}, {"conv6", 512, 1024, 3, 2}};
static int conv_out(int n, int k, int s) { return (n + 2 * ((k - 1) / 2) - k) / s + 1; }

struct DevBuf {
  float* p = nullptr;
  size_t n = 0;  // floats
};

struct odevio_plan {
  odevio_config cfg;
  int device = 0, n_cu = 0;
  int F = 0;               // v_f_len + i_f_len: the width of every feature / state tensor at the boundary
  // The persistent integrator shards columns over INTEG_MEMBERS workgroups, so INSIDE it the state and the ODEFunc's hidden
  // width are rounded up to a multiple of that (Fi, dims[]) with zero weight rows / columns and zero biases - exact: a padded
  // state element starts at 0, its derivative is tanh(0) = 0, its RNN / GRU update is 0; a padded hidden unit (softplus(0) != 0)
  // feeds zero columns.  The reference's own shapes (v_f_len = i_f_len = 200, ode_hidden_dim = 200: scripts/run_training.sh:57-70)
  // need it; for the default shapes Fi == F and nothing changes.
  int Fi = 0;
  int dims_real[INTEG_MAX_LIN + 1] = {};
  bool padded = false;
  // encoder
  float* conv_w[9] = {};
  void* conv_ws[9] = {};   // conv2..conv6 weights as two fp16 pieces (conv_f16x2.hip), [Cout][K-tile][2][32], pre-scaled
  size_t conv_ws_bytes[9] = {}, head_ws_bytes = 0;
  float* conv_scale_h[9] = {};  // BatchNorm scale with the weights' power-of-two pre-scale folded back in
  // The same constants for the production chain, where every stored activation carries a per-layer power-of-two factor
  // 2^act_exp (chosen at plan creation from a variance estimate, see plan_create): keeps the fp16 pieces of the
  // activations in their normal range whatever the checkpoint's BatchNorm statistics are.  Exact: LeakyReLU is
  // positively homogeneous and the factor is divided back out of the next layer's scale.
  float* conv_scale_c[9] = {};
  float* conv_shift_c[9] = {};
  int act_exp[10] = {};
  // train mode (model.train(): BatchNorm with batch statistics + Dropout, bn_train.hip): the BatchNorm affine parameters, an
  // identity epilogue for the convolution kernels (1 / prescale, zero shift, slope 1), scratch for the statistics
  float* conv_gamma[9] = {};
  float* conv_beta[9] = {};
  float* conv_scale_raw[9] = {};
  float* head_scale_raw = nullptr;
  float* zero_vec = nullptr;     // 1024 zeros
  float *bn_scale = nullptr, *bn_shift = nullptr;   // [1024] each: the affine pair of the block being normalised
  float *imu_gamma[3] = {}, *imu_beta[3] = {};
  DevBuf bn_partial;
  // image-encoder backward (enc_bwd.hip): the flipped / transposed filters of the input-gradient convolutions, what a train-mode
  // forward with keep = 1 leaves behind (per block the bare convolution z and the block's output a, both P2; the batch statistics),
  // and scratch
  float* conv_wT[9] = {};                  // [Cin][kh][kw][Cout], taps reversed (blocks 1..8): fp32-input MFMA form (ODEVIO_ENC_BWD_IGEMM)
  void* conv_wTs[9] = {};                  // the same filters as two fp16 pieces for conv_f16x2_kernel: [Cin][K-tile][2][32]
  size_t conv_wTs_bytes[9] = {};
  void* conv_wTp[9][4] = {};               // 5 x 5 stride-2 blocks: the four PARITY sub-filters (3 x 3, one per (row, column) parity of the input pixel)
  size_t conv_wTp_bytes[9] = {};
  size_t conv_wTq_bytes[9][4] = {};        // 3 x 3 stride-2 blocks: the classes' filters differ in size ((1 + py) x (1 + px))
  float conv_wT_inv_prescale[9] = {};
  float* enc_dscale = nullptr;             // [1024] epilogue scale of the current input-gradient convolution + one word for max|D|
  DevBuf enc_z[9], enc_a[9];
  float *enc_mean[9] = {}, *enc_invstd[9] = {};
  int enc_B = 0, enc_S = 0;                // shape of the kept forward (0: none)
  unsigned long long enc_seed = 0, enc_call0 = 0;
  DevBuf enc_gA, enc_D, enc_Dd, enc_part, enc_headT;
  int conv_math = 1;       // 1: fp16x2 operand split on the fp16 MFMA (default); 0: fp32-input MFMA (ODEVIO_CONV_MATH=f32)
  DevBuf pack_tmp, ingest, partial_side;
  // the inertial encoder runs beside the image encoder on its own stream (odevio_forward)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void* zero_page = nullptr;  // what the split kernel's LDS-DMA reads for taps outside the image
  float* conv_scale[9] = {};
  float* conv_shift[9] = {};
  int conv_h[10] = {}, conv_w_sp[10] = {};  // spatial size before conv i (index 0 = image)
  float *head_w = nullptr, *head_b = nullptr;
  void* head_ws = nullptr;       // visual head as a 1x1 convolution for the fp16x2 kernel: [v_f_len][K/32][2][32] fp16
  float* head_scale_h = nullptr; // 1 / prescale per output
  int head_k = 0;
  float *imu_w[3] = {}, *imu_s[3] = {}, *imu_h[3] = {};
  float *imu_wref[3] = {}, *imu_var[3] = {}, *imu_mean[3] = {}, *imu_bias[3] = {};   // backward: reference-layout weights (rows padded to 16), BatchNorm statistics
  float *proj_w = nullptr, *proj_b = nullptr;
  float *fuse_w = nullptr, *fuse_b = nullptr, *fuse_w_t = nullptr;
  unsigned long long seed = 0, rng_calls = 0;   // fuse_method hard: Philox key and the per-call counter block
  float *reg_w0 = nullptr, *reg_b0 = nullptr, *reg_w2 = nullptr, *reg_b2 = nullptr;
  // integrator
  int nlin = 0;
  int dims[INTEG_MAX_LIN + 1] = {};
  float* ode_w[INTEG_MAX_LIN] = {};
  float* ode_wT[INTEG_MAX_LIN] = {};   // the transposes in the same member-slice layout (integrator_adj_kernel: the backward's reverse sweep)
  float* ode_b[INTEG_MAX_LIN] = {};
  float* rnn_w[INTEG_MAX_L] = {};
  float* rnn_b[INTEG_MAX_L] = {};
  int rnn_vcols = 1;
  unsigned long long* xbuf = nullptr;
  int xstride = 0;
  IntegAdjArgs adj_base;               // what every launch of the adjoint twin shares (filled by fill_train_model)
  int* status = nullptr;
  // status words copied to pinned host memory behind every forward (no host synchronisation): the next API call that
  // finds the copy complete reports a failure of the previous forward instead of computing on garbage
  int* status_host = nullptr;
  hipEvent_t ev_status = nullptr;
  bool status_pending = false;
  // backward (train.hip): plain and transposed copies of the ODEFunc / RNN / regressor weights, workspace
  TrainModel train = {};
  DevBuf train_ws, train_log, train_aux;   // train_aux: fusion backward / gradient-norm partials
  // Neural-CDE path (model_type cde)
  CdeModel cde = {};
  float *cde_init_w = nullptr, *cde_init_b = nullptr;
  DevBuf cde_ws, cde_fn_ws;
  hipEvent_t ev_cde[2] = {nullptr, nullptr};   // around the last layer of the most recent odevio_cde_func (stage timers on)
  CdeCtl* cde_ctl_host = nullptr;   // pinned mirror of the device-side controller
  int cde_hint_steps = 0;           // attempts the previous solve needed (first batch of the next one)
  // workspace (grown on demand)
  DevBuf actA, actB, imu_act, fcat, fused, out_seq, reg_hid, partial;
  std::vector<void*> owned;
  // optional per-stage HIP-event timing of odevio_forward (bench.py's roofline figures)
  // A ring of event sets: a caller may run many forwards back to back and read the averages afterwards, so that
  // reading the timers does not put a host synchronisation between the forwards it measures.
  bool prof = false;
  std::vector<hipEvent_t> ev;   // [depth][ODEVIO_N_STAGES + 1]
  int ev_depth = 0, ev_w = 0, ev_n = 0;
};

static void stage_mark(odevio_plan* p, int i, hipStream_t st) {
  if (!p->prof) return;
  (void)hipEventRecord(p->ev[(size_t)p->ev_w * (ODEVIO_N_STAGES + 1) + i], st);
  if (i == ODEVIO_N_STAGES) {   // the forward's last mark: the set is complete
    p->ev_w = (p->ev_w + 1) % p->ev_depth;
    if (p->ev_n < p->ev_depth) ++p->ev_n;
  }
}

static int dev_alloc(odevio_plan* p, void** out, size_t bytes) {
  HIPCHK(hipMalloc(out, bytes));
  p->owned.push_back(*out);
  return 0;
}
static int upload(odevio_plan* p, float** out, const std::vector<float>& h, hipStream_t st) {
  if (*out == nullptr) {   // (a non-null target is a buffer of the same size from an earlier load: odevio_plan_update)
    int rc = dev_alloc(p, (void**)out, h.size() * sizeof(float));
    if (rc) return rc;
  }
  HIPCHK(hipMemcpyAsync(*out, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));  // h may be a temporary
  return 0;
}
// Every plan buffer is followed, inside its allocation, by ODEVIO_ZERO_PAGE_BYTES zero bytes (at b.p + b.n): the conv
// kernels' 32-bit DMA addressing reads them for taps outside the image (common.h, ConvSplitArgs::in_zero_off).  Nothing
// ever writes there: every kernel's extent is b.n.
static int ensure(DevBuf& b, size_t n) {
  if (b.n >= n) return 0;
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = nullptr;
  b.n = 0;
  HIPCHK(hipMalloc((void**)&b.p, n * sizeof(float) + ODEVIO_ZERO_PAGE_BYTES));
  HIPCHK(hipMemset(b.p + n, 0, ODEVIO_ZERO_PAGE_BYTES));
  static const bool zero_all = getenv("ODEVIO_ZERO_BUFFERS") != nullptr;   // timing experiments that skip a producer's stores
  if (zero_all) HIPCHK(hipMemset(b.p, 0, n * sizeof(float)));
  b.n = n;
  return 0;
}

// Bytes from `ptr` to the end of the plan-owned buffer that contains it (what a kernel may touch), or `fallback` for
// memory the caller owns (its extent is the caller's contract).
static size_t extent_of(const odevio_plan* p, const void* ptr, size_t fallback) {
  const DevBuf* bufs[] = {&p->actA, &p->actB, &p->pack_tmp, &p->partial, &p->partial_side, &p->ingest, &p->fcat, &p->fused, &p->out_seq,
                          &p->enc_z[0], &p->enc_z[1], &p->enc_z[2], &p->enc_z[3], &p->enc_z[4], &p->enc_z[5], &p->enc_z[6], &p->enc_z[7], &p->enc_z[8],
                          &p->enc_Dd, &p->enc_a[0], &p->enc_a[1], &p->enc_a[2], &p->enc_a[3], &p->enc_a[4], &p->enc_a[5], &p->enc_a[6], &p->enc_a[7], &p->enc_a[8]};
  const uintptr_t a = (uintptr_t)ptr;
  for (const DevBuf* b : bufs) {
    const uintptr_t lo = (uintptr_t)b->p, hi = lo + b->n * sizeof(float);
    if (b->p && a >= lo && a < hi) return hi - a;
  }
  return fallback;
}

static std::vector<float> transposed(const std::vector<float>& w, int N, int K) {   // [N][K] -> [K][N]
  std::vector<float> t((size_t)N * K);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = w[(size_t)n * K + k];
  return t;
}

// 32-bit DMA addressing (conv_f16x2.hip) when `in` lies in a plan buffer (whose zero tail follows it) and everything is
// below 4 GB; the weights always carry their tail.
static void set_off32(const odevio_plan* p, ConvSplitArgs& a, const void* in) {
  const size_t ext = extent_of(p, in, 0);
  a.off32 = 0;
  static const bool off64 = getenv("ODEVIO_CONV_OFF64") != nullptr;   // diagnostic: the 64-bit addressing form
  if (off64 || ext == 0 || ext + ODEVIO_ZERO_PAGE_BYTES > 0xffffffffull || a.w_bytes + ODEVIO_ZERO_PAGE_BYTES > 0xffffffffull) return;
  a.in_zero_off = (unsigned)ext;
  a.w_zero_off = (unsigned)a.w_bytes;
  a.off32 = 1;
}

struct WeightTable {
  std::map<std::string, std::pair<const void*, int64_t>> m;
  hipStream_t st;
  int get(const std::string& name, int64_t numel, std::vector<float>& out) const {
    auto it = m.find(name);
    if (it == m.end()) return fail(ODEVIO_ERR_MISSING_WEIGHT, "weight '%s' not provided", name.c_str());
    if (it->second.second != numel)
      return fail(ODEVIO_ERR_BAD_ARG, "weight '%s' has %lld elements, expected %lld", name.c_str(),
                  (long long)it->second.second, (long long)numel);
    out.resize((size_t)numel);
    HIPCHK(hipMemcpyAsync(out.data(), it->second.first, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
  }
};

// BatchNorm(eval) as y = x*scale + shift; a preceding conv bias folds into shift.
static int bn_fold(const WeightTable& wt, const std::string& bn, int c, const std::vector<float>* conv_bias,
                   std::vector<float>& scale, std::vector<float>& shift) {
  std::vector<float> g, b, mu, var;
  int rc;
  if ((rc = wt.get(bn + ".weight", c, g))) return rc;
  if ((rc = wt.get(bn + ".bias", c, b))) return rc;
  if ((rc = wt.get(bn + ".running_mean", c, mu))) return rc;
  if ((rc = wt.get(bn + ".running_var", c, var))) return rc;
  scale.resize(c);
  shift.resize(c);
  for (int i = 0; i < c; ++i) {
    const float s = g[i] / std::sqrt(var[i] + 1e-5f);
    const float cb = conv_bias ? (*conv_bias)[i] : 0.f;
    scale[i] = s;
    shift[i] = (cb - mu[i]) * s + b[i];
  }
  return 0;
}

// [Cout][Cin][kh][kw] fp32 -> [Cout][K-tile][2 pieces][32] fp16, K-tile = group * taps + tap (conv_f16x2.hip).
// x * prescale = h + l, h = fp16(.), l = fp16(. - h) (round to nearest even, subnormals kept).  `prescale` is a power of
// two that lifts the layer's largest weight to [2^13, 2^14): the low pieces of all but vanishing weights are then normal
// fp16 numbers, and 1/prescale goes into the BatchNorm scale exactly.
static float split_conv_weights(const std::vector<float>& w, int cout, int cin, int kk, std::vector<uint16_t>& out, float forced_prescale = 0.f) {
  float wmax = 0.f;
  for (float x : w) wmax = std::max(wmax, std::fabs(x));
  int e = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    (void)std::frexp(wmax, &e);   // wmax = f * 2^e, f in [0.5, 1)
    e = 14 - e;                   // wmax * 2^e in [2^13, 2^14)
  }
  e = std::max(-40, std::min(40, e));
  const float prescale = forced_prescale > 0.f ? forced_prescale : std::ldexp(1.0f, e);   // (forced: a part of a filter split with the whole filter's factor)
  const int groups = cin / 32;
  out.assign((size_t)cout * groups * kk * 64, 0);
  for (int n = 0; n < cout; ++n)
    for (int c = 0; c < cin; ++c)
      for (int q = 0; q < kk; ++q) {
        const float x = w[((size_t)n * cin + c) * kk + q] * prescale;
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        const size_t base = (((size_t)n * groups + c / 32) * kk + q) * 64 + (c % 32);
        memcpy(&out[base], &h, 2);
        memcpy(&out[base + 32], &l, 2);
      }
  return prescale;
}

// [N][K] row-major -> per-member slices [member][j][col][ks][4] (integrator.hip, layer()).
// The K axis is given as segments (each padded with zeros to a multiple of 256 = 64 lanes x 4 floats): lane l of
// a wave multiplies inputs 256j + 4l .. 4l+3 of chunk j, so its weights for (chunk, column) are one float4.
static int pad256(int k) { return (k + 255) & ~255; }
static int pad_members(int n) { return (n + INTEG_MEMBERS - 1) / INTEG_MEMBERS * INTEG_MEMBERS; }
static void shard_columns(const std::vector<float>& W, int N, const std::vector<int>& segs, std::vector<float>& out) {
  const int NC = N / INTEG_MEMBERS;
  int K = 0, Kp = 0;
  for (int s : segs) { K += s; Kp += pad256(s); }
  out.assign((size_t)N * Kp, 0.f);
  for (int m = 0; m < INTEG_MEMBERS; ++m)
    for (int col = 0; col < NC; ++col) {
      int k0 = 0, j0 = 0;
      for (int s : segs) {
        for (int k = 0; k < s; ++k) {
          const int j = j0 + k / 256, lane = (k % 256) / 4, e = k % 4;
          out[(size_t)m * NC * Kp + (((size_t)j * NC + col) * 64 + lane) * 4 + e] = W[(size_t)(m * NC + col) * K + k0 + k];
        }
        k0 += s;
        j0 += pad256(s) / 256;
      }
    }
}

static int g_audit_violations = 0;
extern "C" int odevio_version(void) { return ODEVIO_VERSION; }
extern "C" int odevio_audit_violations(void) { return g_audit_violations; }
extern "C" const char* odevio_last_error(void) { return g_err; }

extern "C" void odevio_plan_destroy(odevio_plan* p) {
  if (!p) return;
  for (hipEvent_t e : p->ev)
    if (e) (void)hipEventDestroy(e);
  if (p->status) {   // audit builds: a violation nobody asked about must still be seen (tests/conftest.py)
    int hw[8] = {};
    if (hipMemcpy(hw, p->status, sizeof(hw), hipMemcpyDeviceToHost) == hipSuccess && hw[ODEVIO_STATUS_AUDIT] &&
        !getenv("ODEVIO_AUDIT_SELFTEST")) {
      ++g_audit_violations;
      fprintf(stderr, "libodevio AUDIT: kernel id %d computed an address outside its buffers\n", hw[ODEVIO_STATUS_AUDIT + 1]);
    }
  }
  if (p->status_host) (void)hipHostFree(p->status_host);
  if (p->cde_ctl_host) (void)hipHostFree(p->cde_ctl_host);
  for (hipEvent_t e : p->ev_cde)
    if (e) (void)hipEventDestroy(e);
  if (p->ev_status) (void)hipEventDestroy(p->ev_status);
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  for (void* q : p->owned) (void)hipFree(q);
  for (DevBuf* b : {&p->actA, &p->actB, &p->imu_act, &p->fcat, &p->fused, &p->out_seq, &p->reg_hid, &p->partial,
                    &p->cde_ws, &p->cde_fn_ws, &p->pack_tmp, &p->ingest, &p->partial_side, &p->train_ws, &p->train_log, &p->train_aux, &p->bn_partial,
                    &p->enc_gA, &p->enc_D, &p->enc_Dd, &p->enc_part, &p->enc_headT})
    if (b->p) (void)hipFree(b->p);
  for (int i = 0; i < 9; ++i)
    for (DevBuf* b : {&p->enc_z[i], &p->enc_a[i]})
      if (b->p) (void)hipFree(b->p);
  delete p;
}

static int validate(const odevio_config& c) {
  if (c.struct_size != (int)sizeof(odevio_config)) return fail(ODEVIO_ERR_BAD_ARG, "odevio_config size mismatch");
  if (c.model_type != ODEVIO_MODEL_ODE_RNN && c.model_type != ODEVIO_MODEL_RNN && c.model_type != ODEVIO_MODEL_CDE)
    return fail(ODEVIO_ERR_UNSUPPORTED, "model_type %d not supported", c.model_type);
  if (c.model_type == ODEVIO_MODEL_CDE) {
    // PoseCDE is only dimensionally consistent when cde_hidden_dim == v_f_len + i_f_len (its reduction_net is never
    // applied, reference PoseCDE.py:53-61,83-84,96)
    if (c.cde_hidden_dim != c.v_f_len + c.i_f_len)
      return fail(ODEVIO_ERR_UNSUPPORTED, "cde_hidden_dim (%d) must equal v_f_len + i_f_len (%d)", c.cde_hidden_dim, c.v_f_len + c.i_f_len);
    // (hidden sizes other than 128 / 256 / 512 / 1024 run on the generic last-layer kernel: e.g. the reference's own CDE recipe,
    // v_f_len = i_f_len = 200, cde_hidden_dim = 400, scripts/run_training.sh:57-70)
    if (c.cde_hidden_dim % 16 || c.cde_fn_num_layers < 1 || c.cde_fn_num_layers + 1 > CDE_MAX_LIN)
      return fail(ODEVIO_ERR_UNSUPPORTED, "cde_hidden_dim must be a multiple of 16 and cde_fn_num_layers in 1..%d", CDE_MAX_LIN - 1);
    if (c.cde_activation < 0 || c.cde_activation > 3) return fail(ODEVIO_ERR_BAD_ARG, "Activation function not supported");
    if (c.cde_solver != ODEVIO_DOPRI5 && c.cde_solver != ODEVIO_RK4 && c.cde_solver != ODEVIO_EULER)
      return fail(ODEVIO_ERR_BAD_ARG, "Solver not supported");
    if (c.fuse_method < ODEVIO_FUSE_CAT || c.fuse_method > ODEVIO_FUSE_HARD) return fail(ODEVIO_ERR_BAD_ARG, "Fusion method not supported");
    if (c.img_h < 64 || c.img_w < 64) return fail(ODEVIO_ERR_BAD_ARG, "image size %dx%d too small", c.img_h, c.img_w);
    if (c.v_f_len % 4 || c.i_f_len % 4) return fail(ODEVIO_ERR_UNSUPPORTED, "feature lengths must be multiples of 4");
    return 0;
  }
  if (c.img_h < 64 || c.img_w < 64) return fail(ODEVIO_ERR_BAD_ARG, "image size %dx%d too small", c.img_h, c.img_w);
  if (c.fuse_method < ODEVIO_FUSE_CAT || c.fuse_method > ODEVIO_FUSE_HARD) return fail(ODEVIO_ERR_BAD_ARG, "Fusion method not supported");
  if (c.ode_activation < 0 || c.ode_activation > 3) return fail(ODEVIO_ERR_BAD_ARG, "Activation function not supported");
  if (c.ode_solver < 0 || c.ode_solver > ODEVIO_RK4_CLASSIC) return fail(ODEVIO_ERR_BAD_ARG, "Solver not supported");
  if (c.rnn_type != ODEVIO_RNN_TANH && c.rnn_type != ODEVIO_RNN_GRU) return fail(ODEVIO_ERR_BAD_ARG, "RNN type not supported");
  if (c.rnn_num_layers < 1 || c.rnn_num_layers > INTEG_MAX_L)
    return fail(ODEVIO_ERR_UNSUPPORTED, "rnn_num_layers must be 1..%d", INTEG_MAX_L);
  if (c.ode_fn_num_layers < 1 || c.ode_fn_num_layers + 1 > INTEG_MAX_LIN)
    return fail(ODEVIO_ERR_UNSUPPORTED, "ode_fn_num_layers must be 1..%d", INTEG_MAX_LIN - 1);
  const int F = c.v_f_len + c.i_f_len;
  if (c.v_f_len < 4 || c.i_f_len < 4 || c.v_f_len % 4 || c.i_f_len % 4) return fail(ODEVIO_ERR_UNSUPPORTED, "feature lengths must be multiples of 4");
  if (c.ode_hidden_dim < 1 || F > INTEG_KMAX || c.ode_hidden_dim > INTEG_KMAX)   // (other widths are zero-padded to a multiple of 32 inside the integrator)
    return fail(ODEVIO_ERR_UNSUPPORTED, "v_f_len+i_f_len (%d) and ode_hidden_dim (%d) must be at most %d", F, c.ode_hidden_dim, INTEG_KMAX);
  if (c.ode_substeps < 1) return fail(ODEVIO_ERR_BAD_ARG, "ode_substeps must be >= 1");
  return 0;
}

// The parameters of Pose_net (fusion, regressor, ODEFunc, RNN stack) in the layouts the kernels read: called by
// odevio_plan_create, and again by odevio_plan_update after an optimizer step changed them (buffers are reused).
static int load_pose_net(odevio_plan* p, WeightTable& wt, hipStream_t st) {
  const int F = p->F;
  int rc;
  std::vector<float> w, t, bias;
#define PN(x)            \
  do {                   \
    rc = (x);            \
    if (rc) return rc;   \
  } while (0)
  // ---- fusion, regressor
  if (p->cfg.fuse_method == ODEVIO_FUSE_SOFT) {
    PN(wt.get("Pose_net.fuse.net.0.weight", (int64_t)F * F, w));
    PN(upload(p, &p->fuse_w, w, st));
    PN(upload(p, &p->fuse_w_t, transposed(w, F, F), st));   // backward: g_c += g_w W
    PN(wt.get("Pose_net.fuse.net.0.bias", F, bias));
    PN(upload(p, &p->fuse_b, bias, st));
  }
  if (p->cfg.fuse_method == ODEVIO_FUSE_HARD) {   // Linear(F, 2F): logits (keep, drop) per feature, interleaved (FusionModule.py:14,26)
    PN(wt.get("Pose_net.fuse.net.0.weight", (int64_t)2 * F * F, w));
    PN(upload(p, &p->fuse_w, w, st));
    PN(upload(p, &p->fuse_w_t, transposed(w, 2 * F, F), st));   // backward: g_cat += g_logits W
    PN(wt.get("Pose_net.fuse.net.0.bias", (int64_t)2 * F, bias));
    PN(upload(p, &p->fuse_b, bias, st));
  }
  PN(wt.get("Pose_net.regressor.0.weight", (int64_t)128 * F, w));
  PN(upload(p, &p->reg_w0, w, st));
  {
    float* wt0 = const_cast<float*>(p->train.reg_w0_t);
    PN(upload(p, &wt0, transposed(w, 128, F), st));
    p->train.reg_w0_t = wt0;
  }
  PN(wt.get("Pose_net.regressor.0.bias", 128, bias));
  PN(upload(p, &p->reg_b0, bias, st));
  PN(wt.get("Pose_net.regressor.2.weight", (int64_t)6 * 128, w));
  PN(upload(p, &p->reg_w2, w, st));
  PN(wt.get("Pose_net.regressor.2.bias", 6, bias));
  PN(upload(p, &p->reg_b2, bias, st));
  // ---- ODEFunc (column-sharded)
  if (p->cfg.model_type == ODEVIO_MODEL_ODE_RNN) {
    p->nlin = p->cfg.ode_fn_num_layers + 1;
    p->dims_real[0] = F;
    for (int l = 1; l < p->nlin; ++l) p->dims_real[l] = p->cfg.ode_hidden_dim;
    p->dims_real[p->nlin] = F;
    for (int l = 0; l <= p->nlin; ++l) p->dims[l] = pad_members(p->dims_real[l]);
    for (int l = 0; l < p->nlin; ++l) {
      const std::string pre = "Pose_net.ode_func.net." + std::to_string(2 * l);
      const int N = p->dims_real[l + 1], K = p->dims_real[l];
      const int Ni = p->dims[l + 1], Ki = p->dims[l];
      PN(wt.get(pre + ".weight", (int64_t)N * K, w));
      if (Ni != N || Ki != K) {
        std::vector<float> wp((size_t)Ni * Ki, 0.f);
        for (int n = 0; n < N; ++n) memcpy(&wp[(size_t)n * Ki], &w[(size_t)n * K], K * sizeof(float));
        shard_columns(wp, Ni, {Ki}, t);
      } else {
        shard_columns(w, N, {K}, t);
      }
      PN(upload(p, &p->ode_w[l], t, st));
      {
        // W^T as a layer of the reverse chain: Ki outputs (this Linear's inputs), Ni inputs, zero-padded like the forward's
        std::vector<float> wtp((size_t)Ki * Ni, 0.f);
        for (int n = 0; n < N; ++n)
          for (int k = 0; k < K; ++k) wtp[(size_t)k * Ni + n] = w[(size_t)n * K + k];
        shard_columns(wtp, Ki, {Ni}, t);
        PN(upload(p, &p->ode_wT[l], t, st));
      }
      {
        float *pw = const_cast<float*>(p->train.ode_w[l]), *pwt = const_cast<float*>(p->train.ode_w_t[l]);
        PN(upload(p, &pw, w, st));
        PN(upload(p, &pwt, transposed(w, N, K), st));
        p->train.ode_w[l] = pw;
        p->train.ode_w_t[l] = pwt;
      }
      PN(wt.get(pre + ".bias", N, bias));
      bias.resize(Ni, 0.f);
      PN(upload(p, &p->ode_b[l], bias, st));
    }
  }
  // ---- RNN stack: virtual columns over K = [input | hidden]
  if (p->cfg.model_type != ODEVIO_MODEL_CDE) {
    const int L = p->cfg.rnn_num_layers;
    const bool gru = p->cfg.rnn_type == ODEVIO_RNN_GRU;
    const int gates = gru ? 3 : 1;
    const int V = gru ? 4 : 1;
    p->rnn_vcols = V;
    const int Fi = p->Fi;
    const int NCF = Fi / INTEG_MEMBERS;
    std::vector<float> wih, whh, bih, bhh;
    for (int l = 0; l < L; ++l) {
      const std::string s = std::to_string(l);
      PN(wt.get("Pose_net.rnn.weight_ih_l" + s, (int64_t)gates * F * F, wih));
      PN(wt.get("Pose_net.rnn.weight_hh_l" + s, (int64_t)gates * F * F, whh));
      PN(wt.get("Pose_net.rnn.bias_ih_l" + s, (int64_t)gates * F, bih));
      PN(wt.get("Pose_net.rnn.bias_hh_l" + s, (int64_t)gates * F, bhh));
      {   // plain copies for the backward (train.hip): [gates*F][F] and the transposes [F][gates*F]
        float *a = const_cast<float*>(p->train.rnn_wih[l]), *at = const_cast<float*>(p->train.rnn_wih_t[l]), *b2 = const_cast<float*>(p->train.rnn_whh[l]),
              *bt = const_cast<float*>(p->train.rnn_whh_t[l]), *c = const_cast<float*>(p->train.rnn_bih[l]), *d = const_cast<float*>(p->train.rnn_bhh[l]);
        PN(upload(p, &a, wih, st));
        PN(upload(p, &at, transposed(wih, gates * F, F), st));
        PN(upload(p, &b2, whh, st));
        PN(upload(p, &bt, transposed(whh, gates * F, F), st));
        PN(upload(p, &c, bih, st));
        PN(upload(p, &d, bhh, st));
        p->train.rnn_wih[l] = a; p->train.rnn_wih_t[l] = at; p->train.rnn_whh[l] = b2; p->train.rnn_whh_t[l] = bt;
        p->train.rnn_bih[l] = c; p->train.rnn_bhh[l] = d;
      }
      // virtual matrix [V*Fi][2F], row order: member-major, then v, then local unit (units F .. Fi-1 are padding: zero rows)
      std::vector<float> vm((size_t)V * Fi * 2 * F, 0.f), vb((size_t)V * Fi, 0.f);
      for (int m = 0; m < INTEG_MEMBERS; ++m)
        for (int v = 0; v < V; ++v)
          for (int ul = 0; ul < NCF; ++ul) {
            const int u = m * NCF + ul;
            if (u >= F) continue;
            float* dst = &vm[((size_t)(m * V + v) * NCF + ul) * 2 * F];
            if (!gru) {
              memcpy(dst, &wih[(size_t)u * F], F * sizeof(float));
              memcpy(dst + F, &whh[(size_t)u * F], F * sizeof(float));
              vb[u] = bih[u] + bhh[u];
            } else if (v < 2) {  // r, z
              memcpy(dst, &wih[((size_t)v * F + u) * F], F * sizeof(float));
              memcpy(dst + F, &whh[((size_t)v * F + u) * F], F * sizeof(float));
              vb[(size_t)v * Fi + u] = bih[(size_t)v * F + u] + bhh[(size_t)v * F + u];
            } else if (v == 2) {  // n, input part
              memcpy(dst, &wih[((size_t)2 * F + u) * F], F * sizeof(float));
              vb[(size_t)2 * Fi + u] = bih[(size_t)2 * F + u];
            } else {  // n, hidden part
              memcpy(dst + F, &whh[((size_t)2 * F + u) * F], F * sizeof(float));
              vb[(size_t)3 * Fi + u] = bhh[(size_t)2 * F + u];
            }
          }
      shard_columns(vm, V * Fi, {F, F}, t);
      PN(upload(p, &p->rnn_w[l], t, st));
      PN(upload(p, &p->rnn_b[l], vb, st));
    }
  }
#undef PN
  return 0;
}

extern "C" int odevio_plan_create(const odevio_config* cfg, const odevio_tensor* weights, int32_t n_weights,
                                  void* stream, odevio_plan** out_plan) {
  if (!cfg || !weights || !out_plan || n_weights <= 0) return fail(ODEVIO_ERR_BAD_ARG, "null argument");
  int rc = validate(*cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(ODEVIO_ERR_NO_DEVICE, "no HIP device");
  hipStream_t st = (hipStream_t)stream;
  odevio_plan* p = new odevio_plan();
  p->cfg = *cfg;
  HIPCHK(hipGetDevice(&p->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, p->device));
  p->n_cu = prop.multiProcessorCount;
  const int F = cfg->v_f_len + cfg->i_f_len;
  p->F = F;
  p->Fi = pad_members(F);
  p->padded = cfg->model_type != ODEVIO_MODEL_CDE &&
              (p->Fi != F || (cfg->model_type == ODEVIO_MODEL_ODE_RNN && pad_members(cfg->ode_hidden_dim) != cfg->ode_hidden_dim));
  WeightTable wt;
  wt.st = st;
  for (int i = 0; i < n_weights; ++i)
    if (weights[i].name) wt.m[weights[i].name] = {weights[i].data, weights[i].numel};

#define TRY(x)                 \
  do {                         \
    rc = (x);                  \
    if (rc) {                  \
      odevio_plan_destroy(p);  \
      return rc;               \
    }                          \
  } while (0)

  std::vector<float> w, t, sc, sh, bias;
  // arithmetic of the encoder: the config's `arith` (--dtype), which the diagnostic ODEVIO_CONV_MATH overrides
  if (cfg->arith < 0 || cfg->arith > ODEVIO_ARITH_FP16) {
    odevio_plan_destroy(p);
    return fail(ODEVIO_ERR_BAD_ARG, "odevio_config.arith must be 0 (fp32), 1 (fp32_mfma) or 2 (fp16)");
  }
  p->conv_math = cfg->arith == ODEVIO_ARITH_FP32 ? 1 : (cfg->arith == ODEVIO_ARITH_FP32_MFMA ? 0 : 2);
  if (const char* cm = getenv("ODEVIO_CONV_MATH")) {
    if (!strcmp(cm, "f32")) p->conv_math = 0;
    else if (!strcmp(cm, "f16x2")) p->conv_math = 1;
    else if (!strcmp(cm, "f16")) p->conv_math = 2;   // reduced precision: fp16 operands (h piece only), fp32 accumulate
    else {
      odevio_plan_destroy(p);
      return fail(ODEVIO_ERR_BAD_ARG, "ODEVIO_CONV_MATH must be f16x2, f32 or f16");
    }
  }
  // ---- image encoder
  TRY(dev_alloc(p, &p->zero_page, ODEVIO_ZERO_PAGE_BYTES));
  HIPCHK(hipMemsetAsync(p->zero_page, 0, ODEVIO_ZERO_PAGE_BYTES, st));
  p->conv_h[0] = cfg->img_h;
  p->conv_w_sp[0] = cfg->img_w;
  double ex2_in = 1.0 / 12.0;   // second moment of the layer input in true units; frames are ToTensor() - 0.5: U(-0.5, 0.5)
  for (int i = 0; i < 9; ++i) {
    const ConvSpec& cs = kConvs[i];
    p->conv_h[i + 1] = conv_out(p->conv_h[i], cs.k, cs.stride);
    p->conv_w_sp[i + 1] = conv_out(p->conv_w_sp[i], cs.k, cs.stride);
    const std::string pre = std::string("Image_net.") + cs.name;
    TRY(wt.get(pre + ".0.weight", (int64_t)cs.cout * cs.cin * cs.k * cs.k, w));
    t.resize(w.size());
    if (i == 0) {  // [294][64]: k = c*49 + kh*7 + kw
      for (int n = 0; n < 64; ++n)
        for (int k = 0; k < 294; ++k) t[(size_t)k * 64 + n] = w[(size_t)n * 294 + k];
    } else {  // [Cout][Cin][kh][kw] -> [Cout][kh][kw][Cin]
      const int kk = cs.k * cs.k;
      for (int n = 0; n < cs.cout; ++n)
        for (int c = 0; c < cs.cin; ++c)
          for (int q = 0; q < kk; ++q) t[((size_t)n * kk + q) * cs.cin + c] = w[((size_t)n * cs.cin + c) * kk + q];
    }
    TRY(upload(p, &p->conv_w[i], t, st));
    float prescale = 1.f;
    if (i == 0) {  // fp16x2 conv1: [piece][(c,kh) row][channel][8 kw slots] (kw = 7 is zero), see conv1_f16x2.hip
      float wmax = 0.f;
      for (float x : w) wmax = std::max(wmax, std::fabs(x));
      int e = 0;
      if (wmax > 0.f && std::isfinite(wmax)) {
        (void)std::frexp(wmax, &e);
        e = std::max(-40, std::min(40, 14 - e));
      }
      prescale = std::ldexp(1.0f, e);
      std::vector<uint16_t> ws((size_t)2 * 42 * 64 * 8, 0);
      for (int n = 0; n < 64; ++n)
        for (int c = 0; c < 6; ++c)
          for (int kh = 0; kh < 7; ++kh)
            for (int kw = 0; kw < 7; ++kw) {
              const float x = w[(((size_t)n * 6 + c) * 7 + kh) * 7 + kw] * prescale;
              const _Float16 h = (_Float16)x;
              const _Float16 l = (_Float16)(x - (float)h);
              const size_t base = ((size_t)(c * 7 + kh) * 64 + n) * 8 + kw;
              memcpy(&ws[base], &h, 2);
              memcpy(&ws[(size_t)42 * 64 * 8 + base], &l, 2);
            }
      p->conv_ws_bytes[0] = ws.size() * sizeof(uint16_t);
      TRY(dev_alloc(p, &p->conv_ws[0], ws.size() * sizeof(uint16_t)));
      HIPCHK(hipMemcpyAsync(p->conv_ws[0], ws.data(), ws.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    if (i > 0) {
      std::vector<uint16_t> ws;
      prescale = split_conv_weights(w, cs.cout, cs.cin, cs.k * cs.k, ws);
      p->conv_ws_bytes[i] = ws.size() * sizeof(uint16_t);
      ws.resize(ws.size() + ODEVIO_ZERO_PAGE_BYTES / sizeof(uint16_t), 0);   // zero tail (32-bit DMA addressing)
      TRY(dev_alloc(p, &p->conv_ws[i], ws.size() * sizeof(uint16_t)));
      HIPCHK(hipMemcpyAsync(p->conv_ws[i], ws.data(), ws.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
    }
    TRY(bn_fold(wt, pre + ".1", cs.cout, nullptr, sc, sh));
    {
      std::vector<float> sch(sc);
      for (float& v : sch) v /= prescale;   // exact: prescale is a power of two
      TRY(upload(p, &p->conv_scale_h[i], sch, st));
      std::vector<float> raw((size_t)cs.cout, 1.0f / prescale), gm, bt;   // train mode: z = conv(x) itself
      TRY(upload(p, &p->conv_scale_raw[i], raw, st));
      TRY(wt.get(pre + ".1.weight", cs.cout, gm));
      TRY(wt.get(pre + ".1.bias", cs.cout, bt));
      TRY(upload(p, &p->conv_gamma[i], gm, st));
      TRY(upload(p, &p->conv_beta[i], bt, st));
      std::vector<float> zc((size_t)cs.cout, 0.f);
      TRY(upload(p, &p->enc_mean[i], zc, st));
      TRY(upload(p, &p->enc_invstd[i], zc, st));
      if (i > 0) {   // input gradient = stride-1 convolution with the filter's taps reversed and its channel roles swapped
        const int kk = cs.k * cs.k;
        std::vector<float> wT(w.size());
        for (int n = 0; n < cs.cout; ++n)
          for (int c = 0; c < cs.cin; ++c)
            for (int q = 0; q < kk; ++q) wT[((size_t)c * kk + (kk - 1 - q)) * cs.cout + n] = w[((size_t)n * cs.cin + c) * kk + q];
        TRY(upload(p, &p->conv_wT[i], wT, st));
        // [Cin][Cout][taps reversed] -> pieces, K-tile = (group of Cout, tap)
        std::vector<float> wr(w.size());
        for (int n = 0; n < cs.cout; ++n)
          for (int c = 0; c < cs.cin; ++c)
            for (int q = 0; q < kk; ++q) wr[((size_t)c * cs.cout + n) * kk + (kk - 1 - q)] = w[((size_t)n * cs.cin + c) * kk + q];
        std::vector<uint16_t> wts;
        const float ps = split_conv_weights(wr, cs.cin, cs.cout, kk, wts);
        p->conv_wT_inv_prescale[i] = 1.0f / ps;
        p->conv_wTs_bytes[i] = wts.size() * sizeof(uint16_t);
        wts.resize(wts.size() + ODEVIO_ZERO_PAGE_BYTES / sizeof(uint16_t), 0);
        TRY(dev_alloc(p, &p->conv_wTs[i], wts.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpyAsync(p->conv_wTs[i], wts.data(), wts.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        if (cs.k == 5 && cs.stride == 2) {
          // The input gradient of a stride-2 block without multiplying the zeros of a dilated gradient: input pixel (2i + py, 2j + px)
          // only meets the taps kh = py (mod 2), kw = px (mod 2) (pad 2), so each parity class is a stride-1 3 x 3 correlation of
          // the UNdilated gradient:  g_x[2i+py][2j+px] = sum_t D[i + ty - 1][j + tx - 1] W[2 (2 - ty) + py][2 (2 - tx) + px]
          // (taps that fall outside 0..4 are zero: 9 + 6 + 6 + 4 of 36 slots carry weights - against 25 of 100 slots of the dilated form).
          std::vector<float> wc((size_t)4 * cs.cin * cs.cout * 9, 0.f);
          for (int cl = 0; cl < 4; ++cl)
            for (int c = 0; c < cs.cin; ++c)
              for (int n = 0; n < cs.cout; ++n)
                for (int ty = 0; ty < 3; ++ty)
                  for (int tx = 0; tx < 3; ++tx) {
                    const int kh = 2 * (2 - ty) + (cl >> 1), kw = 2 * (2 - tx) + (cl & 1);
                    if (kh > 4 || kw > 4) continue;
                    wc[(((size_t)cl * cs.cin + c) * cs.cout + n) * 9 + ty * 3 + tx] = w[((size_t)n * cs.cin + c) * 25 + kh * 5 + kw];
                  }
          std::vector<uint16_t> wps;
          const float ps4 = split_conv_weights(wc, 4 * cs.cin, cs.cout, 9, wps);    // one prescale for the four (= the whole filter's: every tap is in one class)
          if (1.0f / ps4 != p->conv_wT_inv_prescale[i]) return fail(ODEVIO_ERR_BAD_ARG, "parity sub-filters: prescale differs from the whole filter's");
          const size_t per = wps.size() / 4;
          p->conv_wTp_bytes[i] = per * sizeof(uint16_t);
          for (int cl = 0; cl < 4; ++cl) {
            std::vector<uint16_t> one(wps.begin() + cl * per, wps.begin() + (cl + 1) * per);
            one.resize(one.size() + ODEVIO_ZERO_PAGE_BYTES / sizeof(uint16_t), 0);
            TRY(dev_alloc(p, &p->conv_wTp[i][cl], one.size() * sizeof(uint16_t)));
            HIPCHK(hipMemcpyAsync(p->conv_wTp[i][cl], one.data(), one.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
          }
        }
        if (cs.k == 3 && cs.stride == 2) {
          // the same for the 3 x 3 stride-2 blocks (pad 1): class (py, px) is a (1 + py) x (1 + px) correlation with NO padding whose output
          // is as large as the gradient (the last row / column sees its second tap hang over the edge = zeros):
          //   g_x[2i][.] = D[i] W[1];   g_x[2i+1][.] = D[i] W[2] + D[i+1] W[0]          (per dimension) - 9 filter slots per four pixels, not 36
          for (int cl = 0; cl < 4; ++cl) {
            const int py = cl >> 1, px = cl & 1, KHc = 1 + py, KWc = 1 + px;
            std::vector<float> wc((size_t)cs.cin * cs.cout * KHc * KWc);
            for (int c = 0; c < cs.cin; ++c)
              for (int n = 0; n < cs.cout; ++n)
                for (int ty = 0; ty < KHc; ++ty)
                  for (int tx = 0; tx < KWc; ++tx) {
                    const int kh = py ? 2 * (1 - ty) : 1, kw = px ? 2 * (1 - tx) : 1;
                    wc[(((size_t)c * cs.cout + n) * KHc + ty) * KWc + tx] = w[((size_t)n * cs.cin + c) * 9 + kh * 3 + kw];
                  }
            std::vector<uint16_t> one;
            (void)split_conv_weights(wc, cs.cin, cs.cout, KHc * KWc, one, ps);
            p->conv_wTq_bytes[i][cl] = one.size() * sizeof(uint16_t);
            one.resize(one.size() + ODEVIO_ZERO_PAGE_BYTES / sizeof(uint16_t), 0);
            TRY(dev_alloc(p, &p->conv_wTp[i][cl], one.size() * sizeof(uint16_t)));
            HIPCHK(hipMemcpyAsync(p->conv_wTp[i][cl], one.data(), one.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
          }
        }
      }
    }
    TRY(upload(p, &p->conv_scale[i], sc, st));
    TRY(upload(p, &p->conv_shift[i], sh, st));
    {
      // Activation exponent of this layer's output from a propagated second moment (inputs treated as independent,
      // zero-mean): E[conv^2] = fan_in * mean(w^2) * E[x^2]; after BatchNorm E[y^2] = mean_c(s_c^2) E[conv^2] + mean_c(t_c^2);
      // LeakyReLU(0.1) of a symmetric y keeps (1 + 0.01) / 2 of it.  The stored activation is out * 2^e with 2^e ~ 1 / rms(out):
      // an fp32 value is carried as two fp16 pieces with an ABSOLUTE resolution of 2^-25, so a layer whose activations are
      // all tiny (or huge) would lose relative accuracy (or leave the fp16 range) without it.  Networks with BatchNorm'd
      // O(1) activations get e = 0 and bit-identical results.
      double mw2 = 0.0, ms2 = 0.0, mt2 = 0.0;
      for (float x : w) mw2 += (double)x * x;
      mw2 /= (double)w.size();
      for (int c = 0; c < cs.cout; ++c) { ms2 += (double)sc[c] * sc[c]; mt2 += (double)sh[c] * sh[c]; }
      ms2 /= cs.cout; mt2 /= cs.cout;
      const double econv2 = (double)cs.cin * cs.k * cs.k * mw2 * ex2_in;
      const double eout2 = (ms2 * econv2 + mt2) * 0.505;
      int e = 0;
      if (eout2 > 0.0 && std::isfinite(eout2)) e = (int)std::lround(-0.5 * std::log2(eout2));
      e = std::max(-24, std::min(24, e));
      if (std::abs(e) <= 2) e = 0;            // already O(1): leave the constants (and the bits) alone
      p->act_exp[i + 1] = e;
      std::vector<float> scc(sc), shc(sh);
      const float fs = std::ldexp(1.0f, e - p->act_exp[i]) / prescale, ft = std::ldexp(1.0f, e);
      for (float& v : scc) v *= fs;
      for (float& v : shc) v *= ft;
      TRY(upload(p, &p->conv_scale_c[i], scc, st));
      TRY(upload(p, &p->conv_shift_c[i], shc, st));
      ex2_in = eout2;
    }
  }
  {
    const int oh = p->conv_h[9], ow = p->conv_w_sp[9];
    p->head_k = 1024 * oh * ow;
    if (p->head_k % 32) {
      odevio_plan_destroy(p);
      return fail(ODEVIO_ERR_UNSUPPORTED, "encoder output size unsupported");
    }
    TRY(wt.get("Image_net.visual_head.weight", (int64_t)cfg->v_f_len * p->head_k, w));
    t.resize(w.size());  // reference flattens (C,H,W); our activations are (H,W,C)
    for (int n = 0; n < cfg->v_f_len; ++n)
      for (int c = 0; c < 1024; ++c)
        for (int s = 0; s < oh * ow; ++s)
          t[(size_t)n * p->head_k + (size_t)s * 1024 + c] = w[(size_t)n * p->head_k + (size_t)c * oh * ow + s];
    TRY(upload(p, &p->head_w, t, st));
    {  // the same matrix as two fp16 pieces (K = (H,W,C) order = the P2 layout of conv6's output, one 'pixel' per pair)
      std::vector<uint16_t> ws;
      const float prescale = split_conv_weights(t, cfg->v_f_len, p->head_k, 1, ws);
      p->head_ws_bytes = ws.size() * sizeof(uint16_t);
      ws.resize(ws.size() + ODEVIO_ZERO_PAGE_BYTES / sizeof(uint16_t), 0);
      TRY(dev_alloc(p, &p->head_ws, ws.size() * sizeof(uint16_t)));
      HIPCHK(hipMemcpyAsync(p->head_ws, ws.data(), ws.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
      HIPCHK(hipStreamSynchronize(st));
      std::vector<float> hs((size_t)cfg->v_f_len, std::ldexp(1.0f, -p->act_exp[9]) / prescale);   // conv6's stored output carries 2^act_exp[9]
      TRY(upload(p, &p->head_scale_h, hs, st));
      std::vector<float> hr((size_t)cfg->v_f_len, 1.0f / prescale);   // train mode: activations carry no exponent
      TRY(upload(p, &p->head_scale_raw, hr, st));
    }
    TRY(wt.get("Image_net.visual_head.bias", cfg->v_f_len, bias));
    TRY(upload(p, &p->head_b, bias, st));
  }
  // ---- inertial encoder
  {
    const int cin[3] = {6, 64, 128}, cout[3] = {64, 128, 256}, idx[3] = {0, 4, 8};
    for (int i = 0; i < 3; ++i) {
      const std::string pre = "Inertial_net.encoder_conv." + std::to_string(idx[i]);
      TRY(wt.get(pre + ".weight", (int64_t)cout[i] * cin[i] * 3, w));
      TRY(wt.get(pre + ".bias", cout[i], bias));
      t.resize(w.size());  // [co][ci][k] -> [(ci,k)][co]
      for (int co = 0; co < cout[i]; ++co)
        for (int ci = 0; ci < cin[i]; ++ci)
          for (int k = 0; k < 3; ++k) t[((size_t)ci * 3 + k) * cout[i] + co] = w[((size_t)co * cin[i] + ci) * 3 + k];
      TRY(upload(p, &p->imu_w[i], t, st));
      TRY(bn_fold(wt, "Inertial_net.encoder_conv." + std::to_string(idx[i] + 1), cout[i], &bias, sc, sh));
      TRY(upload(p, &p->imu_s[i], sc, st));
      TRY(upload(p, &p->imu_h[i], sh, st));
      {   // for odevio_imu_encoder_bwd: [cout][ldk] with ldk = 3 cin rounded up to 16 (zero pad), the conv bias, the running statistics
        const int k3 = 3 * cin[i], ldk = (k3 + 15) / 16 * 16;
        std::vector<float> wr((size_t)cout[i] * ldk, 0.f), mu, var;
        for (int co = 0; co < cout[i]; ++co) memcpy(&wr[(size_t)co * ldk], &w[(size_t)co * k3], k3 * sizeof(float));
        TRY(upload(p, &p->imu_wref[i], wr, st));
        TRY(upload(p, &p->imu_bias[i], bias, st));
        const std::string bn = "Inertial_net.encoder_conv." + std::to_string(idx[i] + 1);
        TRY(wt.get(bn + ".running_mean", cout[i], mu));
        TRY(wt.get(bn + ".running_var", cout[i], var));
        TRY(upload(p, &p->imu_mean[i], mu, st));
        TRY(upload(p, &p->imu_var[i], var, st));
        std::vector<float> gm, bt;
        TRY(wt.get(bn + ".weight", cout[i], gm));
        TRY(wt.get(bn + ".bias", cout[i], bt));
        TRY(upload(p, &p->imu_gamma[i], gm, st));
        TRY(upload(p, &p->imu_beta[i], bt, st));
      }
    }
    TRY(wt.get("Inertial_net.proj.weight", (int64_t)cfg->i_f_len * 2816, w));
    TRY(upload(p, &p->proj_w, w, st));
    TRY(wt.get("Inertial_net.proj.bias", cfg->i_f_len, bias));
    TRY(upload(p, &p->proj_b, bias, st));
  }
  // ---- Pose_net: fusion, regressor, ODEFunc (column-sharded), RNN stack
  TRY(load_pose_net(p, wt, st));
  // ---- Neural-CDE: initial layer, CDEFunc (reference PoseCDE.py:59-66, ODEFunc.py:52-58); reduction_net is unused
  if (cfg->model_type == ODEVIO_MODEL_CDE) {
    const int Hc = cfg->cde_hidden_dim, C = Hc + 1, nh = cfg->cde_fn_num_layers;
    TRY(wt.get("Pose_net.initial.0.weight", (int64_t)Hc * C, w));
    TRY(upload(p, &p->cde_init_w, w, st));
    TRY(wt.get("Pose_net.initial.0.bias", Hc, bias));
    TRY(upload(p, &p->cde_init_b, bias, st));
    p->cde.H = Hc; p->cde.C = C; p->cde.n_hidden = nh; p->cde.act = cfg->cde_activation;
    p->cde.atol = 1e-6f; p->cde.rtol = 1e-4f;  // PoseCDE.py:101
    p->cde.solver = cfg->cde_solver == ODEVIO_DOPRI5 ? 0 : (cfg->cde_solver == ODEVIO_EULER ? 2 : 1);
    p->cde.max_steps = cfg->max_steps;
    for (int l = 0; l <= nh; ++l) {
      const std::string pre = "Pose_net.cde_func.net." + std::to_string(2 * l);
      const int64_t N = l < nh ? Hc : (int64_t)Hc * C;
      float *dw = nullptr, *db = nullptr;
      TRY(wt.get(pre + ".weight", N * Hc, w));
      TRY(upload(p, &dw, w, st));
      TRY(wt.get(pre + ".bias", N, bias));
      TRY(upload(p, &db, bias, st));
      p->cde.w[l] = dw;
      p->cde.b[l] = db;
      if (l == nh && p->conv_math == 2) {
        // reduced-precision mode (--dtype bf16 / fp16): the last layer - 99.9 % of the model's bytes - is ALSO kept as bf16
        // (round to nearest even); the weight-stream kernel then moves half the bytes and widens in registers
        std::vector<uint16_t> h16(w.size());
        for (size_t i = 0; i < w.size(); ++i) {
          uint32_t u;
          memcpy(&u, &w[i], 4);
          if ((u & 0x7fffffffu) > 0x7f800000u) h16[i] = (uint16_t)((u >> 16) | 0x40);   // NaN stays a NaN
          else h16[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        void* d16 = nullptr;
        TRY(dev_alloc(p, &d16, h16.size() * sizeof(uint16_t)));
        HIPCHK(hipMemcpyAsync(d16, h16.data(), h16.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        p->cde.w_last16 = d16;
      }
    }
  }
  {
    std::vector<float> z(1024, 0.f);
    TRY(upload(p, &p->zero_vec, z, st));
    TRY(upload(p, &p->bn_scale, z, st));
    TRY(upload(p, &p->bn_shift, z, st));
    std::vector<float> z2(1024 + 16, 0.f);
    TRY(upload(p, &p->enc_dscale, z2, st));
  }
  // ---- exchange buffers + status
  p->xstride = 8 * INTEG_KMAX;
  TRY(dev_alloc(p, (void**)&p->xbuf, (size_t)INTEG_GROUPS * 2 * p->xstride * sizeof(unsigned long long)));
  TRY(dev_alloc(p, (void**)&p->status, 128));
  HIPCHK(hipMemsetAsync(p->status, 0, 128, st));
  HIPCHK(hipHostMalloc((void**)&p->status_host, 8 * sizeof(int), hipHostMallocDefault));
  memset(p->status_host, 0, 8 * sizeof(int));
  HIPCHK(hipEventCreateWithFlags(&p->ev_status, hipEventDisableTiming));
  HIPCHK(hipStreamSynchronize(st));
#undef TRY
  *out_plan = p;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Split-K factor from a small cost model.  The chip runs 512 workgroups of this kernel at a time (2 per CU); a layer
// whose tile count is not a multiple of that leaves CUs idle in its last round (conv5: 640 tiles = 1.25 rounds,
// conv6: 320 = 0.6).  Splitting K multiplies the workgroups and shortens each; the price is the fp32 slab round trip
// of the deterministic combine.  Units: microseconds, with ~4.1 us per K-tile step of a CU running two workgroups.
static int pick_splitk(int M, int N, int nk) {
  const double tiles = (double)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles >= 2048 || nk < 8) return 1;
  int best = 1;
  double best_cost = 1e30;
  for (int s : {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64}) {
    if (s > 1 && nk / s < 8) break;
    const double rounds = std::ceil(tiles * s / 512.0);
    const double steps = std::ceil((double)nk / s) + 3.0;  // + prologue / epilogue of a workgroup
    double cost = rounds * steps * 4.1;
    if (s > 1) cost += (s + 1.0) * M * N * 4.0 / 4.0e6 + 8.0;  // slab write + read at ~4 TB/s, one more launch
    if (cost < best_cost * 0.97) {  // prefer the smaller factor unless the gain is real
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

static int run_gemm(odevio_plan* p, const float* in, int M, int K, const float* W, int N, const float* scale,
                    const float* shift, const float* mul, int ld_mul, float* out, int ld_out, int act, float slope,
                    hipStream_t st, DevBuf* slabs = nullptr) {
  if (!slabs) slabs = &p->partial;   // split-K slabs; work that runs on the side stream brings its own
  ConvArgs a{};
  a.in = in; a.w = W; a.scale = scale; a.shift = shift; a.mul = mul; a.out = out;
  a.N = M; a.Hi = a.Wi = a.Ho = a.Wo = 1; a.Cin = K; a.Cout = N; a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
  a.M = M; a.ld_out = ld_out; a.ld_mul = ld_mul; a.act = act; a.slope = slope;
  const int nk = K / 32;
  a.splitk = pick_splitk(M, N, nk);
  a.ktiles_per_split = (nk + a.splitk - 1) / a.splitk;
  a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
  if (a.splitk > 1) {
    int rc = ensure(*slabs, (size_t)a.splitk * M * N);
    if (rc) return rc;
    a.partial = slabs->p;
  }
  launch_conv_igemm(a, st);
  return 0;
}

// Same cost model for the fp16x2 kernel: 256 x 128 tiles, one workgroup per CU, K-tiles of 32 channels at ~1.4 us
// (measured: conv3_1 = 10 rounds x 78 steps in 1.2 ms).
static int pick_splitk_h(int M, int N, int nk) {
  const double tiles = (double)((M + 255) / 256) * ((N + 127) / 128);
  if (tiles >= 1024 || nk < 24) return 1;
  int best = 1;
  double best_cost = 1e30;
  for (int s : {1, 2, 3, 4, 6, 8, 12, 16, 32, 64}) {
    if (s > 1 && nk / s < 12) break;
    const double rounds = std::ceil(tiles * s / 256.0);
    const double steps = std::ceil((double)nk / s) + 6.0;
    double cost = rounds * steps * 1.4;
    if (s > 1) cost += (s + 1.0) * M * N * 4.0 / 4.0e6 + 8.0;
    if (cost < best_cost * 0.97) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

// ---- tile plan of one fp16x2 layer.  The kernel has four tile shapes (256 or 192 pixels x 128 or 256 channels); a layer
// is run as ONE shape (optionally split-K), or as R full rounds of the chip in one shape followed by the remaining pixels in
// another, so that no round is left half empty: e.g. conv4 at the bench batch (81,920 pixels x 512 channels) = 2 rounds of
// 256 x 256 tiles + exactly 1 round of 256 x 128 tiles, where 256 x 128 alone needs 5 rounds of twice the staged bytes per
// flop and 256 x 256 alone 2.5 (= 3) rounds.  Costs are the measured K-tile and per-tile times of each shape (DESIGN.md
// section 5.7, tools/conv_stamps.py); the plan only has to rank alternatives.
struct ConvPhase { int wide, bm, m_begin, m_end, splitk; };
struct ConvPlanF { int n; ConvPhase ph[2]; double cost; };
struct ConvShape { int bm, wide; double kt_us, tile_us; };
static const ConvShape kShapes[4] = {{256, 0, 1.10, 9.0}, {256, 1, 1.85, 15.0}, {192, 0, 0.86, 7.5}, {192, 1, 1.45, 12.0}};

static double single_cost(const ConvShape& sh, long M, int N, int nk, int s, int n_cu) {
  const int bn = sh.wide ? 256 : 128;
  const long tiles = ((M + sh.bm - 1) / sh.bm) * ((N + bn - 1) / bn) * s;
  const double rounds = std::ceil((double)tiles / n_cu);
  double cost = rounds * (std::ceil((double)nk / s) * sh.kt_us + sh.tile_us);
  if (s > 1) cost += (s + 1.0) * M * N * 4.0 / 2.0e6 + 15.0;  // slabs written and read (~2 TB/s effective beside the tiles), one more launch
                                                                // (measured: conv6 as 256 x 256 split-K 3 is 0.07 ms slower than 192 x 128 unsplit)
  return cost;
}

static bool parse_shape(const char* t, ConvShape& out) {
  for (const ConvShape& sh : kShapes) {
    char name[8];
    snprintf(name, sizeof name, "%s%s", sh.wide ? "w" : "n", sh.bm == 192 ? "192" : "");
    if (!strcmp(t, name)) { out = sh; return true; }
  }
  return false;
}

static ConvPlanF plan_f16x2(int layer, int M, int N, int nk, int n_cu, bool off32) {
  ConvPlanF best{};
  best.cost = 1e30;
  auto allowed = [&](const ConvShape& sh) { return (!sh.wide || N % 256 == 0) && (sh.bm == 256 || off32); };
  // diagnostic override: ODEVIO_CONV_FORCE="4:w:2:n,5:w192,6:w192:s2" = layer:shape[:rounds:shape2][:sK]
  if (const char* env = getenv("ODEVIO_CONV_FORCE")) {
    std::string all(env);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos);
      if (end == std::string::npos) end = all.size();
      std::vector<std::string> tok;
      std::string item = all.substr(pos, end - pos);
      pos = end + 1;
      size_t q = 0;
      while (q <= item.size()) {
        size_t e = item.find(':', q);
        if (e == std::string::npos) e = item.size();
        tok.push_back(item.substr(q, e - q));
        q = e + 1;
      }
      ConvShape A, B;
      if (tok.size() < 2 || atoi(tok[0].c_str()) != layer || !parse_shape(tok[1].c_str(), A) || !allowed(A)) continue;
      ConvPlanF f{};
      f.n = 1;
      f.ph[0] = {A.wide, A.bm, 0, M, 1};
      if (tok.size() >= 4 && parse_shape(tok[3].c_str(), B) && allowed(B)) {
        const int ntA = N / (A.wide ? 256 : 128);
        const long m1 = (long)atoi(tok[2].c_str()) * n_cu / ntA * A.bm;
        if (m1 > 0 && m1 < M) {
          f.n = 2;
          f.ph[0].m_end = (int)m1;
          f.ph[1] = {B.wide, B.bm, (int)m1, M, 1};
        }
      } else if (tok.size() >= 3 && tok[2][0] == 's') {
        f.ph[0].splitk = std::max(1, atoi(tok[2].c_str() + 1));
      }
      return f;
    }
  }
  for (const ConvShape& sh : kShapes) {
    if (!allowed(sh)) continue;
    for (int s : {1, 2, 3, 4, 6, 8, 12, 16, 32, 64}) {
      if (s > 1 && (nk / s < 12 || nk < 24)) break;
      const double c = single_cost(sh, M, N, nk, s, n_cu);
      if (c < best.cost * 0.97) {
        best.cost = c;
        best.n = 1;
        best.ph[0] = {sh.wide, sh.bm, 0, M, s};
      }
    }
  }
  for (const ConvShape& A : kShapes) {
    if (!allowed(A)) continue;
    const int ntA = (N + (A.wide ? 255 : 127)) / (A.wide ? 256 : 128);
    const long tilesA = (long)((M + A.bm - 1) / A.bm) * ntA;
    for (long R = 1; R * n_cu < tilesA; ++R) {
      if (R * n_cu % ntA) continue;
      const long m1 = R * n_cu / ntA * A.bm;
      if (m1 >= M) break;
      const double c1 = R * (nk * A.kt_us + A.tile_us);
      for (const ConvShape& B : kShapes) {
        if (!allowed(B)) continue;
        const double c = c1 + single_cost(B, M - m1, N, nk, 1, n_cu) + 3.0;   // (+ the gap between two launches)
        if (c < best.cost * 0.97) {
          best.cost = c;
          best.n = 2;
          best.ph[0] = {A.wide, A.bm, 0, (int)m1, 1};
          best.ph[1] = {B.wide, B.bm, (int)m1, M, 1};
        }
      }
    }
  }
  return best;
}

// One encoder block.  Activations between blocks live in the P2 split layout when the fp16x2 kernel is in use
// (in_split / out_split); fp32 NHWC otherwise.
// mode 0: the block alone (activations in true units); 1: the production chain (per-layer activation exponents); 2: train mode -
// an identity epilogue, z = conv(x): BatchNorm with batch statistics, LeakyReLU and Dropout follow as their own passes (bn_train.hip)
static int conv_block(odevio_plan* p, int i, const void* in, int B, int S, void* out, bool in_split, bool out_split,
                      hipStream_t st, bool in_u8 = false, int mode = 0) {
  const bool chain = mode == 1;
  const int P = B * (S - 1);
  const ConvSpec& cs = kConvs[i];
  if (i == 0) {
    Conv1Args a{};
    a.img = (const float*)in; a.wt = p->conv_w[0]; a.scale = p->conv_scale[0]; a.shift = p->conv_shift[0]; a.out = out;
    a.out_split = out_split; a.status = p->status;
    a.B = B; a.S = S; a.H = p->conv_h[0]; a.W = p->conv_w_sp[0]; a.Ho = p->conv_h[1]; a.Wo = p->conv_w_sp[1];
    a.tiles_y = (a.Ho + 7) / 8; a.tiles_x = (a.Wo + 31) / 32; a.n_tiles = P * a.tiles_y * a.tiles_x; a.slope = 0.1f;
    if (p->conv_math != 0) {
      // frames -> zero-bordered fp16-piece planes once per forward (also the uint8 entry: byte / 255 - 0.5)
      IngestArgs g{};
      g.src = in; g.src_u8 = in_u8; g.n_frames = B * S; g.H = a.H; g.W = a.W;
      g.Hp = 16 * a.tiles_y + 8; g.Wp = 64 * a.tiles_x + 8;
      int rc = ensure(p->ingest, (size_t)g.n_frames * 3 * g.Hp * g.Wp);   // 2 pieces x 2 bytes = one float per pixel
      if (rc) return rc;
      g.planes = p->ingest.p; g.planes_bytes = p->ingest.n * sizeof(float); g.status = p->status;
      launch_ingest(g, st);
      a.planes = g.planes; a.zeros = p->zero_page; a.Hp = g.Hp; a.Wp = g.Wp;
      a.planes_bytes = g.planes_bytes;
      a.out_bytes = extent_of(p, out, (size_t)P * a.Ho * a.Wo * 64 * sizeof(float));
      a.wt16 = p->conv_ws[0];
      a.scale = chain ? p->conv_scale_c[0] : p->conv_scale_h[0];
      if (chain) a.shift = p->conv_shift_c[0];
      if (mode == 2) { a.scale = p->conv_scale_raw[0]; a.shift = p->zero_vec; a.slope = 1.0f; }
      a.terms = p->conv_math == 2 ? 1 : 3;
      HIPCHK(launch_conv1_f16x2(a, p->n_cu, st));
    } else {
      if (in_u8) return fail(ODEVIO_ERR_UNSUPPORTED, "uint8 frames need the fp16x2 encoder (unset ODEVIO_CONV_MATH=f32)");
      launch_conv1(a, p->n_cu, st);
    }
    return 0;
  }
  if (in_split) {
    ConvSplitArgs a{};
    a.in = in; a.w = p->conv_ws[i]; a.zeros = p->zero_page; a.out = out; a.status = p->status;
    a.scale = chain ? p->conv_scale_c[i] : p->conv_scale_h[i];
    a.shift = chain ? p->conv_shift_c[i] : p->conv_shift[i];
    a.N = P; a.Hi = p->conv_h[i]; a.Wi = p->conv_w_sp[i]; a.Cin = cs.cin; a.Ho = p->conv_h[i + 1]; a.Wo = p->conv_w_sp[i + 1];
    a.Cout = cs.cout; a.KH = a.KW = cs.k; a.stride = cs.stride; a.pad = (cs.k - 1) / 2;
    a.M = P * a.Ho * a.Wo; a.slope = 0.1f; a.out_split = out_split; a.ld_out = cs.cout; a.terms = p->conv_math == 2 ? 1 : 3;
    if (mode == 2) { a.scale = p->conv_scale_raw[i]; a.shift = p->zero_vec; a.slope = 1.0f; }
    a.in_bytes = extent_of(p, in, (size_t)P * a.Hi * a.Wi * a.Cin * sizeof(float));
    a.w_bytes = p->conv_ws_bytes[i];
    set_off32(p, a, in);
    {
      static const char* sl = getenv("ODEVIO_STAMP_LAYER");   // diagnostic build: which layer's launch writes the phase stamps
      static const char* sw = getenv("ODEVIO_STAMP_WG");      // ... and which workgroup (linear index; default 0: a first-round one)
      a.stamp = (sl ? atoi(sl) == i : 1) ? 1 + (sw ? atoi(sw) : 0) : 0;
    }
    a.out_bytes = extent_of(p, out, (size_t)a.M * a.Cout * sizeof(float));
    const int nk = cs.k * cs.k * cs.cin / 32;
    static const char* legacy = getenv("ODEVIO_CONV_PLAN");   // diagnostic: "legacy" = round 1's rule (one shape per layer, 256-pixel tiles)
    ConvPlanF plan{};
    if (legacy && !strcmp(legacy, "legacy")) {
      const long wide_tiles = (long)((a.M + 255) / 256) * (a.Cout / 256);
      const double rounds = (double)wide_tiles / p->n_cu;
      const bool wide = a.Cout % 256 == 0 && wide_tiles >= 2L * p->n_cu && rounds / std::ceil(rounds) >= 0.9;
      plan.n = 1;
      plan.ph[0] = {wide, 256, 0, a.M, wide ? 1 : pick_splitk_h(a.M, a.Cout, nk)};
    } else {
      plan = plan_f16x2(i, a.M, a.Cout, nk, p->n_cu, a.off32 != 0);
    }
    {
      static const bool print = getenv("ODEVIO_CONV_PLAN_PRINT") != nullptr;   // diagnostic: each layer's plan, once
      static bool printed[16] = {};
      if (print && !printed[i]) {
        printed[i] = true;
        for (int ph = 0; ph < plan.n; ++ph)
          fprintf(stderr, "odevio conv plan: layer %d (M %d, Cout %d, %d K-tiles) phase %d: %d x %d tiles, pixels %d..%d, split-K %d (model %.0f us)\n", i,
                  a.M, a.Cout, nk, ph, plan.ph[ph].bm, plan.ph[ph].wide ? 256 : 128, plan.ph[ph].m_begin, plan.ph[ph].m_end, plan.ph[ph].splitk, plan.cost);
      }
    }
    for (int ph = 0; ph < plan.n; ++ph) {
      const ConvPhase& f = plan.ph[ph];
      a.wide = f.wide; a.bm = f.bm; a.m_begin = f.m_begin; a.m_end = f.m_end;
      a.splitk = f.splitk;
      a.ktiles_per_split = (nk + a.splitk - 1) / a.splitk;
      a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
      if (a.splitk > 1) {
        int rc = ensure(p->partial, (size_t)a.splitk * a.M * a.Cout);
        if (rc) return rc;
        a.partial = p->partial.p;
        a.partial_bytes = p->partial.n * sizeof(float);
      }
      HIPCHK(launch_conv_f16x2(a, st));
    }
    return 0;
  }
  if (out_split) return fail(ODEVIO_ERR_BAD_ARG, "conv_block: fp32-input blocks write fp32");
  ConvArgs a{};
  a.in = (const float*)in; a.w = p->conv_w[i]; a.scale = p->conv_scale[i]; a.shift = p->conv_shift[i]; a.out = (float*)out;
  a.N = P; a.Hi = p->conv_h[i]; a.Wi = p->conv_w_sp[i]; a.Cin = cs.cin; a.Ho = p->conv_h[i + 1]; a.Wo = p->conv_w_sp[i + 1];
  a.Cout = cs.cout; a.KH = a.KW = cs.k; a.stride = cs.stride; a.pad = (cs.k - 1) / 2;
  a.M = P * a.Ho * a.Wo; a.ld_out = cs.cout; a.act = EPI_LEAKY; a.slope = 0.1f;
  const int nk = cs.k * cs.k * cs.cin / 32;
  a.splitk = pick_splitk(a.M, a.Cout, nk);
  a.ktiles_per_split = (nk + a.splitk - 1) / a.splitk;
  a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
  if (a.splitk > 1) {
    int rc = ensure(p->partial, (size_t)a.splitk * a.M * a.Cout);
    if (rc) return rc;
    a.partial = p->partial.p;
  }
  launch_conv_igemm(a, st);
  return 0;
}

// Ping-pong activation buffers of the encoder, in floats (the P2 split layout is 4 bytes per element too).
static int ensure_act(odevio_plan* p, int P) {
  size_t nA = 0, nB = 0;
  for (int i = 0; i < 9; ++i) {
    size_t n = (size_t)P * p->conv_h[i + 1] * p->conv_w_sp[i + 1] * kConvs[i].cout;
    if (i % 2 == 0) nA = std::max(nA, n); else nB = std::max(nB, n);
  }
  int rc;
  if ((rc = ensure(p->actA, nA))) return rc;
  return ensure(p->actB, nB);
}

static int visual_head(odevio_plan* p, const float* cur, int P, float* fv, int ld_fv, const float* head_scale, hipStream_t st);

static int image_encoder(odevio_plan* p, const void* img, int B, int S, float* fv, int ld_fv, hipStream_t st,
                         bool img_u8 = false) {
  RoctxRange range("odevio: ImageEncoder");
  const int P = B * (S - 1);
  int rc;
  if ((rc = ensure_act(p, P))) return rc;
  const bool split = p->conv_math != 0;  // conv1 .. conv5_1 hand their output over in the P2 split layout
  stage_mark(p, 0, st);
  if ((rc = conv_block(p, 0, img, B, S, p->actA.p, false, split, st, img_u8, split ? 1 : 0))) return rc;
  stage_mark(p, 1, st);
  float* cur = p->actA.p;
  for (int i = 1; i < 9; ++i) {
    float* nxt = (cur == p->actA.p) ? p->actB.p : p->actA.p;
    if ((rc = conv_block(p, i, cur, B, S, nxt, split, split, st, false, split ? 1 : 0))) return rc;
    cur = nxt;
  }
  stage_mark(p, 2, st);
  rc = visual_head(p, cur, P, fv, ld_fv, p->head_scale_h, st);
  stage_mark(p, 3, st);
  return rc;
}

static int visual_head(odevio_plan* p, const float* cur, int P, float* fv, int ld_fv, const float* head_scale, hipStream_t st) {
  const bool split = p->conv_math != 0;
  int rc;
  if (split) {
    // visual head = a 1x1 convolution over conv6's P2 output seen as P 'pixels' of head_k channels: the 67 MB weight
    // stream split 64 ways over K so that every CU takes part
    ConvSplitArgs a{};
    a.in = cur; a.w = p->head_ws; a.zeros = p->zero_page; a.scale = head_scale; a.shift = p->head_b; a.out = fv;
    a.status = p->status;
    a.N = P; a.Hi = a.Wi = a.Ho = a.Wo = 1; a.Cin = p->head_k; a.Cout = p->cfg.v_f_len; a.KH = a.KW = 1; a.stride = 1; a.pad = 0;
    a.M = P; a.slope = 1.0f; a.out_split = 0; a.ld_out = ld_fv; a.terms = p->conv_math == 2 ? 1 : 3;
    const int nk = p->head_k / 32;
    a.splitk = pick_splitk_h(a.M, a.Cout, nk);
    a.ktiles_per_split = (nk + a.splitk - 1) / a.splitk;
    a.splitk = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
    if (a.splitk > 1) {
      if ((rc = ensure(p->partial, (size_t)a.splitk * a.M * a.Cout))) return rc;
      a.partial = p->partial.p;
      a.partial_bytes = p->partial.n * sizeof(float);
    }
    a.in_bytes = extent_of(p, cur, (size_t)P * p->head_k * sizeof(float));
    a.w_bytes = p->head_ws_bytes;
    set_off32(p, a, cur);
    a.out_bytes = extent_of(p, fv, ((size_t)(P - 1) * ld_fv + a.Cout) * sizeof(float));
    HIPCHK(launch_conv_f16x2(a, st));
    rc = 0;
  } else {
    rc = run_gemm(p, cur, P, p->head_k, p->head_w, p->cfg.v_f_len, nullptr, p->head_b, nullptr, 0, fv, ld_fv, EPI_NONE, 0.f, st);
  }
  return rc;
}

// Named device tensors a caller hands in for in-place update (BatchNorm running statistics): pointer by name, or null
static float* named_ptr(const odevio_tensor* t, int n, const std::string& name, int64_t numel) {
  for (int i = 0; i < n; ++i)
    if (t[i].name && t[i].data && name == t[i].name && t[i].numel == numel) return (float*)t[i].data;
  return nullptr;
}

// ImageEncoder.forward under model.train() (Encoder.py:97-122 with every block's BatchNorm2d in batch-statistics mode and its
// Dropout(0.2) - conv6: Dropout(0.5) - on, :82-90): per block conv (identity epilogue) -> batch statistics -> normalise +
// LeakyReLU + dropout in place.  Each block's mask is one draw of the plan's random stream (9 draws per call, conv1 first).
static int image_encoder_train(odevio_plan* p, const float* img, int B, int S, float* fv, int ld_fv, const odevio_tensor* stats, int n_stats,
                               bool keep, hipStream_t st) {
  RoctxRange range("odevio: ImageEncoder (train mode)");
  if (p->conv_math == 0) return fail(ODEVIO_ERR_UNSUPPORTED, "train-mode encoders need the fp16x2 encoder (unset ODEVIO_CONV_MATH=f32 / --dtype fp32_mfma)");
  const int P = B * (S - 1);
  int rc;
  if ((rc = ensure_act(p, P)) || (rc = ensure(p->bn_partial, (size_t)4 * BN_MAX_BLOCKS * 1024))) return rc;   // doubles in a float buffer
  p->enc_B = p->enc_S = 0;
  if (keep) {   // every block's bare convolution AND output stay behind for odevio_image_encoder_bwd (allocations before any launch)
    for (int i = 0; i < 9; ++i) {
      const size_t n = (size_t)P * p->conv_h[i + 1] * p->conv_w_sp[i + 1] * kConvs[i].cout;
      if ((rc = ensure(p->enc_z[i], n)) || (rc = ensure(p->enc_a[i], n))) return rc;
    }
    p->enc_seed = p->seed;
    p->enc_call0 = p->rng_calls;
  }
  float* cur = nullptr;
  for (int i = 0; i < 9; ++i) {
    const ConvSpec& cs = kConvs[i];
    float* nxt = (i == 0 || cur == p->actB.p) ? p->actA.p : p->actB.p;
    float* zbuf = keep ? p->enc_z[i].p : nxt;
    float* abuf = keep ? p->enc_a[i].p : nxt;
    if ((rc = conv_block(p, i, i == 0 ? (const void*)img : (const void*)cur, B, S, zbuf, i > 0, true, st, false, 2))) return rc;
    const size_t M = (size_t)P * p->conv_h[i + 1] * p->conv_w_sp[i + 1];
    const std::string bn = std::string("Image_net.") + cs.name + ".1";
    HIPCHK(bn_stats_p2(zbuf, M, cs.cout, reinterpret_cast<double*>(p->bn_partial.p), p->conv_gamma[i], p->conv_beta[i], 1e-5f, 0.1f,
                       named_ptr(stats, n_stats, bn + ".running_mean", cs.cout), named_ptr(stats, n_stats, bn + ".running_var", cs.cout),
                       p->bn_scale, p->bn_shift, keep ? p->enc_mean[i] : nullptr, keep ? p->enc_invstd[i] : nullptr, st));
    const DropoutSpec d = make_dropout(p->seed, p->rng_calls++, i == 8 ? 0.5f : 0.2f);
    HIPCHK(bn_apply_p2(zbuf, abuf, M, cs.cout, p->bn_scale, p->bn_shift, 0.1f, d, p->status, st));
    cur = abuf;
  }
  rc = visual_head(p, cur, P, fv, ld_fv, p->head_scale_raw, st);
  if (!rc && keep) { p->enc_B = B; p->enc_S = S; }
  return rc;
}

// Backward of image_encoder_train (the forward that ran last with keep = 1, same B and S): gradients of every Image_net parameter
// that is asked for, block by block from the head down (enc_bwd.h).  The input gradient of block i is a stride-1 convolution of the
// (zero-dilated, for the stride-2 blocks) gradient D of its bare convolution with the tap-reversed filter, on conv_igemm_kernel.
static int image_encoder_bwd(odevio_plan* p, const float* img, int B, int S, const float* g_fv, int ld_gfv, const odevio_tensor* grads, int n_grads,
                             hipStream_t st) {
  RoctxRange range("odevio: ImageEncoder backward");
  if (p->enc_B != B || p->enc_S != S)
    return fail(ODEVIO_ERR_BAD_ARG, "odevio_image_encoder_bwd: no kept train-mode forward of this shape (odevio_image_encoder_fwd_train with keep = 1 first)");
  const int P = B * (S - 1), V = p->cfg.v_f_len;
  auto grad_ptr = [&](const std::string& name, int64_t numel) { return named_ptr(grads, n_grads, name, numel); };
  for (int i = 0; i < n_grads; ++i) {
    if (!grads[i].name || !grads[i].data) return fail(ODEVIO_ERR_BAD_ARG, "odevio_image_encoder_bwd: gradient %d has no name / pointer", i);
    const std::string nm = grads[i].name;
    int64_t want = -1;
    for (int l = 0; l < 9 && want < 0; ++l) {
      const std::string pre = std::string("Image_net.") + kConvs[l].name;
      if (nm == pre + ".0.weight") want = (int64_t)kConvs[l].cout * kConvs[l].cin * kConvs[l].k * kConvs[l].k;
      else if (nm == pre + ".1.weight" || nm == pre + ".1.bias") want = kConvs[l].cout;
    }
    if (nm == "Image_net.visual_head.weight") want = (int64_t)V * p->head_k;
    if (nm == "Image_net.visual_head.bias") want = V;
    if (want < 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_image_encoder_bwd: '%s' is not a parameter of Image_net", nm.c_str());
    if (want != grads[i].numel) return fail(ODEVIO_ERR_BAD_ARG, "odevio_image_encoder_bwd: gradient '%s' has the wrong size", nm.c_str());
  }
  // ---- workspace, sized before any launch
  size_t n_act = 0, n_dil = 0, n_part = 1;
  for (int i = 0; i < 9; ++i) {
    const ConvSpec& cs = kConvs[i];
    n_act = std::max(n_act, (size_t)P * p->conv_h[i + 1] * p->conv_w_sp[i + 1] * cs.cout);
    if (i > 0) n_dil = std::max(n_dil, (size_t)P * p->conv_h[i] * p->conv_w_sp[i] * cs.cout);   // (dilated to the input's size; stride 1: the same size)
    const int cin_k = i == 0 ? 8 : cs.cin;
    const int M = P * p->conv_h[i + 1] * p->conv_w_sp[i + 1];
    n_part = std::max(n_part, enc_wgrad_partial_floats(cs.cout, cin_k, cs.k * cs.k, enc_wgrad_pick_splits(M, cs.cout, cin_k, cs.k * cs.k, p->conv_w_sp[i + 1])));
  }
  n_act = std::max(n_act, (size_t)P * p->conv_h[0] * p->conv_w_sp[0] * 8);   // conv1's frame pairs as NHWC with 8 slots
  n_act = std::max(n_act, (size_t)V * p->head_k);                             // the head's weight gradient before its permutation
  int rc;
  if ((rc = ensure(p->enc_gA, n_act)) || (rc = ensure(p->enc_D, n_act)) || (rc = ensure(p->enc_Dd, std::max<size_t>(n_dil, 4))) ||
      (rc = ensure(p->enc_part, n_part + 4096)) || (rc = ensure(p->enc_headT, (size_t)V * p->head_k)) ||
      (rc = ensure(p->bn_partial, (size_t)4 * BN_MAX_BLOCKS * 1024)))
    return rc;
  float *gA = p->enc_gA.p, *D = p->enc_D.p;
  float* sums = p->enc_part.p + n_part;   // [2][<= 1024]
  // ---- visual head: fv = flat(a8) W^T + b with the plan's (H, W, C) column order
  {
    const int oh = p->conv_h[9], ow = p->conv_w_sp[9];
    launch_pair_unpack(p->enc_a[8].p, D, (size_t)P * oh * ow, 1024, st);                       // a8 as fp32 [P][head_k]
    if (float* gw = grad_ptr("Image_net.visual_head.weight", (int64_t)V * p->head_k)) {
      skinny_tn(g_fv, ld_gfv, D, p->head_k, gA, p->head_k, P, V, p->head_k, st);                  // [V][(H,W,C)]
      enc_head_grad_permute(gA, gw, V, 1024, oh * ow, st);                                       // -> the reference's (C,H,W) flatten order
    }
    if (float* gb = grad_ptr("Image_net.visual_head.bias", V)) {
      if (ld_gfv != V) return fail(ODEVIO_ERR_BAD_ARG, "odevio_image_encoder_bwd: the head's bias gradient needs contiguous grad_fv rows");
      colsum_rows(g_fv, gb, P, V, st);
    }
    relayout_transpose(p->head_w, p->enc_headT.p, V, p->head_k, st);                             // [V][head_k] -> [head_k][V]
    skinny_linear(g_fv, ld_gfv, p->enc_headT.p, V, nullptr, gA, p->head_k, P, p->head_k, V, st); // g_a8 [P][head_k] = g_fv W
  }
  for (int i = 8; i >= 0; --i) {
    const ConvSpec& cs = kConvs[i];
    const int Ho = p->conv_h[i + 1], Wo = p->conv_w_sp[i + 1], Hi = p->conv_h[i], Wi = p->conv_w_sp[i];
    const size_t M = (size_t)P * Ho * Wo;
    const int pad = (cs.k - 1) / 2;
    const std::string pre = std::string("Image_net.") + cs.name;
    const DropoutSpec d = make_dropout(p->enc_seed, p->enc_call0 + i, i == 8 ? 0.5f : 0.2f);
    HIPCHK(enc_bn_bwd_reduce(gA, p->enc_z[i].p, M, cs.cout, p->enc_mean[i], p->enc_invstd[i], p->conv_gamma[i], p->conv_beta[i], d,
                             reinterpret_cast<double*>(p->bn_partial.p), sums, st));
    if (float* gb = grad_ptr(pre + ".1.bias", cs.cout)) HIPCHK(hipMemcpyAsync(gb, sums, cs.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (float* gg = grad_ptr(pre + ".1.weight", cs.cout)) HIPCHK(hipMemcpyAsync(gg, sums + cs.cout, cs.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(enc_bn_bwd_apply(gA, p->enc_z[i].p, M, cs.cout, p->enc_mean[i], p->enc_invstd[i], p->conv_gamma[i], p->conv_beta[i], d, sums, D, st));
    // ---- weight gradient
    if (float* gw = grad_ptr(pre + ".0.weight", (int64_t)cs.cout * cs.cin * cs.k * cs.k)) {
      WgradArgs a{};
      a.D = D; a.partial = p->enc_part.p; a.dW = gw;
      a.N = P; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.Cout = cs.cout; a.KH = a.KW = cs.k; a.stride = cs.stride; a.pad = pad; a.M = (int)M;
      if (i == 0) {
        enc_pairs_nhwc8(img, gA, B, S, Hi, Wi, st);   // (gA is free: D holds this block's gradient, and block 0 has no input gradient)
        a.x = gA; a.x_f32 = 1; a.ldx = 8; a.Cin = 8; a.cin_out = 6;
      } else {
        a.x = p->enc_a[i - 1].p; a.x_f32 = 0; a.Cin = cs.cin; a.cin_out = cs.cin;
      }
      a.splits = enc_wgrad_pick_splits(a.M, a.Cout, a.Cin, cs.k * cs.k, Wo);
      HIPCHK(enc_wgrad(a, st));
    }
    if (i == 0) break;
    // ---- input gradient -> gA [P * Hi * Wi][Cin]
    const float* din = D;
    int Hd = Ho, Wd = Wo;
    if (cs.stride > 1) {
      Hd = Hi + 2 * pad - cs.k + 1;
      Wd = Wi + 2 * pad - cs.k + 1;
    }
    static const bool igemm = getenv("ODEVIO_ENC_BWD_IGEMM") != nullptr;   // diagnostic: the fp32-input MFMA form (4 x slower)
    // stride-2 blocks on even-sized inputs: four stride-1 convolutions of the UNdilated gradient, one per pixel parity (3 x 3 with zero
    // slots for the 5 x 5 blocks, (1 + py) x (1 + px) for the 3 x 3 blocks), interleaved afterwards (ODEVIO_DGRAD_DILATED: the old form)
    if (!igemm && p->conv_wTp[i][0] && cs.stride == 2 && Hi == 2 * Ho && Wi == 2 * Wo && getenv("ODEVIO_DGRAD_DILATED") == nullptr) {
      unsigned* amax = reinterpret_cast<unsigned*>(p->enc_dscale + 1024);
      enc_pack_dilate(D, p->enc_Dd.p, P, Ho, Wo, Ho, Wo, cs.cout, 1, amax, p->enc_dscale, cs.cin, p->conv_wT_inv_prescale[i], st);
      const size_t Mc = (size_t)P * Ho * Wo;
      for (int cl = 0; cl < 4; ++cl) {
        ConvSplitArgs a{};
        a.in = p->enc_Dd.p; a.w = p->conv_wTp[i][cl]; a.zeros = p->zero_page; a.out = D + (size_t)cl * Mc * cs.cin; a.status = p->status;   // (D itself is packed: free)
        a.scale = p->enc_dscale; a.shift = p->zero_vec;
        a.N = P; a.Hi = Ho; a.Wi = Wo; a.Cin = cs.cout; a.Ho = Ho; a.Wo = Wo; a.Cout = cs.cin; a.stride = 1;
        if (cs.k == 5) { a.KH = a.KW = 3; a.pad = 1; a.w_bytes = p->conv_wTp_bytes[i]; }                            // 3 x 3, zero slots where a tap falls outside the 5 x 5
        else { a.KH = 1 + (cl >> 1); a.KW = 1 + (cl & 1); a.pad = 0; a.w_bytes = p->conv_wTq_bytes[i][cl]; }       // 1 x 1, 1 x 2, 2 x 1, 2 x 2
        a.M = (int)Mc; a.slope = 1.0f; a.out_split = 0; a.ld_out = cs.cin; a.terms = 3;
        a.in_bytes = extent_of(p, a.in, Mc * cs.cout * sizeof(float));
        set_off32(p, a, a.in);
        a.out_bytes = extent_of(p, a.out, Mc * cs.cin * sizeof(float));
        const int nkt = a.KH * a.KW * cs.cout / 32;
        const ConvPlanF plan = plan_f16x2(32 + i, a.M, a.Cout, nkt, p->n_cu, a.off32 != 0);
        for (int ph = 0; ph < plan.n; ++ph) {
          const ConvPhase& f = plan.ph[ph];
          a.wide = f.wide; a.bm = f.bm; a.m_begin = f.m_begin; a.m_end = f.m_end;
          a.splitk = f.splitk;
          a.ktiles_per_split = (nkt + a.splitk - 1) / a.splitk;
          a.splitk = (nkt + a.ktiles_per_split - 1) / a.ktiles_per_split;
          if (a.splitk > 1) {
            if ((rc = ensure(p->partial, (size_t)a.splitk * a.M * a.Cout))) return rc;
            a.partial = p->partial.p;
            a.partial_bytes = p->partial.n * sizeof(float);
          }
          HIPCHK(launch_conv_f16x2(a, st));
        }
      }
      enc_interleave_parity(D, gA, P, Ho, Wo, cs.cin, st);
      continue;
    }
    if (!igemm) {
      // the forward's fp16x2 kernel: D scaled by a per-tensor power of two into the two-piece layout (zero-dilated in the same pass), the
      // tap-reversed filter as pieces, an identity epilogue that divides both factors back out, fp32 output
      unsigned* amax = reinterpret_cast<unsigned*>(p->enc_dscale + 1024);
      enc_pack_dilate(D, p->enc_Dd.p, P, Ho, Wo, Hd, Wd, cs.cout, cs.stride, amax, p->enc_dscale, cs.cin, p->conv_wT_inv_prescale[i], st);
      ConvSplitArgs a{};
      a.in = p->enc_Dd.p; a.w = p->conv_wTs[i]; a.zeros = p->zero_page; a.out = gA; a.status = p->status;
      a.scale = p->enc_dscale; a.shift = p->zero_vec;
      a.N = P; a.Hi = Hd; a.Wi = Wd; a.Cin = cs.cout; a.Ho = Hi; a.Wo = Wi; a.Cout = cs.cin; a.KH = a.KW = cs.k; a.stride = 1; a.pad = cs.k - 1 - pad;
      a.M = P * Hi * Wi; a.slope = 1.0f; a.out_split = 0; a.ld_out = cs.cin; a.terms = 3;
      a.in_bytes = extent_of(p, a.in, (size_t)P * Hd * Wd * cs.cout * sizeof(float));
      a.w_bytes = p->conv_wTs_bytes[i];
      set_off32(p, a, a.in);
      a.out_bytes = extent_of(p, gA, (size_t)a.M * a.Cout * sizeof(float));
      const int nkt = cs.k * cs.k * cs.cout / 32;
      const ConvPlanF plan = plan_f16x2(16 + i, a.M, a.Cout, nkt, p->n_cu, a.off32 != 0);
      for (int ph = 0; ph < plan.n; ++ph) {
        const ConvPhase& f = plan.ph[ph];
        a.wide = f.wide; a.bm = f.bm; a.m_begin = f.m_begin; a.m_end = f.m_end;
        a.splitk = f.splitk;
        a.ktiles_per_split = (nkt + a.splitk - 1) / a.splitk;
        a.splitk = (nkt + a.ktiles_per_split - 1) / a.ktiles_per_split;
        if (a.splitk > 1) {
          if ((rc = ensure(p->partial, (size_t)a.splitk * a.M * a.Cout))) return rc;
          a.partial = p->partial.p;
          a.partial_bytes = p->partial.n * sizeof(float);
        }
        HIPCHK(launch_conv_f16x2(a, st));
      }
      continue;
    }
    if (cs.stride > 1) {
      enc_dilate(D, p->enc_Dd.p, P, Ho, Wo, Hd, Wd, cs.cout, cs.stride, st);
      din = p->enc_Dd.p;
    }
    ConvArgs c{};
    c.in = din; c.w = p->conv_wT[i]; c.out = gA;
    c.N = P; c.Hi = Hd; c.Wi = Wd; c.Cin = cs.cout; c.Ho = Hi; c.Wo = Wi; c.Cout = cs.cin; c.KH = c.KW = cs.k; c.stride = 1; c.pad = cs.k - 1 - pad;
    c.M = P * Hi * Wi; c.ld_out = cs.cin; c.act = EPI_NONE; c.slope = 0.f;
    const int nk = cs.k * cs.k * cs.cout / 32;
    c.splitk = pick_splitk(c.M, c.Cout, nk);
    c.ktiles_per_split = (nk + c.splitk - 1) / c.splitk;
    c.splitk = (nk + c.ktiles_per_split - 1) / c.ktiles_per_split;
    if (c.splitk > 1) {
      if ((rc = ensure(p->partial, (size_t)c.splitk * c.M * c.Cout))) return rc;
      c.partial = p->partial.p;
    }
    launch_conv_igemm(c, st);
  }
  return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "odevio_image_encoder_bwd: a kernel failed to launch");
}

static int imu_encoder(odevio_plan* p, const float* imu, int B, int T, float* fi, int ld_fi, hipStream_t st,
                       DevBuf* slabs = nullptr) {
  RoctxRange range("odevio: InertialEncoder");
  const int pps = (T - 1) / 10;
  const int P = B * pps;
  int rc;
  if ((rc = ensure(p->imu_act, (size_t)P * 2816))) return rc;
  ImuArgs a{};
  a.imu = imu; a.w1t = p->imu_w[0]; a.w2t = p->imu_w[1]; a.w3t = p->imu_w[2];
  a.s1 = p->imu_s[0]; a.h1 = p->imu_h[0]; a.s2 = p->imu_s[1]; a.h2 = p->imu_h[1]; a.s3 = p->imu_s[2]; a.h3 = p->imu_h[2];
  a.out = p->imu_act.p; a.B = B; a.T = T; a.pairs_per_seq = pps;
  launch_imu_convs(a, st);
  static const bool old_gemm = getenv("ODEVIO_IMU_PROJ_IGEMM") != nullptr;   // diagnostic: the implicit-GEMM kernel (split-K 8, 32 workgroups, 0.66 ms)
  if (!old_gemm) {
    // 160 x 2816 -> 256: 160 workgroups of the skinny fp32-MFMA GEMM.  This branch runs on the side stream UNDER conv1, whose
    // persistent workgroups own every CU: the shorter it is, the less it holds back the CUs it shares
    skinny_linear(p->imu_act.p, 2816, p->proj_w, 2816, p->proj_b, fi, ld_fi, P, p->cfg.i_f_len, 2816, st);
    return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "InertialEncoder projection: launch failed");
  }
  return run_gemm(p, p->imu_act.p, P, 2816, p->proj_w, p->cfg.i_f_len, nullptr, p->proj_b, nullptr, 0, fi, ld_fi,
                  EPI_NONE, 0.f, st, slabs);
}

__global__ void concat_kernel(const float* fv, int nv, const float* fi, int ni, float* out, int P) {
  const int F = nv + ni;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)P * F; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / F), c = (int)(i - (size_t)r * F);
    out[i] = c < nv ? fv[(size_t)r * nv + c] : fi[(size_t)r * ni + (c - nv)];
  }
}

// fcat [P][F] (already concatenated) -> fused [P][F]
static int fuse_from_cat(odevio_plan* p, const float* fcat, int P, float* fused, hipStream_t st) {
  const int F = p->F;
  if (p->cfg.fuse_method == ODEVIO_FUSE_CAT) {
    if (fused != fcat) HIPCHK(hipMemcpyAsync(fused, fcat, (size_t)P * F * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (p->cfg.fuse_method == ODEVIO_FUSE_HARD) {
    // logits [P][2F] = cat W^T + b; feature j is kept when logit(2j) + g0 >= logit(2j+1) + g1 with g ~ Gumbel(0, 1): the one-hot
    // forward value of F.gumbel_softmax(..., tau=1, hard=True)[..., 0] (FusionModule.py:24-29; argmax takes index 0 on a tie)
    int rc = ensure(p->train_aux, (size_t)P * 2 * F);
    if (rc) return rc;
    // (the same skinny fp32-MFMA GEMM as odevio_fuse_hard_bwd's recomputation: forward and backward see the same logit bits, so
    // the straight-through gradient can never land on a mask the forward did not apply)
    skinny_linear(fcat, F, p->fuse_w, F, p->fuse_b, p->train_aux.p, 2 * F, P, 2 * F, F, st);
    launch_hard_mask(fcat, p->train_aux.p, fused, (size_t)P * F, p->seed, p->rng_calls++, st);
    return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "hard fusion: launch failed");
  }
  if (F % 32) {   // the implicit-GEMM kernel walks K in 32-wide tiles; other widths (the reference's 200 + 200) take the skinny GEMM
    skinny_linear(fcat, F, p->fuse_w, F, p->fuse_b, fused, F, P, F, F, st);
    mul_inplace(fused, fcat, (size_t)P * F, st);
    return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "soft fusion: launch failed");
  }
  return run_gemm(p, fcat, P, F, p->fuse_w, F, nullptr, p->fuse_b, fcat, F, fused, F, EPI_NONE, 0.f, st);
}

// ------------------------------------------------------------------------------------------------
static void fill_tableau(int solver, IntegTableau& t) {
  memset(&t, 0, sizeof(t));
  auto set_a = [&](int i, std::initializer_list<double> row) {
    int j = 0;
    for (double v : row) t.a[i][j++] = (float)v;
  };
  auto set_v = [&](float* dst, std::initializer_list<double> row) {
    int j = 0;
    for (double v : row) dst[j++] = (float)v;
  };
  switch (solver) {
    case ODEVIO_DOPRI5:
      t.stages = 7; t.fsal = 1; t.has_err = 1; t.order = 5;
      set_a(1, {1 / 5.});
      set_a(2, {3 / 40., 9 / 40.});
      set_a(3, {44 / 45., -56 / 15., 32 / 9.});
      set_a(4, {19372 / 6561., -25360 / 2187., 64448 / 6561., -212 / 729.});
      set_a(5, {9017 / 3168., -355 / 33., 46732 / 5247., 49 / 176., -5103 / 18656.});
      set_a(6, {35 / 384., 0., 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84.});
      set_v(t.b, {35 / 384., 0., 500 / 1113., 125 / 192., -2187 / 6784., 11 / 84., 0.});
      set_v(t.e, {35 / 384. - 5179 / 57600., 0., 500 / 1113. - 7571 / 16695., 125 / 192. - 393 / 640.,
                  -2187 / 6784. + 92097 / 339200., 11 / 84. - 187 / 2100., -1 / 40.});
      break;
    case ODEVIO_TSIT5:
      t.stages = 7; t.fsal = 1; t.has_err = 1; t.order = 5;
      set_a(1, {0.161});
      set_a(2, {-0.008480655492356989, 0.335480655492357});
      set_a(3, {2.8971530571054935, -6.359448489975075, 4.3622954328695815});
      set_a(4, {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525});
      set_a(5, {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383});
      set_a(6, {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774});
      set_v(t.b, {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774, 0.});
      set_v(t.e, {-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
                  0.5823571654525552, -0.45808210592918697, 1 / 66.});
      break;
    case ODEVIO_HEUN:
      t.stages = 2; t.fsal = 0; t.has_err = 1; t.order = 2;
      set_a(1, {1.0});
      set_v(t.b, {0.5, 0.5});
      set_v(t.e, {-0.5, 0.5});
      break;
    case ODEVIO_EULER:
      t.stages = 1; t.fsal = 0; t.has_err = 0; t.order = 1;
      set_v(t.b, {1.0});
      break;
    case ODEVIO_RK4:  // 3/8 rule (torchdiffeq rk4_alt_step_func)
      t.stages = 4; t.fsal = 0; t.has_err = 0; t.order = 4;
      set_a(1, {1 / 3.});
      set_a(2, {-1 / 3., 1.0});
      set_a(3, {1.0, -1.0, 1.0});
      set_v(t.b, {1 / 8., 3 / 8., 3 / 8., 1 / 8.});
      break;
    default:  // ODEVIO_RK4_CLASSIC
      t.stages = 4; t.fsal = 0; t.has_err = 0; t.order = 4;
      set_a(1, {0.5});
      set_a(2, {0., 0.5});
      set_a(3, {0., 0., 1.0});
      set_v(t.b, {1 / 6., 1 / 3., 1 / 3., 1 / 6.});
      break;
  }
}

static bool is_fixed_step(int solver) { return solver == ODEVIO_RK4 || solver == ODEVIO_RK4_CLASSIC; }

// Fills everything that does not depend on the launch's rows and carves LDS for `rt` rows.
static int integ_common(odevio_plan* p, IntegArgs& a, int rt, int solver, int substeps, size_t* lds_bytes) {
  const odevio_config& c = p->cfg;
  memset(&a, 0, sizeof(a));
  a.F = p->Fi; a.Fio = p->F; a.H = pad_members(c.ode_hidden_dim); a.nlin = p->nlin; a.act = c.ode_activation;
  for (int l = 0; l <= p->nlin; ++l) a.dims[l] = p->dims[l];
  for (int l = 0; l < p->nlin; ++l) { a.w[l] = p->ode_w[l]; a.b[l] = p->ode_b[l]; a.w_lds_off[l] = -1; }
  a.rnn_type = c.rnn_type; a.L = c.rnn_num_layers; a.rnn_vcols = p->rnn_vcols;
  for (int l = 0; l < a.L; ++l) { a.rw[l] = p->rnn_w[l]; a.rb[l] = p->rnn_b[l]; }
  fill_tableau(solver, a.tab);
  a.nsub = is_fixed_step(solver) ? substeps : 0;
  a.atol = c.atol; a.rtol = c.rtol; a.dt0 = c.dt0; a.max_steps = c.max_steps;
  a.xbuf = p->xbuf; a.xstride = p->xstride; a.status = p->status;
  a.dbg = (unsigned long long*)(p->status + 8);  // 12 x u64 behind the status words (diagnostic build only)
  {
    // ODEVIO_SAFE_HANDOFF=1 keeps every group on the placement-independent write-through protocol (tests run both)
    const char* e = getenv("ODEVIO_SAFE_HANDOFF");
    a.allow_local = (e && e[0] == '1') ? 0 : 1;
  }
  // LDS carve (floats); every K is padded to a multiple of 256
  int maxdim = pad256(a.F);
  for (int l = 0; l <= p->nlin; ++l) maxdim = std::max(maxdim, pad256(p->dims[l]));
  int off = 0;
  a.lds_xin = off; off += rt * maxdim;
  a.lds_hst = off; off += rt * pad256(a.F);
  a.lds_misc = off; off += 128 * rt + rt * 32 * 2 + 256 + INTEG_MAX_LIN * 32;  // lay, nrm, mv, qb, bia
  off = (off + 3) & ~3;
  a.lds_w = off;
  int budget = (160 * 1024 - 1024) / 4 - off;  // 1 KB left for the kernel's static LDS (__syncthreads_or scratch)
  // keep the largest slices that fit resident; the rest stream from L2
  std::vector<int> order(p->nlin);
  for (int l = 0; l < p->nlin; ++l) order[l] = l;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
    return (size_t)pad256(p->dims[x]) * p->dims[x + 1] > (size_t)pad256(p->dims[y]) * p->dims[y + 1];
  });
  int woff = 0;
  for (int l : order) {
    const int n = (p->dims[l + 1] / INTEG_MEMBERS) * pad256(p->dims[l]);
    if (n <= budget) { a.w_lds_off[l] = woff; woff += n; budget -= n; }
  }
  *lds_bytes = (size_t)(off + woff) * sizeof(float);
  return 0;
}

static int launch_integ(odevio_plan* p, IntegArgs& a, int rt, size_t lds, hipStream_t st) {
  if (p->n_cu < INTEG_GROUPS * INTEG_MEMBERS)
    return fail(ODEVIO_ERR_UNSUPPORTED, "persistent integrator needs %d CUs, device has %d", INTEG_GROUPS * INTEG_MEMBERS, p->n_cu);
  HIPCHK(hipMemsetAsync(p->xbuf, 0, (size_t)INTEG_GROUPS * 2 * p->xstride * sizeof(unsigned long long), st));
  const int e = launch_integrator(a, rt, lds, st);
  if (e != 0) return fail(ODEVIO_ERR_HIP, "integrator launch failed: %s (lds %zu B)", hipGetErrorString((hipError_t)e), lds);
  return 0;
}

// The backward's log of a forward (integrator.h): where the persistent kernel writes it.  All null = no log.
struct StepLog {
  float* dtlog = nullptr;
  int* dtcnt = nullptr;
  float* ylog = nullptr;
  float* yend = nullptr;
  int cap = 0;
};

static int run_sequence(odevio_plan* p, const float* fused, const float* ts, const float* hc, int B, int P,
                        float* out_seq, float* hT, int32_t* stats, hipStream_t st, const StepLog& log = StepLog()) {
  RoctxRange range("odevio: ODE + RNN (persistent integrator)");   // the reference's "ODE" and "RNN" ranges are one launch here
  const int L = p->cfg.rnn_num_layers;
  const int bpg_max = 8 / L;  // rows per group <= 8
  const int chunk = INTEG_GROUPS * bpg_max;
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = std::min(chunk, B - b0);
    const int BPG = (nb + INTEG_GROUPS - 1) / INTEG_GROUPS;
    const int R = L * BPG;
    const int rt = R <= 2 ? 2 : (R <= 4 ? 4 : 8);
    IntegArgs a;
    size_t lds;
    int rc = integ_common(p, a, rt, p->cfg.ode_solver, p->cfg.ode_substeps, &lds);
    if (rc) return rc;
    a.mode = p->cfg.model_type == ODEVIO_MODEL_RNN ? MODE_RNN_ONLY : MODE_ODE_RNN;
    a.B = B; a.P = P; a.b_begin = b0; a.b_end = b0 + nb;
    a.BPG = BPG; a.G = (nb + BPG - 1) / BPG; a.rows_per_group = R;
    a.fused = fused; a.ts = ts; a.ts_relative = hc ? 0 : 1; a.hc = hc; a.out_seq = out_seq; a.hT = hT; a.stats = stats;
    a.dtlog = log.dtlog; a.dtcnt = log.dtcnt; a.dtlog_cap = log.cap; a.ylog = log.ylog; a.yend = log.yend;
    if ((rc = launch_integ(p, a, rt, lds, st))) return rc;
  }
  return 0;
}

static int run_rows(odevio_plan* p, int mode, const float* y, const float* t0, const float* t1, int rows, int solver,
                    int substeps, float* y_out, int32_t* stats, hipStream_t st) {
  if (p->cfg.model_type != ODEVIO_MODEL_ODE_RNN) return fail(ODEVIO_ERR_UNSUPPORTED, "plan has no ODEFunc");
  const int chunk = INTEG_GROUPS * 8;
  for (int r0 = 0; r0 < rows; r0 += chunk) {
    const int nr = std::min(chunk, rows - r0);
    const int BPG = (nr + INTEG_GROUPS - 1) / INTEG_GROUPS;
    const int rt = BPG <= 2 ? 2 : (BPG <= 4 ? 4 : 8);
    IntegArgs a;
    size_t lds;
    int rc = integ_common(p, a, rt, solver, substeps, &lds);
    if (rc) return rc;
    a.mode = mode;
    a.B = rows; a.P = 1; a.b_begin = r0; a.b_end = r0 + nr;
    a.BPG = BPG; a.G = (nr + BPG - 1) / BPG; a.rows_per_group = BPG;
    a.y0 = y; a.t0 = t0; a.t1 = t1; a.y_out = y_out; a.stats = stats;
    if ((rc = launch_integ(p, a, rt, lds, st))) return rc;
  }
  return 0;
}

static int regress(odevio_plan* p, const float* seq, int M, float* poses, hipStream_t st) {
  int rc;
  if ((rc = ensure(p->reg_hid, (size_t)M * 128))) return rc;
  static const bool old_gemm = getenv("ODEVIO_REGRESSOR_IGEMM") != nullptr;   // diagnostic: the implicit-GEMM kernel with split-K (0.057 ms)
  if (!old_gemm || p->F % 32) {
    // Linear(F, 128) + LeakyReLU(0.1) + Linear(128, 6) (PoseODERNN.py:64-67) on the skinny fp32-MFMA GEMM: M is B * (S - 1) rows,
    // three short launches behind the integrator on the critical path
    skinny_linear(seq, p->F, p->reg_w0, p->F, p->reg_b0, p->reg_hid.p, 128, M, 128, p->F, st);
    leaky_inplace(p->reg_hid.p, (size_t)M * 128, 0.1f, st);
    skinny_linear(p->reg_hid.p, 128, p->reg_w2, 128, p->reg_b2, poses, 6, M, 6, 128, st);
    return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "regressor: launch failed");
  }
  if ((rc = run_gemm(p, seq, M, p->F, p->reg_w0, 128, nullptr, p->reg_b0, nullptr, 0, p->reg_hid.p, 128, EPI_LEAKY, 0.1f, st)))
    return rc;
  return run_gemm(p, p->reg_hid.p, M, 128, p->reg_w2, 6, nullptr, p->reg_b2, nullptr, 0, poses, 6, EPI_NONE, 0.f, st);
}

// ------------------------------------------------------------------------------------------------
#define ARGCHK(cond, msg) \
  if (!(cond)) return fail(ODEVIO_ERR_BAD_ARG, msg)

extern "C" int odevio_reserve(odevio_plan* p, int32_t B, int32_t S, void* stream) {
  ARGCHK(p && B > 0 && S > 1, "odevio_reserve: bad argument");
  const int P = B * (S - 1);
  int rc;
  if ((rc = ensure_act(p, P)) || (rc = ensure(p->imu_act, (size_t)P * 2816)) ||
      (rc = ensure(p->fcat, (size_t)P * p->F)) || (rc = ensure(p->fused, (size_t)P * p->F)) ||
      (rc = ensure(p->out_seq, (size_t)P * p->F)) || (rc = ensure(p->reg_hid, (size_t)P * 128)) ||
      (rc = ensure(p->partial, (size_t)64 * P * std::max(p->cfg.v_f_len, 128))))
    return rc;
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" int odevio_profile_enable(odevio_plan* p, int32_t on) {
  ARGCHK(p && on >= 0 && on <= 4096, "odevio_profile_enable: bad argument");
  if (on > p->ev_depth) {
    const size_t have = p->ev.size(), want = (size_t)on * (ODEVIO_N_STAGES + 1);
    p->ev.resize(want, nullptr);
    for (size_t i = have; i < want; ++i) HIPCHK(hipEventCreate(&p->ev[i]));
    p->ev_depth = on;
  }
  p->prof = on != 0;
  p->ev_w = 0;
  p->ev_n = 0;
  return 0;
}

extern "C" int odevio_profile_read(odevio_plan* p, float* ms_out) {
  ARGCHK(p && ms_out && p->prof, "odevio_profile_read: profiling is not enabled");
  if (p->ev_n == 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_profile_read: no complete forward recorded since the last read");
  for (int i = 0; i < ODEVIO_N_STAGES; ++i) ms_out[i] = 0.f;
  for (int k = 1; k <= p->ev_n; ++k) {   // the ev_n most recent sets, newest first
    const hipEvent_t* e = &p->ev[(size_t)((p->ev_w - k + 2 * p->ev_depth) % p->ev_depth) * (ODEVIO_N_STAGES + 1)];
    HIPCHK(hipEventSynchronize(e[ODEVIO_N_STAGES]));
    for (int i = 0; i < ODEVIO_N_STAGES; ++i) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e[i], e[i + 1]));
      ms_out[i] += ms;
    }
  }
  for (int i = 0; i < ODEVIO_N_STAGES; ++i) ms_out[i] /= (float)p->ev_n;
  p->ev_n = 0;
  return 0;
}

extern "C" int odevio_path_accu(const void* poses6, int32_t is_f64, const int64_t* offsets, int32_t n_drives, const double* carry,
                                double* mats, void* stream) {
  ARGCHK(poses6 && offsets && mats && n_drives > 0, "odevio_path_accu: bad argument");
  HIPCHK(launch_path_accu(poses6, is_f64, offsets, n_drives, carry, mats, (hipStream_t)stream));
  return ODEVIO_OK;
}

extern "C" int odevio_resize_u8(const uint8_t* src, int32_t n, int32_t Hin, int32_t Win, uint8_t* dst, int32_t Hout, int32_t Wout,
                                uint8_t* tmp, void* stream) {
  ARGCHK(src && dst && n > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "odevio_resize_u8: bad argument");
  if (Hin != Hout && Win != Wout && !tmp) return fail(ODEVIO_ERR_BAD_ARG, "odevio_resize_u8: a two-pass resize needs n*Hin*Wout*3 bytes of scratch");
  if (resize_u8_launch(src, n, Hin, Win, dst, Hout, Wout, tmp, (hipStream_t)stream)) return fail(ODEVIO_ERR_HIP, "odevio_resize_u8: launch failed");
  return 0;
}

extern "C" int odevio_resize_table(int32_t in_size, int32_t out_size, int32_t* ksize, int32_t* bounds, int32_t* kk, int32_t kk_capacity) {
  const int rc = resize_table_host(in_size, out_size, ksize, bounds, kk, kk_capacity);
  return rc ? fail(rc, "odevio_resize_table: bad argument (sizes > 0, kk_capacity >= out_size * ksize)") : 0;
}

extern "C" int odevio_debug_stamps(odevio_plan* p, uint64_t* out8, void* stream) {
  ARGCHK(p && out8, "odevio_debug_stamps: bad argument");
  HIPCHK(hipMemcpyAsync(out8, p->status + 8, 12 * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));   // (the 128-byte status block: 8 ints + 12 stamps)
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// Turns the device status words into an error (and clears what it reports on the device, stream-ordered).
static int report_status(odevio_plan* p, const int* hw, hipStream_t st) {
  if (hw[ODEVIO_STATUS_AUDIT] != 0) {
    if (!getenv("ODEVIO_AUDIT_SELFTEST")) ++g_audit_violations;
    const int id = hw[ODEVIO_STATUS_AUDIT + 1];
    HIPCHK(hipMemsetAsync(p->status + ODEVIO_STATUS_AUDIT, 0, 2 * sizeof(int), st));
    return fail(ODEVIO_ERR_BOUNDS, "audit build: kernel id %d computed an address outside its buffers (access redirected)", id);
  }
  if (hw[ODEVIO_STATUS_RANGE] != 0) {
    HIPCHK(hipMemsetAsync(p->status + ODEVIO_STATUS_RANGE, 0, sizeof(int), st));
    return fail(ODEVIO_ERR_RANGE, "image encoder: an activation left the fp16x2 range (|x| > 65504 or not finite); "
                                  "set ODEVIO_CONV_MATH=f32 for the fp32-input MFMA path");
  }
  if (hw[ODEVIO_STATUS_RANGE + 1] != 0) {
    HIPCHK(hipMemsetAsync(p->status + ODEVIO_STATUS_RANGE + 1, 0, sizeof(int), st));
    return fail(ODEVIO_ERR_TIMEOUT, "conv1: a bounded in-kernel group barrier gave up");
  }
  const int h = hw[0];
  if (h != 0) {
    HIPCHK(hipMemsetAsync(p->status, 0, sizeof(int), st));
    return fail(h, h == ODEVIO_ERR_TIMEOUT ? "integrator: a bounded in-kernel wait timed out (are all 256 workgroups resident?)"
                                           : "solver exceeded max_steps");
  }
  return 0;
}

// Behind a forward: status words -> pinned host memory, asynchronously.  At an API entry: if that copy has completed and
// shows a failure, report it now (the outputs of that forward are garbage and every later launch would bail out early).
static void post_status(odevio_plan* p, hipStream_t st) {
  if (!p->status_host) return;
  if (hipMemcpyAsync(p->status_host, p->status, 8 * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess &&
      hipEventRecord(p->ev_status, st) == hipSuccess)
    p->status_pending = true;
}
static int poll_status(odevio_plan* p, hipStream_t st) {
  if (!p->status_pending || hipEventQuery(p->ev_status) != hipSuccess) return 0;
  p->status_pending = false;
  int hw[8];
  memcpy(hw, p->status_host, sizeof(hw));
  return report_status(p, hw, st);
}
#define POLL(p, st)                                  \
  do {                                               \
    const int rc_ = poll_status((p), (hipStream_t)(st)); \
    if (rc_) return rc_;                             \
  } while (0)

extern "C" int odevio_check(odevio_plan* p, void* stream) {
  ARGCHK(p, "odevio_check: null plan");
  hipStream_t st = (hipStream_t)stream;
  int hw[8] = {};
  HIPCHK(hipMemcpyAsync(hw, p->status, sizeof(hw), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  p->status_pending = false;
  const int rc = report_status(p, hw, st);
  if (rc) HIPCHK(hipStreamSynchronize(st));
  return rc;
}

extern "C" int odevio_image_encoder_fwd(odevio_plan* p, const float* img, int32_t B, int32_t S, float* fv,
                                        int32_t ld_fv, void* stream) {
  ARGCHK(p && img && fv && B > 0 && S > 1 && ld_fv >= p->cfg.v_f_len, "odevio_image_encoder_fwd: bad argument");
  POLL(p, stream);
  const int rc = image_encoder(p, img, B, S, fv, ld_fv, (hipStream_t)stream);
  post_status(p, (hipStream_t)stream);
  return rc;
}

extern "C" int odevio_conv_block_fwd(odevio_plan* p, int32_t layer, const float* in, int32_t B, int32_t S, float* out,
                                     void* stream) {
  ARGCHK(p && in && out && layer >= 0 && layer < 9 && B > 0 && S > 1, "odevio_conv_block_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (layer == 0 || p->conv_math == 0) return conv_block(p, layer, in, B, S, out, false, false, st);
  // fp32 NHWC at this boundary: split the input into the kernel's two-piece layout first (the encoder itself never
  // converts - each block's epilogue writes the next block's layout)
  const size_t pixels = (size_t)B * (S - 1) * p->conv_h[layer] * p->conv_w_sp[layer];
  const int C = kConvs[layer].cin;
  int rc = ensure(p->pack_tmp, pixels * C);
  if (rc) return rc;
  launch_pair_pack(in, p->pack_tmp.p, pixels, C, p->status, st);
  return conv_block(p, layer, p->pack_tmp.p, B, S, out, true, false, st);
}

extern "C" int odevio_imu_encoder_fwd(odevio_plan* p, const float* imu, int32_t B, int32_t T, float* fi,
                                      int32_t ld_fi, void* stream) {
  ARGCHK(p && imu && fi && B > 0 && T >= 11 && ld_fi >= p->cfg.i_f_len, "odevio_imu_encoder_fwd: bad argument");
  return imu_encoder(p, imu, B, T, fi, ld_fi, (hipStream_t)stream);
}

extern "C" int odevio_fuse_fwd(odevio_plan* p, const float* fv, const float* fi, int32_t P, float* fused, void* stream) {
  ARGCHK(p && fv && fi && fused && P > 0, "odevio_fuse_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  float* cat = fused;
  if (p->cfg.fuse_method != ODEVIO_FUSE_CAT) {
    if ((rc = ensure(p->fcat, (size_t)P * p->F))) return rc;
    cat = p->fcat.p;
  }
  hipLaunchKernelGGL(concat_kernel, dim3(std::min(1024, (P * p->F + 255) / 256)), dim3(256), 0, st, fv, p->cfg.v_f_len, fi,
                     p->cfg.i_f_len, cat, P);
  return fuse_from_cat(p, cat, P, fused, st);
}

extern "C" int odevio_ode_func(odevio_plan* p, const float* y, int32_t rows, float* out, void* stream) {
  ARGCHK(p && y && out && rows > 0, "odevio_ode_func: bad argument");
  return run_rows(p, MODE_FEVAL, y, nullptr, nullptr, rows, p->cfg.ode_solver, 1, out, nullptr, (hipStream_t)stream);
}

extern "C" int odevio_ode_steps(odevio_plan* p, const float* y, const float* t0, const float* t1, int32_t rows,
                                int32_t solver, int32_t substeps, float* y_out, int32_t* stats, void* stream) {
  ARGCHK(p && y && t0 && t1 && y_out && rows > 0, "odevio_ode_steps: bad argument");
  if (solver < 0) solver = p->cfg.ode_solver;
  if (solver > ODEVIO_RK4_CLASSIC) return fail(ODEVIO_ERR_BAD_ARG, "Solver not supported");
  if (substeps <= 0) substeps = p->cfg.ode_substeps;
  POLL(p, stream);
  const int rc = run_rows(p, MODE_ODE_STEPS, y, t0, t1, rows, solver, substeps, y_out, stats, (hipStream_t)stream);
  post_status(p, (hipStream_t)stream);
  return rc;
}

static int ode_rnn_fwd(odevio_plan* p, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                       float* poses, float* h_T, int32_t* stats, hipStream_t st, const StepLog& log = StepLog()) {
  int rc;
  if ((rc = ensure(p->out_seq, (size_t)B * P * p->F))) return rc;
  stage_mark(p, 4, st);
  if ((rc = run_sequence(p, fused, ts, hc_in, B, P, p->out_seq.p, h_T, stats, st, log))) return rc;
  stage_mark(p, 5, st);
  rc = regress(p, p->out_seq.p, B * P, poses, st);
  stage_mark(p, 6, st);
  return rc;
}

extern "C" int odevio_ode_rnn_fwd(odevio_plan* p, const float* fused, const float* ts, const float* hc_in, int32_t B,
                                  int32_t P, float* poses, float* h_T, int32_t* stats, void* stream) {
  ARGCHK(p && fused && ts && poses && h_T && B > 0 && P > 0, "odevio_ode_rnn_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  const int rc = ode_rnn_fwd(p, fused, ts, hc_in, B, P, poses, h_T, stats, st);
  post_status(p, st);
  return rc;
}

extern "C" int odevio_cde_fwd(odevio_plan* p, const float* obs, int32_t B, int32_t L, const double* t_out_host,
                              int32_t n_out, const float* z0_in, float* poses, float* z0_out, int32_t* stats_host,
                              void* stream) {
  ARGCHK(p && obs && t_out_host && poses && z0_out && B > 0 && L > 1 && n_out > 0, "odevio_cde_fwd: bad argument");
  if (p->cfg.model_type != ODEVIO_MODEL_CDE) return fail(ODEVIO_ERR_UNSUPPORTED, "plan is not a Neural-CDE plan");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  const int H = p->cde.H, C = p->cde.C, n = B * H;
  int rc;
  // workspace: [ctl 512 B][t_out doubles][ha hb ytmp y y1][k 7n][interp 5n][z0][sol B*n_out*H]
  const size_t head = (512 + (size_t)n_out * sizeof(double) + 15) / 16 * 4;   // floats, 16-byte aligned
  const size_t need = head + 18 * (size_t)n + (size_t)B * n_out * H;
  static_assert(sizeof(CdeCtl) <= 512, "CdeCtl outgrew its slot");
  if ((rc = ensure(p->cde_ws, need))) return rc;
  if (!p->cde_ctl_host) HIPCHK(hipHostMalloc((void**)&p->cde_ctl_host, sizeof(CdeCtl), hipHostMallocDefault));
  float* q = p->cde_ws.p;
  CdeWork w;
  w.ctl = reinterpret_cast<CdeCtl*>(q);
  w.t_out = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(q) + 512);
  w.ctl_host = p->cde_ctl_host;
  q += head;
  w.ha = q; q += n; w.hb = q; q += n; w.ytmp = q; q += n; w.y = q; q += n; w.y1 = q; q += n;
  w.k = q; q += 7 * (size_t)n;
  w.interp = q; q += 5 * (size_t)n;
  float* z0 = q; q += n;
  float* sol = q;  // [B][n_out][H]
  // z0 = tanh(initial(X(0))) with X(0) = the first observation (PoseCDE.py:96), unless the caller carries one in
  if (z0_in) HIPCHK(hipMemcpyAsync(z0, z0_in, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
  else cde_launch_linear(obs, L * C, p->cde_init_w, p->cde_init_b, z0, B, C, H, 0 /*tanh*/, st);
  HIPCHK(hipMemcpyAsync(z0_out, z0, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, st));
  int stats[2] = {0, 0};
  p->cde.n_cu = p->n_cu;
  rc = cde_solve(p->cde, w, obs, B, L, t_out_host, n_out, z0, sol, stats, p->cde_hint_steps, st);
  if (rc == ODEVIO_ERR_MAX_STEPS) return fail(rc, "cdeint: step budget exhausted");
  if (rc == ODEVIO_ERR_BAD_ARG) return fail(rc, "cdeint: output times must be strictly ascending");
  if (rc) return fail(rc, "cdeint failed: %s", hipGetErrorString(hipGetLastError()));
  p->cde_hint_steps = stats[0];
  if (stats_host) { stats_host[0] = stats[0]; stats_host[1] = stats[1]; }
  rc = regress(p, sol, B * n_out, poses, st);
  post_status(p, st);
  return rc;
}

// Backward of odevio_cde_fwd (cde_bwd.hip): the solve is run once more with a tape, then swept in reverse.
extern "C" int odevio_cde_bwd(odevio_plan* p, const float* obs, int32_t B, int32_t L, const double* t_out_host, int32_t n_out, const float* z0_in,
                              const float* grad_poses, const float* grad_z0_out, float* grad_obs, float* grad_z0_in, const odevio_tensor* grads,
                              int32_t n_grads, int32_t* stats_host, void* stream) {
  ARGCHK(p && obs && t_out_host && grad_poses && grad_obs && B > 0 && L > 1 && n_out > 0 && n_grads >= 0 && (grads || n_grads == 0),
         "odevio_cde_bwd: bad argument");
  if (p->cfg.model_type != ODEVIO_MODEL_CDE) return fail(ODEVIO_ERR_UNSUPPORTED, "plan is not a Neural-CDE plan");
  if (p->cde.w_last16) return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_cde_bwd: the reduced-precision (bf16 last layer) plan has no backward; use --dtype fp32");
  if (grad_z0_in && !z0_in) return fail(ODEVIO_ERR_BAD_ARG, "odevio_cde_bwd: grad_z0_in without z0_in");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  const int H = p->cde.H, C = p->cde.C, n = B * H, nh = p->cde.n_hidden;
  CdeBwdGrads g;
  memset(&g, 0, sizeof(g));
  for (int i = 0; i < n_grads; ++i) {
    if (!grads[i].name || !grads[i].data) return fail(ODEVIO_ERR_BAD_ARG, "odevio_cde_bwd: gradient %d has no name / pointer", i);
    const std::string nm = grads[i].name;
    float* dst = (float*)grads[i].data;
    int64_t want = -1;
    for (int l = 0; l <= nh && want < 0; ++l) {
      const std::string pre = "Pose_net.cde_func.net." + std::to_string(2 * l);
      const int64_t N = l < nh ? H : (int64_t)H * C;
      if (nm == pre + ".weight") { g.w[l] = dst; want = N * H; }
      else if (nm == pre + ".bias") { g.b[l] = dst; want = N; }
    }
    if (want < 0) {
      if (nm == "Pose_net.initial.0.weight") { g.init_w = dst; want = (int64_t)H * C; }
      else if (nm == "Pose_net.initial.0.bias") { g.init_b = dst; want = H; }
      else if (nm == "Pose_net.regressor.0.weight") { g.reg_w0 = dst; want = (int64_t)128 * H; }
      else if (nm == "Pose_net.regressor.0.bias") { g.reg_b0 = dst; want = 128; }
      else if (nm == "Pose_net.regressor.2.weight") { g.reg_w2 = dst; want = 6 * 128; }
      else if (nm == "Pose_net.regressor.2.bias") { g.reg_b2 = dst; want = 6; }
    }
    if (want < 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_cde_bwd: '%s' is not a parameter of the Neural-CDE pose net", nm.c_str());
    if (want != grads[i].numel) return fail(ODEVIO_ERR_BAD_ARG, "odevio_cde_bwd: gradient '%s' has the wrong size", nm.c_str());
    // CDEFunc's gradients accumulate over the vector-field evaluations: start from zero.  z0_in given: the initial layer is not on the path
    HIPCHK(hipMemsetAsync(dst, 0, (size_t)want * sizeof(float), st));
  }
  const int cap = std::max(16, std::min(p->cfg.max_steps, 1024));
  int rc;
  const size_t head = (512 + (size_t)n_out * sizeof(double) + 15) / 16 * 4;
  const size_t need = head + 18 * (size_t)n + (size_t)B * n_out * H;
  if ((rc = ensure(p->cde_ws, need)) || (rc = ensure(p->train_ws, cde_bwd_workspace_floats(p->cde, B, n_out, cap)))) return rc;
  if (!p->cde_ctl_host) HIPCHK(hipHostMalloc((void**)&p->cde_ctl_host, sizeof(CdeCtl), hipHostMallocDefault));
  float* q = p->cde_ws.p;
  CdeWork w;
  w.ctl = reinterpret_cast<CdeCtl*>(q);
  w.t_out = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(q) + 512);
  w.ctl_host = p->cde_ctl_host;
  q += head;
  w.ha = q; q += n; w.hb = q; q += n; w.ytmp = q; q += n; w.y = q; q += n; w.y1 = q; q += n;
  w.k = q; q += 7 * (size_t)n;
  w.interp = q; q += 5 * (size_t)n;
  p->cde.n_cu = p->n_cu;
  int stats[2] = {0, 0};
  rc = cde_backward(p->cde, w, p->train_ws.p, obs, B, L, t_out_host, n_out, z0_in, p->cde_init_w, p->cde_init_b, p->reg_w0, p->train.reg_w0_t,
                    p->reg_b0, p->reg_w2, grad_poses, grad_z0_out, grad_obs, grad_z0_in, g, cap, stats, st);
  if (stats_host) { stats_host[0] = stats[0]; stats_host[1] = stats[1]; }
  if (rc == ODEVIO_ERR_MAX_STEPS) return fail(rc, "odevio_cde_bwd: more than %d accepted steps (the tape's capacity) or the step budget exhausted", cap);
  if (rc == ODEVIO_ERR_BAD_ARG) return fail(rc, "cdeint: output times must be strictly ascending");
  if (rc) return fail(rc, "odevio_cde_bwd failed: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// backward (train.hip)
static int fill_train_model(odevio_plan* p, TrainModel& m) {
  const odevio_config& c = p->cfg;
  if (c.model_type == ODEVIO_MODEL_CDE) return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_ode_rnn_bwd: Neural-CDE plans take odevio_cde_bwd");
  // (the tape runs on plain GEMMs over the REAL widths - only the persistent forward kernel pads them - and needs whole float4 rows)
  if (p->F % 4 || (c.model_type == ODEVIO_MODEL_ODE_RNN && c.ode_hidden_dim % 4))
    return fail(ODEVIO_ERR_UNSUPPORTED, "backward: v_f_len+i_f_len (%d) and ode_hidden_dim (%d) must be multiples of 4", p->F, c.ode_hidden_dim);
  m = p->train;
  m.gru = c.rnn_type == ODEVIO_RNN_GRU;
  m.F = p->F; m.H = c.ode_hidden_dim; m.L = c.rnn_num_layers; m.act = c.ode_activation;
  m.with_ode = c.model_type == ODEVIO_MODEL_ODE_RNN;
  m.nlin = m.with_ode ? p->nlin : 0;
  for (int l = 0; l <= p->nlin; ++l) m.dims[l] = p->dims_real[l];
  for (int l = 0; l < p->nlin; ++l) m.ode_b[l] = p->ode_b[l];
  m.reg_w0 = p->reg_w0; m.reg_b0 = p->reg_b0; m.reg_w2 = p->reg_w2; m.reg_b2 = p->reg_b2;
  m.stages = 1; m.jmax = 1; m.adaptive = 0; m.dtlog = nullptr; m.dtcnt = nullptr; m.dtlog_cap = 0; m.ylog = nullptr; m.yend = nullptr; m.adj = nullptr; m.steps_per_interval = nullptr;
  if (m.with_ode) {
    IntegTableau t;
    fill_tableau(c.ode_solver, t);
    // (euler under torchode's controller takes dt0-sized steps to the end of every interval - a thousand per 0.1 s at the reference's
    // dt0 = 1e-4: replayed like any other logged step sequence, at the cost of that many launches)
    if (!is_fixed_step(c.ode_solver)) m.adaptive = 1;
    // FSAL pairs: the last stage only feeds the error estimate (b_last = 0); the replay does not need it
    m.stages = t.fsal ? t.stages - 1 : t.stages;
    m.jmax = m.adaptive ? 0 : c.ode_substeps;   // adaptive: set by the caller from the forward's step log
    for (int i = 0; i < 8; ++i) {
      m.b[i] = i < m.stages ? t.b[i] : 0.f;
      for (int j = 0; j < 8; ++j) m.a[i][j] = (i < 7 && j < 7) ? t.a[i][j] : 0.f;
    }
    // the reverse sweep of an interval as ONE launch of the integrator's adjoint twin (weights resident, granule exchange) where the
    // device holds the 256 co-resident workgroups it needs; ODEVIO_ADJOINT_LAUNCHES=1 keeps one launch per product (tests compare)
    m.adj = nullptr;
    if (p->n_cu >= INTEG_GROUPS * INTEG_MEMBERS && getenv("ODEVIO_ADJOINT_LAUNCHES") == nullptr) {
      IntegAdjArgs& a = p->adj_base;
      memset(&a, 0, sizeof(a));
      a.F = p->Fi; a.Fio = p->F; a.nlin = p->nlin; a.act = c.ode_activation;
      for (int l = 0; l <= p->nlin; ++l) { a.dims[l] = p->dims[l]; a.dims_io[l] = p->dims_real[l]; }
      for (int l = 0; l < p->nlin; ++l) a.wT[l] = p->ode_wT[l];
      a.S = m.stages;
      for (int i = 0; i < 7; ++i) {
        a.tb[i] = m.b[i];
        for (int j = 0; j < 7; ++j) a.ta[i][j] = (i < m.stages && j < i) ? t.a[i][j] : 0.f;
      }
      a.xbuf = p->xbuf; a.xstride = p->xstride; a.status = p->status;
      const char* e = getenv("ODEVIO_SAFE_HANDOFF");
      a.allow_local = (e && e[0] == '1') ? 0 : 1;
      m.adj = &a;
    }
  }
  return 0;
}

// accepted steps per row and interval the log holds: the reference's tolerances take 4 - 6; a log that overflows is retried with
// 8 x the room (up to TRAIN_DTLOG_CAP_MAX); euler (no error estimate: every dt0 step is accepted) is sized from the longest interval the caller could mean
#define TRAIN_DTLOG_CAP 64
#define TRAIN_DTLOG_CAP_MAX 16384
#define TRAIN_YLOG_MAX_FLOATS ((size_t)256 << 20)   // 1 GiB: beyond it the states are not logged and the tape walks the steps in order

// The log of one forward as one run of floats: [dtlog rows*P*cap][dtcnt rows*P (ints)][yend rows*P*F][ylog rows*P*cap*F]; the last two
// are absent (with_y = false) when they would not fit TRAIN_YLOG_MAX_FLOATS or the plan has no ODE.
struct TapeLayout {
  int cap = 0, R = 0;
  bool with_y = false;
  size_t n_dt = 0, n_cnt = 0, n_yend = 0, n_ylog = 0;
  size_t total() const { return n_dt + n_cnt + n_yend + n_ylog; }
  StepLog carve(float* base) const {
    StepLog l;
    l.cap = cap;
    l.dtlog = base;
    l.dtcnt = reinterpret_cast<int*>(base + n_dt);
    if (with_y) { l.yend = base + n_dt + n_cnt; l.ylog = l.yend + n_yend; }
    return l;
  }
};
static TapeLayout tape_layout(const odevio_plan* p, int B, int P, int cap) {
  TapeLayout t;
  t.cap = cap;
  t.R = p->cfg.rnn_num_layers * B;
  const size_t rp = (size_t)t.R * P;
  t.n_dt = (rp * cap + 3) / 4 * 4;
  t.n_cnt = (rp + 3) / 4 * 4;
  const bool in_order = getenv("ODEVIO_TAPE_IN_ORDER") != nullptr;   // diagnostic / tests: the step-by-step tape although the states would fit (read per call)
  t.with_y = p->cfg.model_type == ODEVIO_MODEL_ODE_RNN && rp * cap * p->F <= TRAIN_YLOG_MAX_FLOATS && !in_order;
  if (t.with_y) { t.n_yend = rp * p->F; t.n_ylog = rp * cap * p->F; }
  return t;
}
// the log's room per interval for this plan's solver: fixed-step solvers take exactly ode_substeps steps
static int tape_default_cap(const odevio_plan* p) {
  if (is_fixed_step(p->cfg.ode_solver)) return std::max(1, p->cfg.ode_substeps);
  IntegTableau tb;
  fill_tableau(p->cfg.ode_solver, tb);
  return tb.has_err ? TRAIN_DTLOG_CAP : std::min(TRAIN_DTLOG_CAP_MAX, std::max(TRAIN_DTLOG_CAP, p->cfg.max_steps));
}

extern "C" int odevio_ode_rnn_tape_floats(const odevio_plan* p, int32_t B, int32_t P, int64_t* n_floats) {
  ARGCHK(p && n_floats && B > 0 && P > 0, "odevio_ode_rnn_tape_floats: bad argument");
  *n_floats = 0;
  if (p->cfg.model_type != ODEVIO_MODEL_ODE_RNN) return ODEVIO_OK;   // nothing a tape would save: the plain pair does the same work
  *n_floats = (int64_t)tape_layout(p, B, P, tape_default_cap(p)).total();
  return ODEVIO_OK;
}

extern "C" int odevio_ode_rnn_fwd_taped(odevio_plan* p, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                                        float* poses, float* h_T, float* tape, int64_t tape_floats, void* stream) {
  ARGCHK(p && fused && ts && poses && h_T && tape && B > 0 && P > 0, "odevio_ode_rnn_fwd_taped: bad argument");
  if (p->cfg.model_type != ODEVIO_MODEL_ODE_RNN) return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_ode_rnn_fwd_taped: ode-rnn plans only");
  const TapeLayout t = tape_layout(p, B, P, tape_default_cap(p));
  if ((int64_t)t.total() != tape_floats)
    return fail(ODEVIO_ERR_BAD_ARG, "odevio_ode_rnn_fwd_taped: tape of %lld floats, odevio_ode_rnn_tape_floats says %lld", (long long)tape_floats, (long long)t.total());
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  const StepLog log = t.carve(tape);
  HIPCHK(hipMemsetAsync(log.dtcnt, 0, t.n_cnt * sizeof(int), st));
  const int rc = ode_rnn_fwd(p, fused, ts, hc_in, B, P, poses, h_T, nullptr, st, log);
  post_status(p, st);
  return rc;
}

static int ode_rnn_bwd_impl(odevio_plan* p, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                            const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc,
                            const odevio_tensor* grads, int32_t n_grads, const float* tape, int64_t tape_floats, void* stream) {
  ARGCHK(p && fused && ts && grad_poses && B > 0 && P > 0 && n_grads >= 0 && (grads || n_grads == 0), "odevio_ode_rnn_bwd: bad argument");
  if (grad_hc && !hc_in) return fail(ODEVIO_ERR_BAD_ARG, "odevio_ode_rnn_bwd: grad_hc without hc_in");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  TrainModel m;
  int rc = fill_train_model(p, m);
  if (rc) return rc;
  TrainGrads g;
  memset(&g, 0, sizeof(g));
  std::vector<int> steps_it;   // (outlives the sweep below: TrainModel points into it)
  const int F = p->F;
  for (int i = 0; i < n_grads; ++i) {
    if (!grads[i].name || !grads[i].data) return fail(ODEVIO_ERR_BAD_ARG, "odevio_ode_rnn_bwd: gradient %d has no name / pointer", i);
    const std::string nm = grads[i].name;
    float* dst = (float*)grads[i].data;
    int64_t want = -1;
    for (int l = 0; l < m.nlin && want < 0; ++l) {
      const std::string pre = "Pose_net.ode_func.net." + std::to_string(2 * l);
      if (nm == pre + ".weight") { g.ode_w[l] = dst; want = (int64_t)m.dims[l + 1] * m.dims[l]; }
      else if (nm == pre + ".bias") { g.ode_b[l] = dst; want = m.dims[l + 1]; }
    }
    for (int l = 0; l < m.L && want < 0; ++l) {
      const std::string sfx = "_l" + std::to_string(l);
      const int64_t GF = (int64_t)(m.gru ? 3 : 1) * F;
      if (nm == "Pose_net.rnn.weight_ih" + sfx) { g.rnn_wih[l] = dst; want = GF * F; }
      else if (nm == "Pose_net.rnn.weight_hh" + sfx) { g.rnn_whh[l] = dst; want = GF * F; }
      else if (nm == "Pose_net.rnn.bias_ih" + sfx) { g.rnn_bih[l] = dst; want = GF; }
      else if (nm == "Pose_net.rnn.bias_hh" + sfx) { g.rnn_bhh[l] = dst; want = GF; }
    }
    if (want < 0) {
      if (nm == "Pose_net.regressor.0.weight") { g.reg_w0 = dst; want = (int64_t)128 * F; }
      else if (nm == "Pose_net.regressor.0.bias") { g.reg_b0 = dst; want = 128; }
      else if (nm == "Pose_net.regressor.2.weight") { g.reg_w2 = dst; want = 6 * 128; }
      else if (nm == "Pose_net.regressor.2.bias") { g.reg_b2 = dst; want = 6; }
    }
    if (want < 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_ode_rnn_bwd: '%s' is not a parameter of the pose path", nm.c_str());
    if (want != grads[i].numel)
      return fail(ODEVIO_ERR_BAD_ARG, "odevio_ode_rnn_bwd: gradient '%s' has %lld elements, expected %lld", nm.c_str(), (long long)grads[i].numel, (long long)want);
  }
  if (m.with_ode) {
    // The log of the forward's accepted steps (their sizes, the states they start from, the evolved state of every interval): the
    // caller's tape from odevio_ode_rnn_fwd_taped, or - without one - the forward once more on the persistent kernel.  The host
    // needs ONE number from it (the largest step count, which sizes the replay).  A log that turns out too short for an interval is
    // written again with 8 x the room.
    int cap = tape_default_cap(p);
    TapeLayout t = tape_layout(p, B, P, cap);
    StepLog log;
    bool have = false;
    steps_it.clear();
    if (tape) {
      if ((int64_t)t.total() != tape_floats)
        return fail(ODEVIO_ERR_BAD_ARG, "odevio_ode_rnn_bwd_taped: tape of %lld floats, odevio_ode_rnn_tape_floats says %lld", (long long)tape_floats, (long long)t.total());
      log = t.carve(const_cast<float*>(tape));
      have = true;
    }
    const bool want_log = m.adaptive || have || t.with_y;   // a fixed-step solve whose states would not fit is replayed in order, from ts alone
    for (int attempt = 0; want_log; ++attempt) {
      if (!have) {
        t = tape_layout(p, B, P, cap);
        if ((rc = ensure(p->train_log, t.total() + (size_t)t.R * F)) || (rc = ensure(p->out_seq, (size_t)B * P * F))) return rc;
        log = t.carve(p->train_log.p);
        float* hT_tmp = p->train_log.p + t.total();
        HIPCHK(hipMemsetAsync(log.dtcnt, 0, t.n_cnt * sizeof(int), st));
        if ((rc = run_sequence(p, fused, ts, hc_in, B, P, p->out_seq.p, hT_tmp, nullptr, st, log))) return rc;
      }
      if (!m.adaptive) break;                                  // fixed-step: ode_substeps steps everywhere, nothing to read back
      const size_t n_cnt = (size_t)t.R * P;
      std::vector<int> cnt(n_cnt);
      HIPCHK(hipMemcpyAsync(cnt.data(), log.dtcnt, n_cnt * sizeof(int), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      if (!have && (rc = odevio_check(p, stream))) return rc;     // that forward's own failures (step budget, ...)
      const int most = *std::max_element(cnt.begin(), cnt.end());
      if (most > t.cap) {
        if (attempt >= 3 || t.cap >= TRAIN_DTLOG_CAP_MAX)
          return fail(ODEVIO_ERR_MAX_STEPS, "odevio_ode_rnn_bwd: an interval took %d accepted steps, the log holds at most %d", most, TRAIN_DTLOG_CAP_MAX);
        cap = std::min(TRAIN_DTLOG_CAP_MAX, t.cap * 8);
        have = false;
        continue;
      }
      m.jmax = std::max(1, most);
      // per interval the most steps any row took: the sweep of an interval starts there (rows are [layer][sequence] x intervals)
      steps_it.assign(P, 0);
      for (size_t r = 0; r < (size_t)t.R; ++r)
        for (int it = 0; it < P; ++it) steps_it[it] = std::max(steps_it[it], cnt[r * P + it]);
      m.steps_per_interval = steps_it.data();
      break;
    }
    if (want_log) {
      m.dtcnt = log.dtcnt; m.dtlog_cap = t.cap;
      m.dtlog = m.adaptive ? log.dtlog : nullptr;
      m.ylog = log.ylog; m.yend = log.yend;
    }
  }
  if ((rc = ensure(p->train_ws, train_workspace_floats(m, B, P)))) return rc;
  rc = train_ode_rnn_bwd(m, p->train_ws.p, fused, ts, hc_in, B, P, grad_poses, grad_hT, grad_fused, grad_hc, g, st);
  if (rc) return fail(rc, "odevio_ode_rnn_bwd: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int odevio_ode_rnn_bwd(odevio_plan* p, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                                  const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc,
                                  const odevio_tensor* grads, int32_t n_grads, void* stream) {
  return ode_rnn_bwd_impl(p, fused, ts, hc_in, B, P, grad_poses, grad_hT, grad_fused, grad_hc, grads, n_grads, nullptr, 0, stream);
}

extern "C" int odevio_ode_rnn_bwd_taped(odevio_plan* p, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                                        const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc,
                                        const odevio_tensor* grads, int32_t n_grads, const float* tape, int64_t tape_floats, void* stream) {
  ARGCHK(tape, "odevio_ode_rnn_bwd_taped: no tape");
  return ode_rnn_bwd_impl(p, fused, ts, hc_in, B, P, grad_poses, grad_hT, grad_fused, grad_hc, grads, n_grads, tape, tape_floats, stream);
}

// odevio_plan_update: the index maps of load_pose_net as device kernels, from the caller's device tensors straight into the
// plan's buffers on the caller's stream (ordered behind the kernels that still read the old values; no host round trip).
extern "C" int odevio_plan_update(odevio_plan* p, const odevio_tensor* weights, int32_t n_weights, void* stream) {
  ARGCHK(p && weights && n_weights > 0, "odevio_plan_update: bad argument");
  if (p->cfg.model_type == ODEVIO_MODEL_CDE) return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_plan_update: ode-rnn / rnn plans only");
  if (p->padded) return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_plan_update: widths that are not multiples of 32 need odevio_plan_create");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  std::map<std::string, std::pair<const float*, int64_t>> m;
  for (int i = 0; i < n_weights; ++i)
    if (weights[i].name && weights[i].data) m[weights[i].name] = {(const float*)weights[i].data, weights[i].numel};
  const int F = p->F;
  int rc = 0;
  auto src = [&](const std::string& name, int64_t numel) -> const float* {
    auto it = m.find(name);
    if (it == m.end()) { rc = fail(ODEVIO_ERR_MISSING_WEIGHT, "odevio_plan_update: weight '%s' not provided", name.c_str()); return nullptr; }
    if (it->second.second != numel) { rc = fail(ODEVIO_ERR_BAD_ARG, "odevio_plan_update: weight '%s' has the wrong size", name.c_str()); return nullptr; }
    return it->second.first;
  };
  auto copy = [&](float* dst, const float* s, size_t n) {   // (a kernel, not hipMemcpyAsync: twenty of these per training step)
    if (!rc) device_copy_f32(dst, s, n, st);
  };
  // first pass: every tensor present with the right size (nothing is written before that is known)
  std::vector<std::pair<std::string, int64_t>> need;
  if (p->cfg.fuse_method == ODEVIO_FUSE_SOFT) { need.push_back({"Pose_net.fuse.net.0.weight", (int64_t)F * F}); need.push_back({"Pose_net.fuse.net.0.bias", F}); }
  if (p->cfg.fuse_method == ODEVIO_FUSE_HARD) { need.push_back({"Pose_net.fuse.net.0.weight", (int64_t)2 * F * F}); need.push_back({"Pose_net.fuse.net.0.bias", (int64_t)2 * F}); }
  need.push_back({"Pose_net.regressor.0.weight", (int64_t)128 * F}); need.push_back({"Pose_net.regressor.0.bias", 128});
  need.push_back({"Pose_net.regressor.2.weight", 6 * 128}); need.push_back({"Pose_net.regressor.2.bias", 6});
  if (p->cfg.model_type == ODEVIO_MODEL_ODE_RNN)
    for (int l = 0; l < p->nlin; ++l) {
      const std::string pre = "Pose_net.ode_func.net." + std::to_string(2 * l);
      need.push_back({pre + ".weight", (int64_t)p->dims[l + 1] * p->dims[l]});
      need.push_back({pre + ".bias", p->dims[l + 1]});
    }
  const bool gru = p->cfg.rnn_type == ODEVIO_RNN_GRU;
  const int64_t GF = (int64_t)(gru ? 3 : 1) * F;
  for (int l = 0; l < p->cfg.rnn_num_layers; ++l) {
    const std::string s = std::to_string(l);
    need.push_back({"Pose_net.rnn.weight_ih_l" + s, GF * F}); need.push_back({"Pose_net.rnn.weight_hh_l" + s, GF * F});
    need.push_back({"Pose_net.rnn.bias_ih_l" + s, GF}); need.push_back({"Pose_net.rnn.bias_hh_l" + s, GF});
  }
  for (const auto& nd : need) {
    (void)src(nd.first, nd.second);
    if (rc) return rc;
  }
  // second pass: the layouts of load_pose_net
  if (p->cfg.fuse_method == ODEVIO_FUSE_SOFT) {
    const float* w = src("Pose_net.fuse.net.0.weight", (int64_t)F * F);
    copy(p->fuse_w, w, (size_t)F * F);
    relayout_transpose(w, p->fuse_w_t, F, F, st);
    copy(p->fuse_b, src("Pose_net.fuse.net.0.bias", F), F);
  }
  if (p->cfg.fuse_method == ODEVIO_FUSE_HARD) {
    const float* wh = src("Pose_net.fuse.net.0.weight", (int64_t)2 * F * F);
    copy(p->fuse_w, wh, (size_t)2 * F * F);
    relayout_transpose(wh, p->fuse_w_t, 2 * F, F, st);
    copy(p->fuse_b, src("Pose_net.fuse.net.0.bias", (int64_t)2 * F), (size_t)2 * F);
  }
  {
    const float* w = src("Pose_net.regressor.0.weight", (int64_t)128 * F);
    copy(p->reg_w0, w, (size_t)128 * F);
    relayout_transpose(w, const_cast<float*>(p->train.reg_w0_t), 128, F, st);
    copy(p->reg_b0, src("Pose_net.regressor.0.bias", 128), 128);
    copy(p->reg_w2, src("Pose_net.regressor.2.weight", 6 * 128), 6 * 128);
    copy(p->reg_b2, src("Pose_net.regressor.2.bias", 6), 6);
  }
  if (p->cfg.model_type == ODEVIO_MODEL_ODE_RNN)
    for (int l = 0; l < p->nlin; ++l) {
      const std::string pre = "Pose_net.ode_func.net." + std::to_string(2 * l);
      const int N = p->dims[l + 1], K = p->dims[l];
      const float* w = src(pre + ".weight", (int64_t)N * K);
      relayout_shard(w, p->ode_w[l], N, K, INTEG_MEMBERS, st);
      copy(const_cast<float*>(p->train.ode_w[l]), w, (size_t)N * K);
      relayout_transpose(w, const_cast<float*>(p->train.ode_w_t[l]), N, K, st);
      relayout_shard(p->train.ode_w_t[l], p->ode_wT[l], K, N, INTEG_MEMBERS, st);   // (behind the transpose on the same stream)
      copy(p->ode_b[l], src(pre + ".bias", N), N);
    }
  for (int l = 0; l < p->cfg.rnn_num_layers; ++l) {
    const std::string s = std::to_string(l);
    const float* wih = src("Pose_net.rnn.weight_ih_l" + s, GF * F);
    const float* whh = src("Pose_net.rnn.weight_hh_l" + s, GF * F);
    const float* bih = src("Pose_net.rnn.bias_ih_l" + s, GF);
    const float* bhh = src("Pose_net.rnn.bias_hh_l" + s, GF);
    copy(const_cast<float*>(p->train.rnn_wih[l]), wih, (size_t)GF * F);
    copy(const_cast<float*>(p->train.rnn_whh[l]), whh, (size_t)GF * F);
    copy(const_cast<float*>(p->train.rnn_bih[l]), bih, GF);
    copy(const_cast<float*>(p->train.rnn_bhh[l]), bhh, GF);
    relayout_transpose(wih, const_cast<float*>(p->train.rnn_wih_t[l]), (int)GF, F, st);
    relayout_transpose(whh, const_cast<float*>(p->train.rnn_whh_t[l]), (int)GF, F, st);
    relayout_rnn(wih, whh, bih, bhh, p->rnn_w[l], p->rnn_b[l], F, gru ? 1 : 0, INTEG_MEMBERS, st);
  }
  if (rc) return rc;
  if (hipGetLastError() != hipSuccess) return fail(ODEVIO_ERR_HIP, "odevio_plan_update: a re-layout kernel failed to launch");
  return 0;
}

extern "C" int odevio_set_seed(odevio_plan* p, uint64_t seed) {
  ARGCHK(p, "odevio_set_seed: bad argument");
  p->seed = seed;
  p->rng_calls = 0;
  return 0;
}

extern "C" int odevio_set_rng_state(odevio_plan* p, uint64_t seed, uint64_t calls) {
  ARGCHK(p, "odevio_set_rng_state: bad argument");
  p->seed = seed;
  p->rng_calls = calls;
  return 0;
}

extern "C" int odevio_rng_state(odevio_plan* p, uint64_t* seed, uint64_t* calls) {
  ARGCHK(p && seed && calls, "odevio_rng_state: bad argument");
  *seed = p->seed;
  *calls = p->rng_calls;
  return 0;
}

extern "C" int odevio_debug_gumbel(uint64_t seed, uint64_t call, int64_t n, float* out, void* stream) {
  ARGCHK(out && n > 0, "odevio_debug_gumbel: bad argument");
  launch_gumbel_dump(out, (size_t)n, seed, call, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "odevio_debug_gumbel: launch failed");
}

// FusionModule "hard": the straight-through backward for the mask drawn by call `call` of seed `seed` (odevio_rng_state before the
// forward); uses train.hip's skinny GEMMs like the soft path
extern "C" int odevio_fuse_hard_bwd(odevio_plan* p, const float* fv, const float* fi, int32_t P, uint64_t seed, uint64_t call,
                                    const float* grad_fused, float* grad_fv, float* grad_fi, const odevio_tensor* grads, int32_t n_grads,
                                    void* stream) {
  ARGCHK(p && fv && fi && grad_fused && P > 0 && n_grads >= 0 && (grads || n_grads == 0), "odevio_fuse_hard_bwd: bad argument");
  if (p->cfg.fuse_method != ODEVIO_FUSE_HARD) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_hard_bwd: the plan's fuse_method is not 'hard'");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  const int F = p->F;
  float *gW = nullptr, *gb = nullptr;
  for (int i = 0; i < n_grads; ++i) {
    if (!grads[i].name || !grads[i].data) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_hard_bwd: gradient %d has no name / pointer", i);
    const std::string nm = grads[i].name;
    int64_t want = -1;
    if (nm == "Pose_net.fuse.net.0.weight") { gW = (float*)grads[i].data; want = (int64_t)2 * F * F; }
    else if (nm == "Pose_net.fuse.net.0.bias") { gb = (float*)grads[i].data; want = (int64_t)2 * F; }
    if (want < 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_hard_bwd: '%s' is not a parameter of this fusion module", nm.c_str());
    if (want != grads[i].numel) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_hard_bwd: gradient '%s' has the wrong size", nm.c_str());
  }
  const size_t n = (size_t)P * F;
  int rc;
  if ((rc = ensure(p->train_aux, 6 * n))) return rc;          // cat, logits [2n], g_logits [2n], g_cat
  float *cat = p->train_aux.p, *logits = cat + n, *gl = logits + 2 * n, *gc = gl + 2 * n;
  rc = train_fuse_hard_bwd(p->fuse_w, p->fuse_w_t, p->fuse_b, cat, logits, gl, gc, fv, p->cfg.v_f_len, fi, p->cfg.i_f_len, P, seed, call, grad_fused,
                           grad_fv, grad_fi, gW, gb, st);
  if (rc) return fail(rc, "odevio_fuse_hard_bwd: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int odevio_fuse_bwd(odevio_plan* p, const float* fv, const float* fi, int32_t P, const float* grad_fused, float* grad_fv,
                               float* grad_fi, const odevio_tensor* grads, int32_t n_grads, void* stream) {
  ARGCHK(p && fv && fi && grad_fused && P > 0 && n_grads >= 0 && (grads || n_grads == 0), "odevio_fuse_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  if (p->cfg.fuse_method == ODEVIO_FUSE_HARD)
    return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_fuse_bwd: fuse_method 'hard' (straight-through gumbel-softmax) has no backward here");
  const int F = p->F;
  const bool soft = p->cfg.fuse_method == ODEVIO_FUSE_SOFT;
  float *gW = nullptr, *gb = nullptr;
  for (int i = 0; i < n_grads; ++i) {
    if (!grads[i].name || !grads[i].data) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_bwd: gradient %d has no name / pointer", i);
    const std::string nm = grads[i].name;
    int64_t want = -1;
    if (soft && nm == "Pose_net.fuse.net.0.weight") { gW = (float*)grads[i].data; want = (int64_t)F * F; }
    else if (soft && nm == "Pose_net.fuse.net.0.bias") { gb = (float*)grads[i].data; want = F; }
    if (want < 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_bwd: '%s' is not a parameter of this fusion module", nm.c_str());
    if (want != grads[i].numel) return fail(ODEVIO_ERR_BAD_ARG, "odevio_fuse_bwd: gradient '%s' has the wrong size", nm.c_str());
  }
  int rc;
  if (soft && (rc = ensure(p->train_aux, train_fuse_workspace_floats(P, F)))) return rc;
  rc = train_fuse_bwd(soft ? 1 : 0, p->fuse_w, p->fuse_w_t, p->fuse_b, p->train_aux.p, fv, p->cfg.v_f_len, fi, p->cfg.i_f_len, P, grad_fused,
                      grad_fv, grad_fi, gW, gb, st);
  if (rc) return fail(rc, "odevio_fuse_bwd: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

static void fill_imu_train(odevio_plan* p, ImuTrain& m);
static int parse_imu_grads(odevio_plan* p, const odevio_tensor* grads, int n_grads, ImuGrads& g, const char* who) {
  const int cin[3] = {6, 64, 128}, cout[3] = {64, 128, 256}, idx[3] = {0, 4, 8};
  for (int j = 0; j < n_grads; ++j) {
    if (!grads[j].name || !grads[j].data) return fail(ODEVIO_ERR_BAD_ARG, "%s: gradient %d has no name / pointer", who, j);
    const std::string nm = grads[j].name;
    float* dst = (float*)grads[j].data;
    int64_t want = -1;
    for (int i = 0; i < 3 && want < 0; ++i) {
      const std::string cv = "Inertial_net.encoder_conv." + std::to_string(idx[i]), bn = "Inertial_net.encoder_conv." + std::to_string(idx[i] + 1);
      if (nm == cv + ".weight") { g.w[i] = dst; want = (int64_t)cout[i] * cin[i] * 3; }
      else if (nm == cv + ".bias") { g.b[i] = dst; want = cout[i]; }
      else if (nm == bn + ".weight") { g.gamma[i] = dst; want = cout[i]; }
      else if (nm == bn + ".bias") { g.beta[i] = dst; want = cout[i]; }
    }
    if (want < 0) {
      if (nm == "Inertial_net.proj.weight") { g.proj_w = dst; want = (int64_t)p->cfg.i_f_len * 2816; }
      else if (nm == "Inertial_net.proj.bias") { g.proj_b = dst; want = p->cfg.i_f_len; }
    }
    if (want < 0) return fail(ODEVIO_ERR_BAD_ARG, "%s: '%s' is not a parameter of Inertial_net", who, nm.c_str());
    if (want != grads[j].numel) return fail(ODEVIO_ERR_BAD_ARG, "%s: gradient '%s' has the wrong size", who, nm.c_str());
  }
  return 0;
}

extern "C" int odevio_imu_encoder_bwd(odevio_plan* p, const float* imu, int32_t B, int32_t T, const float* grad_fi, const odevio_tensor* grads,
                                      int32_t n_grads, void* stream) {
  ARGCHK(p && imu && grad_fi && B > 0 && T >= 11 && (T - 1) % 10 == 0 && n_grads >= 0 && (grads || n_grads == 0), "odevio_imu_encoder_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  ImuTrain m{};
  ImuGrads g{};
  fill_imu_train(p, m);
  if (m.i_f_len % 4 || m.i_f_len > 256)   // (the workspace holds the transposed projection as [2816][<= 256])
    return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_imu_encoder_bwd: i_f_len must be a multiple of 4, at most 256");
  int rc = parse_imu_grads(p, grads, n_grads, g, "odevio_imu_encoder_bwd");
  if (rc) return rc;
  const int P = B * ((T - 1) / 10);
  if ((rc = ensure(p->train_aux, train_imu_workspace_floats(P)))) return rc;
  rc = train_imu_bwd(m, p->train_aux.p, imu, B, T, grad_fi, nullptr, g, st);
  if (rc) return fail(rc, "odevio_imu_encoder_bwd: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

// ---- model.train() forward of the encoders, and the inertial encoder's backward through batch-statistics BatchNorm + Dropout
static void fill_imu_train(odevio_plan* p, ImuTrain& m) {
  const int cin[3] = {6, 64, 128};
  for (int i = 0; i < 3; ++i) {
    m.w[i] = p->imu_wref[i]; m.wt[i] = p->imu_w[i]; m.s[i] = p->imu_s[i]; m.h[i] = p->imu_h[i];
    m.var[i] = p->imu_var[i]; m.mean[i] = p->imu_mean[i]; m.bias[i] = p->imu_bias[i];
    m.ldk[i] = (3 * cin[i] + 15) / 16 * 16;
  }
  m.eps = 1e-5f;
  m.proj_w = p->proj_w;
  m.i_f_len = p->cfg.i_f_len;
}
static void fill_imu_train_mode(odevio_plan* p, ImuTrainMode& tm, float p_drop, unsigned long long seed, unsigned long long call0,
                                const odevio_tensor* stats, int n_stats) {
  const int cout[3] = {64, 128, 256}, idx[3] = {1, 5, 9};
  for (int i = 0; i < 3; ++i) {
    tm.gamma[i] = p->imu_gamma[i]; tm.beta[i] = p->imu_beta[i];
    const std::string bn = "Inertial_net.encoder_conv." + std::to_string(idx[i]);
    tm.run_mean[i] = named_ptr(stats, n_stats, bn + ".running_mean", cout[i]);
    tm.run_var[i] = named_ptr(stats, n_stats, bn + ".running_var", cout[i]);
    tm.drop[i] = make_dropout(seed, call0 + i, p_drop);
  }
  tm.momentum = 0.1f;   // nn.BatchNorm1d default (Encoder.py:45)
}

extern "C" int odevio_image_encoder_fwd_train(odevio_plan* p, const float* img, int32_t B, int32_t S, float* fv, int32_t ld_fv,
                                              const odevio_tensor* stats, int32_t n_stats, int32_t keep, void* stream) {
  ARGCHK(p && img && fv && B > 0 && S > 1 && ld_fv >= p->cfg.v_f_len && n_stats >= 0 && (stats || n_stats == 0),
         "odevio_image_encoder_fwd_train: bad argument");
  POLL(p, stream);
  const int rc = image_encoder_train(p, img, B, S, fv, ld_fv, stats, n_stats, keep != 0, (hipStream_t)stream);
  post_status(p, (hipStream_t)stream);
  return rc;
}

extern "C" int odevio_image_encoder_bwd(odevio_plan* p, const float* img, int32_t B, int32_t S, const float* grad_fv, int32_t ld_gfv,
                                        const odevio_tensor* grads, int32_t n_grads, void* stream) {
  ARGCHK(p && img && grad_fv && B > 0 && S > 1 && ld_gfv >= p->cfg.v_f_len && n_grads >= 0 && (grads || n_grads == 0),
         "odevio_image_encoder_bwd: bad argument");
  POLL(p, stream);
  return image_encoder_bwd(p, img, B, S, grad_fv, ld_gfv, grads, n_grads, (hipStream_t)stream);
}

extern "C" int odevio_imu_encoder_fwd_train(odevio_plan* p, const float* imu, int32_t B, int32_t T, float p_drop, const odevio_tensor* stats,
                                            int32_t n_stats, float* fi, int32_t ld_fi, void* stream) {
  ARGCHK(p && imu && fi && B > 0 && T >= 11 && ld_fi >= p->cfg.i_f_len && p_drop >= 0.f && p_drop < 1.f && n_stats >= 0 && (stats || n_stats == 0),
         "odevio_imu_encoder_fwd_train: bad argument");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  ImuTrain m{};
  ImuTrainMode tm{};
  fill_imu_train(p, m);
  fill_imu_train_mode(p, tm, p_drop, p->seed, p->rng_calls, stats, n_stats);
  p->rng_calls += 3;   // one draw per block, whatever p_drop is (the stream position never depends on a flag)
  const int P = B * ((T - 1) / 10);
  int rc;
  if ((rc = ensure(p->train_aux, train_imu_train_workspace_floats(P)))) return rc;
  rc = train_imu_fwd_train(m, tm, p->train_aux.p, imu, B, T, p->proj_b, fi, ld_fi, st);
  if (rc) return fail(rc, "odevio_imu_encoder_fwd_train: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int odevio_imu_encoder_bwd_train(odevio_plan* p, const float* imu, int32_t B, int32_t T, float p_drop, uint64_t seed, uint64_t call0,
                                            const float* grad_fi, const odevio_tensor* grads, int32_t n_grads, void* stream) {
  ARGCHK(p && imu && grad_fi && B > 0 && T >= 11 && (T - 1) % 10 == 0 && p_drop >= 0.f && p_drop < 1.f && n_grads >= 0 && (grads || n_grads == 0),
         "odevio_imu_encoder_bwd_train: bad argument");
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  ImuTrain m{};
  ImuTrainMode tm{};
  ImuGrads g{};
  fill_imu_train(p, m);
  fill_imu_train_mode(p, tm, p_drop, seed, call0, nullptr, 0);
  if (m.i_f_len % 4 || m.i_f_len > 256) return fail(ODEVIO_ERR_UNSUPPORTED, "odevio_imu_encoder_bwd_train: i_f_len must be a multiple of 4, at most 256");
  int rc = parse_imu_grads(p, grads, n_grads, g, "odevio_imu_encoder_bwd_train");
  if (rc) return rc;
  const int P = B * ((T - 1) / 10);
  if ((rc = ensure(p->train_aux, train_imu_train_workspace_floats(P)))) return rc;
  rc = train_imu_bwd_train(m, tm, p->train_aux.p, imu, B, T, grad_fi, g, st);
  if (rc) return fail(rc, "odevio_imu_encoder_bwd_train: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int odevio_debug_dropout(uint64_t seed, uint64_t call, float p_drop, int64_t n, float* out, void* stream) {
  ARGCHK(out && n > 0 && p_drop >= 0.f && p_drop < 1.f, "odevio_debug_dropout: bad argument");
  launch_dropout_dump(out, (size_t)n, make_dropout(seed, call, p_drop), (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? 0 : fail(ODEVIO_ERR_HIP, "odevio_debug_dropout: launch failed");
}

extern "C" int odevio_grad_clip(odevio_plan* p, const odevio_tensor* grads, int32_t n_grads, float max_norm, float* norm_coef,
                                void* stream) {
  ARGCHK(p && grads && n_grads > 0 && norm_coef && max_norm > 0.f, "odevio_grad_clip: bad argument");
  hipStream_t st = (hipStream_t)stream;
  std::vector<const float*> ptr(n_grads);
  std::vector<size_t> num(n_grads);
  for (int i = 0; i < n_grads; ++i) {
    if (!grads[i].data || grads[i].numel <= 0) return fail(ODEVIO_ERR_BAD_ARG, "odevio_grad_clip: gradient %d is empty", i);
    ptr[i] = (const float*)grads[i].data;
    num[i] = (size_t)grads[i].numel;
  }
  int rc;
  if ((rc = ensure(p->train_aux, 2 * train_grad_clip_workspace_doubles(n_grads)))) return rc;   // doubles in a float buffer
  rc = train_grad_clip(ptr.data(), num.data(), n_grads, max_norm, reinterpret_cast<double*>(p->train_aux.p), norm_coef, st);
  if (rc) return fail(rc, "odevio_grad_clip: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" int odevio_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int32_t step, const float* norm_coef, void* stream) {
  ARGCHK(param && grad && exp_avg && exp_avg_sq && numel > 0 && step >= 1 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f &&
             beta2 < 1.f && eps > 0.f && weight_decay >= 0.f,
         "odevio_adam_step: bad argument");
  if (train_adam_step(param, grad, exp_avg, exp_avg_sq, (size_t)numel, lr, beta1, beta2, eps, weight_decay, step, norm_coef, (hipStream_t)stream))
    return fail(ODEVIO_ERR_HIP, "odevio_adam_step: launch failed");
  return 0;
}

extern "C" int odevio_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t numel, float lr, float momentum, float weight_decay,
                               int32_t step, const float* norm_coef, void* stream) {
  ARGCHK(param && grad && numel > 0 && step >= 1 && lr >= 0.f && momentum >= 0.f && weight_decay >= 0.f && (momentum_buf || momentum == 0.f),
         "odevio_sgd_step: bad argument");
  if (train_sgd_step(param, grad, momentum_buf, (size_t)numel, lr, momentum, weight_decay, step, norm_coef, (hipStream_t)stream))
    return fail(ODEVIO_ERR_HIP, "odevio_sgd_step: launch failed");
  return 0;
}

extern "C" int odevio_optimizer_step(int32_t kind, const odevio_tensor* params, const odevio_tensor* grads, const odevio_tensor* state1,
                                     const odevio_tensor* state2, const float* lrs, int32_t n, float beta1, float beta2, float eps,
                                     float weight_decay, int32_t step, const float* norm_coef, void* stream) {
  ARGCHK((kind == 0 || kind == 1) && params && grads && lrs && n > 0 && step >= 1 && weight_decay >= 0.f && beta1 >= 0.f && beta1 < 1.f,
         "odevio_optimizer_step: bad argument");
  if (kind == 0) ARGCHK(state1 && state2 && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, "odevio_optimizer_step: Adam needs both state tensors, betas in [0, 1), eps > 0");
  if (kind == 1) ARGCHK(state1 || beta1 == 0.f, "odevio_optimizer_step: SGD with momentum needs its buffers");
  for (int i = 0; i < n; ++i) {
    const bool ok = params[i].data && grads[i].data && params[i].numel > 0 && grads[i].numel == params[i].numel && lrs[i] >= 0.f &&
                    (!state1 || (state1[i].data && state1[i].numel == params[i].numel)) && (kind != 0 || (state2[i].data && state2[i].numel == params[i].numel));
    if (!ok) return fail(ODEVIO_ERR_BAD_ARG, "odevio_optimizer_step: tensor %d ('%s'): missing pointer or sizes that differ", i, params[i].name ? params[i].name : "");
  }
  for (int t0 = 0; t0 < n; t0 += OPT_TABLE_MAX) {
    OptTable t;
    t.n = std::min(OPT_TABLE_MAX, n - t0);
    for (int i = 0; i < t.n; ++i) {
      const int k = t0 + i;
      t.e[i].p = (float*)params[k].data; t.e[i].g = (const float*)grads[k].data;
      t.e[i].s1 = state1 ? (float*)state1[k].data : nullptr; t.e[i].s2 = (kind == 0) ? (float*)state2[k].data : nullptr;
      t.e[i].n = (size_t)params[k].numel; t.e[i].lr = lrs[k];
    }
    if (train_optimizer_multi(t, kind, beta1, beta2, eps, weight_decay, step, norm_coef, (hipStream_t)stream))
      return fail(ODEVIO_ERR_HIP, "odevio_optimizer_step: launch failed");
  }
  return 0;
}

extern "C" int odevio_pose_loss(const float* poses, const float* gts, int32_t n_rows, float* loss3, float* grad_poses, void* stream) {
  ARGCHK(poses && gts && loss3 && n_rows > 0, "odevio_pose_loss: bad argument");
  if (train_pose_loss(poses, gts, n_rows, loss3, grad_poses, (hipStream_t)stream)) return fail(ODEVIO_ERR_HIP, "odevio_pose_loss: launch failed");
  return 0;
}

extern "C" int odevio_cde_func(odevio_plan* p, const float* z, const float* obs, int32_t B, int32_t L, int32_t seg, float* out,
                               void* stream) {
  ARGCHK(p && z && obs && out && B > 0 && L > 1 && seg >= 0 && seg <= 2 * L - 3, "odevio_cde_func: bad argument");
  if (p->cfg.model_type != ODEVIO_MODEL_CDE) return fail(ODEVIO_ERR_UNSUPPORTED, "plan is not a Neural-CDE plan");
  hipStream_t st = (hipStream_t)stream;
  const int n = B * p->cde.H;
  int rc;
  if ((rc = ensure(p->cde_fn_ws, 2 * (size_t)n))) return rc;
  p->cde.n_cu = p->n_cu;
  const CdeWhen wh{nullptr, 0, seg, 0};
  const float* x = z;
  float* bufs[2] = {p->cde_fn_ws.p, p->cde_fn_ws.p + n};
  for (int l = 0; l < p->cde.n_hidden; ++l) {
    cde_launch_hidden(wh, x, p->cde.w[l], p->cde.b[l], bufs[l & 1], B, p->cde.H, p->cde.act, st);
    x = bufs[l & 1];
  }
  // with the stage timers on (odevio_profile_enable): HIP events around the last layer alone, for its roofline
  if (p->prof) {
    if (!p->ev_cde[0]) { HIPCHK(hipEventCreate(&p->ev_cde[0])); HIPCHK(hipEventCreate(&p->ev_cde[1])); }
    HIPCHK(hipEventRecord(p->ev_cde[0], st));
  }
  if (cde_launch_last(p->cde, wh, x, obs, B, L, out, st)) return fail(ODEVIO_ERR_HIP, "CDE vector field launch failed");
  if (p->prof) HIPCHK(hipEventRecord(p->ev_cde[1], st));
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int odevio_cde_last_ms(odevio_plan* p, float* ms_out) {
  ARGCHK(p && ms_out && p->prof && p->ev_cde[1], "odevio_cde_last_ms: no timed odevio_cde_func call (enable the stage timers first)");
  HIPCHK(hipEventSynchronize(p->ev_cde[1]));
  HIPCHK(hipEventElapsedTime(ms_out, p->ev_cde[0], p->ev_cde[1]));
  return 0;
}

static int forward_any(odevio_plan* p, const void* img, bool img_u8, const float* imu, int32_t T, const float* ts,
                       const float* hc, int32_t B, int32_t S, float* poses, float* h_T, int32_t* stats, void* stream);

extern "C" int odevio_forward(odevio_plan* p, const float* img, const float* imu, int32_t T, const float* ts,
                              const float* hc, int32_t B, int32_t S, float* poses, float* h_T, int32_t* stats,
                              void* stream) {
  return forward_any(p, img, false, imu, T, ts, hc, B, S, poses, h_T, stats, stream);
}

extern "C" int odevio_forward_u8(odevio_plan* p, const uint8_t* img, const float* imu, int32_t T, const float* ts,
                                 const float* hc, int32_t B, int32_t S, float* poses, float* h_T, int32_t* stats,
                                 void* stream) {
  return forward_any(p, img, true, imu, T, ts, hc, B, S, poses, h_T, stats, stream);
}

static int forward_any(odevio_plan* p, const void* img, bool img_u8, const float* imu, int32_t T, const float* ts,
                       const float* hc, int32_t B, int32_t S, float* poses, float* h_T, int32_t* stats, void* stream) {
  ARGCHK(p && img && imu && ts && poses && h_T && B > 0 && S > 1, "odevio_forward: bad argument");
  if ((T - 1) / 10 != S - 1) return fail(ODEVIO_ERR_BAD_ARG, "imu length %d does not give %d frame pairs", T, S - 1);
  hipStream_t st = (hipStream_t)stream;
  POLL(p, st);
  const int P = B * (S - 1), F = p->F;
  int rc;
  if ((rc = ensure(p->fcat, (size_t)P * F)) || (rc = ensure(p->fused, (size_t)P * F))) return rc;
  // encoders write straight into the concatenated feature rows (torch.cat of FusionModule.py:19 is free).  The
  // inertial encoder (latency-bound: 160 small workgroups + one skinny GEMM) runs on the plan's side stream underneath
  // the image encoder; both join before the fusion.  Allocations first: nothing may (re)allocate while two streams run.
  if ((rc = ensure(p->imu_act, (size_t)P * 2816)) || (rc = ensure(p->partial_side, (size_t)64 * P * p->cfg.i_f_len))) return rc;
  if (!p->side) {
    HIPCHK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(p->ev_fork, st));
  HIPCHK(hipStreamWaitEvent(p->side, p->ev_fork, 0));
  if ((rc = imu_encoder(p, imu, B, T, p->fcat.p + p->cfg.v_f_len, F, p->side, &p->partial_side))) return rc;
  HIPCHK(hipEventRecord(p->ev_join, p->side));
  if ((rc = image_encoder(p, img, B, S, p->fcat.p, F, st, img_u8))) return rc;
  HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
  const float* fused = p->fcat.p;
  if (p->cfg.fuse_method != ODEVIO_FUSE_CAT) {
    if ((rc = fuse_from_cat(p, p->fcat.p, P, p->fused.p, st))) return rc;
    fused = p->fused.p;
  }
  rc = ode_rnn_fwd(p, fused, ts, hc, B, S - 1, poses, h_T, stats, st);
  post_status(p, st);
  return rc;
}
