/*
 * odevio.h - C ABI of libodevio.so: the MI355X (gfx950) implementation of the ODE-VIO hot path.
 *
 * The reference (mc1017/ODE-VIO) is 100% Python and has no FFI layer of its own; the drop-in
 * boundary is its Python class surface, DeepVIO(opt).forward(img, imu, timestamps, hc)
 * (reference src/models/DeepVIO.py:37-68).  This header is the C boundary placed directly UNDER
 * that surface: each entry point replaces one reference method and is what a maintainer would bind
 * from the reference's Python (ctypes stub in INTEGRATION.md).  Plain pointers and sizes only - no
 * torch types.  All tensor pointers are DEVICE pointers to contiguous fp32 unless stated; `stream`
 * is a hipStream_t passed as void* (NULL = the null stream).  The library never allocates or frees
 * caller tensors; it owns only the opaque plan (re-laid-out weights + activation workspace).
 *
 * Error convention: every function returns 0 on success and a negative odevio_status otherwise;
 * odevio_last_error() gives a thread-local message.  The Python host maps
 * ODEVIO_ERR_UNSUPPORTED/BAD_ARG to ValueError like the reference's own constructors
 * (reference src/models/PoseODERNN.py:136,146; src/models/ODEFunc.py:34).
 *
 * Threading: a plan is not re-entrant (one forward at a time per plan); different plans are
 * independent.  No internal threads.  Nothing synchronises the stream except odevio_plan_create
 * (weight re-layout), odevio_reserve, odevio_check and the adaptive solver of odevio_cde_fwd (once per batch of
 * enqueued step attempts, to read the device-side controller's `done` word).
 */
#ifndef ODEVIO_H
#define ODEVIO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODEVIO_VERSION 1

typedef struct odevio_plan odevio_plan;

enum odevio_status {
  ODEVIO_OK = 0,
  ODEVIO_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, unknown enum              */
  ODEVIO_ERR_UNSUPPORTED = -2,  /* valid request the kernels do not cover (dimension rules)   */
  ODEVIO_ERR_MISSING_WEIGHT = -3,
  ODEVIO_ERR_HIP = -4,          /* a HIP runtime call failed                                  */
  ODEVIO_ERR_NO_DEVICE = -5,
  ODEVIO_ERR_TIMEOUT = -6,      /* a bounded in-kernel wait gave up (reported by odevio_check)*/
  ODEVIO_ERR_MAX_STEPS = -7,    /* adaptive solver exceeded max_steps (reported by check)     */
  ODEVIO_ERR_RANGE = -8,        /* an encoder activation left the fp16x2 range (reported by check) */
  ODEVIO_ERR_BOUNDS = -9        /* audit build only (make AUDIT=1): a kernel computed an address outside its buffers */
};

/* reference src/models/ODEFunc.py:23-36 */
enum odevio_activation { ODEVIO_ACT_TANH = 0, ODEVIO_ACT_RELU = 1, ODEVIO_ACT_LEAKY_RELU = 2, ODEVIO_ACT_SOFTPLUS = 3 };
/* reference src/models/PoseODERNN.py:125-137 (first four); RK4 variants are BASELINE extensions */
enum odevio_solver { ODEVIO_DOPRI5 = 0, ODEVIO_HEUN = 1, ODEVIO_TSIT5 = 2, ODEVIO_EULER = 3, ODEVIO_RK4 = 4, ODEVIO_RK4_CLASSIC = 5 };
/* reference src/models/PoseODERNN.py:139-148 */
enum odevio_rnn { ODEVIO_RNN_TANH = 0, ODEVIO_RNN_GRU = 1 };
/* reference src/models/FusionModule.py:17-29.  "hard" is stochastic (F.gumbel_softmax(..., hard=True) under torch's generator):
 * the device path draws the same distribution from its own counter-based generator (odevio_set_seed): no bit-level parity
 * with the reference is possible, the distribution and the arithmetic around the mask are tested. */
enum odevio_fuse { ODEVIO_FUSE_CAT = 0, ODEVIO_FUSE_SOFT = 1, ODEVIO_FUSE_HARD = 2 };
/* reference src/models/DeepVIO.py:45-59 */
enum odevio_model { ODEVIO_MODEL_ODE_RNN = 0, ODEVIO_MODEL_RNN = 1, ODEVIO_MODEL_CDE = 2 };

/* Arithmetic of the image encoder (everything else is plain fp32 in every mode).  Only the first two carry the 1e-4 parity
 * claim.  The environment variable ODEVIO_CONV_MATH = f16x2 | f32 | f16 overrides the field (diagnostics). */
enum odevio_arith {
  ODEVIO_ARITH_FP32 = 0,      /* --dtype fp32: fp32 operands as two fp16 pieces, 3 fp16 MFMAs per product, fp32 accumulate */
  ODEVIO_ARITH_FP32_MFMA = 1, /* --dtype fp32_mfma: the fp32-input MFMA (1/16 of the fp16 rate) */
  ODEVIO_ARITH_FP16 = 2       /* --dtype fp16 | bf16: reduced precision, fp16 operands (one piece), fp32 accumulate */
};

/* The hot-path subset of reference scripts/config.py:29,48-51,59,62-65,68-69 + controller constants
 * of reference src/models/PoseODERNN.py:57,72 (torchode IntegralController(atol, rtol), dt0). */
typedef struct odevio_config {
  int32_t struct_size; /* = sizeof(odevio_config), ABI guard */
  int32_t model_type;  /* odevio_model */
  int32_t img_h, img_w;
  int32_t v_f_len, i_f_len;
  int32_t fuse_method;        /* odevio_fuse */
  int32_t ode_hidden_dim;     /* width H of the ODEFunc MLP */
  int32_t ode_fn_num_layers;  /* n: ODEFunc has n+1 Linears */
  int32_t ode_activation;     /* odevio_activation */
  int32_t ode_solver;         /* odevio_solver */
  int32_t ode_substeps;       /* fixed-step solvers: equal sub-steps per interval (>=1) */
  int32_t rnn_type;           /* odevio_rnn */
  int32_t rnn_num_layers;
  float atol, rtol, dt0;      /* 1e-6, 1e-2, 1e-4 in the reference */
  int32_t max_steps;          /* per-interval step budget of the adaptive solvers */
  /* Neural-CDE variant, reference scripts/config.py:74-78 (used when model_type == ODEVIO_MODEL_CDE) */
  int32_t cde_hidden_dim, cde_fn_num_layers, cde_activation, cde_solver;
  /* build extension `--dtype` (not in the reference, which is fp32-only: scripts/train_model.py:63-66): odevio_arith */
  int32_t arith;
} odevio_config;

/* One named weight, keyed exactly like the reference state_dict, fp32 on device. */
typedef struct odevio_tensor {
  const char* name;
  const void* data;
  int64_t numel;
} odevio_tensor;

/* Per-row integrator statistics written by odevio_ode_rnn_fwd / odevio_ode_steps when `stats` != NULL:
 * stats[2*row+0] = steps attempted, stats[2*row+1] = steps accepted (summed over intervals). */

int odevio_version(void);
const char* odevio_last_error(void);
/* Audit build (make -C odevio_amd/csrc AUDIT=1 -> libodevio_audit.so, selected with ODEVIO_LIB=): number of plans /
 * checks of this process that saw an out-of-bounds address computed by a kernel (the access itself is redirected, so
 * nothing faults).  Always 0 in the production library, whose kernels carry no such checks.  No reference counterpart:
 * the reference's PyTorch ops are bounds-safe by construction; this is how the hand-written LDS-DMA kernels are held
 * to the same standard. */
int odevio_audit_violations(void);

/* Builds a plan: validates the config, finds every weight by name, folds BatchNorm running stats
 * into per-channel (scale, shift), re-lays the convolution weights (two fp16 pieces per fp32 weight in the K-tile
 * order of the fp16-MFMA kernels, pre-scaled by a power of two per layer; plus [Cout][kh][kw][Cin] fp32 for the
 * ODEVIO_CONV_MATH=f32 mode), permutes the visual head to the NHWC flatten order and column-shards the ODEFunc/RNN
 * weights for the persistent integrator.  Replaces: DeepVIO.__init__ + load_state_dict (DeepVIO.py:37-43).
 * Environment: ODEVIO_CONV_MATH = f16x2 (default) | f32 (fp32-input MFMA) | f16 (reduced precision: fp16 encoder
 * operands, outside the fp32 parity claim), read here. */
int odevio_plan_create(const odevio_config* cfg, const odevio_tensor* weights, int32_t n_weights, void* stream,
                       odevio_plan** out_plan);
void odevio_plan_destroy(odevio_plan* plan);
/* Pre-allocates the activation workspace for batches up to (B, S) so that later calls allocate nothing. */
int odevio_reserve(odevio_plan* plan, int32_t B, int32_t S, void* stream);
/* Synchronises `stream` and returns the device status words: an integrator timeout / step-budget overflow, or an
 * encoder activation outside the fp16x2 range (ODEVIO_ERR_RANGE).  Clears what it reports.
 * Without this call a failure still surfaces: every forward copies the status words to pinned host memory behind
 * itself (asynchronously, no host synchronisation), and the next entry point that finds that copy complete returns the
 * error of the forward before it. */
int odevio_check(odevio_plan* plan, void* stream);

/* ImageEncoder.forward (Encoder.py:97-122): img [B,S,3,H,W] -> fv [B,S-1,v_f_len] with row stride ld_fv. */
int odevio_image_encoder_fwd(odevio_plan* plan, const float* img, int32_t B, int32_t S, float* fv, int32_t ld_fv,
                             void* stream);
/* One Conv2d+BatchNorm2d+LeakyReLU block of the ImageEncoder (Encoder.py:8-22), for kernel-level
 * parity tests: layer 0 (conv1) reads img [B,S,3,H,W] and writes NHWC [B*(S-1),H/2,W/2,64]; layer i>0 reads the
 * NHWC output of layer i-1 for `pairs` frame pairs and writes its own NHWC output. */
int odevio_conv_block_fwd(odevio_plan* plan, int32_t layer, const float* in, int32_t B, int32_t S, float* out,
                          void* stream);
/* InertialEncoder.forward (Encoder.py:60-74): imu [B,T,6] -> fi [B,(T-1)/10,i_f_len] with row stride ld_fi. */
int odevio_imu_encoder_fwd(odevio_plan* plan, const float* imu, int32_t B, int32_t T, float* fi, int32_t ld_fi,
                           void* stream);
/* FusionModule.forward (FusionModule.py:17-23): fv [P,v], fi [P,i] -> fused [P,v+i]. */
int odevio_fuse_fwd(odevio_plan* plan, const float* fv, const float* fi, int32_t P, float* fused, void* stream);
/* ODEFunc.forward (ODEFunc.py:38-39): y [rows,F] -> f(y) [rows,F]. */
int odevio_ode_func(odevio_plan* plan, const float* y, int32_t rows, float* out, void* stream);
/* PoseODERNN.evolve_state (PoseODERNN.py:70-75): integrate rows from t0[r] to t1[r].
 * solver < 0 uses the plan's; substeps <= 0 uses the plan's. */
int odevio_ode_steps(odevio_plan* plan, const float* y, const float* t0, const float* t1, int32_t rows,
                     int32_t solver, int32_t substeps, float* y_out, int32_t* stats, void* stream);
/* PoseODERNN.forward / PoseRNN.forward (PoseODERNN.py:88-123) AFTER fusion:
 * fused [B,P,F], ts [B,P+1], hc_in NULL or [L,B,F] -> poses [B,P,6], h_T [L,B,F]. */
int odevio_ode_rnn_fwd(odevio_plan* plan, const float* fused, const float* ts, const float* hc_in, int32_t B,
                       int32_t P, float* poses, float* h_T, int32_t* stats, void* stream);
/* PoseCDE.forward after fusion (reference src/models/PoseCDE.py:94-103): obs [B,L,1+F] = [time | fused features] of every
 * observation so far (device), t_out = the output times ts[0,1:] (HOST doubles, n_out of them, strictly ascending),
 * z0_in NULL or a carried [B,H] state -> poses [B,n_out,6], z0_out [B,H] (the reference returns the INITIAL state).
 * cdeint's controller (torchdiffeq dopri5: error norm, accept / reject, step size, knot clipping, dense output) runs on
 * the DEVICE; the host enqueues step attempts ahead and reads the controller's `done` word once per batch of attempts
 * (typically once per call), never per step.  The fixed-grid solvers (euler, rk4) do not synchronise at all.
 * stats_host = {steps, accepted} or NULL. */
int odevio_cde_fwd(odevio_plan* plan, const float* obs, int32_t B, int32_t L, const double* t_out_host, int32_t n_out,
                   const float* z0_in, float* poses, float* z0_out, int32_t* stats_host, void* stream);
/* Backward of odevio_cde_fwd - loss.backward() through PoseCDE.forward with adjoint = False (PoseCDE.py:98-101: autograd through
 * torchcde's cdeint -> torchdiffeq's odeint): the solve is run once more with a tape of its accepted steps and swept in reverse
 * (discretise-then-optimise; step sizes are constants of the differentiation; dense-output interpolation, knot re-evaluations and
 * the three solvers included).  torchcde / torchdiffeq are absent offline: checked against autograd through the oracle's
 * restatement, parity with the real libraries UNPINNED.  grad_poses [B,n_out,6]; grad_z0_out (or NULL): gradient of the returned z0;
 * grad_obs [B,L,1+F] out (channel 0 = the time channel); grad_z0_in out when z0_in is given.  grads: "Pose_net.cde_func.net.<2l>.weight /
 * .bias", "Pose_net.initial.0.*", "Pose_net.regressor.{0,2}.*".  One host read (the tape's step records).  fp32 plans only. */
int odevio_cde_bwd(odevio_plan* plan, const float* obs, int32_t B, int32_t L, const double* t_out_host, int32_t n_out, const float* z0_in,
                   const float* grad_poses, const float* grad_z0_out, float* grad_obs, float* grad_z0_in, const odevio_tensor* grads,
                   int32_t n_grads, int32_t* stats_host, void* stream);
/* Backward of odevio_ode_rnn_fwd: what `loss.backward()` reaches below the encoders in the reference's training step
 * (scripts/train_model.py:69-78; autograd through torchode's AutoDiffAdjoint = backpropagation through the solver's own
 * operations, "discretise-then-optimise").  Inputs as in the forward plus grad_poses [B,P,6] and grad_hT [L,B,F] or NULL;
 * outputs grad_fused [B,P,F] (or NULL), grad_hc [L,B,F] (or NULL; needs hc_in) and the weight gradients listed in
 * `grads`: name = the reference state_dict key (Pose_net.ode_func.net.{0,2,..}.{weight,bias},
 * Pose_net.rnn.{weight_ih,weight_hh,bias_ih,bias_hh}_l{k}, Pose_net.regressor.{0,2}.{weight,bias}), data = DEVICE pointer
 * the gradient is WRITTEN to (same shape as the parameter), numel checked.  Nothing is kept from odevio_ode_rnn_fwd: the
 * forward runs once more on the persistent kernel, logging per row and interval every ACCEPTED step (its size and the state it
 * starts from) and the evolved state; from that log every stage of every step is rebuilt in one batch, then swept in reverse.
 * Fixed-step solvers (rk4, rk4_classic, any ode_substeps) are differentiated as they stand; for the adaptive ones (dopri5,
 * tsit5, heun, euler) the accepted steps are replayed with their sizes held constant (one host read: the largest step
 * count).  nn.RNN (tanh) and nn.GRU.  ODEVIO_ERR_UNSUPPORTED for the Neural-CDE path (odevio_cde_bwd). */
int odevio_ode_rnn_bwd(odevio_plan* plan, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                       const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc,
                       const odevio_tensor* grads, int32_t n_grads, void* stream);
/* The same pair with the log kept by the CALLER between forward and backward, so that a training step runs the persistent kernel
 * once (what autograd's saved tensors are to `poses = model(...)` ... `loss.backward()`, scripts/train_model.py:69-78):
 *   odevio_ode_rnn_tape_floats  -> *n_floats = size of the device buffer (fp32 elements) a taped forward of (B, P) fills; 0 for
 *                                  plans without an ODE (use the plain pair);
 *   odevio_ode_rnn_fwd_taped    = odevio_ode_rnn_fwd + the log written to `tape`;
 *   odevio_ode_rnn_bwd_taped    = odevio_ode_rnn_bwd reading that log instead of running the forward again (same plan, same
 *                                  inputs, weights unchanged in between).  A log that proved too short for an interval (more than 64
 *                                  accepted steps) is rebuilt inside with more room, like the plain backward does. */
int odevio_ode_rnn_tape_floats(const odevio_plan* plan, int32_t B, int32_t P, int64_t* n_floats);
int odevio_ode_rnn_fwd_taped(odevio_plan* plan, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                             float* poses, float* h_T, float* tape, int64_t tape_floats, void* stream);
int odevio_ode_rnn_bwd_taped(odevio_plan* plan, const float* fused, const float* ts, const float* hc_in, int32_t B, int32_t P,
                             const float* grad_poses, const float* grad_hT, float* grad_fused, float* grad_hc,
                             const odevio_tensor* grads, int32_t n_grads, const float* tape, int64_t tape_floats, void* stream);
/* The reference's training loss and its gradient (scripts/train_model.py:72-77): loss3 (device, 3 floats) =
 * {100 * angle_loss + translation_loss, angle_loss, translation_loss} with MSE over the first / last three pose columns of
 * n_rows = B*P rows; grad_poses [n_rows,6] = d loss3[0] / d poses, or NULL. */
int odevio_pose_loss(const float* poses, const float* gts, int32_t n_rows, float* loss3, float* grad_poses, void* stream);

/* ---- model.train() semantics of the encoders (the reference trains under model.train(), scripts/train_model.py:219: every
 * BatchNorm of both encoders - the frozen Image_net's too - normalises with BATCH statistics and updates its running statistics,
 * every Dropout is on; src/models/Encoder.py:8-22,43-57,82-90).
 * ImageEncoder.forward in train mode: per block conv -> batch statistics over (N,H,W) -> gamma (z - mean) / sqrt(var + eps) + beta ->
 * LeakyReLU(0.1) -> Dropout(0.2; conv6: 0.5), then the visual head.  `stats`: device tensors named like the module's buffers
 * ("Image_net.conv3_1.1.running_mean" / ".running_var", fp32 [Cout]) that are updated IN PLACE as torch does (momentum 0.1, unbiased
 * variance); buffers not listed are left alone (num_batches_tracked is the caller's counter).  Each block's dropout mask is one
 * draw of the plan's random stream (9 draws, conv1 first; odevio_rng_state BEFORE the call gives the first).  fp32 frames only.
 * keep = 1: what odevio_image_encoder_bwd needs stays behind (about 8 bytes per activation element of the nine blocks). */
int odevio_image_encoder_fwd_train(odevio_plan* plan, const float* img, int32_t B, int32_t S, float* fv, int32_t ld_fv,
                                   const odevio_tensor* stats, int32_t n_stats, int32_t keep, void* stream);
/* Backward of the forward above when it ran with keep = 1 (every block's bare convolution, output and batch statistics stay in
 * plan-owned memory until the next train-mode forward): gradients of the Image_net parameters named in `grads`
 * ("Image_net.conv2.0.weight" [Cout,Cin,k,k], "....1.weight" / "....1.bias" = BatchNorm gamma / beta, "Image_net.visual_head.weight" in
 * the reference's (C,H,W) column order, ".bias") from grad_fv [B*(S-1)][ld_gfv] - what loss.backward() leaves on Image_net when
 * --freeze_encoder is off, the gradients that then count in clip_grad_norm_ (scripts/train_model.py:78,84).  Through Dropout (the
 * same Philox masks), LeakyReLU, batch-statistics BatchNorm (torch's batch_norm backward with training = True) and the convolutions
 * (weight gradients: contraction over every pixel on the fp32 MFMA; input gradients: stride-1 convolutions of the zero-dilated
 * gradient with the reversed filters).  All fp32; `img` = the frames of that forward. */
int odevio_image_encoder_bwd(odevio_plan* plan, const float* img, int32_t B, int32_t S, const float* grad_fv, int32_t ld_gfv,
                             const odevio_tensor* grads, int32_t n_grads, void* stream);
/* InertialEncoder.forward in train mode: BatchNorm1d over the (pair, time) rows of the batch, Dropout(p_drop = opt.imu_dropout)
 * after every block; three draws of the random stream (consumed whatever p_drop is); `stats` as above
 * ("Inertial_net.encoder_conv.1.running_mean", ...). */
int odevio_imu_encoder_fwd_train(odevio_plan* plan, const float* imu, int32_t B, int32_t T, float p_drop, const odevio_tensor* stats,
                                 int32_t n_stats, float* fi, int32_t ld_fi, void* stream);
/* Its backward: gradients of every Inertial_net parameter through the batch-statistics BatchNorm (torch's batch_norm backward with
 * training=True) and the dropout masks of draws call0 .. call0+2 of `seed` (what odevio_rng_state returned before the forward). */
int odevio_imu_encoder_bwd_train(odevio_plan* plan, const float* imu, int32_t B, int32_t T, float p_drop, uint64_t seed, uint64_t call0,
                                 const float* grad_fi, const odevio_tensor* grads, int32_t n_grads, void* stream);
/* Test hook: the factor nn.Dropout(p_drop) applies to each of n elements under draw `call` of `seed` (0 or 1 / (1 - p_drop)) -> out [n]
 * (device).  Element order: image encoder NHWC (pixel * C + channel), inertial encoder [pair][channel][time]. */
int odevio_debug_dropout(uint64_t seed, uint64_t call, float p_drop, int64_t n, float* out, void* stream);

/* Seed of the plan's random stream (Philox 4x32-10; fuse_method "hard" draws its Gumbel noise from it, one counter block per
 * call).  The same seed gives the same sequence of masks; plans start at seed 0. */
int odevio_set_seed(odevio_plan* plan, uint64_t seed);

/* State of the plan's random stream: the seed and the number of draws so far.  The call index a "hard" fusion forward will use is
 * `calls` read BEFORE it; odevio_fuse_hard_bwd regenerates the same noise from (seed, call). */
int odevio_rng_state(odevio_plan* plan, uint64_t* seed, uint64_t* calls);
/* Restores a state read with odevio_rng_state - a caller that rebuilds its plan (new weights after load_state_dict or an
 * optimizer step: odevio_amd.DeepVIO._ensure_plan) carries the stream over, so that every forward keeps drawing FRESH noise
 * like the reference's F.gumbel_softmax under torch's generator (FusionModule.py:27) instead of replaying draw 0. */
int odevio_set_rng_state(odevio_plan* plan, uint64_t seed, uint64_t calls);
/* Test hook: the Gumbel(0,1) pair of each of n elements for draw `call` of `seed` -> out [n][2] (device). */
int odevio_debug_gumbel(uint64_t seed, uint64_t call, int64_t n, float* out, void* stream);
/* FusionModule "hard" backward (FusionModule.py:24-29): the straight-through estimator of F.gumbel_softmax(..., hard=True) - the
 * forward value is the one-hot mask, the gradient is y_soft's - for the mask of draw (seed, call).  Gradients of
 * Pose_net.fuse.net.0.weight [2F,F] / .bias [2F] as named in `grads`. */
int odevio_fuse_hard_bwd(odevio_plan* plan, const float* fv, const float* fi, int32_t P, uint64_t seed, uint64_t call,
                         const float* grad_fused, float* grad_fv, float* grad_fi, const odevio_tensor* grads, int32_t n_grads,
                         void* stream);

/* FusionModule backward (reference src/models/FusionModule.py:17-23; autograd in scripts/train_model.py:78): fv [P,v], fi [P,i],
 * grad_fused [P,v+i] -> grad_fv [P,v], grad_fi [P,i] (either may be NULL) and, for fuse_method "soft", the gradients of
 * Pose_net.fuse.net.0.weight / .bias named in `grads` (device pointers, reference shapes).  ("hard" is not a device
 * fusion mode: the host-side wrapper refuses its backward.) */
int odevio_fuse_bwd(odevio_plan* plan, const float* fv, const float* fi, int32_t P, const float* grad_fused, float* grad_fv,
                    float* grad_fi, const odevio_tensor* grads, int32_t n_grads, void* stream);

/* InertialEncoder backward (reference src/models/Encoder.py:41-74, BatchNorm in eval mode like odevio_imu_encoder_fwd, imu_dropout 0):
 * imu [B,T,6], grad_fi [B*(T-1)/10, i_f_len] contiguous -> the gradients of the Inertial_net parameters named in `grads`
 * (encoder_conv.{0,4,8}.weight/bias, encoder_conv.{1,5,9}.weight/bias, proj.weight/bias; device pointers, reference shapes).
 * In the reference's recipe (--freeze_encoder freezes Image_net only) these gradients enter the step through
 * clip_grad_norm_(model.parameters()); its optimizer does not hold them (utils/utils.py:116-119). */
int odevio_imu_encoder_bwd(odevio_plan* plan, const float* imu, int32_t B, int32_t T, const float* grad_fi, const odevio_tensor* grads,
                           int32_t n_grads, void* stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) as the reference calls it before optimizer.step()
 * (scripts/train_model.py:84): the total L2 norm over the n gradient tensors and the factor min(1, max_norm / (norm + 1e-6)),
 * written to the device pair norm_coef = {norm, factor}.  Nothing is scaled here and nothing returns to the host:
 * odevio_adam_step multiplies by the factor. */
int odevio_grad_clip(odevio_plan* plan, const odevio_tensor* grads, int32_t n_grads, float max_norm, float* norm_coef, void* stream);

/* One torch.optim.Adam update of one parameter tensor, the optimizer the reference builds over Pose_net's parameters
 * (utils/utils.py:115-130: betas (0.9, 0.999), eps 1e-8, weight_decay added to the gradient, amsgrad off):
 * param, exp_avg, exp_avg_sq are updated in place; `step` counts from 1; norm_coef = the pair of odevio_grad_clip or NULL. */
int odevio_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int32_t step, const float* norm_coef, void* stream);
/* One torch.optim.SGD update of one tensor - the reference's other optimizer (utils/utils.py:120-121: SGD(param_groups, lr=1e-4,
 * momentum=0.9), each group's own lr = lr_warmup): g = clip * grad + weight_decay * p; buf = g at step 1, else momentum * buf + g;
 * p -= lr * buf.  norm_coef: the device pair written by odevio_grad_clip, or NULL. */
int odevio_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t numel, float lr, float momentum, float weight_decay,
                    int32_t step, const float* norm_coef, void* stream);
/* The same updates for a whole list of tensors in ONE call and one launch (what a training step does: the reference's optimizer.step()
 * over Pose_net's ~20 parameter tensors, utils/utils.py:115-130): kind 0 = torch.optim.Adam (state1 = exp_avg, state2 = exp_avg_sq, beta1 /
 * beta2), kind 1 = torch.optim.SGD (state1 = momentum buffers or NULL, beta1 = momentum; beta2 / eps unused).  params / grads / state*: n
 * device tensors each (odevio_tensor.data, .numel; sizes must agree), lrs: n HOST floats (the reference's two learning-rate groups),
 * step >= 1 (1-based, shared), norm_coef: the device pair written by odevio_grad_clip, or NULL. */
int odevio_optimizer_step(int32_t kind, const odevio_tensor* params, const odevio_tensor* grads, const odevio_tensor* state1,
                          const odevio_tensor* state2, const float* lrs, int32_t n, float beta1, float beta2, float eps, float weight_decay,
                          int32_t step, const float* norm_coef, void* stream);

/* After an optimizer step: re-reads the parameters of Pose_net (fusion, regressor, ODEFunc, RNN; every key of the
 * reference's Pose_net state_dict, device pointers) into the plan's kernel layouts, in place.  The encoders are not
 * touched: the reference's optimizer does not hold their parameters either (utils/utils.py:116-119). */
int odevio_plan_update(odevio_plan* plan, const odevio_tensor* weights, int32_t n_weights, void* stream);

/* The Neural-CDE vector field for one piece of the control path (CDEFunc.forward, reference src/models/ODEFunc.py:76-83,
 * contracted with dX/dt as torchcde's cdeint does): z [B,H], obs [B,L,1+F], seg = piece 0 .. 2L-3 of the rectilinear
 * path (even: the time channel moves, odd: the features) -> out [B,H] = reshape(CDEFunc(z), [B,H,H+1]) . dX/dt(seg).
 * The unit the adaptive solver calls 6 times per step; bench.py times it for the HBM roofline of the weight stream. */
int odevio_cde_func(odevio_plan* plan, const float* z, const float* obs, int32_t B, int32_t L, int32_t seg, float* out,
                    void* stream);

/* Measurement only: duration (ms, HIP events on the launch stream) of the LAST LAYER of the most recent odevio_cde_func
 * call - on an odd piece that is the weight-stream kernel alone - while the stage timers are on (odevio_profile_enable).
 * Waits for that call.  bench.py --model cde prices it against the HBM roofline. */
int odevio_cde_last_ms(odevio_plan* plan, float* ms_out);

/* DeepVIO.forward (DeepVIO.py:61-68): img [B,S,3,H,W], imu [B,T,6], ts [B,S], hc NULL or [L,B,F]
 * -> poses [B,S-1,6], h_T [L,B,F].  Asynchronous on `stream`; the inertial encoder runs on a stream owned by the plan,
 * forked from and joined back into `stream` with events (nothing for the caller to do). */
int odevio_forward(odevio_plan* plan, const float* img, const float* imu, int32_t T, const float* ts,
                   const float* hc, int32_t B, int32_t S, float* poses, float* h_T, int32_t* stats, void* stream);

/* Trajectory accumulation for the streaming evaluator: path_accu / pose_accu / pose_6DoF_to_matrix of the reference
 * (src/data/utils.py:93-161, called from kitti_eval, src/data/KITTI_eval.py:231-232).  No plan needed; every pointer is
 * a DEVICE pointer.  poses6 = relative poses [N,6] (angles x,y,z then translation) of n_drives drives back to back,
 * float32 (is_f64 = 0) or float64; offsets[n_drives+1] = first pose of every drive (int64); carry = NULL (start every
 * drive at the identity) or [n_drives,4,4] float64 row-major start poses (the last matrix of the previous window);
 * mats = [(N + n_drives),4,4] float64: drive d writes offsets[d+1]-offsets[d]+1 matrices starting at row
 * offsets[d]+d, the first being its start pose.  A float32 input has its per-pose rotation built in float32, as
 * numpy does for the reference; the running product is float64. */
int odevio_path_accu(const void* poses6, int32_t is_f64, const int64_t* offsets, int32_t n_drives, const double* carry,
                     double* mats, void* stream);

/* The same forward from the loader's uint8 frames (reference src/data/KITTI_eval.py:97-110, src/data/utils.py:355-374:
 * PIL image -> resize -> ToTensor() - 0.5): img_u8 [B,S,H,W,3] uint8 (HWC, already resized to img_h x img_w); the
 * normalisation float(byte) / 255 - 0.5 is fused into the encoder's ingest pass (69 MB instead of 277 MB at B = 16).
 * Model types ode-rnn / rnn; needs the default fp16x2 encoder. */
int odevio_forward_u8(odevio_plan* plan, const uint8_t* img_u8, const float* imu, int32_t T, const float* ts,
                      const float* hc, int32_t B, int32_t S, float* poses, float* h_T, int32_t* stats, void* stream);

/* The loader's frame resize on the device (reference src/data/KITTI_eval.py:101, src/data/utils.py:366-371:
 * torchvision TF.resize of a PIL image = PIL.Image.resize(size, BILINEAR)): src [n,Hin,Win,3] uint8 HWC ->
 * dst [n,Hout,Wout,3] uint8, BIT-IDENTICAL to Pillow (8-bit fixed-point triangle filter with antialiasing when shrinking,
 * horizontal then vertical pass).  tmp = n*Hin*Wout*3 bytes of device scratch (may be NULL when only one dimension
 * changes).  No plan needed.  The output feeds odevio_forward_u8 directly. */
int odevio_resize_u8(const uint8_t* src, int32_t n, int32_t Hin, int32_t Win, uint8_t* dst, int32_t Hout, int32_t Wout,
                     uint8_t* tmp, void* stream);

/* Host-only (no GPU needed): the fixed-point coefficient table odevio_resize_u8 uses for one axis - Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter.  bounds [out_size][2] = (first input index, count),
 * kk [out_size][*ksize]; kk_capacity = ints available in kk (>= out_size * (2 * ceil(max(in/out, 1)) + 1)).
 * Test hook: compared with the oracle on the CPU and run under AddressSanitizer (make ASAN=1). */
int odevio_resize_table(int32_t in_size, int32_t out_size, int32_t* ksize, int32_t* bounds, int32_t* kk, int32_t kk_capacity);

/* Per-stage timing of odevio_forward with HIP events recorded on the caller's stream (used by bench.py for
 * the roofline figures).  Stages: 0 conv1, 1 conv2..conv6 (implicit-GEMM kernel), 2 visual head,
 * 3 inertial encoder + fusion, 4 persistent ODE-RNN integrator, 5 pose regressor. */
#define ODEVIO_N_STAGES 6
/* on = 0 switches the timing off; on = d > 0 keeps a ring of d event sets, so up to d forwards can be issued back to
 * back before the timers are read (reading them must not put a host synchronisation between the forwards). */
int odevio_profile_enable(odevio_plan* plan, int32_t on);
/* Waits for the last recorded forward and writes ODEVIO_N_STAGES durations in milliseconds, averaged over the forwards
 * recorded since the previous read (at most the ring depth). */
int odevio_profile_read(odevio_plan* plan, float* ms_out);

/* Diagnostic build only (make STAMPS=1 -> libodevio_stamps.so): in-kernel phase totals of the last integrator
 * launch, in shader clocks: [0] kernel, [1] waiting in all-gathers, [2] ODEFunc layer products, [3] RNN layer
 * products, [4] number of all-gathers, [5] per-group placement bits, [6] post-product barriers, [7] owner epilogues, [8] launch
 * to first interval, [9] vector-field evaluations, [10] error norm + controller, [11] RNN phases.  out12: TWELVE words.  The
 * production library leaves them at zero. */
int odevio_debug_stamps(odevio_plan* plan, uint64_t* out12, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ODEVIO_H */
