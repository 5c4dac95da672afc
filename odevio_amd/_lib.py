"""ctypes binding of libodevio.so (include/odevio.h).  There is NO fallback: if the library is
missing or a symbol is absent, importing the product path raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODEVIO_LIB", os.path.join(_HERE, "libodevio.so"))  # ODEVIO_LIB: diagnostic builds

ODEVIO_OK = 0
ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_MISSING_WEIGHT, ERR_HIP, ERR_NO_DEVICE, ERR_TIMEOUT, ERR_MAX_STEPS, ERR_RANGE, ERR_BOUNDS = range(-1, -10, -1)

ACTIVATIONS = {"tanh": 0, "relu": 1, "leaky_relu": 2, "softplus": 3}
SOLVERS = {"dopri5": 0, "heun": 1, "tsit5": 2, "euler": 3, "rk4": 4, "runge_kutta": 4, "rk4_classic": 5}
RNN_TYPES = {"rnn": 0, "gru": 1}
FUSE_METHODS = {"cat": 0, "soft": 1, "hard": 2}
MODEL_TYPES = {"ode-rnn": 0, "rnn": 1, "cde": 2}
# --dtype -> odevio_arith: fp32 = fp32-grade products on the fp16 MFMA (two-piece operands), fp32_mfma = the fp32-input MFMA,
# fp16 / bf16 = the reduced-precision encoder (its operand type on gfx950 is fp16: same MFMA rate as bf16, 3 more bits)
DTYPES = {"fp32": 0, "fp32_mfma": 1, "fp16": 2, "bf16": 2}

# every symbol include/odevio.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "odevio_version", "odevio_last_error", "odevio_plan_create", "odevio_plan_destroy", "odevio_reserve",
    "odevio_check", "odevio_conv_block_fwd", "odevio_image_encoder_fwd", "odevio_imu_encoder_fwd", "odevio_fuse_fwd", "odevio_ode_func",
    "odevio_ode_steps", "odevio_ode_rnn_fwd", "odevio_cde_fwd", "odevio_forward", "odevio_profile_enable", "odevio_profile_read", "odevio_debug_stamps",
    "odevio_path_accu", "odevio_forward_u8", "odevio_audit_violations", "odevio_cde_func", "odevio_cde_last_ms",
    "odevio_ode_rnn_bwd", "odevio_pose_loss", "odevio_resize_u8", "odevio_resize_table",
    "odevio_fuse_bwd", "odevio_grad_clip", "odevio_adam_step", "odevio_plan_update", "odevio_imu_encoder_bwd", "odevio_set_seed",
    "odevio_rng_state", "odevio_debug_gumbel", "odevio_fuse_hard_bwd", "odevio_set_rng_state",
    "odevio_image_encoder_fwd_train", "odevio_imu_encoder_fwd_train", "odevio_imu_encoder_bwd_train", "odevio_debug_dropout",
    "odevio_sgd_step", "odevio_image_encoder_bwd", "odevio_cde_bwd",
    "odevio_ode_rnn_tape_floats", "odevio_ode_rnn_fwd_taped", "odevio_ode_rnn_bwd_taped", "odevio_optimizer_step",
]


def source_sha():
    """sha256 (first 16 hex digits) of the library's sources (csrc/*.hip, *.h, include/odevio.h): profiles are stamped with
    it, and bench.py only quotes a committed PMC figure when the stamp matches the sources it runs."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "odevio.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


class OdevioConfig(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("model_type", ctypes.c_int32),
        ("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32),
        ("v_f_len", ctypes.c_int32), ("i_f_len", ctypes.c_int32),
        ("fuse_method", ctypes.c_int32),
        ("ode_hidden_dim", ctypes.c_int32), ("ode_fn_num_layers", ctypes.c_int32),
        ("ode_activation", ctypes.c_int32), ("ode_solver", ctypes.c_int32), ("ode_substeps", ctypes.c_int32),
        ("rnn_type", ctypes.c_int32), ("rnn_num_layers", ctypes.c_int32),
        ("atol", ctypes.c_float), ("rtol", ctypes.c_float), ("dt0", ctypes.c_float),
        ("max_steps", ctypes.c_int32),
        ("cde_hidden_dim", ctypes.c_int32), ("cde_fn_num_layers", ctypes.c_int32),
        ("cde_activation", ctypes.c_int32), ("cde_solver", ctypes.c_int32),
        ("arith", ctypes.c_int32),
    ]


class OdevioTensor(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("data", ctypes.c_void_p), ("numel", ctypes.c_int64)]


class OdevioError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libodevio error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libodevio.so once; raise (never fall back) if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C odevio_amd/csrc`). The HIP library is the product; there is no CPU fallback.")
    import torch  # noqa: F401  - load PyTorch's HIP runtime first so that libodevio.so binds to the same libamdhip64
    lib = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise ImportError(f"libodevio.so does not export {s}")
    vp, i32, fp = ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p
    lib.odevio_version.restype = ctypes.c_int
    lib.odevio_audit_violations.restype = ctypes.c_int
    lib.odevio_last_error.restype = ctypes.c_char_p
    lib.odevio_plan_create.argtypes = [ctypes.POINTER(OdevioConfig), ctypes.POINTER(OdevioTensor), i32, vp,
                                       ctypes.POINTER(vp)]
    lib.odevio_plan_destroy.argtypes = [vp]
    lib.odevio_plan_destroy.restype = None
    lib.odevio_reserve.argtypes = [vp, i32, i32, vp]
    lib.odevio_check.argtypes = [vp, vp]
    lib.odevio_conv_block_fwd.argtypes = [vp, i32, fp, i32, i32, fp, vp]
    lib.odevio_image_encoder_fwd.argtypes = [vp, fp, i32, i32, fp, i32, vp]
    lib.odevio_imu_encoder_fwd.argtypes = [vp, fp, i32, i32, fp, i32, vp]
    lib.odevio_fuse_fwd.argtypes = [vp, fp, fp, i32, fp, vp]
    lib.odevio_ode_func.argtypes = [vp, fp, i32, fp, vp]
    lib.odevio_ode_steps.argtypes = [vp, fp, fp, fp, i32, i32, i32, fp, vp, vp]
    lib.odevio_ode_rnn_fwd.argtypes = [vp, fp, fp, fp, i32, i32, fp, fp, vp, vp]
    lib.odevio_cde_fwd.argtypes = [vp, fp, i32, i32, vp, i32, fp, fp, fp, vp, vp]
    lib.odevio_cde_last_ms.argtypes = [vp, fp]
    lib.odevio_cde_bwd.argtypes = [vp, fp, i32, i32, vp, i32, fp, fp, fp, fp, fp, ctypes.POINTER(OdevioTensor), i32, vp, vp]
    lib.odevio_ode_rnn_bwd.argtypes = [vp, fp, fp, fp, i32, i32, fp, fp, fp, fp, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_ode_rnn_tape_floats.argtypes = [vp, i32, i32, ctypes.POINTER(ctypes.c_int64)]
    lib.odevio_ode_rnn_fwd_taped.argtypes = [vp, fp, fp, fp, i32, i32, fp, fp, fp, ctypes.c_int64, vp]
    lib.odevio_ode_rnn_bwd_taped.argtypes = [vp, fp, fp, fp, i32, i32, fp, fp, fp, fp, ctypes.POINTER(OdevioTensor), i32, fp, ctypes.c_int64, vp]
    lib.odevio_pose_loss.argtypes = [fp, fp, i32, fp, fp, vp]
    f32 = ctypes.c_float
    lib.odevio_fuse_bwd.argtypes = [vp, fp, fp, i32, fp, fp, fp, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_grad_clip.argtypes = [vp, ctypes.POINTER(OdevioTensor), i32, f32, fp, vp]
    lib.odevio_adam_step.argtypes = [fp, fp, fp, fp, ctypes.c_int64, f32, f32, f32, f32, f32, i32, fp, vp]
    lib.odevio_sgd_step.argtypes = [fp, fp, fp, ctypes.c_int64, f32, f32, f32, i32, fp, vp]
    tp = ctypes.POINTER(OdevioTensor)
    lib.odevio_optimizer_step.argtypes = [i32, tp, tp, tp, tp, ctypes.POINTER(f32), i32, f32, f32, f32, f32, i32, fp, vp]
    lib.odevio_plan_update.argtypes = [vp, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_imu_encoder_bwd.argtypes = [vp, fp, i32, i32, fp, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_set_seed.argtypes = [vp, ctypes.c_uint64]
    u64 = ctypes.c_uint64
    lib.odevio_rng_state.argtypes = [vp, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.odevio_set_rng_state.argtypes = [vp, u64, u64]
    lib.odevio_image_encoder_fwd_train.argtypes = [vp, fp, i32, i32, fp, i32, ctypes.POINTER(OdevioTensor), i32, i32, vp]
    lib.odevio_image_encoder_bwd.argtypes = [vp, fp, i32, i32, fp, i32, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_imu_encoder_fwd_train.argtypes = [vp, fp, i32, i32, f32, ctypes.POINTER(OdevioTensor), i32, fp, i32, vp]
    lib.odevio_imu_encoder_bwd_train.argtypes = [vp, fp, i32, i32, f32, u64, u64, fp, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_debug_dropout.argtypes = [u64, u64, f32, ctypes.c_int64, fp, vp]
    lib.odevio_debug_gumbel.argtypes = [u64, u64, ctypes.c_int64, fp, vp]
    lib.odevio_fuse_hard_bwd.argtypes = [vp, fp, fp, i32, u64, u64, fp, fp, fp, ctypes.POINTER(OdevioTensor), i32, vp]
    lib.odevio_resize_table.argtypes = [i32, i32, vp, vp, vp, i32]
    lib.odevio_resize_u8.argtypes = [vp, i32, i32, i32, vp, i32, i32, vp, vp]
    lib.odevio_cde_func.argtypes = [vp, fp, fp, i32, i32, i32, fp, vp]
    lib.odevio_forward.argtypes = [vp, fp, fp, i32, fp, fp, i32, i32, fp, fp, vp, vp]
    lib.odevio_forward_u8.argtypes = [vp, fp, fp, i32, fp, fp, i32, i32, fp, fp, vp, vp]
    lib.odevio_profile_enable.argtypes = [vp, i32]
    lib.odevio_profile_read.argtypes = [vp, fp]
    lib.odevio_debug_stamps.argtypes = [vp, fp, vp]
    lib.odevio_path_accu.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    for s in SYMBOLS[2:]:
        if s != "odevio_plan_destroy":
            getattr(lib, s).restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != ODEVIO_OK:
        msg = load().odevio_last_error().decode("utf-8", "replace")
        if rc in (ERR_BAD_ARG, ERR_UNSUPPORTED):
            raise ValueError(msg)  # the reference raises ValueError for unsupported options
        raise OdevioError(rc, msg)
