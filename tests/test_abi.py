"""The C-ABI library loads and exports every symbol include/odevio.h declares (no compute: CPU only)."""
import ctypes
import os
import re

from odevio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "odevio.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(odevio_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for s in header_symbols():
        assert hasattr(lib, s), s
    assert lib.odevio_version() == 1


def test_config_struct_layout_matches_header():
    # 20 int32 + 3 float fields, no padding: the ABI guard the library checks via struct_size
    assert ctypes.sizeof(_lib.OdevioConfig) == 23 * 4
    src = open(os.path.join(ROOT, "include", "odevio.h")).read()
    body = src[src.index("typedef struct odevio_config {"):src.index("} odevio_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int32_t|float)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [f[0] for f in _lib.OdevioConfig._fields_]


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = _lib.load()
    plan = ctypes.c_void_p()
    rc = lib.odevio_plan_create(None, None, 0, None, ctypes.byref(plan))
    assert rc == _lib.ERR_BAD_ARG and b"null" in lib.odevio_last_error()
    cfg = _lib.OdevioConfig()
    cfg.struct_size = 4  # wrong ABI size
    arr = (_lib.OdevioTensor * 1)()
    rc = lib.odevio_plan_create(ctypes.byref(cfg), arr, 1, None, ctypes.byref(plan))
    assert rc == _lib.ERR_BAD_ARG and b"size mismatch" in lib.odevio_last_error()


def test_training_entry_points_reject_bad_arguments_without_a_gpu():
    """The optimizer-step entry points check their arguments before they touch the device."""
    lib = _lib.load()
    f = ctypes.c_float
    one = ctypes.c_void_p(16)   # never dereferenced: the checks below fail first
    assert lib.odevio_adam_step(None, None, None, None, 0, f(1e-4), f(0.9), f(0.999), f(1e-8), f(0.0), 1, None, None) == _lib.ERR_BAD_ARG
    assert lib.odevio_adam_step(one, one, one, one, 8, f(1e-4), f(0.9), f(0.999), f(1e-8), f(0.0), 0, None, None) == _lib.ERR_BAD_ARG   # step counts from 1
    assert lib.odevio_adam_step(one, one, one, one, 8, f(1e-4), f(1.0), f(0.999), f(1e-8), f(0.0), 1, None, None) == _lib.ERR_BAD_ARG   # beta1 < 1
    assert b"odevio_adam_step" in lib.odevio_last_error()
    assert lib.odevio_grad_clip(None, None, 0, f(5.0), None, None) == _lib.ERR_BAD_ARG
    assert lib.odevio_plan_update(None, None, 0, None) == _lib.ERR_BAD_ARG
    assert lib.odevio_fuse_bwd(None, None, None, 0, None, None, None, None, 0, None) == _lib.ERR_BAD_ARG
    assert lib.odevio_imu_encoder_bwd(None, None, 0, 0, None, None, 0, None) == _lib.ERR_BAD_ARG
