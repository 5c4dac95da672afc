"""AddressSanitizer run of the host side of the C ABI (CPU container only: GPU sanitizers are not available on the pool).
Builds libodevio_asan.so (host code instrumented, device code as usual) and tests/asan_driver.c, runs the driver."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_host_side_of_the_abi_under_asan(tmp_path):
    csrc = os.path.join(ROOT, "odevio_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "ASAN=1", "-j4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "odevio_amd", "libodevio_asan.so")
    exe = str(tmp_path / "asan_driver")
    clang = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin", "clang")   # the compiler hipcc drives: same ASan runtime
    r = subprocess.run([clang, os.path.join(ROOT, "tests", "asan_driver.c"), "-fsanitize=address", "-shared-libsan", "-g",
                        "-o", exe, lib, f"-Wl,-rpath,{os.path.dirname(lib)}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    # the sanitizer runtime clang links against
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if rt and os.path.exists(rt):
        env["LD_LIBRARY_PATH"] = os.path.dirname(rt) + ":" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and "asan driver: ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr
