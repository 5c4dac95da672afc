"""bench.py started the way the driver starts it for N > 1 without a launcher (`python bench.py --gpus N`): it must
hand over to `torch.distributed.run` children BEFORE touching the GPU, and refuse a WORLD_SIZE / --gpus mismatch."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plain_multi_gpu_start_spawns_a_launcher(monkeypatch):
    bench = _load_bench()
    calls = []

    class Done:
        returncode = 0

    def fake_run(cmd, **kw):
        calls.append(cmd)
        return Done()

    import torch
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: pytest.fail("the parent must not touch the GPU"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd = calls[0]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_world_size_mismatch_is_refused(monkeypatch):
    bench = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code not in (0, None)


@pytest.mark.parametrize("world", [1, 2])
def test_launcher_to_rank0_line_end_to_end_over_gloo(world):
    """`python bench.py --gpus N --stub-model`: the REAL launcher path (self-started torch.distributed.run children, rank
    environment, barrier-bracketed step loop, pose all-gather, max-over-ranks clock, rank 0's JSON line) with a stand-in for the
    model on CPU tensors over gloo.  What a future SCALE record is checked by: `ranks.collective_world` (the sum of ones over a
    real all-reduce) must equal n_gpus, every rank reports its own time and device, and `ms_per_step` is the slowest rank's."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--stub-model", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 4 and out["warmup"] == 1 and out["data"] == "stub" and out["scaling"] == "weak"
    rk = out["ranks"]
    assert rk["collective_world"] == world and len(rk["per_rank_ms_per_step"]) == world and len(rk["devices"]) == world
    assert abs(out["ms_per_step"] - max(rk["per_rank_ms_per_step"])) < 1e-3
    assert abs(out["value"] - world * 16 * 11 / (out["ms_per_step"] * 1e-3)) <= 0.01 * out["value"]
    if world > 1:
        assert rk["backend"] == "gloo"
