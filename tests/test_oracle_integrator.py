"""Known-answer tests for the oracle's integrator.  CPU only.

torchode / torchdiffeq are not installable offline, so ``evolve_state`` is UNPINNED against the
real libraries; these tests pin it against independent truths instead:
 (i)   SciPy RK45 = the same Dormand-Prince pair with the same (0.9, 0.2, 10, -1/5) controller,
 (ii)  linear ODE vs the matrix exponential,
 (iii) empirical order of convergence of every tableau,
 (iv)  rows are independent (own t, dt, accept): solving rows together == solving them alone.
"""
import numpy as np
import pytest
import scipy.integrate
import scipy.linalg
import torch

from odevio_amd import weights
from odevio_amd.config import default_opt
from oracle import odevio_oracle as oc


def _field(seed=0, F=768, H=512, n=3, act="tanh", dtype=torch.float64):
    opt = default_opt(v_f_len=F - 256 if F > 256 else F // 2, i_f_len=256 if F > 256 else F - F // 2,
                      ode_hidden_dim=H, ode_fn_num_layers=n, ode_activation_fn=act)
    sd = oc._sd(weights.make_state_dict(opt, seed=seed), dtype)
    return lambda y: oc.mlp_tanh_out(sd, "Pose_net.ode_func.net", n, y, act)


def test_dopri5_matches_scipy_rk45_step_sequence():
    f = _field()
    g = torch.Generator().manual_seed(1)
    y0 = torch.randn(3, 768, generator=g, dtype=torch.float64) * 0.5
    t0 = torch.zeros(3, dtype=torch.float64)
    t1 = torch.tensor([0.1, 0.3, 0.5], dtype=torch.float64)
    tr = {}
    y = oc.evolve_state(f, y0, t0, t1, "dopri5", trace=tr)
    for r in range(3):
        sol = scipy.integrate.solve_ivp(
            lambda t, v: f(torch.from_numpy(v)[None])[0].numpy(), (0.0, float(t1[r])), y0[r].numpy(),
            method="RK45", first_step=oc.DT0, rtol=oc.RTOL, atol=oc.ATOL)
        ours = [d for d, acc in tr["dts"][r] if acc]
        theirs = np.diff(sol.t)
        rejected = [d for d, acc in tr["dts"][r] if not acc]
        if not rejected:  # SciPy caps growth after a rejection (not part of the I-controller spec)
            assert len(ours) == len(theirs)
            np.testing.assert_allclose(ours, theirs, rtol=1e-9)
            np.testing.assert_allclose(y[r].numpy(), sol.y[:, -1], rtol=1e-9, atol=1e-12)
    # the 0.1 s interval takes the 4-step 1e-4, 1e-3, 1e-2, rest sequence
    seq = [d for d, _ in tr["dts"][0]]
    np.testing.assert_allclose(seq[:3], [1e-4, 1e-3, 1e-2], rtol=1e-12)
    assert len(seq) == 4 and abs(sum(seq) - 0.1) < 1e-12


@pytest.mark.parametrize("method,tol", [("dopri5", 2e-3), ("tsit5", 2e-3), ("heun", 5e-2),
                                        ("rk4", 1e-6), ("rk4_classic", 1e-6)])
def test_linear_ode_vs_expm(method, tol):
    g = torch.Generator().manual_seed(2)
    A = torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.8
    y0 = torch.randn(4, 6, generator=g, dtype=torch.float64)
    T = torch.tensor([0.1, 0.2, 0.35, 0.5], dtype=torch.float64)
    y = oc.evolve_state(lambda v: v @ A.T, y0, torch.zeros(4, dtype=torch.float64), T, method, substeps=8)
    for r in range(4):
        ref = scipy.linalg.expm(A.numpy() * float(T[r])) @ y0[r].numpy()
        assert np.abs(y[r].numpy() - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("tab,order", [(oc.RK4_38, 4), (oc.RK4_CLASSIC, 4), (oc.DOPRI5, 5), (oc.TSIT5, 5),
                                       (oc.HEUN, 2), (oc.EULER, 1)])
def test_convergence_order(tab, order):
    f = _field(F=64, H=32, n=2)
    g = torch.Generator().manual_seed(3)
    y0 = torch.randn(2, 64, generator=g, dtype=torch.float64)

    def solve(nsteps):
        y, h = y0.clone(), torch.full((2,), 0.8 / nsteps, dtype=torch.float64)
        for _ in range(nsteps):
            y, _, _ = oc.rk_stages(f, tab, y, h)
        return y

    ref = solve(2048) if order < 4 else None
    if ref is None:
        yy, h = y0.clone(), torch.full((2,), 0.8 / 512, dtype=torch.float64)
        for _ in range(512):
            yy, _, _ = oc.rk_stages(f, oc.DOPRI5, yy, h)
        ref = yy
    e1 = (solve(4) - ref).abs().max().item()
    e2 = (solve(8) - ref).abs().max().item()
    rate = np.log2(e1 / e2)
    assert rate > order - 0.5, (rate, e1, e2)


def test_embedded_error_orders():
    # the embedded estimate of an order-p pair shrinks like h^p (dopri5/tsit5: 5, heun: 2)
    f = _field(F=64, H=32, n=2)
    y0 = torch.randn(1, 64, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    for tab, p in ((oc.DOPRI5, 5), (oc.TSIT5, 5), (oc.HEUN, 2)):
        e = [oc.rk_stages(f, tab, y0, torch.tensor([h], dtype=torch.float64))[1].abs().max().item()
             for h in (0.2, 0.1)]
        assert np.log2(e[0] / e[1]) > p - 0.6


@pytest.mark.parametrize("method", ["dopri5", "tsit5", "heun"])
def test_rows_are_independent(method):
    f = _field(F=64, H=32, n=2, dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    y0 = torch.randn(5, 64, generator=g)
    t0 = torch.tensor([0.0, 1.0, 2.5, 100.0, 0.3])
    t1 = t0 + torch.tensor([0.1, 0.2, 0.1, 0.5, 0.3])
    tr = {}
    together = oc.evolve_state(f, y0, t0, t1, method, trace=tr)
    for r in range(5):
        tr1 = {}
        alone = oc.evolve_state(f, y0[r:r + 1], t0[r:r + 1], t1[r:r + 1], method, trace=tr1)
        # same step sequence; values equal up to BLAS blocking differences between batch sizes
        assert int(tr1["n_steps"][0]) == int(tr["n_steps"][r])
        assert int(tr1["n_accepted"][0]) == int(tr["n_accepted"][r])
        assert oc.rel_err(alone[0], together[r]) < 1e-5
    assert tr["n_steps"].min() >= 2


def test_euler_keeps_dt0_until_clipped():
    # no embedded estimate -> the controller accepts everything and never changes dt
    tr = {}
    oc.evolve_state(lambda y: -y, torch.ones(1, 4), torch.zeros(1), torch.tensor([0.00105]), "euler", trace=tr)
    seq = [d for d, _ in tr["dts"][0]]
    assert len(seq) == 11 and abs(seq[0] - 1e-4) < 1e-10 and seq[-1] < 0.6e-4
