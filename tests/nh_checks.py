"""Checks of the non-hydrostatic acoustic steps shared by the host-emulation (test_emul_nh.py) and
the MI355X (test_gpu_nh.py) runs: product fv3lm_dyn_core with hydrostatic = 0 against oracle/nh.hpp's dyn_core_nh on the
doubly-periodic tile — tangent-linear outputs, adjoint inputs, and the dot-product identity.
Tolerances: relative L-inf 1e-11 on every trajectory, tangent and adjoint field (BASELINE.md §6 asks 1e-10 after a full step;
measured agreement is 1e-13 .. 1e-14), dot-product identity 1e-11.  All cases run with the library's DEFAULT options, i.e. the
first-order sponge schemes (hord_*_ks_* = 1 for k < n_sponge_pert) switched on, and npz > n_sponge_pert so that sponge and
regular levels are both present."""
import numpy as np
from common import relerr
from oracle import NL, TL, AD
from test_oracle_nh import nh_state

INS = ["u", "v", "pt", "delp", "w", "delz"]
OUTS = ["u", "v", "pt", "delp", "w", "delz", "pe", "peln", "pk", "zh"]


def put(c, T, P=None):
    for n in ("mfx", "mfy", "cx", "cy", "pkz"):      # accumulators / leftovers of an earlier sweep: no adjoint comes in through them here
        c.dy.put(n, np.zeros((1, c.npz, c.ny + 7, c.nx + 7)), 1)
    c.dy.put("ws", np.zeros((1, 1, c.ny + 7, c.nx + 7)), 1)
    for n, t in zip(INS, T):
        c.dy.put(n, t[None], 0)
    if P is not None:
        for n, p in zip(INS, P):
            c.dy.put(n, p[None], 1)


def check_nh_tangent(c, tol_traj=1e-11, tol=1e-11):
    T, P = nh_state(c)
    ot, op = c.oracle.dyn_core_nh(TL, c.dims.dt, c.dims.n_split, T, P)
    put(c, T, P)
    c.dy.dyn_core(TL)
    A = c.rect(1, c.nx, 1, c.ny)
    for n, a, b in zip(OUTS, ot, op):
        e1, e2 = relerr(c.dy.get(n, 0)[0][A], a[A]), relerr(c.dy.get(n, 1)[0][A], b[A])
        assert e1 < tol_traj, (n, "traj", e1)
        assert e2 < tol, (n, "tl", e2)


def check_nh_adjoint(c, tol=1e-11):
    T, P = nh_state(c)
    rng = np.random.default_rng(11)
    A = c.rect(1, c.nx, 1, c.ny)
    seeds = []
    for n in OUTS:
        s = np.zeros((c.dy.levels(n), c.ny + 7, c.nx + 7))
        if n != "zh":                       # the heights are internal to dyn_core: no adjoint comes in through them
            s[A] = rng.standard_normal(s[A].shape)
        seeds.append(s)
    _, iad = c.oracle.dyn_core_nh(AD, c.dims.dt, c.dims.n_split, T, None, seeds)
    put(c, T)
    c.dy.dyn_core(NL)
    for n, s in zip(OUTS, seeds):
        c.dy.put(n, s[None], 1)
    c.dy.dyn_core(AD)
    for n, a in zip(INS, iad):
        e = relerr(c.dy.get(n, 1)[0], a)
        assert e < tol, (n, "ad", e)


def check_nh_dot_product(c, tol=1e-11):
    T, P = nh_state(c)
    put(c, T, P)
    c.dy.dyn_core(TL)
    A = c.rect(1, c.nx, 1, c.ny)
    outs = [n for n in OUTS if n != "zh"]
    y = {n: c.dy.get(n, 1)[0].copy() for n in outs}
    lhs = sum(float(np.sum(y[n][A] ** 2)) for n in outs)
    put(c, T)
    c.dy.dyn_core(NL)
    for n in OUTS:
        s = np.zeros((c.dy.levels(n), c.ny + 7, c.nx + 7))
        if n != "zh":
            s[A] = y[n][A]
        c.dy.put(n, s[None], 1)
    c.dy.dyn_core(AD)
    rhs = 0.0
    for n, p in zip(INS, P):
        rhs += float(np.sum(c.dy.get(n, 1)[0] * p))      # the halos of u, v, pt, delp are read as given: independent inputs
    assert abs(lhs - rhs) <= tol * abs(lhs), (lhs, rhs)


# ---- whole time step (fv_dynamics: pkz from the equation of state, k_split x (acoustic steps, tracers, vertical remap))
def fv_names(c):
    return ["u", "v", "pt", "delp", "w", "delz"] + ["q%d" % (n + 1) for n in range(c.nq)]


def fv_put(c, T, P=None):
    for n, t in zip(fv_names(c), T):
        c.dy.put(n, t[None], 0)
        c.dy.put(n, (P[fv_names(c).index(n)] if P is not None else np.zeros_like(t))[None], 1)


def fv_dom(c, n):
    return c.rect(1, c.nx, 1, c.ny + 1) if n == "u" else c.rect(1, c.nx + 1, 1, c.ny) if n == "v" else c.rect(1, c.nx, 1, c.ny)


def check_nh_fv_tangent(c, tol_traj=1e-11, tol=1e-11):
    from test_oracle_nh import nh_state_fv
    T, P = nh_state_fv(c)
    ot, op = c.oracle.fv_dynamics_nh(TL, c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split, T, P)
    fv_put(c, T, P)
    c.dy.fv_dynamics(TL)
    for n, a, b in zip(fv_names(c), ot, op):
        A = fv_dom(c, n)
        e1, e2 = relerr(c.dy.get(n, 0)[0][A], a[A]), relerr(c.dy.get(n, 1)[0][A], b[A])
        assert e1 < tol_traj, (n, "traj", e1)
        assert e2 < tol, (n, "tl", e2)


def fv_seeds(c, rng, like):
    seeds = []
    for n, y in zip(fv_names(c), like):
        s = np.zeros_like(y)
        A = fv_dom(c, n)
        s[A] = rng.standard_normal(y[A].shape)
        seeds.append(s)
    return seeds


def check_nh_fv_adjoint(c, tol=1e-11):
    from test_oracle_nh import nh_state_fv
    T, P = nh_state_fv(c)
    seeds = fv_seeds(c, np.random.default_rng(13), T)
    _, iad = c.oracle.fv_dynamics_nh(AD, c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split, T, None, seeds)
    fv_put(c, T)
    c.dy.fv_dynamics(NL)
    for n, s in zip(fv_names(c), seeds):
        c.dy.put(n, s[None], 1)
    c.dy.fv_dynamics(AD)
    for n, a in zip(fv_names(c), iad):
        e = relerr(c.dy.get(n, 1)[0], a)
        assert e < tol, (n, "ad", e)


def check_nh_fv_dot_product(c, tol=1e-11):
    from test_oracle_nh import nh_state_fv
    T, P = nh_state_fv(c)
    fv_put(c, T, P)
    c.dy.fv_dynamics(TL)
    y = {n: c.dy.get(n, 1)[0].copy() for n in fv_names(c)}
    lhs = sum(float(np.sum(y[n][fv_dom(c, n)] ** 2)) for n in fv_names(c))
    fv_put(c, T)
    c.dy.fv_dynamics(NL)
    for n in fv_names(c):
        s = np.zeros_like(y[n]); A = fv_dom(c, n); s[A] = y[n][A]
        c.dy.put(n, s[None], 1)
    c.dy.fv_dynamics(AD)
    rhs = sum(float(np.sum(c.dy.get(n, 1)[0] * p)) for n, p in zip(fv_names(c), P))
    assert abs(lhs - rhs) <= tol * abs(lhs), (lhs, rhs)


# ---- six faces (tests/common.py CubeCase, oracle/nh.hpp fv_dynamics_nh_cube)
from fv3_jedi_linearmodel_amd.harness import cube_nh_state      # noqa: E402,F401


def cube_put(c, T, P=None):
    for n, t in zip(fv_names(c), T):
        c.dy.put(n, t, 0)
        c.dy.put(n, P[fv_names(c).index(n)] if P is not None else np.zeros_like(t), 1)


def cube_check_nh_fv(c, mode, tol_traj=1e-11, tol=1e-11):
    T, P = cube_nh_state(c)
    a = (c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split)
    if mode == TL:
        ot, op = c.oracle.fv_dynamics_nh(TL, *a, T, P)
        cube_put(c, T, P)
        c.dy.fv_dynamics(TL)
        for n, x, y in zip(fv_names(c), ot, op):
            A = fv_dom(c, n)
            e1, e2 = relerr(c.dy.get(n, 0)[A], x[A]), relerr(c.dy.get(n, 1)[A], y[A])
            assert e1 < tol_traj, (n, "traj", e1)
            assert e2 < tol, (n, "tl", e2)
        return
    seeds = fv_seeds(c, np.random.default_rng(17), T)
    _, iad = c.oracle.fv_dynamics_nh(AD, *a, T, None, seeds)
    cube_put(c, T)
    c.dy.fv_dynamics(NL)
    for n, s in zip(fv_names(c), seeds):
        c.dy.put(n, s, 1)
    c.dy.fv_dynamics(AD)
    for n, x in zip(fv_names(c), iad):
        e = relerr(c.dy.get(n, 1), x)
        assert e < tol, (n, "ad", e)


def cube_check_nh_dot_product(c, tol=1e-11):
    T, P = cube_nh_state(c)
    dx = []
    for n, p in zip(fv_names(c), P):           # the halos are overwritten by the exchange: perturb the compute domain only
        s = np.zeros_like(p); A = fv_dom(c, n); s[A] = p[A]; dx.append(s)
    cube_put(c, T, dx)
    c.dy.fv_dynamics(TL)
    y = {n: c.dy.get(n, 1).copy() for n in fv_names(c)}
    lhs = sum(float(np.sum(y[n][fv_dom(c, n)] ** 2)) for n in fv_names(c))
    cube_put(c, T)
    c.dy.fv_dynamics(NL)
    for n in fv_names(c):
        s = np.zeros_like(y[n]); A = fv_dom(c, n); s[A] = y[n][A]
        c.dy.put(n, s, 1)
    c.dy.fv_dynamics(AD)
    rhs = sum(float(np.sum(c.dy.get(n, 1) * p)) for n, p in zip(fv_names(c), dx))
    assert abs(lhs - rhs) <= tol * abs(lhs), (lhs, rhs)


def nh_adjoint_fields(c):
    """input adjoints of two acoustic steps for a fixed random output adjoint (used to compare the hand-written adjoints of
    the implicit solvers with the taped run of the generic code, FV3LM_NH_TAPE=1)"""
    T, _ = nh_state(c)
    rng = np.random.default_rng(23)
    A = c.rect(1, c.nx, 1, c.ny)
    put(c, T)
    c.dy.dyn_core(NL)
    for n in OUTS:
        s = np.zeros((c.dy.levels(n), c.ny + 7, c.nx + 7))
        if n != "zh":
            s[A] = rng.standard_normal(s[A].shape)
        c.dy.put(n, s[None], 1)
    c.dy.put("ws", rng.standard_normal((1, 1, c.ny + 7, c.nx + 7)), 1)     # exercises the surface-w output of the last step too
    c.dy.dyn_core(AD)
    return {n: c.dy.get(n, 1)[0].copy() for n in INS}
