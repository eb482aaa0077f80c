"""Checks of the non-hydrostatic acoustic steps shared by the host-emulation (test_emul_nh.py) and
the MI355X (test_gpu_nh.py) runs: product fv3lm_dyn_core with hydrostatic = 0 against oracle/nh.hpp's dyn_core_nh on the
doubly-periodic tile — tangent-linear outputs, adjoint inputs, and the dot-product identity."""
import numpy as np
from common import relerr
from oracle import NL, TL, AD
from test_oracle_nh import nh_state

INS = ["u", "v", "pt", "delp", "w", "delz"]
OUTS = ["u", "v", "pt", "delp", "w", "delz", "pe", "peln", "pk", "zh"]


def put(c, T, P=None):
    for n in ("mfx", "mfy", "cx", "cy", "pkz"):      # accumulators / leftovers of an earlier sweep: no adjoint comes in through them here
        c.dy.put(n, np.zeros((1, c.npz, c.ny + 7, c.nx + 7)), 1)
    for n, t in zip(INS, T):
        c.dy.put(n, t[None], 0)
    if P is not None:
        for n, p in zip(INS, P):
            c.dy.put(n, p[None], 1)


def check_nh_tangent(c, tol_traj=1e-11, tol=1e-9):
    T, P = nh_state(c)
    ot, op = c.oracle.dyn_core_nh(TL, c.dims.dt, c.dims.n_split, T, P)
    put(c, T, P)
    c.dy.dyn_core(TL)
    A = c.rect(1, c.nx, 1, c.ny)
    for n, a, b in zip(OUTS, ot, op):
        e1, e2 = relerr(c.dy.get(n, 0)[0][A], a[A]), relerr(c.dy.get(n, 1)[0][A], b[A])
        assert e1 < tol_traj, (n, "traj", e1)
        assert e2 < tol, (n, "tl", e2)


def check_nh_adjoint(c, tol=1e-9):
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
