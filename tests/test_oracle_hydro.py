"""CPU (-m "not gpu") self-consistency of the hydrostatic oracle at the level of the whole time step (oracle/fv_dynamics.hpp,
oracle/cube.hpp): the dual-number tangent equals centred finite differences of the nonlinear restatement, and the taped
adjoint is the transpose of the tangent.  (The reference damps the perturbation vorticity with its own coefficients, so the
finite-difference comparison runs with the vorticity damping off on both sides, as in test_oracle_nh.py.)"""
import numpy as np
import pytest
from common import Case, CubeCase
from groups import step_state, cube_step_state
from oracle import NL, TL, AD

INS = ["u", "v", "pt", "delp", "pe", "peln", "pk", "pkz"]


def _inputs(T, P, nq):
    names = INS + ["q%d" % (n + 1) for n in range(nq)]
    return [T[n] for n in names], [P[n] for n in names]


@pytest.fixture(scope="module")
def hcase():
    return Case(nx=10, ny=8, npz=8, n_split=2, k_split=2, dt=1200.0, nq=2, backend="none", do_vort_damp=0, do_vort_damp_pert=0)


def test_hydrostatic_step_tangent_matches_finite_differences(hcase):
    c = hcase
    T, P = step_state(c)
    i_t, i_p = _inputs(T, P, c.nq)
    a = (c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split)
    _, tl = c.oracle.fv_dynamics(TL, *a, i_t, i_p)
    eps = 1e-6
    up, _ = c.oracle.fv_dynamics(NL, *a, [t + eps * p for t, p in zip(i_t, i_p)])
    dn, _ = c.oracle.fv_dynamics(NL, *a, [t - eps * p for t, p in zip(i_t, i_p)])
    A = c.rect(1, c.nx, 1, c.ny)
    for n in range(len(tl)):
        fd = (up[n][A] - dn[n][A]) / (2 * eps)
        assert np.max(np.abs(fd - tl[n][A])) / max(1e-30, np.max(np.abs(tl[n][A]))) < 5e-5, n


def test_hydrostatic_step_adjoint_is_the_transpose(hcase):
    c = hcase
    T, P = step_state(c)
    i_t, i_p = _inputs(T, P, c.nq)
    a = (c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split)
    _, tl = c.oracle.fv_dynamics(TL, *a, i_t, i_p)
    rng = np.random.default_rng(3)
    A = c.rect(1, c.nx, 1, c.ny)
    seeds = []
    for y in tl:
        s = np.zeros_like(y); s[A] = rng.standard_normal(y[A].shape) / max(1e-30, np.max(np.abs(y[A]))); seeds.append(s)
    _, ad = c.oracle.fv_dynamics(AD, *a, i_t, None, seeds)
    lhs = sum(float(np.sum(y * s)) for y, s in zip(tl, seeds))
    rhs = sum(float(np.sum(x * p)) for x, p in zip(ad, i_p))
    assert abs(lhs - rhs) <= 1e-10 * abs(lhs), (lhs, rhs)


def test_six_face_step_tangent_matches_finite_differences():
    c = CubeCase(n=8, npz=6, n_split=2, k_split=1, dt=900.0, nq=1, backend="none", oracle=True, do_vort_damp=0, do_vort_damp_pert=0)
    T, P = cube_step_state(c)
    i_t, i_p = _inputs(T, P, c.nq)
    a = (c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split)
    _, tl = c.oracle.fv_dynamics(TL, *a, i_t, i_p)
    eps = 1e-6
    up, _ = c.oracle.fv_dynamics(NL, *a, [t + eps * p for t, p in zip(i_t, i_p)])
    dn, _ = c.oracle.fv_dynamics(NL, *a, [t - eps * p for t, p in zip(i_t, i_p)])
    A = c.rect(1, c.n, 1, c.n)
    for n in range(len(tl)):
        fd = (up[n][A] - dn[n][A]) / (2 * eps)
        assert np.max(np.abs(fd - tl[n][A])) / max(1e-30, np.max(np.abs(tl[n][A]))) < 5e-5, n
