"""CPU (-m "not gpu") self-consistency checks of the non-hydrostatic oracle (oracle/nh.hpp: update_dz_c/d, riem_solver_c/3,
SIM1/SIM solvers, nh_p_grad, w transport): groundwork for SURVEY.md §8 row a7, whose product side is not built yet.  The
restatement is checked against itself: the dual-number tangent equals centred finite differences of the nonlinear code, and
the taped adjoint is the transpose of the tangent (dot-product identity)."""
import numpy as np
import pytest
from common import Case
from oracle import NL, TL, AD


def nh_state(c, seed=5):
    """hydrostatically balanced w = small, delz from delp, pt (theta_v) so that the non-hydrostatic pressure equals the layer mean"""
    o = c.opt
    delp, pt = c.traj["delp"][0], c.traj["pt"][0]
    pe = np.concatenate([np.full_like(delp[:1], o.ptop), o.ptop + np.cumsum(delp, axis=0)], axis=0)
    pm = delp / np.diff(np.log(pe), axis=0)
    delz = -(delp / o.grav) * o.rdgas * pt / pm ** (1.0 - o.akap)
    rng = np.random.default_rng(seed)
    from fv3_jedi_linearmodel_amd.grid import _smooth_field, halo_fill_periodic
    w = halo_fill_periodic(_smooth_field(rng, (1,) + delp.shape, c.nx, c.ny, 0.05), c.nx, c.ny)[0]
    T = [c.traj["u"][0], c.traj["v"][0], pt, delp, w, delz]
    P = [c.pert["u"][0], c.pert["v"][0], c.pert["pt"][0], c.pert["delp"][0],
         halo_fill_periodic(_smooth_field(rng, (1,) + delp.shape, c.nx, c.ny, 0.01), c.nx, c.ny)[0],
         halo_fill_periodic(_smooth_field(rng, (1,) + delp.shape, c.nx, c.ny, 0.5), c.nx, c.ny)[0]]
    return T, P


def nh_state_fv(c, seed=5):
    """fv_dynamics-level state: pt is temperature; delz hydrostatically balanced for T_v; tracers from the case"""
    o = c.opt
    delp = c.traj["delp"][0]
    pe = np.concatenate([np.full_like(delp[:1], o.ptop), o.ptop + np.cumsum(delp, axis=0)], axis=0)
    qv = c.qtraj[0][0] if c.nq > 0 else 0.0
    pkz = np.diff(pe ** o.akap, axis=0) / (o.akap * np.diff(np.log(pe), axis=0))
    tt = c.traj["pt"][0] * pkz / (1.0 + o.zvir * qv)          # the case's pt is theta_v; the time step takes temperature
    delz = -(o.rdgas / o.grav) * tt * (1.0 + o.zvir * qv) * np.diff(np.log(pe), axis=0)
    T0, P0 = nh_state(c, seed)
    T = [c.traj["u"][0], c.traj["v"][0], tt, delp, T0[4], delz] + [q[0] for q in c.qtraj]
    P = P0[:2] + [20.0 * P0[2]] + P0[3:6] + [q[0] for q in c.qpert]
    return T, P


@pytest.fixture(scope="module")
def nhcase():
    return Case(nx=10, ny=8, npz=8, n_split=2, dt=600.0, backend="none")


@pytest.fixture(scope="module")
def nhcase_fd():
    """The reference damps the vorticity of the perturbation with its own coefficients (sw_core_tlm.F90:2436-2452), so its
    tangent is the derivative of the nonlinear code only where both sets coincide: vorticity damping off on both sides here."""
    return Case(nx=10, ny=8, npz=8, n_split=2, dt=600.0, backend="none", do_vort_damp=0,
                do_vort_damp_pert=0)


def test_nh_nonlinear_is_quiet(nhcase):
    c = nhcase
    T, P = nh_state(c)
    out, _ = c.oracle.dyn_core_nh(NL, c.dims.dt, c.dims.n_split, T)
    A = c.rect(1, c.nx, 1, c.ny)
    for a in out:
        assert np.all(np.isfinite(a[A]))
    assert np.max(np.abs(out[4][A])) < 5.0            # w stays small from a balanced start
    assert np.max(np.abs(out[5][A] / T[5][A] - 1.0)) < 0.05   # layer thickness changes by a few per cent at most


def test_nh_tangent_matches_finite_differences(nhcase_fd):
    c = nhcase_fd
    T, P = nh_state(c)
    _, tl = c.oracle.dyn_core_nh(TL, c.dims.dt, c.dims.n_split, T, P)
    eps = 1e-6
    up, _ = c.oracle.dyn_core_nh(NL, c.dims.dt, c.dims.n_split, [t + eps * p for t, p in zip(T, P)])
    dn, _ = c.oracle.dyn_core_nh(NL, c.dims.dt, c.dims.n_split, [t - eps * p for t, p in zip(T, P)])
    A = c.rect(1, c.nx, 1, c.ny)
    for n in range(6):
        fd = (up[n][A] - dn[n][A]) / (2 * eps)
        scale = max(1e-30, np.max(np.abs(tl[n][A])))
        assert np.max(np.abs(fd - tl[n][A])) / scale < 2e-5, n


def test_nh_adjoint_dot_product(nhcase):
    c = nhcase
    T, P = nh_state(c)
    _, tl = c.oracle.dyn_core_nh(TL, c.dims.dt, c.dims.n_split, T, P)
    rng = np.random.default_rng(9)
    A = c.rect(1, c.nx, 1, c.ny)
    seeds = []
    for n, a in enumerate(tl):
        s = np.zeros_like(a)
        if n < 6:
            s[A] = rng.standard_normal(a[A].shape) / max(1e-30, np.max(np.abs(a[A])))
        seeds.append(s)
    _, ad = c.oracle.dyn_core_nh(AD, c.dims.dt, c.dims.n_split, T, None, seeds)
    lhs = sum(float(np.sum(a * s)) for a, s in zip(tl, seeds))
    rhs = sum(float(np.sum(a * p)) for a, p in zip(ad, P))
    assert abs(lhs - rhs) <= 1e-10 * abs(lhs), (lhs, rhs)


@pytest.fixture(scope="module")
def nhcase_fv():
    return Case(nx=10, ny=8, npz=8, n_split=2, k_split=2, dt=1200.0, nq=2, backend="none", do_vort_damp=0,
                do_vort_damp_pert=0)


def test_nh_fv_dynamics_tangent_matches_finite_differences(nhcase_fv):
    """whole non-hydrostatic time step (pkz from the equation of state, k_split x (acoustic steps, tracers, remap of T, w, delz))"""
    c = nhcase_fv
    T, P = nh_state_fv(c)
    a = (c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split)
    nl, tl = c.oracle.fv_dynamics_nh(TL, *a, T, P)
    A = c.rect(1, c.nx, 1, c.ny)
    assert np.max(np.abs(nl[5][A] / T[5][A] - 1.0)) < 0.25          # layer thickness: remapped to the reference levels, changes moderately
    eps = 1e-6
    up, _ = c.oracle.fv_dynamics_nh(NL, *a, [t + eps * p for t, p in zip(T, P)])
    dn, _ = c.oracle.fv_dynamics_nh(NL, *a, [t - eps * p for t, p in zip(T, P)])
    for n in range(len(T)):
        fd = (up[n][A] - dn[n][A]) / (2 * eps)
        scale = max(1e-30, np.max(np.abs(tl[n][A])))
        assert np.max(np.abs(fd - tl[n][A])) / scale < 5e-5, n


def test_nh_fv_dynamics_adjoint_dot_product(nhcase_fv):
    c = nhcase_fv
    T, P = nh_state_fv(c)
    a = (c.nq, c.dims.dt, c.dims.n_split, c.dims.k_split)
    _, tl = c.oracle.fv_dynamics_nh(TL, *a, T, P)
    rng = np.random.default_rng(9)
    A = c.rect(1, c.nx, 1, c.ny)
    seeds = []
    for y in tl:
        s = np.zeros_like(y)
        s[A] = rng.standard_normal(y[A].shape) / max(1e-30, np.max(np.abs(y[A])))
        seeds.append(s)
    _, ad = c.oracle.fv_dynamics_nh(AD, *a, T, None, seeds)
    lhs = sum(float(np.sum(y * s)) for y, s in zip(tl, seeds))
    rhs = sum(float(np.sum(x * p)) for x, p in zip(ad, P))
    assert abs(lhs - rhs) <= 1e-10 * abs(lhs), (lhs, rhs)


def test_nh_sim1_dispatch_tangent_matches_finite_differences():
    """a_imp > 0.999: RIEM_SOLVER3 runs SIM1_SOLVER (nh_core_tlm.F90:176-181); the restated dispatch against finite differences"""
    c = Case(nx=10, ny=8, npz=8, n_split=2, dt=600.0, backend="none", do_vort_damp=0, do_vort_damp_pert=0, a_imp=1.0, scale_z=0.3)
    test_nh_tangent_matches_finite_differences(c)
    c075 = Case(nx=10, ny=8, npz=8, n_split=2, dt=600.0, backend="none", do_vort_damp=0, do_vort_damp_pert=0, a_imp=0.75, scale_z=0.3)
    T, P = nh_state(c)
    a, _ = c.oracle.dyn_core_nh(NL, c.dims.dt, c.dims.n_split, T)
    b, _ = c075.oracle.dyn_core_nh(NL, c.dims.dt, c.dims.n_split, T)
    A = c.rect(1, c.nx, 1, c.ny)
    assert np.max(np.abs(a[4][A] - b[4][A])) > 1e-6         # the two solvers do differ
