"""CPU (-m "not gpu") tests of the product's stage logic: the HIP sources are compiled with
-DFV3LM_HOST_EMUL (host loops instead of kernel launches; test-only library, never loaded by the
package) and compared with the oracle.  The same checks run through the real HIP library in
test_gpu_parity.py.  Tolerance: relative L-inf <= 1e-12 per kernel group."""
import numpy as np
import pytest
from common import Case, relerr
from oracle import NL, TL, AD

TOL = 1e-12


@pytest.fixture(scope="module")
def case():
    return Case(nx=12, ny=10, npz=10, n_split=2, dt=1800.0, backend="emul")


def _csw_product(c, mode, pert=None):
    c.put_state(pert=pert)
    c.dy.run_group("c_sw", mode)
    names = ["delpc", "ptc", "uc1", "vc1", "ua", "va", "utf", "vtf", "divgd"]
    return {n: c.dy.get(n, 0)[0] for n in names}, ({n: c.dy.get(n, 1)[0] for n in names} if mode == TL else None)


def test_c_sw_tl(case):
    c = case
    ins = [c.traj[n][0] for n in ("delp", "pt", "u", "v")]
    ins_p = [c.pert[n][0] for n in ("delp", "pt", "u", "v")]
    ot, op = c.oracle.c_sw(TL, 0.5 * c.dt_ac, ins, ins_p)
    pt_, pp_ = _csw_product(c, TL, c.pert)
    nx, ny = c.nx, c.ny
    rects = {"delpc": (0, nx + 1, 0, ny + 1), "ptc": (0, nx + 1, 0, ny + 1), "uc1": (1, nx + 1, 1, ny), "vc1": (1, nx, 1, ny + 1),
             "ua": (0, nx + 1, 0, ny + 1), "va": (0, nx + 1, 0, ny + 1), "utf": (0, nx + 2, 0, ny + 1), "vtf": (0, nx + 1, 0, ny + 2),
             "divgd": (1, nx + 1, 1, ny + 1)}
    onames = ["delpc", "ptc", "uc1", "vc1", "ua", "va", "utf", "vtf", "divgd"]
    for n, a, b in zip(onames, ot, op):
        r = c.rect(*rects[n])
        assert relerr(pt_[n][r], a[r]) < TOL, n
        assert relerr(pp_[n][r], b[r]) < TOL, n + "_tl"


from groups import check_group

GROUPS = ["c_sw", "geopk_c", "p_grad_c", "d_sw", "geopk_d", "one_grad_p"]


@pytest.mark.parametrize("group", GROUPS)
def test_group_tl(case, group):
    check_group(case, group, TL, TOL)


@pytest.mark.parametrize("group", GROUPS)
def test_group_ad(case, group):
    check_group(case, group, AD, 1e-11)


from groups import check_dyn_core, dot_product_test


def test_dyn_core_tl(case):
    check_dyn_core(case, TL, 1e-10)      # full-step tolerance, BASELINE.md §6


def test_dyn_core_ad(case):
    check_dyn_core(case, AD, 1e-10)


def test_dot_product(case):
    lhs, rhs = dot_product_test(case)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


from groups import check_tracer, check_remap, check_fv_dynamics, dot_product_step


@pytest.fixture(scope="module")
def case_q():
    return Case(nx=12, ny=10, npz=10, n_split=2, k_split=2, dt=1800.0, backend="emul", nq=3)


@pytest.mark.parametrize("mode", [TL, AD])
def test_tracer_2d(case_q, mode):
    check_tracer(case_q, mode, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
@pytest.mark.parametrize("last", [0, 1])
def test_remap(case_q, mode, last):
    check_remap(case_q, mode, last, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_fv_dynamics(case_q, mode):
    check_fv_dynamics(case_q, mode, 1e-10)


def test_dot_product_step(case_q):
    lhs, rhs = dot_product_step(case_q)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_step_c12l64_tl_and_dot_product():
    """BASELINE config 1 (C12 L64 hydrostatic, one tile, one TL step) on the host emulation of the stage code"""
    from common import Case
    from groups import dot_product_step, check_fv_dynamics
    from oracle import TL
    c = Case(nx=12, ny=12, npz=64, n_split=4, k_split=1, dt=900.0, backend="emul", nq=1)
    check_fv_dynamics(c, TL, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_repeated_adjoint_on_one_forward_sweep(case_q):
    from groups import repeated_adjoint
    assert repeated_adjoint(case_q) < 1e-13


def test_tracer_subcycling(case_q):
    """accumulated Courant number >= 1: nsplt = 2 with per-level sub-step counts (fv_tracer2d_tlm.F90:1306-1345)"""
    check_tracer(case_q, TL, 1e-11, scale=10.0)
    check_tracer(case_q, AD, 1e-10, scale=10.0)
    assert case_q.dy.lib.L.fv3lm_tracer_nsplt(case_q.dy.h) >= 2


@pytest.mark.parametrize("slots", ["0", "1", "2"])
def test_trajectory_slots(slots, monkeypatch):
    """The backward sweep either finds a step's intermediates in a trajectory slot or recomputes them from the 4-field
    checkpoint (FV3LM_TRAJ_SLOTS caps the number of slots; default: as many as fit): same adjoint either way."""
    monkeypatch.setenv("FV3LM_TRAJ_SLOTS", slots)
    c = Case(nx=12, ny=10, npz=6, n_split=3, dt=1800.0, backend="emul")
    check_dyn_core(c, AD, 1e-10)
    lhs, rhs = dot_product_test(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_step_nl_matches_the_oracle():
    """the step_nl entry point (reference: fv3jedi_lm_dynamics_type%step_nl, DYN/fv3jedi_lm_dynamics_mod.F90:268-343, with the schemes
    the TL/AD path implements) against the oracle's nonlinear step"""
    from groups import check_step_nl
    c = Case(nx=12, ny=10, npz=10, n_split=2, k_split=2, dt=1800.0, backend="emul", nq=3)
    check_step_nl(c, 1e-12)
