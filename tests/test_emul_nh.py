"""Non-hydrostatic acoustic steps: the product's stage + column-tape implementation compiled for the
host (FV3LM_HOST_EMUL: same kernel bodies, loops instead of launches) against oracle/nh.hpp (checks in nh_checks.py)."""
import pytest
from common import Case
import nh_checks as N


@pytest.fixture(scope="module")
def nhc():
    return Case(nx=10, ny=8, npz=12, n_split=2, dt=600.0, backend="emul", hydrostatic=0)


def test_nh_dyn_core_tangent_matches_oracle(nhc):
    N.check_nh_tangent(nhc)


def test_nh_dyn_core_adjoint_matches_oracle(nhc):
    N.check_nh_adjoint(nhc)


def test_nh_dot_product(nhc):
    N.check_nh_dot_product(nhc)


@pytest.fixture(scope="module")
def nhfv():
    return Case(nx=10, ny=8, npz=12, n_split=2, k_split=2, dt=1200.0, nq=2, backend="emul", hydrostatic=0)


def test_nh_fv_dynamics_tangent_matches_oracle(nhfv):
    N.check_nh_fv_tangent(nhfv)


def test_nh_fv_dynamics_adjoint_matches_oracle(nhfv):
    N.check_nh_fv_adjoint(nhfv)


def test_nh_fv_dynamics_dot_product(nhfv):
    N.check_nh_fv_dot_product(nhfv)


@pytest.fixture(scope="module")
def nhcube():
    from common import CubeCase
    return CubeCase(n=12, npz=11, n_split=2, k_split=2, dt=1200.0, nq=2, backend="emul", oracle=True, hydrostatic=0)


def test_nh_cube_tangent_matches_oracle(nhcube):
    from oracle import TL
    N.cube_check_nh_fv(nhcube, TL)


def test_nh_cube_adjoint_matches_oracle(nhcube):
    from oracle import AD
    N.cube_check_nh_fv(nhcube, AD)


def test_nh_cube_dot_product(nhcube):
    N.cube_check_nh_dot_product(nhcube)


def test_nh_hand_written_adjoints_match_the_taped_run(monkeypatch):
    """riem_solver_c / riem_solver3: the hand-written reverse sweeps (csrc/nh_ad.h, the default) against the taped run of the
    generic column code (FV3LM_NH_TAPE=1, csrc/coltape.h)"""
    kw = dict(nx=10, ny=8, npz=12, n_split=2, dt=600.0, backend="emul", oracle=False, hydrostatic=0)
    monkeypatch.delenv("FV3LM_NH_TAPE", raising=False)
    hand = N.nh_adjoint_fields(Case(**kw))
    monkeypatch.setenv("FV3LM_NH_TAPE", "1")
    tape = N.nh_adjoint_fields(Case(**kw))
    from common import relerr
    for n in hand:
        assert relerr(hand[n], tape[n]) < 1e-11, n
        assert abs(hand[n]).max() > 0


@pytest.fixture(scope="module")
def nhc_sim1():
    """a_imp = 1 (BASELINE.md config 3's first setting): RIEM_SOLVER3 dispatches to SIM1_SOLVER (nh_core_tlm.F90:176-181); scale_z is
    set to show that SIM1 ignores it (the SIM path would not)"""
    from common import Case
    return Case(nx=10, ny=8, npz=12, n_split=2, dt=600.0, backend="emul", hydrostatic=0, a_imp=1.0, scale_z=0.3)


def test_nh_sim1_tangent_matches_oracle(nhc_sim1):
    N.check_nh_tangent(nhc_sim1)


def test_nh_sim1_adjoint_matches_oracle(nhc_sim1):
    N.check_nh_adjoint(nhc_sim1)


def test_nh_sim1_dot_product(nhc_sim1):
    N.check_nh_dot_product(nhc_sim1)
