"""Non-hydrostatic acoustic steps through the C-ABI of the HIP library on an MI355X, against
oracle/nh.hpp (checks in nh_checks.py).  Tolerances (nh_checks.py): relative L-inf 1e-11 on trajectory, tangent and adjoint fields,
dot product 1e-11; default options (sponge schemes on), npz > n_sponge_pert."""
import pytest
import nh_checks as N

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nhc():
    from common import Case
    return Case(nx=10, ny=8, npz=12, n_split=2, dt=600.0, backend="hip", hydrostatic=0)


def test_nh_dyn_core_tangent_matches_oracle(nhc):
    N.check_nh_tangent(nhc)


def test_nh_dyn_core_adjoint_matches_oracle(nhc):
    N.check_nh_adjoint(nhc)


def test_nh_dot_product(nhc):
    N.check_nh_dot_product(nhc)


def test_nh_dot_product_c48l72():
    """size-independent invariant at BASELINE config-2 size, no oracle"""
    from common import Case
    c = Case(nx=48, ny=48, npz=72, n_split=3, dt=300.0, backend="hip", oracle=False, hydrostatic=0)
    N.check_nh_dot_product(c)


@pytest.fixture(scope="module")
def nhfv():
    from common import Case
    return Case(nx=10, ny=8, npz=12, n_split=2, k_split=2, dt=1200.0, nq=2, backend="hip", hydrostatic=0)


def test_nh_fv_dynamics_tangent_matches_oracle(nhfv):
    N.check_nh_fv_tangent(nhfv)


def test_nh_fv_dynamics_adjoint_matches_oracle(nhfv):
    N.check_nh_fv_adjoint(nhfv)


def test_nh_fv_dynamics_dot_product(nhfv):
    N.check_nh_fv_dot_product(nhfv)


def test_nh_fv_dynamics_dot_product_c48l72():
    """whole non-hydrostatic step at BASELINE config-2 size (k_split 2, n_split 3, 3 tracers): dot-product identity, no oracle"""
    from common import Case
    c = Case(nx=48, ny=48, npz=72, n_split=3, k_split=2, dt=600.0, nq=3, backend="hip", oracle=False,
             hydrostatic=0)
    N.check_nh_fv_dot_product(c)


@pytest.fixture(scope="module")
def nhcube():
    from common import CubeCase
    return CubeCase(n=12, npz=11, n_split=2, k_split=2, dt=1200.0, nq=2, backend="hip", oracle=True, hydrostatic=0)


def test_nh_cube_tangent_matches_oracle(nhcube):
    from oracle import TL
    N.cube_check_nh_fv(nhcube, TL)


def test_nh_cube_adjoint_matches_oracle(nhcube):
    from oracle import AD
    N.cube_check_nh_fv(nhcube, AD)


def test_nh_cube_dot_product(nhcube):
    N.cube_check_nh_dot_product(nhcube)


def test_nh_cube_dot_product_c48l72():
    """six faces of C48 L72, whole non-hydrostatic step: dot-product identity, no oracle"""
    from common import CubeCase
    c = CubeCase(n=48, npz=72, n_split=3, k_split=2, dt=600.0, nq=3, backend="hip", hydrostatic=0)
    N.cube_check_nh_dot_product(c)


def test_nh_cube_dot_product_c96l127():
    """BASELINE config 3: C96 L127 non-hydrostatic TL+AD on one MI355X (six faces resident), dot-product identity"""
    from common import CubeCase
    c = CubeCase(n=96, npz=127, n_split=6, k_split=1, dt=225.0, nq=0, backend="hip", hydrostatic=0)
    N.cube_check_nh_dot_product(c)


def test_nh_hand_written_adjoints_match_the_taped_run(monkeypatch):
    """csrc/nh_ad.h (default) against the taped run of the generic column code (FV3LM_NH_TAPE=1) on the device"""
    from common import Case, relerr
    kw = dict(nx=16, ny=12, npz=24, n_split=2, dt=600.0, backend="hip", oracle=False, hydrostatic=0)
    monkeypatch.delenv("FV3LM_NH_TAPE", raising=False)
    hand = N.nh_adjoint_fields(Case(**kw))
    monkeypatch.setenv("FV3LM_NH_TAPE", "1")
    tape = N.nh_adjoint_fields(Case(**kw))
    for n in hand:
        assert relerr(hand[n], tape[n]) < 1e-10, n


@pytest.fixture(scope="module")
def nhc_sim1():
    """a_imp = 1 (BASELINE.md config 3's first setting): RIEM_SOLVER3 dispatches to SIM1_SOLVER (nh_core_tlm.F90:176-181); scale_z is
    set to show that SIM1 ignores it (the SIM path would not)"""
    from common import Case
    return Case(nx=10, ny=8, npz=12, n_split=2, dt=600.0, backend="hip", hydrostatic=0, a_imp=1.0, scale_z=0.3)


def test_nh_sim1_tangent_matches_oracle(nhc_sim1):
    N.check_nh_tangent(nhc_sim1)


def test_nh_sim1_adjoint_matches_oracle(nhc_sim1):
    N.check_nh_adjoint(nhc_sim1)


def test_nh_sim1_dot_product(nhc_sim1):
    N.check_nh_dot_product(nhc_sim1)
