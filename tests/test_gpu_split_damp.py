"""split_damp on the MI355X (-m gpu): the checks of test_emul_split_damp.py through the C-ABI of the HIP library -- periodic tile, one cube
face, six faces -- against the oracle, with the trajectory's nord 1, 2 and 3 beside a perturbation nord of 1 or 0 and every damping
coefficient different between the two, plus the dot-product identity at a size the oracle does not reach."""
import pytest
from oracle import TL, AD
from test_emul_split_damp import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["coef", "nord0p", "nord2", "nord3", "same", "nord2_h10"])
def test_periodic_tile(name):
    from common import Case
    from groups import check_group, check_fv_dynamics, dot_product_step, check_step_nl
    c = Case(nx=24, ny=20, npz=12, n_split=2, k_split=2, dt=1800.0, backend="hip", nq=1, **CASES[name])
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)
    check_fv_dynamics(c, TL, 1e-10)
    check_fv_dynamics(c, AD, 1e-10)
    check_step_nl(c, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("name", ["coef", "nord0p", "nord2", "nord3"])
def test_face_groups(name):
    from common import Case
    from groups import check_group
    c = Case(nx=12, ny=12, npz=12, n_split=2, dt=1800.0, backend="hip", face=2, nq=0, **CASES[name])
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)


@pytest.mark.parametrize("name", ["nord2", "nord3"])
def test_six_faces_against_the_oracle(name):
    from common import CubeCase
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = CubeCase(n=16, npz=12, n_split=2, k_split=2, backend="hip", oracle=True, nq=1, **CASES[name])
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_dot_product_c96l32_six_faces():
    """the operational pairing: trajectory hord 10 / nord 2, perturbation hord 2 (1 in the sponge) / nord 1"""
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=96, npz=32, n_split=3, k_split=2, dt=900.0, backend="hip", oracle=False, nq=2, **CASES["nord2_h10"])
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)
