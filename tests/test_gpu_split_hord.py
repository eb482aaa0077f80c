"""split_hord on the MI355X (-m gpu): the checks of test_emul_split_hord.py through the C-ABI of the HIP library -- periodic tile,
six cube faces, non-hydrostatic -- against the oracle, plus the dot-product identity at a size the oracle does not reach (C96 L32 six
faces with the operational pairing: trajectory hord 10, perturbation 2, 1 in the sponge)."""
import pytest
from oracle import TL, AD

pytestmark = pytest.mark.gpu

SPLIT10 = dict(hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10, hord_tr=10)
SPLIT8 = dict(hord_mt=8, hord_vt=8, hord_tm=8, hord_dp=8, hord_tr=8, hord_mt_pert=333, hord_vt_pert=333, hord_tm_pert=333, hord_dp_pert=333, hord_tr_pert=333)


@pytest.mark.parametrize("kw", [SPLIT10, SPLIT8], ids=["h10", "h8"])
def test_periodic_tile(kw):
    from common import Case
    from groups import check_group, check_fv_dynamics, check_tracer, dot_product_step, check_step_nl
    c = Case(nx=24, ny=20, npz=12, n_split=2, k_split=2, dt=1800.0, backend="hip", nq=2, **kw)
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_fv_dynamics(c, TL, 1e-10)
    check_fv_dynamics(c, AD, 1e-10)
    check_step_nl(c, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_face_groups():
    from common import Case
    from groups import check_group, check_tracer
    c = Case(nx=12, ny=12, npz=12, n_split=2, dt=1800.0, backend="hip", face=2, nq=2, **SPLIT10)
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_tracer(c, AD, 1e-10)


def test_six_faces_against_the_oracle():
    from common import CubeCase
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = CubeCase(n=16, npz=12, n_split=2, k_split=2, backend="hip", oracle=True, nq=2, **SPLIT10)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_nonhydrostatic_cube():
    from common import CubeCase
    import nh_checks as N
    c = CubeCase(n=12, npz=11, n_split=2, k_split=1, dt=600.0, nq=1, backend="hip", oracle=True, hydrostatic=0, **SPLIT10)
    N.cube_check_nh_fv(c, TL)
    N.cube_check_nh_fv(c, AD)
    N.cube_check_nh_dot_product(c)


def test_dot_product_c96l32_six_faces():
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=96, npz=32, n_split=3, k_split=2, dt=900.0, backend="hip", oracle=False, nq=2, **SPLIT10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("kord", [8, 9, 10, 11, 12, 13, 14, 15])
def test_split_kord(kord):
    """trajectory remapped with the limited profiles, perturbation with the linear one: remap alone and the whole step"""
    from common import Case
    from groups import check_remap, check_fv_dynamics, dot_product_step, check_step_nl
    c = Case(nx=24, ny=20, npz=14, n_split=2, k_split=2, dt=1800.0, backend="hip", nq=3, kord_tm=-kord, kord_mt=kord, kord_tr=kord)
    check_remap(c, TL, 0, 1e-11)
    check_remap(c, AD, 1, 1e-11)
    check_fv_dynamics(c, TL, 1e-10)
    check_fv_dynamics(c, AD, 1e-10)
    check_step_nl(c, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_operational_pairing_six_faces():
    """trajectory hord 10 / kord 9, perturbation hord 2 (1 in the sponge) / kord 17"""
    from common import CubeCase
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = CubeCase(n=16, npz=12, n_split=2, k_split=2, backend="hip", oracle=True, nq=2, kord_tm=-9, kord_mt=9, kord_tr=9, **SPLIT10)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_nonhydrostatic_split_kord():
    """non-hydrostatic, trajectory hord 10 / kord 9 (kord_wz 9 for w), perturbation linear: six faces against the oracle"""
    from common import CubeCase
    import nh_checks as N
    c = CubeCase(n=12, npz=12, n_split=2, k_split=2, dt=1200.0, nq=1, backend="hip", oracle=True, hydrostatic=0, kord_tm=-9, kord_mt=9, kord_tr=9, kord_wz=9, **SPLIT10)
    N.cube_check_nh_fv(c, TL)
    N.cube_check_nh_fv(c, AD)
    N.cube_check_nh_dot_product(c)


def test_dot_product_c192l127_operational_pairing():
    """the headline size with the operational pairing (trajectory hord 10 / kord 9, perturbation 2 / 17): TL/AD dot-product identity"""
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=192, npz=127, n_split=6, k_split=2, dt=450.0, backend="hip", oracle=False, nq=4, kord_tm=-9, kord_mt=9, kord_tr=9, **SPLIT10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)
