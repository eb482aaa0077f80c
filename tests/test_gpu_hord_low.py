"""-m gpu: trajectory advection schemes 3 .. 7 and 9, 11, 12, 13 with split_hord (hord_low_checks.py) through the C-ABI of the HIP library against the oracle on
rough fields -- periodic tile, a tile wider than one 64-column block, one face, six faces -- and the dot-product identity at C96 L32 with a
GFS-like pairing (trajectory 5 / 5 / 5 / 6 / 8, perturbation 2, 1 in the sponge) at a size the oracle does not reach."""
import pytest
from oracle import TL, AD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h", [3, 4, 5, 6, 7, 9, 11, 12, 13])
def test_periodic_tile(h):
    from common import Case
    from groups import check_group, check_fv_dynamics, check_tracer, dot_product_step, check_step_nl
    from hord_low_checks import hord_kw, roughen
    c = roughen(Case(nx=24, ny=20, npz=12, n_split=2, k_split=2, dt=900.0, backend="hip", nq=2, **hord_kw(h)), qamp=1.2 if h in (9, 13) else 0.3)
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_fv_dynamics(c, TL, 1e-10)
    check_fv_dynamics(c, AD, 1e-10)
    check_step_nl(c, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("h", [5, 7, 13])
def test_wider_than_one_block(h):
    from common import Case
    from groups import check_group, check_tracer
    from hord_low_checks import hord_kw, roughen
    c = roughen(Case(nx=70, ny=20, npz=3, n_split=2, k_split=1, dt=900.0, backend="hip", nq=2, **hord_kw(h)), qamp=1.2 if h == 13 else 0.3)
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_tracer(c, AD, 1e-10)


@pytest.mark.parametrize("h", [3, 5, 6, 7, 9, 11, 13])
def test_face_groups(h):
    from common import Case
    from groups import check_group, check_tracer
    from hord_low_checks import hord_kw, roughen
    c = roughen(Case(nx=12, ny=12, npz=12, n_split=2, dt=900.0, backend="hip", face=2, nq=2, **hord_kw(h, pert=333 if h == 6 else 2)), periodic=False, qamp=1.2 if h in (9, 13) else 0.3)
    check_group(c, "d_sw", TL, 1e-12)
    check_group(c, "d_sw", AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_tracer(c, AD, 1e-10)


@pytest.mark.parametrize("h", [5, 6, 9, 12])
def test_six_faces_against_the_oracle(h):
    from common import CubeCase
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    from hord_low_checks import hord_kw, roughen
    c = roughen(CubeCase(n=16, npz=12, n_split=2, k_split=2, dt=900.0, backend="hip", oracle=True, nq=2, **hord_kw(h)), periodic=False)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_dot_product_c96l32_gfs_like_pairing():
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=96, npz=32, n_split=3, k_split=2, dt=900.0, backend="hip", oracle=False, nq=2, hord_mt=5, hord_vt=5, hord_tm=5, hord_dp=6, hord_tr=8)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)
