"""BASELINE.md §4's configurations with their STATED flags on one MI355X (-m gpu).  Sizes the CPU oracle finishes in seconds
are compared with it field by field (relative L-inf, 1e-10 after a full step); at the full sizes the
size-independent invariant is the TL/AD dot-product identity |<M dx, dy> - <dx, M^T dy>| <= 1e-11 |<M dx, dy>|.
Config 5 (C384 L127, 24 sub-face tiles on 8 GPUs, non-hydrostatic, 4 tracers) does not fit one GPU at 127 levels: its partition, flags and
horizontal size run here with 16 levels (the acoustic step is level-parallel; the column operators see sponge and regular levels)."""
import pytest
from oracle import TL, AD

pytestmark = pytest.mark.gpu


def test_config1_c12l64_against_the_oracle():
    """config 1: C12 L64 hydrostatic, one doubly-periodic tile, k_split 1, n_split 4, dt 1800 s, 4 tracers -- whole step TL and AD
    against the oracle, and the dot-product identity"""
    from common import Case
    from groups import check_fv_dynamics, dot_product_step
    c = Case(nx=12, ny=12, npz=64, n_split=4, k_split=1, dt=1800.0, backend="hip", nq=4)
    check_fv_dynamics(c, TL, 1e-10)
    check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_six_faces_c24l16_against_the_oracle():
    """the largest six-face whole step the CPU oracle finishes in about a minute: C24 L16 (sponge and regular levels), k_split 2,
    n_split 3, 2 tracers -- TL and AD field by field"""
    from common import CubeCase
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = CubeCase(n=24, npz=16, n_split=3, k_split=2, dt=900.0, backend="hip", oracle=True, nq=2)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("a_imp", [1.0, 0.75])
def test_config3_c96l127_nonhydrostatic(a_imp):
    """config 3: C96 L127 non-hydrostatic (six faces resident), k_split 1, n_split 6, dt 450 s, 4 tracers, a_imp = 1 (SIM1) then 0.75
    (SIM), default sponge flags"""
    from common import CubeCase
    import nh_checks as N
    c = CubeCase(n=96, npz=127, n_split=6, k_split=1, dt=450.0, nq=4, backend="hip", hydrostatic=0, a_imp=a_imp)
    N.cube_check_nh_dot_product(c)


def test_config4_c192l127_headline_hydrostatic():
    """config 4 / bench.py's workload: C192 L127 hydrostatic, six faces on one GPU, k_split 2, n_split 6, dt 450 s, 4 tracers"""
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=192, npz=127, n_split=6, k_split=2, dt=450.0, nq=4, backend="hip")
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_config4_c192l127_nonhydrostatic():
    """config 4, non-hydrostatic variant (a_imp = 1): same size, w and delz prognostic"""
    from common import CubeCase
    import nh_checks as N
    c = CubeCase(n=192, npz=127, n_split=6, k_split=2, dt=450.0, nq=4, backend="hip", hydrostatic=0, a_imp=1.0)
    N.cube_check_nh_dot_product(c)


def test_config5_c384_24_subface_tiles_nonhydrostatic_l16():
    """config 5's shape on one GPU: C384, layout 2 x 2 = 24 tiles of 192 x 192, non-hydrostatic (a_imp = 1), k_split 2, n_split 6,
    dt 225 s, 4 tracers; 16 levels instead of 127 (memory of ONE GPU; on the 8-GPU node each GPU holds three of the tiles)"""
    from common import CubeCase
    import nh_checks as N
    c = CubeCase(n=384, npz=16, n_split=6, k_split=2, dt=225.0, nq=4, backend="hip", hydrostatic=0, a_imp=1.0, layout=2)
    N.cube_check_nh_dot_product(c)
