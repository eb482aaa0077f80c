"""-m gpu: sub-face tiles (layout > 1 x 1) through the C-ABI of the HIP library: 24 tiles of a 2 x 2 layout and 54 of a 3 x 3 one on one GPU
must give what the six whole faces give (layout_checks.py) -- hydrostatic with tracers, split_damp / split_hord, non-hydrostatic -- and the
TL/AD dot-product identity holds at C96 with 24 tiles."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout,n", [(2, 32), (3, 48)])
def test_hydrostatic_with_tracers(layout, n):
    from common import CubeCase
    from layout_checks import check_layout_equals_whole_faces
    check_layout_equals_whole_faces(lambda L: CubeCase(n=n, npz=6, n_split=2, k_split=2, dt=600.0, backend="hip", nq=2, layout=L), layout)


def test_split_damp_nord3_and_split_hord():
    from common import CubeCase
    from layout_checks import check_layout_equals_whole_faces
    kw = dict(split_damp=1, nord=3, nord_pert=1, dddmp=0.35, d4_bg=0.11, vtdm4=0.03, n_sponge_pert=2, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10, hord_tr=10)
    check_layout_equals_whole_faces(lambda L: CubeCase(n=32, npz=6, n_split=2, k_split=1, dt=300.0, backend="hip", nq=1, layout=L, **kw), 2)


def test_nonhydrostatic():
    from common import CubeCase
    from layout_checks import check_layout_equals_whole_faces
    check_layout_equals_whole_faces(lambda L: CubeCase(n=32, npz=8, n_split=2, k_split=1, dt=150.0, backend="hip", nq=1, layout=L, hydrostatic=0), 2, tol=1e-11)


def test_boundary_copies_with_tiles():
    from common import CubeCase
    from boundary_checks import check_boundary_copies
    check_boundary_copies(CubeCase(n=32, npz=6, n_split=2, k_split=1, backend="hip", nq=1, layout=2), cube=True)


def test_dot_product_c96l32_24_tiles():
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=96, npz=32, n_split=3, k_split=2, dt=900.0, backend="hip", oracle=False, nq=2, layout=2)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)
