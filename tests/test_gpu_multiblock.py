"""-m gpu: the multi-block oracle comparisons of test_emul_multiblock.py through the C-ABI of the HIP library (nx = 70: two block columns
of every launch, ny = 20 / 70: two and five tiles of the fused fv_tp_2d kernels), every kernel group, tangent and adjoint, + tracer_2d."""
import pytest
from oracle import TL, AD

pytestmark = pytest.mark.gpu
GROUPS = ["c_sw", "geopk_c", "p_grad_c", "d_sw", "geopk_d", "one_grad_p"]


def test_periodic_tile_70x20():
    from common import Case
    from groups import check_group, check_tracer
    c = Case(nx=70, ny=20, npz=3, n_split=2, dt=600.0, backend="hip", nq=1)
    for g in GROUPS:
        check_group(c, g, TL, 1e-12)
        check_group(c, g, AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_tracer(c, AD, 1e-10)


def test_cube_face_70x70():
    from common import Case
    from groups import check_group, check_tracer
    c = Case(nx=70, ny=70, npz=2, n_split=2, dt=300.0, backend="hip", face=2, nq=1)
    for g in GROUPS:
        check_group(c, g, TL, 1e-12)
        check_group(c, g, AD, 1e-11)
    check_tracer(c, TL, 1e-11)
    check_tracer(c, AD, 1e-10)


def test_non_hydrostatic_70x20():
    from common import Case
    import nh_checks as N
    c = Case(nx=70, ny=20, npz=6, n_split=2, k_split=1, dt=300.0, nq=1, backend="hip", hydrostatic=0)
    N.check_nh_fv_tangent(c)
    N.check_nh_fv_adjoint(c)
