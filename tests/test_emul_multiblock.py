"""Oracle comparisons beyond one 64-wide block in x (every product-vs-oracle test had nx <= 24, so the multi-block
paths -- tile + halo loads of the LDS-staged launches, the XCD block renumbering, the thin last block column, the 64 x 16 tiles of the
fused fv_tp_2d kernels and their hand-written adjoints -- were covered only by fused-vs-staged self-comparisons and by the dot-product
identity, which cannot see a tangent and an adjoint that are wrong consistently).  nx = 70: two block columns, the second six cells
wide; ny = 20: two 16-row tiles of the fused kernels, five / three rows of the generic launches' blocks.  Host-emulation build here,
the HIP library in test_gpu_multiblock.py.  Oracle time: seconds."""
import pytest
from common import Case
from oracle import TL, AD
from groups import check_group, check_tracer

GROUPS = ["c_sw", "geopk_c", "p_grad_c", "d_sw", "geopk_d", "one_grad_p"]


@pytest.fixture(scope="module")
def pcase():
    return Case(nx=70, ny=20, npz=3, n_split=2, dt=600.0, backend="emul", nq=1)


@pytest.fixture(scope="module")
def fcase():
    return Case(nx=70, ny=70, npz=2, n_split=2, dt=300.0, backend="emul", face=2, nq=1)


@pytest.mark.parametrize("group", GROUPS)
@pytest.mark.parametrize("mode", [TL, AD])
def test_periodic_groups(pcase, group, mode):
    check_group(pcase, group, mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_periodic_tracer(pcase, mode):
    check_tracer(pcase, mode, 1e-11 if mode == TL else 1e-10)


@pytest.mark.parametrize("group", GROUPS)
@pytest.mark.parametrize("mode", [TL, AD])
def test_face_groups(fcase, group, mode):
    check_group(fcase, group, mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_face_tracer(fcase, mode):
    check_tracer(fcase, mode, 1e-11 if mode == TL else 1e-10)
