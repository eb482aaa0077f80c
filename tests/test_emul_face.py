"""CPU (-m "not gpu") tests of the cube-face edge and corner branches of the stage library (host-emulation build of
the HIP sources) against the oracle: one whole face of a C12 cube with its real gnomonic metrics and arbitrary
smooth halo data, every kernel group of the acoustic step, tangent and adjoint.  The same checks run through the
HIP library in test_gpu_parity.py."""
import pytest
from common import Case
from oracle import TL, AD
from groups import check_group

GROUPS = ["c_sw", "geopk_c", "p_grad_c", "d_sw", "geopk_d", "one_grad_p"]
OPTS = {"default": {}, "hord2": dict(hord_ks_traj=0, hord_ks_pert=0), "nord0_hord333": dict(nord=0, nord_pert=0, hord_mt=333, hord_vt=333, hord_tm=333, hord_dp=333, hord_mt_pert=333,
                                              hord_vt_pert=333, hord_tm_pert=333, hord_dp_pert=333, hord_ks_traj=0, hord_ks_pert=0)}


@pytest.fixture(scope="module", params=list(OPTS))
def fcase(request):
    return Case(nx=12, ny=12, npz=6, n_split=2, dt=1800.0, backend="emul", face=2, **OPTS[request.param])


@pytest.mark.parametrize("group", GROUPS)
def test_face_group_tl(fcase, group):
    check_group(fcase, group, TL, 1e-12)


@pytest.mark.parametrize("group", GROUPS)
def test_face_group_ad(fcase, group):
    check_group(fcase, group, AD, 1e-11)


from groups import check_tracer


@pytest.fixture(scope="module")
def fcase_q():
    return Case(nx=12, ny=12, npz=6, n_split=2, dt=1800.0, backend="emul", face=4, nq=2)


def test_face_tracer_tl(fcase_q):
    check_tracer(fcase_q, TL, 1e-11)


def test_face_tracer_ad(fcase_q):
    check_tracer(fcase_q, AD, 1e-10)
