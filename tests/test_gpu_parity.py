"""GPU parity tests (-m gpu): the same kernel-group, dyn_core and dot-product checks as
test_emul_parity.py, but through the C-ABI of the real HIP library (libfv3lm_hip.so) on an MI355X.
Tolerances: relative L-inf <= 1e-12 per kernel group, <= 1e-10 for a full dyn_core sweep,
dot-product residual <= 1e-12."""
import numpy as np
import pytest
from oracle import TL, AD

pytestmark = pytest.mark.gpu
GROUPS = ["c_sw", "geopk_c", "p_grad_c", "d_sw", "geopk_d", "one_grad_p"]


@pytest.fixture(scope="module")
def case():
    from common import Case
    return Case(nx=12, ny=10, npz=10, n_split=2, dt=1800.0, backend="hip")


@pytest.fixture(scope="module")
def case_big():
    from common import Case
    return Case(nx=24, ny=24, npz=16, n_split=3, dt=900.0, backend="hip")


@pytest.mark.parametrize("group", GROUPS)
def test_group_tl(case, group):
    from groups import check_group
    check_group(case, group, TL, 1e-12)


@pytest.mark.parametrize("group", GROUPS)
def test_group_ad(case, group):
    from groups import check_group
    check_group(case, group, AD, 1e-11)


def test_dyn_core_tl(case_big):
    from groups import check_dyn_core
    check_dyn_core(case_big, TL, 1e-10)


def test_dyn_core_ad(case_big):
    from groups import check_dyn_core
    check_dyn_core(case_big, AD, 1e-10)


def test_dot_product_small(case):
    from groups import dot_product_test
    lhs, rhs = dot_product_test(case)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_dot_product_c48l72():
    """BASELINE config 2 size (C48 L72, n_split=6): size-independent invariant, no oracle involved."""
    from common import Case
    from groups import dot_product_test
    c = Case(nx=48, ny=48, npz=72, n_split=6, dt=900.0, backend="hip", oracle=False)
    lhs, rhs = dot_product_test(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


@pytest.fixture(scope="module")
def case_q():
    from common import Case
    return Case(nx=12, ny=10, npz=10, n_split=2, k_split=2, dt=1800.0, backend="hip", nq=3)


def test_repeated_adjoint_on_one_forward_sweep(case_q):
    from groups import repeated_adjoint
    assert repeated_adjoint(case_q) < 1e-13


@pytest.mark.parametrize("mode", [TL, AD])
def test_tracer_2d(case_q, mode):
    from groups import check_tracer
    check_tracer(case_q, mode, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
@pytest.mark.parametrize("last", [0, 1])
def test_remap(case_q, mode, last):
    from groups import check_remap
    check_remap(case_q, mode, last, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_fv_dynamics(case_q, mode):
    from groups import check_fv_dynamics
    check_fv_dynamics(case_q, mode, 1e-10)


def test_dot_product_step(case_q):
    from groups import dot_product_step
    lhs, rhs = dot_product_step(case_q)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_step_c12l64_tl_and_dot_product():
    """BASELINE config 1 (C12 L64 hydrostatic, one tile, one TL step): tangent-linear step against the oracle's whole
    step, and the TL/AD dot-product identity"""
    from common import Case
    from groups import dot_product_step, check_fv_dynamics
    c = Case(nx=12, ny=12, npz=64, n_split=4, k_split=1, dt=900.0, backend="hip", nq=1)
    check_fv_dynamics(c, TL, 1e-10)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_dot_product_step_c48l72():
    """BASELINE config 2 (C48 L72 hydrostatic, 4 tracers, k_split 1, n_split 6, dt 900 s): the TL/AD
    dot-product identity of the whole step at full size — a size-independent invariant."""
    from common import Case
    from groups import dot_product_step
    c = Case(nx=48, ny=48, npz=72, n_split=6, k_split=1, dt=900.0, backend="hip", oracle=False, nq=4)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


# ------------------------------------------------------------------ cube faces (edge / corner branches, exchange)
FACE_OPTS = {"default": {}, "hord2": dict(hord_ks_traj=0, hord_ks_pert=0),
             "nord0_hord333": dict(nord=0, nord_pert=0, hord_mt=333, hord_vt=333, hord_tm=333, hord_dp=333, hord_mt_pert=333, hord_vt_pert=333,
                                   hord_tm_pert=333, hord_dp_pert=333, hord_ks_traj=0, hord_ks_pert=0)}


@pytest.fixture(scope="module", params=list(FACE_OPTS))
def fcase(request):
    from common import Case
    return Case(nx=12, ny=12, npz=6, n_split=2, dt=1800.0, backend="hip", face=2, **FACE_OPTS[request.param])


@pytest.mark.parametrize("group", GROUPS)
def test_face_group_tl(fcase, group):
    from groups import check_group
    check_group(fcase, group, TL, 1e-12)


@pytest.mark.parametrize("group", GROUPS)
def test_face_group_ad(fcase, group):
    from groups import check_group
    check_group(fcase, group, AD, 1e-11)


def test_face_tracer():
    from common import Case
    from groups import check_tracer
    c = Case(nx=12, ny=12, npz=6, n_split=2, dt=1800.0, backend="hip", face=4, nq=2)
    check_tracer(c, TL, 1e-11)
    check_tracer(c, AD, 1e-10)


@pytest.fixture(scope="module")
def cube_case():
    from common import CubeCase
    return CubeCase(n=8, npz=5, n_split=2, backend="hip", oracle=True)


KINDS = [("cell", "delp", ""), ("dvec", "u", "v"), ("cvec", "uc", "vc"), ("corner", "divgd", ""), ("dedge", "u_o", "v_o")]


@pytest.mark.parametrize("kind,f0,f1", KINDS)
def test_cube_exchange(cube_case, kind, f0, f1):
    """device exchange == table applied in numpy (bit-exact: pure data motion), and <E x, y> = <x, E^T y>"""
    from fv3_jedi_linearmodel_amd import cube
    c = cube_case; rng = np.random.default_rng(3)
    shp = c.dy.shape(f0); names = [f0] + ([f1] if f1 else [])
    x = [rng.standard_normal(shp) for _ in names]; y = [rng.standard_normal(shp) for _ in names]
    for n, a in zip(names, x):
        c.dy.put(n, a, 0); c.dy.put(n, a, 1)
    c.dy.halo(kind, f0, f1, 1)
    ref = [a.copy() for a in x]
    cube.apply_table(c.tables[kind], ref[0], ref[1] if f1 else None)
    for n, r in zip(names, ref):
        assert np.array_equal(c.dy.get(n, 0), r) and np.array_equal(c.dy.get(n, 1), r)
    Ex = [c.dy.get(n, 1) for n in names]
    for n, a in zip(names, y):
        c.dy.put(n, a, 1)
    c.dy.halo(kind, f0, f1, 2)
    Ety = [c.dy.get(n, 1) for n in names]
    lhs = sum(float(np.sum(p * q)) for p, q in zip(Ex, y)); rhs = sum(float(np.sum(p * q)) for p, q in zip(x, Ety))
    assert abs(lhs - rhs) <= 1e-13 * abs(lhs)


def test_cube_dyn_core(cube_case):
    from groups import cube_check_dyn_core, cube_dot_product
    cube_check_dyn_core(cube_case, TL, 1e-10)
    cube_check_dyn_core(cube_case, AD, 1e-10)
    lhs, rhs = cube_dot_product(cube_case)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_cube_fv_dynamics():
    from common import CubeCase
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="hip", oracle=True, nq=2)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_cube_step_dot_product_c48l72():
    """Six faces of a C48 L72 cube resident on one GPU (BASELINE config 2 resolution): step_tl / step_ad identity."""
    from common import CubeCase
    from groups import cube_dot_product_step
    c = CubeCase(n=48, npz=72, n_split=6, k_split=2, dt=900.0, backend="hip", nq=2)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


def test_rccl_single_rank_comm():
    """The RCCL binding used for faces on other GPUs (dlopen of the process's RCCL, unique id, communicator): one rank."""
    import fv3_jedi_linearmodel_amd as fv3
    from fv3_jedi_linearmodel_amd._lib import comm_init_rccl
    lib = fv3.load_hip_library()
    comm_init_rccl(lib, 0, 1, lambda data: data)
    try:
        # with a communicator up, tracer_2d's max Courant numbers go through ncclAllReduce(max) on the library stream (one rank: identity)
        from common import CubeCase
        from groups import cube_check_tracer
        cc = CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="hip", oracle=True, nq=2)
        cube_check_tracer(cc, TL, 1e-11, scale=80.0)
        assert cc.dy.lib.L.fv3lm_tracer_nsplt(cc.dy.h) >= 2
    finally:
        assert lib.L.fv3lm_comm_destroy() == 0


def test_tracer_subcycling_gpu():
    """max Courant number >= 1: tracer_2d sub-cycles (periodic tile; six faces with the q exchange between sub-steps)"""
    from common import Case, CubeCase
    from groups import check_tracer, cube_check_tracer
    c = Case(nx=12, ny=10, npz=10, n_split=2, k_split=2, dt=1800.0, backend="hip", nq=3)
    check_tracer(c, TL, 1e-11, scale=10.0)
    check_tracer(c, AD, 1e-10, scale=10.0)
    assert c.dy.lib.L.fv3lm_tracer_nsplt(c.dy.h) >= 2
    cc = CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="hip", oracle=True, nq=2)
    cube_check_tracer(cc, TL, 1e-11, scale=80.0)
    cube_check_tracer(cc, AD, 1e-10, scale=80.0)
    assert cc.dy.lib.L.fv3lm_tracer_nsplt(cc.dy.h) >= 2


# ---- fv_tp_2d as one LDS-tiled launch (csrc/tpfused.h) against the staged launches (tp_fused_checks.py): to rounding on the device, bit for bit in the host emulation
def test_fused_tp_equals_staged_periodic():
    from common import Case
    from tp_fused_checks import check_fused_equals_staged
    check_fused_equals_staged(lambda: Case(nx=70, ny=20, npz=3, n_split=2, k_split=1, dt=900.0, backend="hip", oracle=False, nq=2), rtol=1e-12)


def test_fused_tp_equals_staged_cube():
    from common import CubeCase
    from tp_fused_checks import check_fused_equals_staged
    check_fused_equals_staged(lambda: CubeCase(n=66, npz=10, n_split=1, k_split=1, dt=225.0, backend="hip", nq=1), rtol=1e-12)


def test_fused_tp_equals_staged_nonhydrostatic():
    from common import Case
    from tp_fused_checks import check_fused_equals_staged
    check_fused_equals_staged(lambda: Case(nx=66, ny=18, npz=10, n_split=1, k_split=1, dt=300.0, backend="hip", oracle=False, hydrostatic=0), rtol=1e-12)


def test_boundary_copies_on_the_device():
    """fv3lm_traj_to_fv3 / _pert_to_fv3 / _fv3_to_pert (compact host arrays, halos / edge rows / pressures done by the library) against the
    whole-field put / get sequence: step_tl and step_ad bit for bit (boundary_checks.py)"""
    from common import Case, CubeCase
    from boundary_checks import check_boundary_copies
    check_boundary_copies(Case(nx=24, ny=20, npz=16, n_split=2, k_split=1, dt=900.0, backend="hip", oracle=False, nq=2))
    check_boundary_copies(CubeCase(n=12, npz=8, n_split=2, k_split=2, backend="hip", nq=2), cube=True)


def test_boundary_copies_against_the_numpy_oracle():
    """... and against tests/boundary_oracle.py (DYN/fv3jedi_lm_dynamics_mod.F90:717-809, :386-399, :651-665, :846-933 in numpy, independent
    exchange tables): traj_to_fv3 field by field, the perturbation edge fill of step_tl, its adjoint in step_ad"""
    from common import CubeCase
    from boundary_checks import check_boundary_oracle
    check_boundary_oracle(CubeCase(n=12, npz=8, n_split=2, k_split=2, backend="hip", nq=2))


def test_step_nl_matches_the_oracle(case_q):
    from groups import check_step_nl
    check_step_nl(case_q, 1e-11)
