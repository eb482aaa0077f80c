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


def test_dot_product_step_c48l72():
    """BASELINE config 2 (C48 L72 hydrostatic, 4 tracers, k_split 1, n_split 6, dt 900 s): the TL/AD
    dot-product identity of the whole step at full size — a size-independent invariant."""
    from common import Case
    from groups import dot_product_step
    c = Case(nx=48, ny=48, npz=72, n_split=6, k_split=1, dt=900.0, backend="hip", oracle=False, nq=4)
    lhs, rhs = dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)
