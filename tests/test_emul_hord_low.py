"""Trajectory advection schemes 3 .. 7 and 9, 11, 12, 13 with split_hord (hord_low_checks.py) on the host-emulation build against the oracle (tp_mono.hpp
ppm_line_low / uv_line_low, ppm_line_mono / uv_line_mono).  No reference-held fixtures for these routines: parity unpinned, like the rest of the path."""
import numpy as np
import pytest
from common import Case, CubeCase
from groups import check_group, check_fv_dynamics, check_tracer, dot_product_step, check_step_nl
from hord_low_checks import hord_kw, roughen, nl_step
from oracle import TL, AD


@pytest.fixture(scope="module", params=[3, 4, 5, 6, 7, 9, 11, 12, 13])
def case(request):
    return roughen(Case(nx=12, ny=10, npz=12, n_split=2, k_split=2, dt=900.0, backend="emul", nq=2, **hord_kw(request.param)), qamp=1.2 if request.param in (9, 13) else 0.3)


@pytest.mark.parametrize("mode", [TL, AD])
def test_d_sw_group(case, mode):
    check_group(case, "d_sw", mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_tracer_2d(case, mode):
    check_tracer(case, mode, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_fv_dynamics(case, mode):
    check_fv_dynamics(case, mode, 1e-10)


def test_step_nl_and_dot_product(case):
    check_step_nl(case, 1e-10)
    lhs, rhs = dot_product_step(case)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_every_scheme_is_its_own():
    """on rough fields the twelve trajectory schemes 2 .. 13 give pairwise different nonlinear steps: the tests that tell them apart fire
    (tracer with zero crossings: the positive-definite constraint is what separates 12 from 13)"""
    out = [nl_step(roughen(Case(nx=12, ny=10, npz=12, n_split=2, k_split=1, dt=900.0, backend="emul", oracle=False, nq=1, **hord_kw(h)), qamp=1.2)) for h in range(2, 14)]
    for a in range(len(out)):
        for b in range(a + 1, len(out)):
            d = max(np.max(np.abs(out[a][n] - out[b][n])) / np.max(np.abs(out[b][n])) for n in ("pt", "delp", "q1"))
            assert d > 1e-9, (a + 2, b + 2, d)


# ---- one cube face with smooth halo data plus noise everywhere: the edge values next to a face edge (and their clipping for scheme 7)
@pytest.fixture(scope="module", params=[3, 5, 6, 7, 9, 11, 13])
def fcase(request):
    return roughen(Case(nx=12, ny=12, npz=12, n_split=2, dt=900.0, backend="emul", face=2, nq=2, **hord_kw(request.param, pert=333 if request.param == 6 else 2)), periodic=False,
                   qamp=1.2 if request.param in (9, 13) else 0.3)


@pytest.mark.parametrize("mode", [TL, AD])
def test_face_d_sw_group(fcase, mode):
    check_group(fcase, "d_sw", mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_face_tracer(fcase, mode):
    check_tracer(fcase, mode, 1e-11 if mode == TL else 1e-10)


@pytest.mark.parametrize("h", [5, 6, 9, 12])
def test_cube_fv_dynamics(h):
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = roughen(CubeCase(n=8, npz=12, n_split=2, k_split=2, dt=900.0, backend="emul", oracle=True, nq=2, **hord_kw(h)), periodic=False)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_nonhydrostatic():
    """w transport (hord_vt) and the height transport of update_dz_d with a limited trajectory scheme"""
    import nh_checks as N
    c = Case(nx=10, ny=8, npz=12, n_split=2, k_split=2, dt=600.0, nq=2, backend="emul", hydrostatic=0, **hord_kw(5))
    N.check_nh_fv_tangent(c)
    N.check_nh_fv_adjoint(c)
    N.check_nh_fv_dot_product(c)
