"""CPU (-m "not gpu") tests of the six-face path: DYN_CORE and the full fv_dynamics sweep on a C8 cube (all six faces
resident, table-driven exchange) — host-emulation build of the HIP sources vs the six-face oracle (oracle/cube.hpp),
tangent and adjoint, plus the dot-product identity.  The same checks run through the HIP library in test_gpu_parity.py."""
import numpy as np
import pytest
from common import CubeCase, relerr
from oracle import NL, TL, AD
from groups import cube_check_dyn_core, cube_dot_product, cube_check_fv_dynamics, cube_dot_product_step


@pytest.fixture(scope="module")
def ccase():
    return CubeCase(n=8, npz=5, n_split=2, backend="emul", oracle=True)


def test_cube_dyn_core_tl(ccase):
    cube_check_dyn_core(ccase, TL, 1e-10)


def test_cube_dyn_core_ad(ccase):
    cube_check_dyn_core(ccase, AD, 1e-10)


def test_cube_dot_product(ccase):
    lhs, rhs = cube_dot_product(ccase)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


@pytest.fixture(scope="module")
def ccase_q():
    return CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="emul", oracle=True, nq=2)


def test_cube_fv_dynamics_tl(ccase_q):
    cube_check_fv_dynamics(ccase_q, TL, 1e-10)


def test_cube_fv_dynamics_ad(ccase_q):
    cube_check_fv_dynamics(ccase_q, AD, 1e-10)


def test_cube_step_dot_product(ccase_q):
    lhs, rhs = cube_dot_product_step(ccase_q)
    assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)


from groups import cube_check_tracer


def test_cube_tracer(ccase_q):
    cube_check_tracer(ccase_q, TL, 1e-11)
    cube_check_tracer(ccase_q, AD, 1e-10)


def test_cube_tracer_subcycling(ccase_q):
    """max Courant number over the whole cube >= 1: sub-steps with the q halo exchange in between"""
    cube_check_tracer(ccase_q, TL, 1e-11, scale=80.0)
    cube_check_tracer(ccase_q, AD, 1e-10, scale=80.0)
    assert ccase_q.dy.lib.L.fv3lm_tracer_nsplt(ccase_q.dy.h) >= 2


def test_cube_loopback_message_path():
    """Every row between two faces travels as a message with the rank as its own peer (cube.split_table loopback): pack -> transport ->
    unpack, the adjoint back the other way — the N > 1 path of exchange.h on one rank, against the six-face oracle.  The same test runs
    through ncclSend / ncclRecv on the device (test_gpu_parity.py)."""
    import ctypes as C
    from fv3_jedi_linearmodel_amd._lib import set_transport_callback, TRANSPORT_FN
    c = CubeCase(n=8, npz=6, n_split=2, k_split=2, backend="emul", oracle=True, nq=2, loopback=True)
    calls = [0]

    def transport(peers, sbufs, rbufs):
        assert peers == [0]
        for s, r in zip(sbufs, rbufs):
            assert s.size == r.size
            r[:] = s
        calls[0] += 1
    set_transport_callback(c.lib, transport)
    try:
        cube_check_fv_dynamics(c, TL, 1e-10)
        cube_check_fv_dynamics(c, AD, 1e-10)
        cube_check_tracer(c, TL, 1e-11, scale=80.0)
        lhs, rhs = cube_dot_product_step(c)
        assert abs(lhs - rhs) <= 1e-11 * abs(lhs), (lhs, rhs)
        assert calls[0] > 0
    finally:
        c.lib.L.fv3lm_set_transport_callback(C.cast(None, TRANSPORT_FN), None)
