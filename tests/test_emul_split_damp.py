"""split_damp (fv_arrays_tlmadm.F90:76 -- the reference's DEFAULT): the perturbation takes its own damping coefficients and sponge
rules (nord_pert, dddmp_pert, d2_bg_pert, d4_bg_pert, vtdm4_pert; dyn_core_tlm.F90:835-921), the trajectory values the trajectory's
(sw_core_tlm.F90:1664-1682 mass, :1787-1803 heat, :2341-2369 divergence damping, :2436-2452 vorticity damping) -- whose nord may then be
2 or 3 (fill_corners, fv_mp_nlm_mod.F90:1046-1303; the multi-pass loop sw_core_tlm.F90:8463-8530).  Product (differentiated stage chain
with the perturbation's coefficients + the values-only trajectory kernels of csrc/dampt.h) against the oracle (sw_core.hpp d_sw: the _TLM
routine on *_tj copies, then the nonlinear routine), host-emulation build.  No reference-held fixtures: parity unpinned, like the path."""
import numpy as np
import pytest
import fv3_jedi_linearmodel_amd as fv3
from common import Case, CubeCase
from groups import check_group, check_dyn_core, check_fv_dynamics, dot_product_step, check_step_nl
from oracle import TL, AD, NL

# every trajectory coefficient differs from its perturbation counterpart
COEF = dict(split_damp=1, dddmp=0.35, dddmp_pert=0.2, d4_bg=0.11, d4_bg_pert=0.15, d2_bg=0.02, d2_bg_pert=0.015, vtdm4=0.03, vtdm4_pert=0.0005,
            d2_bg_k1=0.18, d2_bg_k2=0.1)
CASES = {
    "coef": dict(COEF),                                    # nord = nord_pert = 1, coefficients differ
    "nord0p": dict(COEF, nord=1, nord_pert=0),             # trajectory del-4, perturbation del-2
    "nord2": dict(COEF, nord=2, nord_pert=1),              # operational: trajectory del-6
    "nord3": dict(COEF, nord=3, nord_pert=1, n_sponge_pert=4),
    "same": dict(split_damp=1),                            # equal namelists still differ: perturbation sponge (9 levels) vs trajectory sponge
    "nord2_h10": dict(COEF, nord=2, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10, hord_tr=10),   # with split_hord on top
}


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    return Case(nx=12, ny=10, npz=12, n_split=2, k_split=2, dt=1800.0, backend="emul", nq=1, **CASES[request.param])


@pytest.mark.parametrize("mode", [TL, AD])
def test_d_sw_group(case, mode):
    check_group(case, "d_sw", mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_dyn_core(case, mode):
    check_dyn_core(case, mode, 1e-10)


def test_dot_product(case):
    lhs, rhs = dot_product_step(case)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


@pytest.fixture(scope="module", params=["coef", "nord2", "nord3", "nord0p"])
def fcase(request):
    return Case(nx=12, ny=12, npz=12, n_split=2, dt=1800.0, backend="emul", face=2, nq=0, **CASES[request.param])


@pytest.mark.parametrize("mode", [TL, AD])
def test_face_d_sw_group(fcase, mode):
    """cube face: fill_corners of divg_d and of the difference pair (nord 2, 3), copy_corners views of the iterated Laplacians"""
    check_group(fcase, "d_sw", mode, 1e-12 if mode == TL else 1e-11)


@pytest.fixture(scope="module", params=["nord2", "nord3"])
def ccase(request):
    return CubeCase(n=8, npz=12, n_split=2, k_split=2, backend="emul", oracle=True, nq=1, **CASES[request.param])


@pytest.mark.parametrize("mode", [TL, AD])
def test_cube_fv_dynamics(ccase, mode):
    from groups import cube_check_fv_dynamics
    cube_check_fv_dynamics(ccase, mode, 1e-10)


def test_cube_step_dot_product(ccase):
    from groups import cube_dot_product_step
    lhs, rhs = cube_dot_product_step(ccase)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_step_nl():
    check_step_nl(Case(nx=12, ny=10, npz=12, n_split=2, k_split=2, dt=1800.0, backend="emul", nq=1, **CASES["nord2"]), 1e-10)


def test_the_perturbation_coefficients_are_read():
    """dddmp_pert / d4_bg_pert were fields nothing read.  Same trajectory, two perturbation coefficient sets: the
    tangent must differ, the values must not."""
    from groups import step_state
    out = []
    for kw in (dict(COEF), dict(COEF, dddmp_pert=0.5, d4_bg_pert=0.05)):
        c = Case(nx=12, ny=10, npz=12, n_split=2, k_split=1, dt=1800.0, backend="emul", oracle=False, **kw)
        T, P = step_state(c)
        for n in ("u", "v", "pt", "delp"):
            c.dy.put(n, T[n][None], 0); c.dy.put(n, P[n][None], 1)
        c.dy.step_tl()
        out.append(({n: c.dy.get(n, 0)[0].copy() for n in ("u", "v")}, {n: c.dy.get(n, 1)[0].copy() for n in ("u", "v")}))
    (v0, t0), (v1, t1) = out
    I = (Ellipsis, slice(3, 3 + 10), slice(3, 3 + 12))
    assert max(np.max(np.abs(v0[n][I] - v1[n][I])) for n in v0) == 0.0
    assert max(np.max(np.abs(t0[n][I] - t1[n][I])) / np.max(np.abs(t0[n][I])) for n in t0) > 1e-6


def test_split_damp_off_refuses_two_coefficient_sets():
    """split_damp = 0: run_setup_pert has made the two sets equal (fv_control_tlmadm.F90:220-229); anything else is refused, not guessed"""
    for kw in (dict(dddmp_pert=0.35), dict(d4_bg_pert=0.11), dict(nord=0), dict(vtdm4=0.01), dict(d2_bg_k1=0.3)):
        with pytest.raises(RuntimeError, match="split_damp"):
            Case(nx=12, ny=10, npz=6, backend="emul", oracle=False, **kw)


def test_refusals():
    for kw, msg in ((dict(split_damp=1, nord=0, nord_pert=1), "nord = 0"), (dict(split_damp=1, nord_pert=2, nord=2), "nord_pert"),
                    (dict(split_damp=1, nord=4), "nord in 0..3"), (dict(split_damp=1, nord=2, hydrostatic=0), "non-hydrostatic")):
        with pytest.raises(RuntimeError, match=msg):
            Case(nx=12, ny=10, npz=6, backend="emul", oracle=False, **kw)
