"""split_hord: a trajectory advection scheme that differs from the perturbation scheme -- incl. the monotone
iord 8 / 10 of the nonlinear xppm / yppm (tp_core_tlm.F90:592-955) -- for the fv_tp_2d transports.  The reference runs the _TLM
routine with the perturbation scheme, then the nonlinear routine with the trajectory scheme for the values
(sw_core_tlm.F90:1664-1682); product (fused kernel + a values-only second pass) against the oracle (tp_core.hpp fv_tp_2d_split,
tp_mono.hpp), on the host-emulation build.  The monotone schemes have no reference-held fixtures: parity unpinned, like the path."""
import numpy as np
import pytest
from common import Case, CubeCase
from groups import check_group, check_dyn_core, check_fv_dynamics, check_tracer, dot_product_step
from oracle import TL, AD, NL

SPLIT10 = dict(hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10, hord_tr=10)      # perturbation schemes stay at their default (2; 1 in the sponge)
SPLIT8 = dict(hord_mt=8, hord_vt=8, hord_tm=8, hord_dp=8, hord_tr=8, hord_mt_pert=333, hord_vt_pert=333, hord_tm_pert=333, hord_dp_pert=333, hord_tr_pert=333)


@pytest.fixture(scope="module", params=["h10", "h8"])
def case(request):
    kw = SPLIT10 if request.param == "h10" else SPLIT8
    return Case(nx=12, ny=10, npz=12, n_split=2, k_split=2, dt=1800.0, backend="emul", nq=2, **kw)      # 12 levels: sponge + regular


@pytest.mark.parametrize("mode", [TL, AD])
def test_d_sw_group(case, mode):
    check_group(case, "d_sw", mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_dyn_core(case, mode):
    check_dyn_core(case, mode, 1e-10)


@pytest.mark.parametrize("mode", [TL, AD])
def test_tracer_2d(case, mode):
    check_tracer(case, mode, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_fv_dynamics(case, mode):
    check_fv_dynamics(case, mode, 1e-10)


def test_dot_product(case):
    lhs, rhs = dot_product_step(case)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_step_nl(case):
    """f2: the nonlinear step runs the TRAJECTORY schemes (monotone here), as the reference's step_nl does"""
    from groups import check_step_nl
    check_step_nl(case, 1e-10)


def test_trajectory_differs_from_the_unsplit_one():
    """the second pass does something: the nonlinear step with hord 10 is not the one with hord 2"""
    from groups import step_state
    out = []
    for kw in (SPLIT10, {}):
        c = Case(nx=12, ny=10, npz=12, n_split=2, k_split=1, dt=1800.0, backend="emul", oracle=False, **kw)
        T, _ = step_state(c)
        for n in ("u", "v", "pt", "delp"):
            c.dy.put(n, T[n][None], 0)
        c.dy.step_nl()
        out.append({n: c.dy.get(n, 0)[0].copy() for n in ("pt", "delp")})
    a, b = out
    assert max(np.max(np.abs(a[n] - b[n])) / np.max(np.abs(b[n])) for n in a) > 1e-7


# ---- cube faces: the three cells each side of a cube edge have their own monotone slopes (tp_core_tlm.F90:828-942)
@pytest.fixture(scope="module", params=["h10", "h8"])
def fcase(request):
    kw = SPLIT10 if request.param == "h10" else SPLIT8
    return Case(nx=12, ny=12, npz=12, n_split=2, dt=1800.0, backend="emul", face=2, nq=2, **kw)


@pytest.mark.parametrize("mode", [TL, AD])
def test_face_d_sw_group(fcase, mode):
    check_group(fcase, "d_sw", mode, 1e-12 if mode == TL else 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_face_tracer(fcase, mode):
    check_tracer(fcase, mode, 1e-11 if mode == TL else 1e-10)


@pytest.fixture(scope="module")
def ccase():
    return CubeCase(n=8, npz=12, n_split=2, k_split=2, backend="emul", oracle=True, nq=2, **SPLIT10)


@pytest.mark.parametrize("mode", [TL, AD])
def test_cube_fv_dynamics(ccase, mode):
    from groups import cube_check_fv_dynamics
    cube_check_fv_dynamics(ccase, mode, 1e-10)


def test_cube_step_dot_product(ccase):
    from groups import cube_dot_product_step
    lhs, rhs = cube_dot_product_step(ccase)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


# ---- non-hydrostatic: w transport (hord_vt) and the height transport of update_dz_d (the global hord_tm, nh_utils_tlm.F90:496-560)
@pytest.fixture(scope="module")
def nhcase():
    return Case(nx=10, ny=8, npz=12, n_split=2, k_split=2, dt=1200.0, nq=2, backend="emul", hydrostatic=0, **SPLIT10)


def test_nh_fv_dynamics(nhcase):
    import nh_checks as N
    N.check_nh_fv_tangent(nhcase)
    N.check_nh_fv_adjoint(nhcase)
    N.check_nh_fv_dot_product(nhcase)


def test_nh_cube():
    import nh_checks as N
    c = CubeCase(n=12, npz=11, n_split=2, k_split=1, dt=600.0, nq=1, backend="emul", oracle=True, hydrostatic=0, **SPLIT10)
    N.cube_check_nh_fv(c, TL)
    N.cube_check_nh_fv(c, AD)
    N.cube_check_nh_dot_product(c)


# ---- split_kord: the trajectory remapped with the limited profiles (cs_profile / scalar_profile, kord 9 / 10 / 11), the perturbation with
#      the linear one (fv_mapz_tlm.F90:494-523, :596-637, :780-827)
@pytest.fixture(scope="module", params=[8, 9, 10, 11, 12, 13, 14, 15])
def kcase(request):
    k = request.param
    return Case(nx=12, ny=10, npz=14, n_split=2, k_split=2, dt=1800.0, backend="emul", nq=3, kord_tm=-k, kord_mt=k, kord_tr=k, kord_wz=17)


@pytest.mark.parametrize("mode", [TL, AD])
@pytest.mark.parametrize("last", [0, 1])
def test_kord_remap(kcase, mode, last):
    from groups import check_remap
    check_remap(kcase, mode, last, 1e-11)


@pytest.mark.parametrize("mode", [TL, AD])
def test_kord_fv_dynamics(kcase, mode):
    check_fv_dynamics(kcase, mode, 1e-10)


def test_kord_step_nl_and_dot_product(kcase):
    from groups import check_step_nl
    check_step_nl(kcase, 1e-10)
    lhs, rhs = dot_product_step(kcase)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


def test_operational_pairing_on_the_cube():
    """trajectory hord 10 / kord 9, perturbation hord 2 (1 in the sponge) / kord 17: six faces, whole step"""
    from groups import cube_check_fv_dynamics, cube_dot_product_step
    c = CubeCase(n=8, npz=12, n_split=2, k_split=2, backend="emul", oracle=True, nq=2, kord_tm=-9, kord_mt=9, kord_tr=9, **SPLIT10)
    cube_check_fv_dynamics(c, TL, 1e-10)
    cube_check_fv_dynamics(c, AD, 1e-10)
    lhs, rhs = cube_dot_product_step(c)
    assert abs(lhs - rhs) <= 1e-12 * abs(lhs), (lhs, rhs)


@pytest.mark.parametrize("kord", [8, 9, 10, 11, 12, 13, 14, 15])
def test_kord_remap_noisy_columns(monkeypatch, kord):
    """columns with 2-delta-z noise, local extrema, zero and negative tracer values: every branch of the limited profiles (extrema tests,
    Huynh's constraint, the positive-definite limiter, q < qmin) -- where the profiles 8 .. 15 differ from each other"""
    import groups
    c = Case(nx=12, ny=10, npz=16, n_split=1, k_split=1, dt=900.0, backend="emul", nq=3, kord_tm=-kord, kord_mt=kord, kord_tr=kord)
    rng = np.random.default_rng(5)
    orig = groups._dyn_outputs

    def noisy(cc):
        T, P = orig(cc)
        saw = np.where(np.arange(T["pt"].shape[0])[:, None, None] % 2 == 0, 1.0, -1.0)
        T["pt"] = T["pt"] * (1.0 + 0.02 * saw * rng.random(T["pt"].shape))
        T["pt"][3:6] *= 0.55                                   # below t_min = 184 K in places (T_v = pt * pkz)
        T["u"] = T["u"] + 4.0 * saw[: T["u"].shape[0]] * rng.random(T["u"].shape) - 2.0
        T["v"] = T["v"] * (1.0 - 1.5 * rng.random(T["v"].shape))     # sign changes
        return T, P
    monkeypatch.setattr(groups, "_dyn_outputs", noisy)
    for n in range(c.nq):
        q = c.qtraj[n][0]
        saw = np.where(np.arange(q.shape[0])[:, None, None] % 2 == 0, 1.0, -1.0)
        q *= (1.0 + 0.9 * saw * rng.random(q.shape))
        q[rng.random(q.shape) < 0.1] = 0.0
        q[rng.random(q.shape) < 0.03] *= -1.0
    groups.check_remap(c, TL, 0, 1e-11)
    groups.check_remap(c, TL, 1, 1e-11)
    groups.check_remap(c, AD, 1, 1e-11)


def test_nh_split_kord():
    """non-hydrostatic remap with limited trajectory profiles: T (map_scalar), tracers, delz (iv = 1) and w (iv = -2, kord_wz)"""
    import nh_checks as N
    c = Case(nx=10, ny=8, npz=12, n_split=2, k_split=2, dt=1200.0, nq=2, backend="emul", hydrostatic=0, kord_tm=-9, kord_mt=9, kord_tr=9, kord_wz=9, **SPLIT10)
    N.check_nh_fv_tangent(c)
    N.check_nh_fv_adjoint(c)
    N.check_nh_fv_dot_product(c)
    c = Case(nx=10, ny=8, npz=12, n_split=2, k_split=1, dt=600.0, nq=1, backend="emul", hydrostatic=0, kord_tm=-10, kord_mt=10, kord_tr=11, kord_wz=10)
    N.check_nh_fv_tangent(c)
    N.check_nh_fv_adjoint(c)
    c = Case(nx=10, ny=8, npz=12, n_split=2, k_split=1, dt=600.0, nq=1, backend="emul", hydrostatic=0, kord_tm=-13, kord_mt=12, kord_tr=14, kord_wz=8)
    N.check_nh_fv_tangent(c)
    N.check_nh_fv_adjoint(c)


def test_every_limited_profile_is_its_own():
    """on columns with grid-scale noise the trajectory profiles 8 .. 14 give pairwise different nonlinear steps; 15 takes the branch of 11
    (model/fv_mapz_nlm.F90:2425: the ELSE of the chain) and equals it"""
    from hord_low_checks import roughen, nl_step
    out = {}
    for k in (8, 9, 10, 11, 12, 13, 14, 15, 17):
        c = roughen(Case(nx=12, ny=10, npz=16, n_split=1, k_split=1, dt=900.0, backend="emul", oracle=False, nq=1, kord_tm=-k, kord_mt=k, kord_tr=k), qamp=0.8)
        out[k] = nl_step(c)
    ks = sorted(out)
    for a in ks:
        for b in ks:
            if a < b:
                d = max(np.max(np.abs(out[a][n] - out[b][n])) / np.max(np.abs(out[b][n])) for n in ("pt", "u", "q1"))
                if (a, b) == (11, 15):
                    assert d == 0.0, d
                else:
                    assert d > 1e-9, (a, b, d)
