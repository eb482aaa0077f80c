"""fv_tp_2d as one LDS-tiled launch (csrc/tpfused.h) against the staged launches it replaces in the nonlinear and tangent modes
(FV3LM_TP_FUSED=0): the two forms run the same arithmetic in the same order, so every field must agree BIT FOR BIT -- step_tl
outputs (trajectory and tangent), and step_ad inputs (the adjoint stays staged and reads the intermediates the fused nonlinear
launch stored).  Tile sizes are chosen so that a face is cut into several 64 x 16 blocks in both directions."""
import os
import numpy as np
from groups import step_state, cube_step_state


FUSED_VALUE = "1"        # which fused form stands against the staged launches (1: tiled default, 3: marching)


def _run(make_case, fused):
    old = os.environ.get("FV3LM_TP_FUSED")
    os.environ["FV3LM_TP_FUSED"] = FUSED_VALUE if fused else "0"
    try:
        c = make_case()
    finally:
        if old is None:
            os.environ.pop("FV3LM_TP_FUSED", None)
        else:
            os.environ["FV3LM_TP_FUSED"] = old
    cube = getattr(c, "face", None) == "cube"
    T, P = cube_step_state(c) if cube else step_state(c)
    if not cube:
        T = {k: v[None] for k, v in T.items()}; P = {k: v[None] for k, v in P.items()}
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
    if not c.opt.hydrostatic:          # w, delz prognostic: balanced state of the non-hydrostatic checks
        import nh_checks
        from test_oracle_nh import nh_state_fv
        names = nh_checks.fv_names(c)
        Tl, Pl = nh_state_fv(c)
        T = {n: t[None] for n, t in zip(names, Tl)}; P = {n: p[None] for n, p in zip(names, Pl)}
    for n in names:
        c.dy.put(n, T[n], 0); c.dy.put(n, P[n], 1)
    c.dy.step_tl()
    out = {("tl", n, w): c.dy.get(n, w).copy() for n in names for w in (0, 1)}
    rng = np.random.default_rng(5)
    for n in names:
        c.dy.put(n, T[n], 0)
    c.dy.step_nl()
    for n in names:
        c.dy.put(n, rng.standard_normal(c.dy.shape(n)), 1)
    c.dy.step_ad()
    out.update({("ad", n, 1): c.dy.get(n, 1).copy() for n in names})
    launches = c.dy.launch_count() if hasattr(c.dy, "launch_count") else None
    return out, launches


def check_fused_equals_staged(make_case, rtol=0.0):
    """rtol = 0: bit for bit (host emulation: one compiler, no contraction).  On the device the two forms are different kernels and
    hipcc contracts a*b+c into fused multiply-adds per kernel, so agreement there is to rounding (rtol 1e-12), not bitwise."""
    a, la = _run(make_case, True)
    b, lb = _run(make_case, False)
    worst = 0.0
    for key in a:
        assert np.isfinite(a[key]).all() and np.abs(a[key]).max() > 0, key
        if rtol == 0.0 and key[0] != "ad":
            assert np.array_equal(a[key], b[key]), key
        elif rtol == 0.0:       # the fused outer adjoint sums in another order than the staged gather: rounding, not bits
            e = float(np.max(np.abs(a[key] - b[key])) / np.max(np.abs(b[key])))
            assert e <= 1e-13, (key, e)
        else:
            e = float(np.max(np.abs(a[key] - b[key])) / np.max(np.abs(b[key])))
            assert e <= rtol, (key, e)
            worst = max(worst, e)
    return worst
