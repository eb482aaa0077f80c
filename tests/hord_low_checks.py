"""Shared by test_emul_hord_low.py / test_gpu_hord_low.py: trajectory advection schemes 3 .. 7 (the limited low-order schemes of the nonlinear
xppm / yppm, tp_core_tlm.F90:442-590, and xtp_u / ytp_v, sw_core_tlm.F90:4483-4612) and 9, 11, 12, 13 (the variants of the monotone schemes:
tp_core_tlm.F90:679-715, :826-828; sw_core_tlm.F90:4710-4780, :4890-4896) beside a perturbation scheme of {1, 2, 333} (split_hord).
Which cells keep their parabola is decided by tests on the slopes (smt5 / smt6), so the checks run on ROUGH fields: grid-scale noise on the
trajectory makes every branch of every scheme fire (asserted below by the schemes giving pairwise different steps)."""
import numpy as np


def hord_kw(h, pert=2):
    kw = dict(hord_mt=h, hord_vt=h, hord_tm=h, hord_dp=h, hord_tr=h)
    if pert != 2:
        kw.update({"hord_%s_pert" % n: pert for n in ("mt", "vt", "tm", "dp", "tr")})
    return kw


def roughen(c, seed=7, periodic=True, qamp=0.3):
    """grid-scale noise on the trajectory of a single-tile case (periodic images kept equal) or of a six-face case (compute domains; the halos are
    exchanged by product and oracle alike)"""
    rng = np.random.default_rng(seed)

    def noise(a):
        nk, pj, pi = a.shape[-3:]
        if periodic:
            r = rng.standard_normal((nk, c.ny, c.nx))
            jj = (np.arange(pj) - 3) % c.ny; ii = (np.arange(pi) - 3) % c.nx
            return r[:, jj][:, :, ii]
        return rng.standard_normal(a.shape)
    for n in ("pt", "delp"):
        c.traj[n] = c.traj[n] * (1.0 + 0.01 * noise(c.traj[n]))
    for n in ("u", "v"):
        r = noise(c.traj[n])
        if not periodic:      # the D-grid rows two faces share must stay one value on both: no noise on them
            if n == "u":
                r[..., 3, :] = 0.0; r[..., 3 + c.ny, :] = 0.0
            else:
                r[..., :, 3] = 0.0; r[..., :, 3 + c.nx] = 0.0
        c.traj[n] = c.traj[n] + 2.0 * r
    for m in range(c.nq):
        c.qtraj[m] = c.qtraj[m] * (1.0 + qamp * noise(c.qtraj[m]))      # qamp > 1: zero crossings, for the positive-definite constraint of 9 / 13
    return c


def nl_step(c):
    from groups import step_state
    T, _ = step_state(c)
    names = ["u", "v", "pt", "delp"] + ["q%d" % (m + 1) for m in range(c.nq)]
    for n in names:
        c.dy.put(n, T[n][None], 0)
    c.dy.step_nl()
    return {n: c.dy.get(n, 0)[0].copy() for n in names}
