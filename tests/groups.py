"""Kernel-group tables shared by the emulation (CPU) and GPU parity tests: for each group of the
acoustic step the product field names, the oracle entry point and the index ranges that carry
defined data (Fortran index ranges of the reference routines, cited in oracle/*.hpp)."""
import numpy as np
from oracle import NL, TL, AD
from common import relerr


def rects(c):
    nx, ny = c.nx, c.ny
    isd, ied, jsd, jed = -2, nx + 3, -2, ny + 3
    full = (isd, ied + 1, jsd, jed + 1)
    return dict(
        A=(1, nx, 1, ny), Ah=(0, nx + 1, 0, ny + 1), Ah2=(-1, nx + 2, -1, ny + 2),
        U=(1, nx, 1, ny + 1), V=(1, nx + 1, 1, ny), B=(1, nx + 1, 1, ny + 1),
        CX=(1, nx + 1, jsd, jed), CY=(isd, ied, 1, ny + 1), UTF=(0, nx + 2, 0, ny + 1), VTF=(0, nx + 1, 0, ny + 2),
        full=full)


# group -> (oracle method, scalar maker, [(oracle-in order) product field], [(product out field, rect key)])
def table(c):
    dt = c.dt_ac
    return {
        "c_sw": ("c_sw", lambda: (0.5 * dt,), ["delp", "pt", "u", "v"],
                 [("delpc", "Ah"), ("ptc", "Ah"), ("uc1", "V"), ("vc1", "U"), ("ua", "Ah"), ("va", "Ah"), ("utf", "UTF"),
                  ("vtf", "VTF"), ("divgd", "B")]),
        "geopk_c": ("geopk", lambda: (1,), ["delpc", "ptc"],
                    [("pe_c", "Ah"), ("peln_c", "A"), ("pkc", "Ah"), ("gz", "Ah"), (None, None)]),
        "p_grad_c": ("p_grad_c", lambda: (0.5 * dt,), ["pkc", "gz", "uc1", "vc1"], [("uc", "V"), ("vc", "U")]),
        "d_sw": ("d_sw", lambda: (dt,), ["delp", "pt", "u", "v", "uc", "vc", "ua", "va", "divgd", "mfx", "mfy", "cx", "cy"],
                 [("delp_o", "A"), ("pt_o", "A"), ("u_m", "U"), ("v_m", "V"), ("mfx", "V"), ("mfy", "U"), ("cx", "CX"),
                  ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"), ("yfx", "CY")]),
        "geopk_d": ("geopk", lambda: (0,), ["delp_o", "pt_o"],
                    [("pe", "Ah"), ("peln", "A"), ("pk", "Ah2"), ("gzd", "Ah2"), ("pkz", "A")]),
        "one_grad_p": ("one_grad_p", lambda: (dt,), ["u_m", "v_m", "pk", "gzd"], [("u_o", "U"), ("v_o", "V")]),
    }


def masked(c, arr, rkey):
    out = np.zeros_like(arr)
    r = c.rect(*rects(c)[rkey])
    out[r] = arr[r]
    return out


def drop_face_corners(c, name, arr):
    """ua/va at the four corner-halo cells of a face: the reference overwrites them in place with the rotated
    neighbour values while d2a2c_vect runs (sw_core_tlm.F90:6662-6677, :6745-6760) and nothing reads them
    afterwards; the product reads those views through an index map and leaves the stored value alone."""
    if getattr(c, "face", None) is None:
        return arr
    if name in ("pe_c", "pkc", "gz", "pe", "pk", "gzd"):    # geopk: corner-halo columns hold no exchanged data, product skips them
        out = arr.copy()
        for js in (slice(0, 3), slice(c.ny + 3, None)):
            for isl in (slice(0, 3), slice(c.nx + 3, None)):
                out[..., js, isl] = 0.0
        return out
    if name not in ("ua", "va"):
        return arr
    out = arr.copy()
    for (i, j) in ((0, 0), (c.nx + 1, 0), (0, c.ny + 1), (c.nx + 1, c.ny + 1)):
        out[..., j + 2, i + 2] = 0.0
    return out


def make_inputs(c, group, seed=7):
    """Physically plausible inputs of a group = outputs of the preceding groups run by the oracle
    (NL+TL), halo-filled where the reference would have exchanged them."""
    from fv3_jedi_linearmodel_amd.grid import halo_fill_periodic as hf
    nx, ny, dt = c.nx, c.ny, c.dt_ac
    T = {n: c.traj[n][0] for n in ("delp", "pt", "u", "v")}
    P = {n: c.pert[n][0] for n in ("delp", "pt", "u", "v")}
    if group == "c_sw":
        return T, P
    ot, op = c.oracle.c_sw(TL, 0.5 * dt, [T[n] for n in ("delp", "pt", "u", "v")], [P[n] for n in ("delp", "pt", "u", "v")])
    for n, a, b in zip(["delpc", "ptc", "uc1", "vc1", "ua", "va", "utf", "vtf", "divgd"], ot, op):
        T[n], P[n] = a, b
    T["divgd"], P["divgd"] = hf(T["divgd"], nx, ny), hf(P["divgd"], nx, ny)
    if group == "geopk_c":
        return T, P
    ot, op = c.oracle.geopk(TL, 1, [T["delpc"], T["ptc"]], [P["delpc"], P["ptc"]])
    T["pkc"], P["pkc"], T["gz"], P["gz"] = ot[2], op[2], ot[3], op[3]
    if group == "p_grad_c":
        return T, P
    ot, op = c.oracle.p_grad_c(TL, 0.5 * dt, [T["pkc"], T["gz"], T["uc1"], T["vc1"]], [P["pkc"], P["gz"], P["uc1"], P["vc1"]])
    T["uc"], P["uc"], T["vc"], P["vc"] = hf(ot[0], nx, ny), hf(op[0], nx, ny), hf(ot[1], nx, ny), hf(op[1], nx, ny)
    rng = np.random.default_rng(seed)
    for n in ("mfx", "mfy", "cx", "cy"):
        T[n] = rng.standard_normal(T["u"].shape); P[n] = rng.standard_normal(T["u"].shape)
    if group == "d_sw":
        return T, P
    names = ["delp", "pt", "u", "v", "uc", "vc", "ua", "va", "divgd", "mfx", "mfy", "cx", "cy"]
    ot, op = c.oracle.d_sw(TL, dt, [T[n] for n in names], [P[n] for n in names])
    for n, a, b in zip(["delp_o", "pt_o", "u_m", "v_m"], ot[:4], op[:4]):
        T[n], P[n] = a, b
    for n in ("delp_o", "pt_o"):
        T[n], P[n] = hf(T[n], nx, ny), hf(P[n], nx, ny)
    if group == "geopk_d":
        return T, P
    ot, op = c.oracle.geopk(TL, 0, [T["delp_o"], T["pt_o"]], [P["delp_o"], P["pt_o"]])
    T["pk"], P["pk"], T["gzd"], P["gzd"] = ot[2], op[2], ot[3], op[3]
    return T, P


def check_group(c, group, mode, tol):
    meth, scal, ins, outs = table(c)[group]
    T, P = make_inputs(c, group)
    fn = getattr(c.oracle, meth)
    i_t = [T[n] for n in ins]; i_p = [P[n] for n in ins]
    if mode == TL:
        ot, op = fn(TL, *scal(), i_t, i_p)
        for n in ins:
            c.dy.put(n, T[n][None], 0); c.dy.put(n, P[n][None], 1)
        c.dy.run_group(group, TL)
        worst = 0.0
        for (n, rk), a, b in zip(outs, ot, op):
            if n is None:
                continue
            r = c.rect(*rects(c)[rk])
            D = lambda x: drop_face_corners(c, n, x)
            e1, e2 = relerr(D(c.dy.get(n, 0)[0])[r], D(a)[r]), relerr(D(c.dy.get(n, 1)[0])[r], D(b)[r])
            assert e1 < tol, (group, n, "traj", e1)
            assert e2 < tol, (group, n, "tl", e2)
            worst = max(worst, e1, e2)
        return worst
    # adjoint: random output adjoints on the defined ranges
    rng = np.random.default_rng(11)
    nks = {n: c.dy.levels(n) for n, _ in outs if n}
    seeds = []
    for n, rk in outs:
        if n is None:
            seeds.append(np.zeros((c.npz, c.ny + 7, c.nx + 7)))
        else:
            seeds.append(drop_face_corners(c, n, masked(c, rng.standard_normal((nks[n], c.ny + 7, c.nx + 7)), rk)))
    _, iad = fn(AD, *scal(), i_t, None, seeds)
    for n in ins:
        c.dy.put(n, T[n][None], 0)
    c.dy.run_group(group, NL)
    c.dy.zero_work_adjoint()
    for n in set(ins) | {o for o, _ in outs if o}:
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for (n, rk), s in zip(outs, seeds):
        if n:
            c.dy.put(n, s[None], 1)
    c.dy.run_group(group, AD)
    worst = 0.0
    for n, a in zip(ins, iad):
        got = c.dy.get(n, 1)[0]
        if n in [o for o, _ in outs]:      # accumulators: in-place, adjoint passes through
            pass
        e = relerr(got, a)
        assert e < tol, (group, n, "ad", e)
        worst = max(worst, e)
    return worst


DC_OUT = [("u", "full"), ("v", "full"), ("pt", "full"), ("delp", "full"), ("mfx", "V"), ("mfy", "U"), ("cx", "CX"),
          ("cy", "CY"), ("pe", "Ah"), ("peln", "A"), ("pk", "A"), ("pkz", "A")]


def check_dyn_core(c, mode, tol):
    """n_split acoustic steps: product fv3lm_dyn_core vs oracle orc_dyn_core (DYN_CORE_TLM /
    DYN_CORE_FWD+BWD)."""
    ins = ["u", "v", "pt", "delp"]
    i_t = [c.traj[n][0] for n in ins]; i_p = [c.pert[n][0] for n in ins]
    bdt, ns = c.dims.dt / c.dims.k_split, c.dims.n_split
    if mode == TL:
        ot, op = c.oracle.dyn_core(TL, bdt, ns, i_t, i_p)
        c.put_state(pert=c.pert)
        c.dy.dyn_core(TL)
        worst = 0.0
        for (n, rk), a, b in zip(DC_OUT, ot, op):
            r = c.rect(*rects(c)[rk])
            e1, e2 = relerr(c.dy.get(n, 0)[0][r], a[r]), relerr(c.dy.get(n, 1)[0][r], b[r])
            assert e1 < tol, (n, "traj", e1)
            assert e2 < tol, (n, "tl", e2)
            worst = max(worst, e1, e2)
        return worst
    rng = np.random.default_rng(5)
    seeds = [masked(c, rng.standard_normal((c.dy.levels(n), c.ny + 7, c.nx + 7)), rk) for n, rk in DC_OUT]
    _, iad = c.oracle.dyn_core(AD, bdt, ns, i_t, None, seeds)
    c.put_state()
    c.dy.dyn_core(NL)            # forward sweep: stores the per-step checkpoints
    for (n, rk), s in zip(DC_OUT, seeds):
        c.dy.put(n, s[None], 1)
    c.dy.dyn_core(AD)
    worst = 0.0
    for n, a in zip(ins, iad):
        e = relerr(c.dy.get(n, 1)[0], a)
        assert e < tol, (n, "ad", e)
        worst = max(worst, e)
    return worst


def dot_product_test(c, seed=3):
    """<M dx, dy> = <dx, M^T dy> for the n_split-step dyn_core operator (the JEDI LinearModel test)."""
    rng = np.random.default_rng(seed)
    ins = ["u", "v", "pt", "delp"]
    outs = [("u", "U"), ("v", "V"), ("pt", "A"), ("delp", "A")]
    c.put_state(pert=c.pert)
    c.dy.dyn_core(TL)
    Mdx = {n: c.dy.get(n, 1)[0] for n, _ in outs}
    dy = {n: masked(c, rng.standard_normal(Mdx[n].shape), rk) for n, rk in outs}
    lhs = sum(float(np.sum(Mdx[n] * dy[n])) for n, _ in outs)
    c.put_state()
    c.dy.dyn_core(NL)
    for n in ("mfx", "mfy", "cx", "cy", "pe", "peln", "pk", "pkz"):
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for n, _ in outs:
        c.dy.put(n, dy[n][None], 1)
    c.dy.dyn_core(AD)
    rhs = sum(float(np.sum(c.dy.get(n, 1)[0] * c.pert[n][0])) for n in ins)
    return lhs, rhs


# ------------------------------------------------------------------------------ fv_dynamics level
def np_pressures(c, delp):
    """compute_fv3_pressures (fv_pressure.F90:23-72) in numpy on the padded plane (all points)."""
    k = c.opt.akap
    pe = np.concatenate([np.full_like(delp[:1], c.opt.ptop), c.opt.ptop + np.cumsum(delp, axis=0)], axis=0)
    peln = np.log(pe); pk = np.exp(k * peln)
    pkz = (pk[1:] - pk[:-1]) / (k * (peln[1:] - peln[:-1]))
    return pe, peln, pk, pkz


def _dyn_outputs(c):
    """oracle dyn_core (TL) from the case state: realistic mfx.., pe.., state after the acoustic steps."""
    ins = ["u", "v", "pt", "delp"]
    ot, op = c.oracle.dyn_core(TL, c.dims.dt / c.dims.k_split, c.dims.n_split, [c.traj[n][0] for n in ins], [c.pert[n][0] for n in ins])
    names = ["u", "v", "pt", "delp", "mfx", "mfy", "cx", "cy", "pe", "peln", "pk", "pkz"]
    return dict(zip(names, ot)), dict(zip(names, op))


def check_tracer(c, mode, tol, scale=1.0):
    """scale > 1 multiplies the accumulated Courant numbers and mass fluxes so that max Courant >= 1 and tracer_2d
    sub-cycles (nsplt > 1, levels with different sub-step counts)."""
    T, P = _dyn_outputs(c)
    nq = c.nq
    ins_n = ["dp1", "mfx", "mfy", "cx", "cy"] + ["q%d" % (n + 1) for n in range(nq)]
    for n in ("mfx", "mfy", "cx", "cy"):
        T[n], P[n] = scale * T[n], scale * P[n]
    T["dp1"], P["dp1"] = c.traj["delp"][0], c.pert["delp"][0]
    for n in range(nq):
        T["q%d" % (n + 1)], P["q%d" % (n + 1)] = c.qtraj[n][0], c.qpert[n][0]
    i_t = [T[n] for n in ins_n]; i_p = [P[n] for n in ins_n]
    A = c.rect(*rects(c)["A"])
    for n in ins_n:
        c.dy.put(n, T[n][None], 0)
    if mode == TL:
        ot, op = c.oracle.tracer_2d(TL, nq, i_t, i_p)
        for n in ins_n:
            c.dy.put(n, P[n][None], 1)
        c.dy.tracer_2d(TL)
        for n in range(nq):
            nm = "q%d" % (n + 1)
            assert relerr(c.dy.get(nm, 0)[0][A], ot[n][A]) < tol, (nm, "traj")
            assert relerr(c.dy.get(nm, 1)[0][A], op[n][A]) < tol, (nm, "tl")
        return
    rng = np.random.default_rng(21)
    seeds = [masked(c, rng.standard_normal(T["dp1"].shape), "A") for _ in range(nq)]
    _, iad = c.oracle.tracer_2d(AD, nq, i_t, None, seeds)
    if scale != 1.0:      # forward sweep first: it stores the sub-step trajectory the adjoint replays
        c.dy.tracer_2d(NL)
        for n in ins_n:
            c.dy.put(n, T[n][None], 0)
    for n in ins_n:
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for n in range(nq):
        c.dy.put("q%d" % (n + 1), seeds[n][None], 1)
    c.dy.tracer_2d(AD)
    for n, a in zip(ins_n, iad):
        e = relerr(c.dy.get(n, 1)[0], a)
        assert e < tol, (n, "ad", e)


def check_remap(c, mode, last_step, tol):
    T, P = _dyn_outputs(c)
    nq = c.nq
    ins_n = ["pe", "peln", "pk", "pt", "delp", "u", "v"] + ["q%d" % (n + 1) for n in range(nq)]
    for n in range(nq):
        T["q%d" % (n + 1)], P["q%d" % (n + 1)] = c.qtraj[n][0], c.qpert[n][0]
    # the oracle's pe is defined on 0..nx+1, peln/pk on the compute domain: zero the rest on both sides
    for d in (T, P):
        d["pe"] = masked(c, d["pe"], "Ah"); d["peln"] = masked(c, d["peln"], "A"); d["pk"] = masked(c, d["pk"], "A")
    T["pe"] = np.where(T["pe"] == 0, 1.0, T["pe"]); T["peln"] = np.where(T["peln"] == 0, 1.0, T["peln"])
    i_t = [T[n] for n in ins_n]; i_p = [P[n] for n in ins_n]
    outs = [("pe", "A"), ("peln", "A"), ("pk", "A"), ("pkz", "A"), ("pt", "A"), ("delp", "A"), ("u", "U"), ("v", "V")] + \
           [("q%d" % (n + 1), "A") for n in range(nq)]
    for n in ins_n:
        c.dy.put(n, T[n][None], 0)
    if mode == TL:
        ot, op = c.oracle.remap(TL, nq, last_step, i_t, i_p)
        for n in ins_n:
            c.dy.put(n, P[n][None], 1)
        c.dy.remap(TL, last_step)
        for (n, rk), a, b in zip(outs, ot, op):
            r = c.rect(*rects(c)[rk])
            e1, e2 = relerr(c.dy.get(n, 0)[0][r], a[r]), relerr(c.dy.get(n, 1)[0][r], b[r])
            assert e1 < tol, (n, "traj", e1)
            assert e2 < tol, (n, "tl", e2)
        return
    rng = np.random.default_rng(31)
    seeds = []
    for n, rk in outs:
        s = masked(c, rng.standard_normal((c.dy.levels(n), c.ny + 7, c.nx + 7)), rk)
        if n == "pe":
            s[:] = 0.0       # pe after the remap is dead in fv_dynamics (overwritten by the next geopk)
        seeds.append(s)
    _, iad = c.oracle.remap(AD, nq, last_step, i_t, None, seeds)
    for n in set(ins_n) | {o for o, _ in outs}:
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for (n, rk), s in zip(outs, seeds):
        c.dy.put(n, s[None], 1)
    c.dy.remap(AD, last_step)
    for n, a in zip(ins_n, iad):
        e = relerr(c.dy.get(n, 1)[0], a)
        assert e < tol, (n, "ad", e)


from fv3_jedi_linearmodel_amd.harness import step_state, cube_step_state      # noqa: E402,F401  (the package's own state builders)


def check_fv_dynamics(c, mode, tol):
    T, P = step_state(c)
    nq = c.nq
    ins_n = ["u", "v", "pt", "delp", "pe", "peln", "pk", "pkz"] + ["q%d" % (n + 1) for n in range(nq)]
    outs = [("u", "U"), ("v", "V"), ("pt", "A"), ("delp", "A")] + [("q%d" % (n + 1), "A") for n in range(nq)]
    i_t = [T[n] for n in ins_n]; i_p = [P[n] for n in ins_n]
    for n in ins_n:
        c.dy.put(n, T[n][None], 0)
    if mode == TL:
        ot, op = c.oracle.fv_dynamics(TL, nq, c.dims.dt, c.dims.n_split, c.dims.k_split, i_t, i_p)
        for n in ins_n:
            c.dy.put(n, P[n][None], 1)
        c.dy.fv_dynamics(TL)
        worst = 0.0
        for (n, rk), a, b in zip(outs, ot, op):
            r = c.rect(*rects(c)[rk])
            e1, e2 = relerr(c.dy.get(n, 0)[0][r], a[r]), relerr(c.dy.get(n, 1)[0][r], b[r])
            assert e1 < tol, (n, "traj", e1)
            assert e2 < tol, (n, "tl", e2)
            worst = max(worst, e1, e2)
        return worst
    rng = np.random.default_rng(41)
    seeds = [masked(c, rng.standard_normal(T["u"].shape), rk) for n, rk in outs]
    _, iad = c.oracle.fv_dynamics(AD, nq, c.dims.dt, c.dims.n_split, c.dims.k_split, i_t, None, seeds)
    c.dy.fv_dynamics(NL)
    for n in ins_n:
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for (n, rk), s in zip(outs, seeds):
        c.dy.put(n, s[None], 1)
    c.dy.fv_dynamics(AD)
    worst = 0.0
    for n, a in zip(ins_n, iad):
        if n in ("pe", "peln", "pk"):
            continue        # not used by the first dyn_core (geopk recomputes them): adjoint 0 on both sides
        e = relerr(c.dy.get(n, 1)[0], a)
        assert e < tol, (n, "ad", e)
        worst = max(worst, e)
    return worst


def dot_product_step(c, seed=13):
    """<M dx, dy> = <dx, M^T dy> for the complete dynamics step (step_tl / step_ad), the test the JEDI
    LinearModel applies downstream.  x = (u, v, T, delp, q) on the compute domain."""
    T, P = step_state(c)
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
    rk = {"u": "U", "v": "V"}
    rng = np.random.default_rng(seed)
    dx = {n: masked(c, P[n], rk.get(n, "A")) for n in names}
    for n in names:
        c.dy.put(n, T[n][None], 0); c.dy.put(n, dx[n][None], 1)
    c.dy.step_tl()
    Mdx = {n: c.dy.get(n, 1)[0] for n in names}
    dy = {n: masked(c, rng.standard_normal(Mdx[n].shape) * (1.0 / max(1e-30, np.abs(Mdx[n]).max())), rk.get(n, "A")) for n in names}
    lhs = sum(float(np.sum(masked(c, Mdx[n], rk.get(n, "A")) * dy[n])) for n in names)
    for n in names:
        c.dy.put(n, T[n][None], 0)
    c.dy.step_nl()
    for n in names:
        c.dy.put(n, dy[n][None], 1)
    c.dy.step_ad()
    rhs = sum(float(np.sum(c.dy.get(n, 1)[0] * dx[n])) for n in names)
    return lhs, rhs


# ------------------------------------------------------------------------------ six-face cube (CubeCase)
CUBE_DC_OUT = [("u", "U"), ("v", "V"), ("pt", "A"), ("delp", "A"), ("mfx", "V"), ("mfy", "U"), ("cx", "CX"), ("cy", "CY"),
               ("pe", "A"), ("peln", "A"), ("pk", "A"), ("pkz", "A")]


def cube_check_dyn_core(c, mode, tol):
    """fv3lm_dyn_core with six resident faces vs oracle/cube.hpp dyn_core_cube (state compared on the compute
    domain: after the last acoustic step only the shared edge rows of u, v are exchanged, dyn_core_tlm.F90:2418-2431)."""
    ins = ["u", "v", "pt", "delp"]
    i_t = [c.traj[n] for n in ins]; i_p = [c.pert[n] for n in ins]
    bdt, ns = c.dims.dt / c.dims.k_split, c.dims.n_split
    if mode == TL:
        ot, op = c.oracle.dyn_core(TL, bdt, ns, i_t, i_p)
        c.put_state(pert=c.pert)
        c.dy.dyn_core(TL)
        worst = 0.0
        for (n, rk), a, b in zip(CUBE_DC_OUT, ot, op):
            r = c.rect(*rects(c)[rk])
            e1, e2 = relerr(c.dy.get(n, 0)[r], a[r]), relerr(c.dy.get(n, 1)[r], b[r])
            assert e1 < tol, (n, "traj", e1)
            assert e2 < tol, (n, "tl", e2)
            worst = max(worst, e1, e2)
        return worst
    rng = np.random.default_rng(5)
    seeds = [masked(c, rng.standard_normal(c.dy.shape(n)), rk) for n, rk in CUBE_DC_OUT]
    _, iad = c.oracle.dyn_core(AD, bdt, ns, i_t, None, seeds)
    c.put_state()
    c.dy.dyn_core(NL)
    for (n, rk), s in zip(CUBE_DC_OUT, seeds):
        c.dy.put(n, s, 1)
    c.dy.dyn_core(AD)
    worst = 0.0
    for n, a in zip(ins, iad):
        e = relerr(c.dy.get(n, 1), a)
        assert e < tol, (n, "ad", e)
        worst = max(worst, e)
    return worst


def cube_dot_product(c, seed=3):
    rng = np.random.default_rng(seed)
    ins = ["u", "v", "pt", "delp"]
    outs = [("u", "U"), ("v", "V"), ("pt", "A"), ("delp", "A")]
    c.put_state(pert=c.pert)
    c.dy.dyn_core(TL)
    Mdx = {n: c.dy.get(n, 1) for n, _ in outs}
    dy = {n: masked(c, rng.standard_normal(Mdx[n].shape), rk) for n, rk in outs}
    lhs = sum(float(np.sum(Mdx[n] * dy[n])) for n, _ in outs)
    c.put_state()
    c.dy.dyn_core(NL)
    for n in ("mfx", "mfy", "cx", "cy", "pe", "peln", "pk", "pkz"):
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for n, _ in outs:
        c.dy.put(n, dy[n], 1)
    c.dy.dyn_core(AD)
    rhs = sum(float(np.sum(c.dy.get(n, 1) * c.pert[n])) for n in ins)
    return lhs, rhs


def cube_check_fv_dynamics(c, mode, tol):
    T, P = cube_step_state(c)
    nq = c.nq
    ins_n = ["u", "v", "pt", "delp", "pe", "peln", "pk", "pkz"] + ["q%d" % (n + 1) for n in range(nq)]
    outs = [("u", "U"), ("v", "V"), ("pt", "A"), ("delp", "A")] + [("q%d" % (n + 1), "A") for n in range(nq)]
    i_t = [T[n] for n in ins_n]; i_p = [P[n] for n in ins_n]
    for n in ins_n:
        c.dy.put(n, T[n], 0)
    if mode == TL:
        ot, op = c.oracle.fv_dynamics(TL, nq, c.dims.dt, c.dims.n_split, c.dims.k_split, i_t, i_p)
        for n in ins_n:
            c.dy.put(n, P[n], 1)
        c.dy.fv_dynamics(TL)
        worst = 0.0
        for (n, rk), a, b in zip(outs, ot, op):
            r = c.rect(*rects(c)[rk])
            e1, e2 = relerr(c.dy.get(n, 0)[r], a[r]), relerr(c.dy.get(n, 1)[r], b[r])
            assert e1 < tol, (n, "traj", e1)
            assert e2 < tol, (n, "tl", e2)
            worst = max(worst, e1, e2)
        return worst
    rng = np.random.default_rng(41)
    seeds = [masked(c, rng.standard_normal(T["u"].shape), rk) for n, rk in outs]
    _, iad = c.oracle.fv_dynamics(AD, nq, c.dims.dt, c.dims.n_split, c.dims.k_split, i_t, None, seeds)
    c.dy.fv_dynamics(NL)
    for n in ins_n:
        c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for (n, rk), s in zip(outs, seeds):
        c.dy.put(n, s, 1)
    c.dy.fv_dynamics(AD)
    worst = 0.0
    for n, a in zip(ins_n, iad):
        if n in ("pe", "peln", "pk"):
            continue
        e = relerr(c.dy.get(n, 1), a)
        assert e < tol, (n, "ad", e)
        worst = max(worst, e)
    return worst


def cube_dot_product_step(c, seed=13):
    T, P = cube_step_state(c)
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
    rk = {"u": "U", "v": "V"}
    rng = np.random.default_rng(seed)
    dx = {n: masked(c, P[n], rk.get(n, "A")) for n in names}
    for n in names:
        c.dy.put(n, T[n], 0); c.dy.put(n, dx[n], 1)
    c.dy.step_tl()
    Mdx = {n: c.dy.get(n, 1) for n in names}
    dy = {n: masked(c, rng.standard_normal(Mdx[n].shape) * (1.0 / max(1e-30, np.abs(Mdx[n]).max())), rk.get(n, "A")) for n in names}
    lhs = sum(float(np.sum(masked(c, Mdx[n], rk.get(n, "A")) * dy[n])) for n in names)
    for n in names:
        c.dy.put(n, T[n], 0)
    c.dy.step_nl()
    for n in names:
        c.dy.put(n, dy[n], 1)
    c.dy.step_ad()
    rhs = sum(float(np.sum(c.dy.get(n, 1) * dx[n])) for n in names)
    return lhs, rhs


def cube_check_tracer(c, mode, tol, scale=1.0):
    """tracer_2d on six faces (global sub-step count, halo exchange between sub-steps) vs oracle/cube.hpp tracer_2d_cube"""
    from fv3_jedi_linearmodel_amd import cube
    ins = ["u", "v", "pt", "delp"]
    ot, op = c.oracle.dyn_core(TL, c.dims.dt / c.dims.k_split, c.dims.n_split, [c.traj[n] for n in ins], [c.pert[n] for n in ins])
    T = dict(dp1=c.traj["delp"], mfx=scale * ot[4], mfy=scale * ot[5], cx=scale * ot[6], cy=scale * ot[7])
    P = dict(dp1=c.pert["delp"], mfx=scale * op[4], mfy=scale * op[5], cx=scale * op[6], cy=scale * op[7])
    nq = c.nq
    for n in range(nq):
        qt, qp = c.qtraj[n].copy(), c.qpert[n].copy()
        cube.apply_table(c.tables["cell"], qt); cube.apply_table(c.tables["cell"], qp)
        T["q%d" % (n + 1)], P["q%d" % (n + 1)] = qt, qp
    ins_n = ["dp1", "mfx", "mfy", "cx", "cy"] + ["q%d" % (n + 1) for n in range(nq)]
    i_t = [T[n] for n in ins_n]; i_p = [P[n] for n in ins_n]
    A = c.rect(*rects(c)["A"])
    for n in ins_n:
        c.dy.put(n, T[n], 0)
    if mode == TL:
        ot, op = c.oracle.tracer_2d(TL, nq, i_t, i_p)
        for n in ins_n:
            c.dy.put(n, P[n], 1)
        c.dy.tracer_2d(TL)
        for n in range(nq):
            nm = "q%d" % (n + 1)
            assert relerr(c.dy.get(nm, 0)[A], ot[n][A]) < tol, (nm, "traj")
            assert relerr(c.dy.get(nm, 1)[A], op[n][A]) < tol, (nm, "tl")
        return
    rng = np.random.default_rng(21)
    seeds = [masked(c, rng.standard_normal(T["dp1"].shape), "A") for _ in range(nq)]
    _, iad = c.oracle.tracer_2d(AD, nq, i_t, None, seeds)
    c.dy.tracer_2d(NL)
    for n in ins_n:
        c.dy.put(n, T[n], 0); c.dy.put(n, np.zeros(c.dy.shape(n)), 1)
    for n in range(nq):
        c.dy.put("q%d" % (n + 1), seeds[n], 1)
    c.dy.tracer_2d(AD)
    for n, a in zip(ins_n, iad):
        e = relerr(c.dy.get(n, 1), a)
        assert e < tol, (n, "ad", e)


def repeated_adjoint(c, seed=17):
    """One forward sweep (step_nl), several backward sweeps: the checkpoints and trajectory slots of the forward sweep stay
    valid, so further adjoint applications on the same trajectory need no new forward sweep (the role of cp_iter in the
    reference, utils/tapenade/tapenade_iter.F90).  Returns the worst relative difference between the second backward sweep
    and a fresh step_nl + step_ad with the same adjoint input."""
    T, _ = step_state(c)
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
    rk = {"u": "U", "v": "V"}
    rng = np.random.default_rng(seed)
    ya = {n: masked(c, rng.standard_normal(T[n].shape), rk.get(n, "A")) for n in names}
    yb = {n: masked(c, rng.standard_normal(T[n].shape), rk.get(n, "A")) for n in names}

    def backward(y):
        for n in names:
            c.dy.put(n, y[n][None], 1)
        c.dy.step_ad()
        return {n: c.dy.get(n, 1)[0].copy() for n in names}
    for n in names:
        c.dy.put(n, T[n][None], 0)
    c.dy.step_nl()
    backward(ya)
    second = backward(yb)           # no step_nl in between
    for n in names:
        c.dy.put(n, T[n][None], 0)
    c.dy.step_nl()
    fresh = backward(yb)
    return max(relerr(second[n], fresh[n]) for n in names)


def check_step_nl(c, tol):
    """fv3lm_step_nl (nonlinear propagation of the trajectory with the schemes in force) against the oracle's nonlinear fv_dynamics:
    the state after one step, field by field"""
    from oracle import NL
    T, P = step_state(c)
    nq = c.nq
    ins_n = ["u", "v", "pt", "delp", "pe", "peln", "pk", "pkz"] + ["q%d" % (n + 1) for n in range(nq)]
    outs = [("u", "U"), ("v", "V"), ("pt", "A"), ("delp", "A")] + [("q%d" % (n + 1), "A") for n in range(nq)]
    ot, _ = c.oracle.fv_dynamics(NL, nq, c.dims.dt, c.dims.n_split, c.dims.k_split, [T[n] for n in ins_n])
    for n, _rk in outs:
        c.dy.put(n, T[n][None], 0)
    c.dy.step_nl()
    worst = 0.0
    for (n, rk), a in zip(outs, ot):
        r = c.rect(*rects(c)[rk])
        e = relerr(c.dy.get(n, 0)[0][r], a[r])
        assert e < tol, (n, "nl", e)
        worst = max(worst, e)
    return worst
