"""The host's boundary copies on the device (fv3lm_traj_to_fv3 / _pert_to_fv3 / _fv3_to_pert; reference traj_to_fv3, pert_to_fv3,
fv3_to_pert, DYN/fv3jedi_lm_dynamics_mod.F90:717-933) against the same sequence done by hand with whole-field put / get + the
library's own exchanges: step_tl and step_ad results must agree bit for bit."""
import numpy as np


def check_boundary_copies(c, cube=False):
    from groups import step_state, cube_step_state
    T, P = cube_step_state(c) if cube else step_state(c)
    if not cube:
        T = {k: v[None] for k, v in T.items()}; P = {k: v[None] for k, v in P.items()}
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
    nx, ny = c.nx, c.ny
    I = (slice(None), slice(None), slice(3, 3 + ny), slice(3, 3 + nx))          # compute domain of the padded plane

    def interior_only(a):
        z = np.zeros_like(a); z[I] = a[I]; return z
    # ---- by hand: zero halos, put, D-grid edge rows from the neighbours, pressures happen inside step_tl
    def by_hand(step):
        for n in names:
            c.dy.put(n, interior_only(T[n]), 0); c.dy.put(n, interior_only(P[n]), 1)
        c.dy.halo("dedge", "u", "v", 0)
        if step == "tl":
            c.dy.step_tl()
        else:
            c.dy.step_nl(); c.dy.step_ad()
        return {n: c.dy.get(n, 1)[I].copy() for n in names}
    # ---- through the boundary entry points, compact arrays
    def by_boundary(step):
        c.dy.traj_to_fv3({n: T[n][I] for n in names})
        c.dy.pert_to_fv3({n: P[n][I] for n in names})
        if step == "tl":
            c.dy.step_tl()
        else:
            c.dy.step_nl(); c.dy.step_ad()
        out = c.dy.fv3_to_pert(names)
        for n in names:                      # fv3_to_pert clears the device perturbation, as the reference does
            assert not c.dy.get(n, 1).any(), n
        return out
    for step in ("tl", "ad"):
        a, b = by_hand(step), by_boundary(step)
        for n in names:
            assert np.array_equal(a[n], b[n]), (step, n)
            assert np.isfinite(a[n]).all() and np.abs(a[n]).max() > 0, (step, n)


def check_boundary_oracle(c):
    """fv3lm_traj_to_fv3 / _pert_to_fv3 / _fv3_to_pert and the edge fill of step_tl / its adjoint in step_ad against the numpy restatement
    of DYN/fv3jedi_lm_dynamics_mod.F90:717-809, :386-399, :651-665, :846-933 (tests/boundary_oracle.py), whose exchange tables are the
    independently derived ones of oracle/cube_topology.py.  c: a CubeCase (six faces)."""
    from groups import cube_step_state
    from boundary_oracle import BoundaryOracle, interior, pad
    T, P = cube_step_state(c)
    n = c.n
    names = ["u", "v", "pt", "delp"] + ["q%d" % (m + 1) for m in range(c.nq)]
    B = BoundaryOracle(n)
    Tc = {k: interior(T[k], n) for k in names}; Pc = {k: interior(P[k], n) for k in names}
    phis_c = interior(c.phis, n)
    # ---- traj_to_fv3, field by field
    c.dy.traj_to_fv3(dict(Tc, phis=phis_c))
    ref, ph = B.traj_to_fv3(Tc, phis_c)
    for k in names:
        assert np.array_equal(c.dy.get(k, 0), ref[k]), ("traj_to_fv3", k)
    assert np.array_equal(c.dy.get("phis", 0), ph), "traj_to_fv3 phis halo"
    from boundary_oracle import pressures
    pe, peln, pk, pkz = pressures(Tc["delp"], c.opt.akap, c.opt.ptop)
    for k, r in (("pe", pe), ("peln", peln), ("pk", pk), ("pkz", pkz)):
        got = interior(c.dy.get(k, 0), n)
        assert np.max(np.abs(got - r)) <= 1e-13 * np.max(np.abs(r)), ("traj_to_fv3", k)
    # ---- pert_to_fv3: halos and edge rows zero, interior copied
    c.dy.pert_to_fv3(Pc)
    pin = B.pert_in(Pc, fill_edges=False)
    for k in names:
        assert np.array_equal(c.dy.get(k, 1), pin[k]), ("pert_to_fv3", k)
    # ---- step_tl through the compact path ...
    c.dy.step_tl()
    out_tl = c.dy.fv3_to_pert(names)
    # ... against: whole-field put of the oracle's arrays (edge rows of the perturbation filled by the oracle), then the two calls
    # step_tl replaces -- compute_fv3_pressures_tlm + FV_DYNAMICS_TLM -- without the library's own edge fill
    pin = B.pert_in(Pc, fill_edges=True)
    assert np.abs(pin["u"][:, :, 3 + n, 3:3 + n]).max() > 0 and np.abs(pin["v"][:, :, 3:3 + n, 3 + n]).max() > 0
    c.dy.traj_to_fv3(dict(Tc, phis=phis_c))
    for k in names:
        c.dy.put(k, pin[k], 1)
    c.dy.pressures(1); c.dy.fv_dynamics(1)
    for k in names:
        a, b = out_tl[k], interior(c.dy.get(k, 1), n)
        assert np.array_equal(a, b), ("step_tl", k, float(np.max(np.abs(a - b))))
    # the fill matters: without it the result differs along the north / east edge of every face
    c.dy.traj_to_fv3(dict(Tc, phis=phis_c))
    pz = B.pert_in(Pc, fill_edges=False)
    for k in names:
        c.dy.put(k, pz[k], 1)
    c.dy.pressures(1); c.dy.fv_dynamics(1)
    assert np.max(np.abs(interior(c.dy.get("u", 1), n) - out_tl["u"])) > 1e-6 * np.max(np.abs(out_tl["u"]))
    # ---- step_ad through the compact path ...
    c.dy.traj_to_fv3(dict(Tc, phis=phis_c))
    c.dy.pert_to_fv3(Pc)
    c.dy.step_nl(); c.dy.step_ad()
    out_ad = c.dy.fv3_to_pert(names)
    # ... against FV_DYNAMICS_BWD + compute_fv3_pressures_bwd by hand and the oracle's mpp_get_boundary_ad
    c.dy.traj_to_fv3(dict(Tc, phis=phis_c))
    for k in names:
        c.dy.put(k, pz[k], 1)
    c.dy.step_nl()
    c.dy.fv_dynamics(2); c.dy.pressures(0); c.dy.pressures(2)
    full = {k: c.dy.get(k, 1) for k in names}
    assert np.abs(full["u"][:, :, 3 + n, 3:3 + n]).max() > 0, "adjoint mass on the shared edge row"
    ref_ad = B.pert_out_ad(full)
    for k in names:
        a, b = out_ad[k], ref_ad[k]
        assert np.max(np.abs(a - b)) <= 1e-14 * np.max(np.abs(b)), ("step_ad", k, float(np.max(np.abs(a - b))))
