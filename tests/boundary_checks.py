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
