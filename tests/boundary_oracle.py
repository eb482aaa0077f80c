"""TEST INFRASTRUCTURE -- numpy restatement of the host's boundary copies of the reference, src/dynamics/fv3jedi_lm_dynamics_mod.F90:

  traj_to_fv3   :717-809   halos zeroed (:730-753), interior copy (:758-778), D-grid edge rows u(:, jec+1), v(iec+1, :) from the
                           neighbour faces (mpp_get_boundary :783-797), halo of phis (mpp_update_domains :802), pressures (:807-809)
  step_tl       :386-399   "Edge of pert always needs to be filled": the same edge rows of the perturbation, before the pressures
  step_ad       :651-665   its adjoint (mpp_get_boundary_ad) after FV_DYNAMICS_BWD and compute_fv3_pressures_bwd
  pert_to_fv3   :846-889   halos (and edge rows) zeroed, interior copy
  fv3_to_pert   :893-933   interior copy out

The data motion of mpp_get_boundary / mpp_update_domains lives in FMS (absent); here it is driven by the exchange tables of
oracle/cube_topology.py -- derived from the reference's contact lines by index arithmetic alone, nothing imported from the package -- so
the product's own tables (fv3_jedi_linearmodel_amd/cube.py) are not in the loop.  Arrays: compact [6, nk, n, n] in, padded planes
[6, nk, n+7, n+7] out (include/fv3lm.h)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import cube_topology as topo       # noqa: E402

NG = 3


def pad(a, n):
    """compact (isc:iec, jsc:jec) -> padded plane with everything else zero (:730-753, :758-778)"""
    z = np.zeros(a.shape[:-2] + (n + 7, n + 7))
    z[..., NG:NG + n, NG:NG + n] = a
    return z


def interior(a, n):
    return np.ascontiguousarray(a[..., NG:NG + n, NG:NG + n])


def apply_rows(rows, f0, f1=None):
    """forward exchange: halo element <- sign * source element, every level"""
    fs = [f0, f1]
    flat = [None if f is None else f.reshape(f.shape[0], f.shape[1], -1) for f in fs]
    vals = [(r, flat[r[3]][r[4], :, r[5]] * r[6]) for r in rows]          # gather first: sources are never destinations of the same table
    for r, v in vals:
        flat[r[0]][r[1], :, r[2]] = v


def apply_rows_ad(rows, f0, f1=None):
    """adjoint: source += sign * halo, halo cleared"""
    fs = [f0, f1]
    flat = [None if f is None else f.reshape(f.shape[0], f.shape[1], -1) for f in fs]
    for r in rows:
        flat[r[3]][r[4], :, r[5]] += r[6] * flat[r[0]][r[1], :, r[2]]
        flat[r[0]][r[1], :, r[2]] = 0.0


def pressures(delp, akap, ptop):
    """compute_fv3_pressures (model_tlmadm/fv_pressure.F90:21-72) on the compute domain: pe, peln, pk [npz+1], pkz [npz]"""
    nt, npz = delp.shape[:2]
    pe = np.zeros((nt, npz + 1) + delp.shape[2:]); pe[:, 0] = ptop
    for k in range(npz):
        pe[:, k + 1] = pe[:, k] + delp[:, k]
    peln = np.log(pe); pk = np.exp(akap * peln)
    pkz = (pk[:, 1:] - pk[:, :-1]) / (akap * (peln[:, 1:] - peln[:, :-1]))
    return pe, peln, pk, pkz


class BoundaryOracle:
    def __init__(self, n):
        self.n = n
        self.dedge = topo.boundary_table(n)
        self.cell = topo.exchange_table(n, "cell")

    def traj_to_fv3(self, T, phis):
        """T: dict of compact arrays; returns padded u v pt delp q*, phis [6, 1, pj, pi]"""
        n = self.n
        out = {k: pad(v, n) for k, v in T.items()}
        apply_rows(self.dedge, out["u"], out["v"])
        ph = pad(phis[:, None], n)
        apply_rows(self.cell, ph)
        return out, ph

    def pert_in(self, P, fill_edges):
        """pert_to_fv3 (+ the edge fill of step_tl when fill_edges)"""
        out = {k: pad(v, self.n) for k, v in P.items()}
        if fill_edges:
            apply_rows(self.dedge, out["u"], out["v"])
        return out

    def pert_out_ad(self, full):
        """mpp_get_boundary_ad on the adjoint winds, then fv3_to_pert"""
        u, v = full["u"].copy(), full["v"].copy()
        apply_rows_ad(self.dedge, u, v)
        out = {k: interior(a, self.n) for k, a in full.items()}
        out["u"], out["v"] = interior(u, self.n), interior(v, self.n)
        return out
