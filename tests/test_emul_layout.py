"""CPU (-m "not gpu"): sub-face tiles on the host-emulation build (layout_checks.py) + the tiled exchange tables against a direct
application of the face tables."""
import numpy as np
import pytest
from common import CubeCase
from layout_checks import check_layout_equals_whole_faces


def test_tiled_tables_reproduce_the_windows_of_global_fields():
    from fv3_jedi_linearmodel_amd import cube
    n, L = 16, 2
    nt = n // L
    _, _, _, _, _, geo = cube.cubed_sphere_metrics(n)
    f = cube.cube_fields(n, 2, geo, 5, "pert")
    tl = cube.tiles(n, L)
    T = cube.tiled_tables(n, L)
    for kind, names in (("cell", ["pt"]), ("dvec", ["u", "v"]), ("dedge", ["u", "v"])):
        w = [cube.tile_window(f[nm], tl, nt) for nm in names]
        ref = [x.copy() for x in w]
        fl = [x.reshape(x.shape[0], x.shape[1], -1) for x in w]
        for r in T[kind]:
            fl[r[0]][r[1], :, r[2]] = np.nan
        src = [fl[r[3]][r[4], :, r[5]] * r[6] for r in T[kind]]
        assert not any(np.isnan(v).any() for v in src), "a source that is itself a destination"
        for r, v in zip(T[kind], src):
            fl[r[0]][r[1], :, r[2]] = v
        for a_, b_ in zip(w, ref):
            assert np.max(np.abs(a_ - b_)) < 1e-12


@pytest.mark.parametrize("layout", [2, 3])
def test_hydrostatic_with_tracers(layout):
    n = 16 if layout == 2 else 24
    check_layout_equals_whole_faces(lambda L: CubeCase(n=n, npz=5, n_split=2, k_split=2, dt=900.0, backend="emul", nq=2, layout=L), layout)


def test_split_damp_nord3_and_split_hord():
    kw = dict(split_damp=1, nord=3, nord_pert=1, dddmp=0.35, d4_bg=0.11, vtdm4=0.03, n_sponge_pert=2, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10, hord_tr=10)
    check_layout_equals_whole_faces(lambda L: CubeCase(n=16, npz=5, n_split=2, k_split=1, dt=600.0, backend="emul", nq=1, layout=L, **kw), 2)


def test_nonhydrostatic():
    check_layout_equals_whole_faces(lambda L: CubeCase(n=16, npz=6, n_split=2, k_split=1, dt=300.0, backend="emul", nq=1, layout=L, hydrostatic=0), 2, tol=1e-11)


def test_boundary_copies_with_tiles():
    """fv3lm_traj_to_fv3 / _pert_to_fv3 / _fv3_to_pert with compact (isc:iec, jsc:jec) arrays per TILE: the D-grid edge rows of an interior
    tile boundary come from the neighbour tile of the same face"""
    from boundary_checks import check_boundary_copies
    check_boundary_copies(CubeCase(n=16, npz=4, n_split=2, k_split=1, backend="emul", nq=1, layout=2), cube=True)
