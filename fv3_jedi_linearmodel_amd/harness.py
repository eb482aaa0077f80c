"""Synthetic cases for the product: one tile or the six faces of a cube with the metrics, the smooth synthetic state and perturbation of
BASELINE.md section 4, and a HIP-library instance holding them.  Used by bench.py, __graft_entry__.smoke() and -- through tests/common.py, which
adds the oracle and the test-only host-emulation backend -- by the parity tests.  Nothing here imports from tests/ or oracle/."""
import numpy as np
from . import default_options, Dims, synthetic_tile_metrics, load_hip_library
from . import grid as G
from ._lib import Dycore


def relerr(a, b):
    s = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (s if s > 0 else 1.0))


class Case:
    def __init__(self, nx=12, ny=12, npz=8, n_split=2, k_split=1, dt=1800.0, backend="hip", seed=20250114, oracle=False, nq=0,
                 face=None, **optkw):
        """face=None: the doubly-periodic tile with no cube edge.  face=t (0..5): one whole face of a C<nx> cube with
        the real gnomonic metrics of that face and arbitrary smooth halo data (kernel-group tests)."""
        self.nx, self.ny, self.npz = nx, ny, npz
        self.opt = default_options(**optkw)
        self.face = face
        if face is None:
            self.metrics, self.da_min, self.da_min_c = synthetic_tile_metrics(nx, ny)
        else:
            from . import cube
            assert nx == ny
            m6, self.da_min, self.da_min_c, edge, ecorner, _ = cube.cubed_sphere_metrics(nx)
            self.metrics = {k: np.ascontiguousarray(v[face:face + 1]) for k, v in m6.items()}
            self.edge, self.ecorner = np.ascontiguousarray(edge[face:face + 1]), np.ascontiguousarray(ecorner[face:face + 1])
        self.traj, self.phis, self.ak, self.bk = G.synthetic_state(nx, ny, npz, self.opt, seed=seed)
        self.pert = G.synthetic_pert(nx, ny, npz, seed=seed + 1)
        for d in (self.traj, self.pert):
            for k in d:
                d[k] = G.halo_fill_periodic(d[k], nx, ny)
        self.phis = G.halo_fill_periodic(self.phis, nx, ny)
        self.dims = Dims(nx=nx, ny=ny, npz=npz, ntile=1, nq=nq, n_split=n_split, k_split=k_split, face=0 if face is None else 1, dt=dt)
        self.dt_ac = dt / n_split / k_split
        self.nq = nq
        rng = np.random.default_rng(seed + 7)
        from .grid import _smooth_field
        shp = (1, npz, ny + 7, nx + 7)
        self.qtraj = [G.halo_fill_periodic(np.abs(1e-3 * (n + 1) + _smooth_field(rng, shp, nx, ny, 3e-4)), nx, ny) for n in range(nq)]
        self.qpert = [G.halo_fill_periodic(_smooth_field(rng, shp, nx, ny, 1e-4), nx, ny) for n in range(nq)]
        self.oracle = self._make_oracle() if oracle else None
        if backend == "none":        # no product instance (the CPU baseline leg of bench.py)
            self.lib = self.dy = None
            return
        self.lib = self._load_library(backend)
        self.dy = Dycore(self.lib, self.dims, self.opt, self.metrics, self.da_min, self.da_min_c, self.phis, self.ak, self.bk)
        if face is not None:
            self.dy.set_face_data(self.edge, self.ecorner)
        self._after_create(backend)

    # hooks of tests/common.py (oracle as checker, host-emulation backend); the package itself knows the HIP library only
    def _make_oracle(self):
        raise RuntimeError("the oracle is test infrastructure: use tests/common.py")

    def _load_library(self, backend):
        if backend != "hip":
            raise RuntimeError("backend %r: the package drives the HIP library only" % backend)
        return load_hip_library()

    def _after_create(self, backend):
        pass

    # helpers -------------------------------------------------------------------------------
    def put_state(self, traj=None, pert=None):
        traj = traj or self.traj
        for n in ("u", "v", "delp", "pt"):
            self.dy.put(n, traj[n], 0)
            if pert is not None:
                self.dy.put(n, pert[n], 1)

    def rect(self, i0, i1, j0, j1):
        """numpy slices of the padded plane for Fortran index ranges i0..i1, j0..j1."""
        return (Ellipsis, slice(j0 + 2, j1 + 3), slice(i0 + 2, i1 + 3))

    def rng_field(self, nk, seed, amp=1.0):
        rng = np.random.default_rng(seed)
        return amp * rng.standard_normal((1, nk, self.ny + 7, self.nx + 7))


class CubeCase:
    """All six faces of a C<n> cube resident in one product instance (face mode), exchange tables from cube.py, smooth global fields
    as state.  layout = L > 1: every face cut into L x L tiles (fv_control_nlm.F90:556), 6 L^2 tiles of n/L cells; the fields of the
    case are then the tiles' windows [ntile, nk, n/L+7, n/L+7] of the same global fields (gather() puts results back on faces)."""

    def __init__(self, n=12, npz=6, n_split=2, k_split=1, dt=1800.0, backend="hip", seed=20250114, nq=0, oracle=False, rank=0, world=1,
                 layout=1, loopback=False, **optkw):
        """rank/world: this process holds only cube.faces_of(rank, world, ntiles) (one process per GPU); state and metrics are the
        corresponding slices of the same global fields, so results can be compared with a single-process run.
        loopback: the rows between this rank's own tiles go through the message path too (cube.split_table)."""
        from . import cube
        self.n = n
        self.layout = layout
        self.nt = self.nx = self.ny = n // layout
        self.npz, self.nq = npz, nq
        self.opt = default_options(**optkw)
        self.metrics, self.da_min, self.da_min_c, self.edge, self.ecorner, self.geo = cube.cubed_sphere_metrics(n)
        self.traj, self.phis, self.ak, self.bk = cube.cube_fields(n, npz, self.geo, seed, "traj", self.opt)
        self.pert = cube.cube_fields(n, npz, self.geo, seed + 1, "pert")
        aux = cube.cube_fields(n, npz, self.geo, seed + 11, "pert") if nq else None
        self.qtraj = [1e-3 * (m + 1) + 1e-2 * np.abs(aux["pt"] if m % 2 == 0 else 2e-3 * aux["delp"]) * (1.0 + 0.25 * m) for m in range(nq)]
        self.qpert = [1e-4 * (aux["delp"] if m % 2 == 0 else 500.0 * aux["pt"]) * (1.0 + 0.5 * m) for m in range(nq)]
        self.tiles = cube.tiles(n, layout)
        ntiles = len(self.tiles)
        if layout > 1:      # the tiles' windows of everything
            nt, tl = self.nt, self.tiles
            W = lambda a: cube.tile_window(a, tl, nt)
            self.face_fields = dict(traj=self.traj, pert=self.pert, phis=self.phis, qtraj=self.qtraj, qpert=self.qpert)      # on whole faces (the oracle's view)
            self.metrics = {k: W(v) for k, v in self.metrics.items()}
            pt_ = nt + 7
            self.edge = np.ascontiguousarray(np.stack([np.stack([self.edge[f, e, (j0 if e < 2 else i0) - 1:(j0 if e < 2 else i0) - 1 + pt_] for e in range(4)]) for (f, i0, j0) in tl]))
            self.ecorner = np.ascontiguousarray(np.stack([self.ecorner[f] for (f, _, _) in tl]))
            self.phis = W(self.phis)
            self.traj = {k: W(v) for k, v in self.traj.items()}; self.pert = {k: W(v) for k, v in self.pert.items()}
            self.qtraj = [W(v) for v in self.qtraj]; self.qpert = [W(v) for v in self.qpert]
            self.tables = cube.tiled_tables(n, layout)
        else:
            self.tables = cube.all_tables(n)
        self.faces = cube.faces_of(rank, world, ntiles)
        self.ntiles = ntiles
        if world > 1:
            F = self.faces
            self.metrics = {k: np.ascontiguousarray(v[F]) for k, v in self.metrics.items()}
            self.edge, self.ecorner, self.phis = np.ascontiguousarray(self.edge[F]), np.ascontiguousarray(self.ecorner[F]), np.ascontiguousarray(self.phis[F])
            self.traj = {k: np.ascontiguousarray(v[F]) for k, v in self.traj.items()}; self.pert = {k: np.ascontiguousarray(v[F]) for k, v in self.pert.items()}
            self.qtraj = [np.ascontiguousarray(v[F]) for v in self.qtraj]; self.qpert = [np.ascontiguousarray(v[F]) for v in self.qpert]
        self.dims = Dims(nx=self.nt, ny=self.nt, npz=npz, ntile=len(self.faces), nq=nq, n_split=n_split, k_split=k_split, face=1, dt=dt)
        if layout > 1:
            import ctypes as C
            self._ij0 = (C.c_int * (2 * len(self.faces)))(*[v for t in self.faces for v in self.tiles[t][1:]])
            self.dims.nface = n
            self.dims.tile_ij0 = C.cast(self._ij0, C.POINTER(C.c_int))
        self.dt_ac = dt / n_split / k_split
        self.face = "cube"
        self.oracle = self._make_oracle() if oracle else None
        if backend == "none":
            self.lib = self.dy = None
            return
        self.lib = self._load_library(backend)
        self.dy = Dycore(self.lib, self.dims, self.opt, self.metrics, self.da_min, self.da_min_c, self.phis, self.ak, self.bk)
        self.dy.set_face_data(self.edge, self.ecorner)
        self._after_create(backend)
        for k, t in self.tables.items():
            if world > 1 or loopback:
                self.dy.set_exchange_split(k, t, rank, world, ntiles, loopback)
            else:
                self.dy.set_exchange(k, t)

    def gather(self, a):
        """tile windows [ntile, ..., nt+7, nt+7] -> face arrays [6, ..., n+7, n+7] on the compute domains (whole faces: unchanged)"""
        if self.layout == 1:
            return a
        from . import cube
        return cube.tile_gather(a, self.tiles, self.n, self.nt)

    _make_oracle = Case._make_oracle
    _load_library = Case._load_library
    _after_create = Case._after_create

    def put_state(self, traj=None, pert=None):
        traj = traj or self.traj
        for n in ("u", "v", "delp", "pt"):
            self.dy.put(n, traj[n], 0)
            if pert is not None:
                self.dy.put(n, pert[n], 1)
        for m in range(self.nq):
            self.dy.put("q%d" % (m + 1), self.qtraj[m], 0)
            if pert is not None:
                self.dy.put("q%d" % (m + 1), self.qpert[m], 1)

    def rect(self, i0, i1, j0, j1):
        return (Ellipsis, slice(j0 + 2, j1 + 3), slice(i0 + 2, i1 + 3))


def _pressures_np(delp, ptop, akap, axis):
    sl = [slice(None)] * delp.ndim
    first = list(sl); first[axis] = slice(0, 1)
    pe = np.concatenate([np.full_like(delp[tuple(first)], ptop), ptop + np.cumsum(delp, axis=axis)], axis=axis)
    peln = np.log(pe); pk = np.exp(akap * peln)
    hi = list(sl); hi[axis] = slice(1, None); lo = list(sl); lo[axis] = slice(None, -1)
    pkz = (pk[tuple(hi)] - pk[tuple(lo)]) / (akap * (peln[tuple(hi)] - peln[tuple(lo)]))
    return pe, peln, pk, pkz


def step_state(c):
    """Temperature-based state of the single tile as the host hands it over (pt = T), plus numpy pressures and their exact linearisation"""
    k = c.opt.akap
    delp = c.traj["delp"][0]
    pe, peln, pk, pkz = _pressures_np(delp, c.opt.ptop, k, 0)
    qv = c.qtraj[0][0] if c.nq else 0.0
    T = dict(u=c.traj["u"][0], v=c.traj["v"][0], pt=c.traj["pt"][0] * pkz / (1.0 + c.opt.zvir * qv), delp=delp, pe=pe, peln=peln, pk=pk, pkz=pkz)
    P = dict(u=c.pert["u"][0], v=c.pert["v"][0], pt=20.0 * c.pert["pt"][0], delp=c.pert["delp"][0])
    pe_p = np.concatenate([np.zeros_like(P["delp"][:1]), np.cumsum(P["delp"], axis=0)], axis=0)
    peln_p = pe_p / pe; pk_p = k * peln_p * pk
    den = k * (peln[1:] - peln[:-1])
    pkz_p = ((pk_p[1:] - pk_p[:-1]) * den - (pk[1:] - pk[:-1]) * k * (peln_p[1:] - peln_p[:-1])) / den ** 2
    P.update(pe=pe_p, peln=peln_p, pk=pk_p, pkz=pkz_p)
    for n in range(c.nq):
        T["q%d" % (n + 1)], P["q%d" % (n + 1)] = c.qtraj[n][0], c.qpert[n][0]
    return T, P


def cube_step_state(c):
    """Temperature-based state as the host hands it over (pt = T) on six faces, plus numpy pressures and their exact linearisation"""
    k = c.opt.akap
    delp = c.traj["delp"]
    pe, peln, pk, pkz = _pressures_np(delp, c.opt.ptop, k, 1)
    qv = c.qtraj[0] if c.nq else 0.0
    T = dict(u=c.traj["u"], v=c.traj["v"], pt=c.traj["pt"] * pkz / (1.0 + c.opt.zvir * qv), delp=delp, pe=pe, peln=peln, pk=pk, pkz=pkz)
    P = dict(u=c.pert["u"], v=c.pert["v"], pt=20.0 * c.pert["pt"], delp=c.pert["delp"])
    pe_p = np.concatenate([np.zeros_like(delp[:, :1]), np.cumsum(P["delp"], axis=1)], axis=1)
    peln_p = pe_p / pe; pk_p = k * peln_p * pk
    den = k * (peln[:, 1:] - peln[:, :-1])
    pkz_p = ((pk_p[:, 1:] - pk_p[:, :-1]) * den - (pk[:, 1:] - pk[:, :-1]) * k * (peln_p[:, 1:] - peln_p[:, :-1])) / den ** 2
    P.update(pe=pe_p, peln=peln_p, pk=pk_p, pkz=pkz_p)
    for n in range(c.nq):
        T["q%d" % (n + 1)], P["q%d" % (n + 1)] = c.qtraj[n], c.qpert[n]
    return T, P


def cube_nh_state(c):
    """non-hydrostatic: + w and the hydrostatic delz of the same state, lists in the order u v pt delp w delz q*"""
    from . import cube
    T0, P0 = cube_step_state(c)
    o = c.opt
    qv = c.qtraj[0] if c.nq else 0.0
    delz = -(o.rdgas / o.grav) * T0["pt"] * (1.0 + o.zvir * qv) * np.diff(T0["peln"], axis=1)
    aux = cube.cube_fields(c.n, c.npz, c.geo, 20250135, "pert")
    aux2 = cube.cube_fields(c.n, c.npz, c.geo, 20250136, "pert")
    if c.layout > 1:
        aux = {k: cube.tile_window(v, c.tiles, c.nt) for k, v in aux.items()}; aux2 = {k: cube.tile_window(v, c.tiles, c.nt) for k, v in aux2.items()}
    if len(c.faces) != c.ntiles:
        aux = {k: v[c.faces] for k, v in aux.items()}; aux2 = {k: v[c.faces] for k, v in aux2.items()}
    w = 0.05 * aux["pt"]; w_p = 0.01 * aux2["pt"]; dz_p = 1e-3 * aux2["delp"]
    names = ["u", "v", "pt", "delp"]
    T = [T0[n] for n in names] + [w, delz] + [T0["q%d" % (n + 1)] for n in range(c.nq)]
    P = [P0[n] for n in names] + [w_p, dz_p] + [P0["q%d" % (n + 1)] for n in range(c.nq)]
    return T, P
