"""Shared case builder for the parity tests: one synthetic tile, the oracle and a product instance
(HIP library on the GPU box, or the test-only host-emulation build of the same stage code)."""
import os
import subprocess
import numpy as np
import fv3_jedi_linearmodel_amd as fv3
from fv3_jedi_linearmodel_amd._lib import Fv3LmLibrary, Dycore
from fv3_jedi_linearmodel_amd import grid as G
from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_SO = os.path.join(ROOT, "tests", "_emul", "libfv3lm_emul.so")
CSRC = os.path.join(ROOT, "fv3_jedi_linearmodel_amd", "csrc")


def build_emul():
    """g++ build of the product's stage code with -DFV3LM_HOST_EMUL (tests only)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "fv3lm.h")]
    if os.path.exists(EMUL_SO) and all(os.path.getmtime(EMUL_SO) >= os.path.getmtime(s) for s in srcs):
        return EMUL_SO
    os.makedirs(os.path.dirname(EMUL_SO), exist_ok=True)
    import fcntl
    with open(EMUL_SO + ".lock", "w") as lock:        # pytest -n: one worker builds, the others wait and find it done
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not (os.path.exists(EMUL_SO) and all(os.path.getmtime(EMUL_SO) >= os.path.getmtime(s) for s in srcs)):
            tmp = EMUL_SO + ".%d.tmp" % os.getpid()
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-DFV3LM_HOST_EMUL", "-shared", "-o", tmp,
                                   os.path.join(CSRC, "fv3lm_capi.cpp")])
            os.replace(tmp, EMUL_SO)
    return EMUL_SO


class Case:
    def __init__(self, nx=12, ny=12, npz=8, n_split=2, k_split=1, dt=1800.0, backend="emul", seed=20250114, oracle=True, nq=0,
                 face=None, **optkw):
        """face=None: the doubly-periodic tile with no cube edge.  face=t (0..5): one whole face of a C<nx> cube with
        the real gnomonic metrics of that face and arbitrary smooth halo data (kernel-group tests)."""
        self.nx, self.ny, self.npz = nx, ny, npz
        self.opt = fv3.default_options(**optkw)
        self.face = face
        if face is None:
            self.metrics, self.da_min, self.da_min_c = fv3.synthetic_tile_metrics(nx, ny)
        else:
            from fv3_jedi_linearmodel_amd import cube
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
        self.dims = fv3.Dims(nx=nx, ny=ny, npz=npz, ntile=1, nq=nq, n_split=n_split, k_split=k_split, face=0 if face is None else 1, dt=dt)
        self.dt_ac = dt / n_split / k_split
        self.nq = nq
        rng = np.random.default_rng(seed + 7)
        from fv3_jedi_linearmodel_amd.grid import _smooth_field
        shp = (1, npz, ny + 7, nx + 7)
        self.qtraj = [G.halo_fill_periodic(np.abs(1e-3 * (n + 1) + _smooth_field(rng, shp, nx, ny, 3e-4)), nx, ny) for n in range(nq)]
        self.qpert = [G.halo_fill_periodic(_smooth_field(rng, shp, nx, ny, 1e-4), nx, ny) for n in range(nq)]
        self.oracle = Oracle(nx, ny, npz, nq, self.metrics, self.opt, self.da_min, self.da_min_c, self.phis, self.ak, self.bk) if oracle else None
        if self.oracle is not None and face is not None:
            self.oracle.set_face(self.edge[0], self.ecorner[0])
        if backend == "none":        # oracle only (bench.py's cpu_baseline leg)
            self.lib = self.dy = None
            return
        if backend == "emul":
            self.lib = Fv3LmLibrary(build_emul())
            self.lib.L.fv3lm_emul_check_boxes.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
        else:
            self.lib = fv3.load_hip_library()
        self.dy = Dycore(self.lib, self.dims, self.opt, self.metrics, self.da_min, self.da_min_c, self.phis, self.ak, self.bk)
        if face is not None:
            self.dy.set_face_data(self.edge, self.ecorner)
        if backend == "emul":
            self.lib.L.fv3lm_emul_check_boxes(self.dy.h, 1)

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


def relerr(a, b):
    s = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (s if s > 0 else 1.0))


class CubeCase:
    """All six faces of a C<n> cube resident in one product instance (ntile = 6, face mode), exchange tables from
    cube.py, smooth global fields as state.  oracle=True adds the six-face oracle (tests/oracle.py CubeOracle)."""

    def __init__(self, n=12, npz=6, n_split=2, k_split=1, dt=1800.0, backend="emul", seed=20250114, nq=0, oracle=False, rank=0, world=1,
                 **optkw):
        """rank/world: this process holds only cube.faces_of(rank, world) (one process per GPU); state and metrics are the
        corresponding slices of the same global fields, so results can be compared with a single-process run."""
        from fv3_jedi_linearmodel_amd import cube
        self.n = self.nx = self.ny = n
        self.npz, self.nq = npz, nq
        self.opt = fv3.default_options(**optkw)
        self.metrics, self.da_min, self.da_min_c, self.edge, self.ecorner, self.geo = cube.cubed_sphere_metrics(n)
        self.tables = cube.all_tables(n)
        self.traj, self.phis, self.ak, self.bk = cube.cube_fields(n, npz, self.geo, seed, "traj", self.opt)
        self.pert = cube.cube_fields(n, npz, self.geo, seed + 1, "pert")
        rng = np.random.default_rng(seed + 7)
        aux = cube.cube_fields(n, npz, self.geo, seed + 11, "pert") if nq else None
        self.qtraj = [1e-3 * (m + 1) + 1e-2 * np.abs(aux["pt"] if m % 2 == 0 else 2e-3 * aux["delp"]) * (1.0 + 0.25 * m) for m in range(nq)]
        self.qpert = [1e-4 * (aux["delp"] if m % 2 == 0 else 500.0 * aux["pt"]) * (1.0 + 0.5 * m) for m in range(nq)]
        self.faces = cube.faces_of(rank, world)
        if world > 1:
            F = self.faces
            self.metrics = {k: np.ascontiguousarray(v[F]) for k, v in self.metrics.items()}
            self.edge, self.ecorner, self.phis = np.ascontiguousarray(self.edge[F]), np.ascontiguousarray(self.ecorner[F]), np.ascontiguousarray(self.phis[F])
            self.traj = {k: np.ascontiguousarray(v[F]) for k, v in self.traj.items()}; self.pert = {k: np.ascontiguousarray(v[F]) for k, v in self.pert.items()}
            self.qtraj = [np.ascontiguousarray(v[F]) for v in self.qtraj]; self.qpert = [np.ascontiguousarray(v[F]) for v in self.qpert]
        self.dims = fv3.Dims(nx=n, ny=n, npz=npz, ntile=len(self.faces), nq=nq, n_split=n_split, k_split=k_split, face=1, dt=dt)
        self.dt_ac = dt / n_split / k_split
        self.face = "cube"
        self.oracle = None
        if oracle:
            from oracle import CubeOracle
            self.oracle = CubeOracle(n, npz, nq, self.metrics, self.opt, self.da_min, self.da_min_c, self.phis, self.ak, self.bk,
                                     self.edge, self.ecorner, self.tables)
        if backend == "none":
            self.lib = self.dy = None
            return
        if backend == "emul":
            self.lib = Fv3LmLibrary(build_emul())
            self.lib.L.fv3lm_emul_check_boxes.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
        else:
            self.lib = fv3.load_hip_library()
        self.dy = Dycore(self.lib, self.dims, self.opt, self.metrics, self.da_min, self.da_min_c, self.phis, self.ak, self.bk)
        self.dy.set_face_data(self.edge, self.ecorner)
        for k, t in self.tables.items():
            if world > 1:
                self.dy.set_exchange_split(k, t, rank, world)
            else:
                self.dy.set_exchange(k, t)

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
