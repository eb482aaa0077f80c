"""ctypes binding of the C-ABI in include/fv3lm.h.

`load_hip_library()` loads the product library libfv3lm_hip.so and raises if it is missing — the
package has no CPU path.  `Fv3LmLibrary(path)` is the thin object wrapper the tests also use to
drive the test-only host-emulation build of the same sources.
"""
import ctypes as C
import os
import numpy as np
from .config import Options, Dims

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfv3lm_hip.so")
_dp = C.POINTER(C.c_double)


class Fv3LmError(RuntimeError):
    pass


class Fv3LmLibrary:
    def __init__(self, path):
        if not os.path.exists(path):
            raise Fv3LmError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        L = self.L = C.CDLL(path)
        L.fv3lm_last_error.restype = C.c_char_p
        L.fv3lm_metric_names.restype = C.c_char_p
        L.fv3lm_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Dims), C.POINTER(Options), C.POINTER(_dp),
                                   C.c_double, C.c_double, _dp, _dp, _dp]
        L.fv3lm_destroy.argtypes = [C.c_void_p]
        L.fv3lm_field_put.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _dp]
        L.fv3lm_field_get.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _dp]
        L.fv3lm_field_levels.argtypes = [C.c_void_p, C.c_char_p]
        L.fv3lm_run_group.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.fv3lm_dyn_core.argtypes = [C.c_void_p, C.c_int]
        L.fv3lm_zero_work_adjoint.argtypes = [C.c_void_p]
        L.fv3lm_sync.argtypes = [C.c_void_p]
        L.fv3lm_launch_count.argtypes = [C.c_void_p]
        L.fv3lm_launch_count.restype = C.c_long
        L.fv3lm_level_params.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), _dp]
        for fn in ("fv3lm_pressures", "fv3lm_tracer_2d", "fv3lm_fv_dynamics"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_int]
        L.fv3lm_remap.argtypes = [C.c_void_p, C.c_int, C.c_int]
        for fn in ("fv3lm_step_tl", "fv3lm_step_nl", "fv3lm_step_ad"):
            getattr(L, fn).argtypes = [C.c_void_p]

    def err(self):
        return self.L.fv3lm_last_error().decode()

    def metric_names(self):
        return self.L.fv3lm_metric_names().decode().split(",")


def load_hip_library():
    """The product library.  FV3LM_LIB names another build of the same HIP sources (kernel-tuning variants, tools/mkvariant.sh); a
    host-emulation build (tests/_emul, it exports fv3lm_emul_check_boxes) is refused: the package has no CPU path."""
    lib = Fv3LmLibrary(os.environ.get("FV3LM_LIB", LIB_PATH))
    if hasattr(lib.L, "fv3lm_emul_check_boxes"):
        raise Fv3LmError("%s is a host-emulation build (tests only); the package runs the HIP library alone" % lib.path)
    return lib


TRANSPORT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_dp), C.POINTER(C.c_long), C.POINTER(_dp), C.POINTER(C.c_long))


def rccl_library_path():
    """The RCCL this process already uses: torch's bundled copy if torch is imported, else the ROCm one."""
    import sys
    if "torch" in sys.modules:
        p = os.path.join(os.path.dirname(sys.modules["torch"].__file__), "lib", "librccl.so")
        if os.path.exists(p):
            return p
    return "/opt/rocm/lib/librccl.so"


def comm_init_rccl(lib, rank, world, broadcast_bytes):
    """One RCCL communicator over all ranks for the face exchange.  broadcast_bytes(buf: bytearray|None) -> bytes must hand
    rank 0's 128-byte unique id to every rank (e.g. through torch.distributed)."""
    path = rccl_library_path().encode()
    uid = C.create_string_buffer(128)
    if rank == 0:
        lib.L.fv3lm_comm_unique_id.argtypes = [C.c_char_p, C.c_void_p]
        if lib.L.fv3lm_comm_unique_id(path, uid) != 0:
            raise Fv3LmError(lib.err())
    data = broadcast_bytes(bytes(uid.raw) if rank == 0 else None)
    lib.L.fv3lm_comm_init.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    if lib.L.fv3lm_comm_init(path, data, world, rank) != 0:
        raise Fv3LmError(lib.err())


ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, C.c_int)


def set_allreduce_callback(lib, fn):
    """fn(buf: numpy view) must replace buf by its element-wise max over the ranks that hold faces (tracer_2d's
    mp_reduce_max of the per-level Courant numbers, fv_tracer2d_tlm.F90:1306).  Not needed with one rank."""
    def tramp(user, buf, n):
        fn(np.ctypeslib.as_array(buf, shape=(n,)))
    cb = ALLREDUCE_FN(tramp)
    lib._allreduce_cb = cb
    lib.L.fv3lm_set_allreduce_callback.argtypes = [ALLREDUCE_FN, C.c_void_p]
    lib.L.fv3lm_set_allreduce_callback(cb, None)


def set_transport_callback(lib, fn):
    """fn(peers, sendbufs, recvbufs): lists of numpy views (host memory in the emulation build).  Test transports (gloo)."""
    def tramp(user, npeers, peers, sb, sc, rb, rc):
        P = [peers[i] for i in range(npeers)]
        S = [np.ctypeslib.as_array(sb[i], shape=(sc[i],)) if sc[i] else np.zeros(0) for i in range(npeers)]
        R = [np.ctypeslib.as_array(rb[i], shape=(rc[i],)) if rc[i] else np.zeros(0) for i in range(npeers)]
        fn(P, S, R)
    cb = TRANSPORT_FN(tramp)
    lib._transport_cb = cb          # keep alive
    lib.L.fv3lm_set_transport_callback.argtypes = [TRANSPORT_FN, C.c_void_p]
    lib.L.fv3lm_set_transport_callback(cb, None)


def _ptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


class Dycore:
    """Device-resident dycore instance (one per GPU / MPI rank), cf. fv3jedi_lm_dynamics_type."""

    NL, TL, AD = 0, 1, 2

    def __init__(self, lib, dims, options, metrics, da_min, da_min_c, phis, ak, bk):
        self.lib, self.dims, self.options = lib, dims, options
        names = lib.metric_names()
        self.pj, self.pi = dims.ny + 7, dims.nx + 7
        arrs = []
        for n in names:
            a = np.ascontiguousarray(metrics[n], dtype=np.float64)
            assert a.shape == (dims.ntile, self.pj, self.pi), (n, a.shape)
            arrs.append(a)
        self._keep = arrs
        mp = (_dp * len(arrs))(*[_ptr(a) for a in arrs])
        self.h = C.c_void_p()
        phis = np.ascontiguousarray(phis, dtype=np.float64)
        ak = np.ascontiguousarray(ak, dtype=np.float64); bk = np.ascontiguousarray(bk, dtype=np.float64)
        rc = lib.L.fv3lm_create(C.byref(self.h), C.byref(dims), C.byref(options), mp, da_min, da_min_c,
                                _ptr(phis), _ptr(ak), _ptr(bk))
        if rc != 0:
            raise Fv3LmError(lib.err())

    def close(self):
        if self.h:
            self.lib.L.fv3lm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def levels(self, name):
        n = self.lib.L.fv3lm_field_levels(self.h, name.encode())
        if n < 0:
            raise Fv3LmError(self.lib.err())
        return n

    def shape(self, name):
        return (self.dims.ntile, self.levels(name), self.pj, self.pi)

    def put(self, name, arr, which=0):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        assert a.shape == self.shape(name), (name, a.shape, self.shape(name))
        if self.lib.L.fv3lm_field_put(self.h, name.encode(), which, _ptr(a)) != 0:
            raise Fv3LmError(self.lib.err())

    def get(self, name, which=0):
        a = np.empty(self.shape(name), dtype=np.float64)
        if self.lib.L.fv3lm_field_get(self.h, name.encode(), which, _ptr(a)) != 0:
            raise Fv3LmError(self.lib.err())
        return a

    def run_group(self, group, mode):
        if self.lib.L.fv3lm_run_group(self.h, group.encode(), mode) != 0:
            raise Fv3LmError(self.lib.err())

    def dyn_core(self, mode):
        if self.lib.L.fv3lm_dyn_core(self.h, mode) != 0:
            raise Fv3LmError(self.lib.err())

    def _chk(self, rc):
        if rc != 0:
            raise Fv3LmError(self.lib.err())

    def pressures(self, mode):
        self._chk(self.lib.L.fv3lm_pressures(self.h, mode))

    def tracer_2d(self, mode):
        self._chk(self.lib.L.fv3lm_tracer_2d(self.h, mode))

    def remap(self, mode, last_step):
        self._chk(self.lib.L.fv3lm_remap(self.h, mode, int(last_step)))

    def fv_dynamics(self, mode):
        self._chk(self.lib.L.fv3lm_fv_dynamics(self.h, mode))

    def step_tl(self):
        """fv3jedi_lm_dynamics_type%step_tl (DYN/fv3jedi_lm_dynamics_mod.F90:347-456), device part."""
        self._chk(self.lib.L.fv3lm_step_tl(self.h))

    def step_nl(self):
        self._chk(self.lib.L.fv3lm_step_nl(self.h))

    def step_ad(self):
        """%step_ad (DYN/fv3jedi_lm_dynamics_mod.F90:460-689): call after step_nl() (the FV_DYNAMICS_FWD
        role: nonlinear sweep storing the stage checkpoints)."""
        self._chk(self.lib.L.fv3lm_step_ad(self.h))

    # ---- the host's boundary copies on the device (compact arrays [ntile, nk, ny, nx], no halo) ----
    def _cptrs(self, d, names, out=False):
        keep = []
        required = ["u", "v", "pt", "delp"] + ["q%d" % (m + 1) for m in range(self.dims.nq)] + ([] if self.options.hydrostatic else ["w", "delz"])
        missing = [n for n in required if n not in d or d[n] is None]
        if missing:
            raise Fv3LmError("boundary copy: missing array(s) %s" % ", ".join(missing))
        def one(n):
            if n not in d or d[n] is None:
                return None
            a = d[n] if out else np.ascontiguousarray(d[n], dtype=np.float64)
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], n
            keep.append(a)
            return a.ctypes.data_as(_dp)
        nq = self.dims.nq
        qarr = (_dp * max(1, nq))(*[one("q%d" % (m + 1)) for m in range(nq)])
        return [one(n) for n in names], qarr, keep

    def traj_to_fv3(self, d):
        """d: dict of compact arrays u v pt delp q1.. (w delz) and optionally phis [ntile, ny, nx] (traj_to_fv3, :717-807)"""
        (u, v, t, dp, w, dz, ph), q, keep = self._cptrs(d, ["u", "v", "pt", "delp", "w", "delz", "phis"])
        f = self.lib.L.fv3lm_traj_to_fv3
        f.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.POINTER(_dp), _dp, _dp, _dp]
        self._chk(f(self.h, u, v, t, dp, q, w, dz, ph))

    def pert_to_fv3(self, d):
        (u, v, t, dp, w, dz), q, keep = self._cptrs(d, ["u", "v", "pt", "delp", "w", "delz"])
        f = self.lib.L.fv3lm_pert_to_fv3
        f.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.POINTER(_dp), _dp, _dp]
        self._chk(f(self.h, u, v, t, dp, q, w, dz))

    def fv3_to_pert(self, names):
        shp = (self.dims.ntile, self.dims.npz, self.dims.ny, self.dims.nx)
        d = {n: np.empty(shp) for n in names}
        (u, v, t, dp, w, dz), q, keep = self._cptrs(d, ["u", "v", "pt", "delp", "w", "delz"], out=True)
        f = self.lib.L.fv3lm_fv3_to_pert
        f.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, C.POINTER(_dp), _dp, _dp]
        self._chk(f(self.h, u, v, t, dp, q, w, dz))
        return d

    def state_save(self):
        self.lib.L.fv3lm_state_save.argtypes = [C.c_void_p]
        self._chk(self.lib.L.fv3lm_state_save(self.h))

    def state_restore(self):
        self.lib.L.fv3lm_state_restore.argtypes = [C.c_void_p]
        self._chk(self.lib.L.fv3lm_state_restore(self.h))

    def profile_begin(self):
        self.lib.L.fv3lm_profile_begin.argtypes = [C.c_void_p]
        self.lib.L.fv3lm_profile_begin(self.h)

    def profile_end(self):
        """-> {kernel: (launches, total_ms, algorithmic_bytes)} measured with HIP events on the library stream."""
        self.lib.L.fv3lm_profile_end.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        buf = C.create_string_buffer(1 << 16)
        self.lib.L.fv3lm_profile_end(self.h, buf, len(buf))
        out = {}
        for line in buf.value.decode().splitlines():
            n, cnt, ms, by = line.split()
            out[n] = (int(float(cnt)), float(ms), float(by))
        return out

    def zero_work_adjoint(self):
        self.lib.L.fv3lm_zero_work_adjoint(self.h)

    def sync(self):
        self.lib.L.fv3lm_sync(self.h)

    def launch_count(self):
        return self.lib.L.fv3lm_launch_count(self.h)

    # ---- face mode (dims.face = 1) -------------------------------------------------------
    def set_face_data(self, edge, ecorner):
        """a2b_ord4 edge weights [ntile,4,pj] and extrap_corner factors [ntile,4,3] (cube.cubed_sphere_metrics)."""
        e = np.ascontiguousarray(edge, dtype=np.float64); c = np.ascontiguousarray(ecorner, dtype=np.float64)
        assert e.shape == (self.dims.ntile, 4, self.pj) and c.shape == (self.dims.ntile, 4, 3)
        self.lib.L.fv3lm_set_face_data.argtypes = [C.c_void_p, _dp, _dp]
        self._chk(self.lib.L.fv3lm_set_face_data(self.h, _ptr(e), _ptr(c)))

    HALO_KINDS = {"cell": 0, "dvec": 1, "cvec": 2, "corner": 3, "dedge": 4}

    def set_exchange(self, kind, rows):
        """Exchange table of one kind (cube.exchange_table / cube.boundary_table), int32 [n,7]."""
        r = np.ascontiguousarray(rows, dtype=np.int32)
        assert r.ndim == 2 and r.shape[1] == 7
        self.lib.L.fv3lm_set_exchange.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        self._chk(self.lib.L.fv3lm_set_exchange(self.h, self.HALO_KINDS[kind], r.ctypes.data_as(C.POINTER(C.c_int)), r.shape[0]))

    def set_exchange_split(self, kind, table, rank, world, ntiles=6, loopback=False):
        """Install one exchange kind for this rank's tiles: local rows + per-peer send/receive lists (cube.split_table)."""
        from . import cube
        local, peers, send, recv = cube.split_table(table, rank, world, ntiles, loopback)
        self.set_exchange(kind, local)
        ip = C.POINTER(C.c_int)
        pe = np.array(peers, dtype=np.int32)
        ns = np.array([send[p].shape[0] for p in peers], dtype=np.int32); nr = np.array([recv[p].shape[0] for p in peers], dtype=np.int32)
        sr = np.ascontiguousarray(np.concatenate([send[p] for p in peers], axis=0) if peers else np.zeros((0, 3), np.int32))
        rr = np.ascontiguousarray(np.concatenate([recv[p] for p in peers], axis=0) if peers else np.zeros((0, 4), np.int32))
        self.lib.L.fv3lm_set_exchange_remote.argtypes = [C.c_void_p, C.c_int, C.c_int, ip, ip, ip, ip, ip]
        P = lambda a: a.ctypes.data_as(ip)
        self._chk(self.lib.L.fv3lm_set_exchange_remote(self.h, self.HALO_KINDS[kind], len(peers), P(pe), P(ns), P(sr), P(nr), P(rr)))

    def halo(self, kind, name0, name1="", mode=0):
        self.lib.L.fv3lm_halo.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
        self._chk(self.lib.L.fv3lm_halo(self.h, self.HALO_KINDS[kind], name0.encode(), name1.encode(), mode))

    def level_params(self, k):
        ip = (C.c_int * 10)(); rp = (C.c_double * 6)()
        if self.lib.L.fv3lm_level_params(self.h, k, ip, rp) != 0:
            raise Fv3LmError(self.lib.err())
        return list(ip), list(rp)
