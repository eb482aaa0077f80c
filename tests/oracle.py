"""ctypes wrapper of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_dp = C.POINTER(C.c_double)
NL, TL, AD = 0, 1, 2


def build_oracle():
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".hpp", ".cpp"))]
    if not os.path.exists(ORACLE_SO) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return ORACLE_SO


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _nh_opts(opt, a_imp, p_fac, scale_z):
    """solver options default to the case's fv3lm_options (the same struct the product was created with)"""
    return (opt.a_imp if a_imp is None else a_imp, opt.p_fac if p_fac is None else p_fac, opt.scale_z if scale_z is None else scale_z)


class Oracle:
    def __init__(self, nx, ny, npz, nq, metrics, opt, da_min, da_min_c, phis, ak, bk):
        L = self.L = C.CDLL(build_oracle())
        L.orc_create.restype = C.c_void_p
        L.orc_metric_names.restype = C.c_char_p
        names = L.orc_metric_names().decode().split(",")
        self.nx, self.ny, self.npz = nx, ny, npz
        self.pj, self.pi = ny + 7, nx + 7
        self.opt = opt
        arrs = [np.ascontiguousarray(metrics[n][0], dtype=np.float64) for n in names]
        mp = (_dp * len(arrs))(*[_ptr(a) for a in arrs])
        il = opt.int_list()
        # orc_create iopt: 28 scheme ints + kord_tm, kord_mt, kord_wz, kord_tr + the four kord_*_pert + split_damp
        iopt = (C.c_int * 37)(*(il[:36] + [opt.split_damp]))
        rl = opt.real_list()
        ropt = (C.c_double * 29)(*(rl + [da_min, da_min_c]))
        assert len(rl) == 27
        self._keep = (arrs, np.ascontiguousarray(phis[0]), np.ascontiguousarray(ak), np.ascontiguousarray(bk))
        L.orc_create.argtypes = [C.c_int] * 4 + [C.POINTER(_dp), C.POINTER(C.c_int), _dp, _dp, _dp, _dp]
        self.h = C.c_void_p(L.orc_create(nx, ny, npz, nq, mp, iopt, ropt, _ptr(self._keep[1]), _ptr(self._keep[2]),
                                         _ptr(self._keep[3])))

    def set_face(self, edge, ecorner):
        """switch the oracle tile to a whole cube face: a2b edge weights [4,pj], extrap_corner factors [4,3]"""
        e = np.ascontiguousarray(edge, dtype=np.float64); c = np.ascontiguousarray(ecorner, dtype=np.float64)
        self.L.orc_set_face.argtypes = [C.c_void_p, _dp, _dp]
        self.L.orc_set_face.restype = None
        self.L.orc_set_face(self.h, _ptr(e), _ptr(c))

    def level_params(self, k):
        ip = (C.c_int * 10)(); rp = (C.c_double * 6)()
        self.L.orc_level_params.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), _dp]
        ok = self.L.orc_level_params(self.h, k, ip, rp)
        return ok, list(ip), list(rp)

    def _call(self, fname, mode, scalars, ins, ins_p, out_nk, outs_p=None):
        """ins: list of (nk,pj,pi) arrays; ins_p: tangents (TL) or None; outs_p: output adjoints (AD).
        Returns (outs_t, outs_p) for NL/TL, or (outs_t, ins_adjoint) for AD."""
        fn = getattr(self.L, fname)
        ins = [np.ascontiguousarray(a, dtype=np.float64) for a in ins]
        outs = [np.zeros((nk, self.pj, self.pi)) for nk in out_nk]
        if mode == NL:
            ip = None; op = None
        elif mode == TL:
            ipa = [np.ascontiguousarray(a, dtype=np.float64) for a in ins_p]
            opa = [np.zeros_like(o) for o in outs]
        else:
            ipa = [np.zeros_like(a) for a in ins]
            opa = [np.ascontiguousarray(a, dtype=np.float64) for a in outs_p]
        it = (_dp * len(ins))(*[_ptr(a) for a in ins])
        ot = (_dp * len(outs))(*[_ptr(a) for a in outs])
        if mode == NL:
            ip = C.POINTER(_dp)(); op = C.POINTER(_dp)()
        else:
            ip = (_dp * len(ins))(*[_ptr(a) for a in ipa])
            op = (_dp * len(outs))(*[_ptr(a) for a in opa])
        args = [self.h, C.c_int(mode)] + scalars + [it, ip, ot, op]
        fn.restype = None
        fn(*args)
        if mode == NL:
            return outs, None
        if mode == TL:
            return outs, opa
        return outs, ipa

    def c_sw(self, mode, dt2, ins, ins_p=None, outs_p=None):
        return self._call("orc_c_sw", mode, [C.c_double(dt2)], ins, ins_p, [self.npz] * 9, outs_p)

    def d_sw(self, mode, dt, ins, ins_p=None, outs_p=None):
        return self._call("orc_d_sw", mode, [C.c_double(dt)], ins, ins_p, [self.npz] * 12, outs_p)

    def geopk(self, mode, cg, ins, ins_p=None, outs_p=None):
        n = self.npz
        return self._call("orc_geopk", mode, [C.c_int(cg)], ins, ins_p, [n + 1, n + 1, n + 1, n + 1, n], outs_p)

    def p_grad_c(self, mode, dt2, ins, ins_p=None, outs_p=None):
        return self._call("orc_p_grad_c", mode, [C.c_double(dt2)], ins, ins_p, [self.npz] * 2, outs_p)

    def one_grad_p(self, mode, dt, ins, ins_p=None, outs_p=None):
        return self._call("orc_one_grad_p", mode, [C.c_double(dt)], ins, ins_p, [self.npz] * 2, outs_p)

    def dyn_core(self, mode, bdt, n_split, ins, ins_p=None, outs_p=None):
        n = self.npz
        return self._call("orc_dyn_core", mode, [C.c_double(bdt), C.c_int(n_split)], ins, ins_p,
                          [n] * 8 + [n + 1, n + 1, n + 1, n], outs_p)

    def dyn_core_nh(self, mode, bdt, n_split, ins, ins_p=None, outs_p=None, a_imp=None, p_fac=None, scale_z=None):
        n = self.npz
        a_imp, p_fac, scale_z = _nh_opts(self.opt, a_imp, p_fac, scale_z)
        return self._call("orc_dyn_core_nh", mode, [C.c_double(bdt), C.c_int(n_split), C.c_double(a_imp), C.c_double(p_fac), C.c_double(scale_z)],
                          ins, ins_p, [n] * 6 + [n + 1] * 4, outs_p)

    def fv_tp_2d(self, mode, hord, nord, damp_c, use_mf, use_mass, ins, ins_p=None, outs_p=None):
        return self._call("orc_fv_tp_2d", mode, [C.c_int(hord), C.c_int(nord), C.c_double(damp_c), C.c_int(use_mf),
                                                 C.c_int(use_mass)], ins, ins_p, [1, 1], outs_p)

    def tracer_2d(self, mode, nq, ins, ins_p=None, outs_p=None):
        return self._call("orc_tracer_2d", mode, [C.c_int(nq)], ins, ins_p, [self.npz] * nq, outs_p)

    def remap(self, mode, nq, last_step, ins, ins_p=None, outs_p=None):
        n = self.npz
        return self._call("orc_remap", mode, [C.c_int(nq), C.c_int(int(last_step))], ins, ins_p,
                          [n + 1, n + 1, n + 1] + [n] * (5 + nq), outs_p)

    def fv_dynamics(self, mode, nq, bdt, n_split, k_split, ins, ins_p=None, outs_p=None):
        return self._call("orc_fv_dynamics", mode, [C.c_int(nq), C.c_double(bdt), C.c_int(n_split), C.c_int(k_split)],
                          ins, ins_p, [self.npz] * (4 + nq), outs_p)

    def fv_dynamics_nh(self, mode, nq, bdt, n_split, k_split, ins, ins_p=None, outs_p=None, a_imp=None, p_fac=None, scale_z=None):
        a_imp, p_fac, scale_z = _nh_opts(self.opt, a_imp, p_fac, scale_z)
        return self._call("orc_fv_dynamics_nh", mode, [C.c_int(nq), C.c_double(bdt), C.c_int(n_split), C.c_int(k_split), C.c_double(a_imp),
                                                       C.c_double(p_fac), C.c_double(scale_z)], ins, ins_p, [self.npz] * (6 + nq), outs_p)


class CubeOracle:
    """Six-face oracle: one Oracle per face (face mode) + the exchange tables (oracle/cube.hpp)."""
    KINDS = ["cell", "dvec", "cvec", "corner", "dedge"]

    def __init__(self, n, npz, nq, metrics, opt, da_min, da_min_c, phis, ak, bk, edge, ecorner, tables):
        self.n, self.npz, self.nq = n, npz, nq
        self.opt = opt
        self.pj = self.pi = n + 7
        self.faces = []
        for t in range(6):
            o = Oracle(n, n, npz, nq, {k: v[t:t + 1] for k, v in metrics.items()}, opt, da_min, da_min_c, phis[t:t + 1], ak, bk)
            o.set_face(edge[t], ecorner[t])
            self.faces.append(o)
        L = self.L = self.faces[0].L
        self._tabs = [np.ascontiguousarray(tables[k], dtype=np.int32) for k in self.KINDS]
        ip = C.POINTER(C.c_int)
        tp = (ip * 5)(*[t.ctypes.data_as(ip) for t in self._tabs])
        nr = (C.c_int * 5)(*[t.shape[0] for t in self._tabs])
        fh = (C.c_void_p * 6)(*[o.h for o in self.faces])
        L.orc_cube_create.restype = C.c_void_p
        L.orc_cube_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(ip), C.POINTER(C.c_int)]
        self.h = C.c_void_p(L.orc_cube_create(fh, tp, nr))

    def _call(self, fname, mode, scalars, ins, ins_p, out_nk, outs_p=None):
        """fields are [6, nk, pj, pi]; same conventions as Oracle._call"""
        fn = getattr(self.L, fname)
        ins = [np.ascontiguousarray(a, dtype=np.float64) for a in ins]
        outs = [np.zeros((6, nk, self.pj, self.pi)) for nk in out_nk]
        it = (_dp * len(ins))(*[_ptr(a) for a in ins]); ot = (_dp * len(outs))(*[_ptr(a) for a in outs])
        if mode == NL:
            ip = C.POINTER(_dp)(); op = C.POINTER(_dp)(); ipa = opa = None
        else:
            ipa = [np.ascontiguousarray(a, dtype=np.float64) for a in ins_p] if mode == TL else [np.zeros_like(a) for a in ins]
            opa = [np.zeros_like(o) for o in outs] if mode == TL else [np.ascontiguousarray(a, dtype=np.float64) for a in outs_p]
            ip = (_dp * len(ins))(*[_ptr(a) for a in ipa]); op = (_dp * len(outs))(*[_ptr(a) for a in opa])
        fn.restype = None
        fn(self.h, C.c_int(mode), *scalars, it, ip, ot, op)
        return (outs, None) if mode == NL else (outs, opa) if mode == TL else (outs, ipa)

    def dyn_core(self, mode, bdt, n_split, ins, ins_p=None, outs_p=None):
        n = self.npz
        return self._call("orc_cube_dyn_core", mode, [C.c_double(bdt), C.c_int(n_split)], ins, ins_p, [n] * 8 + [n + 1, n + 1, n + 1, n], outs_p)

    def tracer_2d(self, mode, nq, ins, ins_p=None, outs_p=None):
        return self._call("orc_cube_tracer_2d", mode, [C.c_int(nq)], ins, ins_p, [self.npz] * nq, outs_p)

    def fv_dynamics(self, mode, nq, bdt, n_split, k_split, ins, ins_p=None, outs_p=None):
        return self._call("orc_cube_fv_dynamics", mode, [C.c_int(nq), C.c_double(bdt), C.c_int(n_split), C.c_int(k_split)], ins, ins_p,
                          [self.npz] * (4 + nq), outs_p)

    def fv_dynamics_nh(self, mode, nq, bdt, n_split, k_split, ins, ins_p=None, outs_p=None, a_imp=None, p_fac=None, scale_z=None):
        a_imp, p_fac, scale_z = _nh_opts(self.opt, a_imp, p_fac, scale_z)
        return self._call("orc_cube_fv_dynamics_nh", mode, [C.c_int(nq), C.c_double(bdt), C.c_int(n_split), C.c_int(k_split), C.c_double(a_imp),
                                                            C.c_double(p_fac), C.c_double(scale_z)], ins, ins_p, [self.npz] * (6 + nq), outs_p)
