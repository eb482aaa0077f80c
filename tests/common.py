"""Shared case builder for the parity tests: the package's synthetic cases (fv3_jedi_linearmodel_amd/harness.py) + the oracle as checker and
a second backend, the test-only host-emulation build of the product's stage code."""
import os
import subprocess
import numpy as np
import fv3_jedi_linearmodel_amd as fv3
from fv3_jedi_linearmodel_amd._lib import Fv3LmLibrary, Dycore
from fv3_jedi_linearmodel_amd import grid as G
from fv3_jedi_linearmodel_amd import harness as H
from fv3_jedi_linearmodel_amd.harness import relerr      # noqa: F401
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


class _TestHooks:
    def _load_library(self, backend):
        if backend == "emul":
            lib = Fv3LmLibrary(build_emul())
            lib.L.fv3lm_emul_check_boxes.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
            return lib
        return fv3.load_hip_library()

    def _after_create(self, backend):
        if backend == "emul":
            self.lib.L.fv3lm_emul_check_boxes(self.dy.h, 1)


class Case(_TestHooks, H.Case):
    def __init__(self, *a, backend="emul", oracle=True, **kw):
        super().__init__(*a, backend=backend, oracle=oracle, **kw)

    def _make_oracle(self):
        o = Oracle(self.nx, self.ny, self.npz, self.nq, self.metrics, self.opt, self.da_min, self.da_min_c, self.phis, self.ak, self.bk)
        if self.face is not None:
            o.set_face(self.edge[0], self.ecorner[0])
        return o


class CubeCase(_TestHooks, H.CubeCase):
    def __init__(self, *a, backend="emul", oracle=False, **kw):
        super().__init__(*a, backend=backend, oracle=oracle, **kw)

    def _make_oracle(self):
        from oracle import CubeOracle
        return CubeOracle(self.n, self.npz, self.nq, self.metrics, self.opt, self.da_min, self.da_min_c, self.phis, self.ak, self.bk,
                          self.edge, self.ecorner, self.tables)
