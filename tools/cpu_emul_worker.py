"""One worker of bench.py's cpu_baseline leg, second comparator: the PRODUCT's own stage code with its hand-written adjoints, compiled for the
host (g++ -O3 -DFV3LM_HOST_EMUL: host loops where the HIP build launches kernels; tests/_emul/, never loaded by the package) on one host
core, one doubly-periodic tile of the workload's shape that does not fit the core's L2.  Runs step_tl + step_nl + step_ad like the timed
region of bench.py and prints {"t_tl", "t_nl", "t_ad", "columns"}.  The reference Fortran cannot be built here (FMS absent); this is the
closest thing to it the repository owns: the same algorithm, source-level adjoints instead of the oracle port's operation tape."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    so = sys.argv[1]
    nx, npz, n_split, k_split, nq = (int(a) for a in sys.argv[2:7])
    dt = float(sys.argv[7]); nh = int(sys.argv[8]); seed = int(sys.argv[9])
    from fv3_jedi_linearmodel_amd import harness as H
    from fv3_jedi_linearmodel_amd._lib import Fv3LmLibrary

    class HostCase(H.Case):
        def _load_library(self, backend):
            return Fv3LmLibrary(so)
    kw = dict(hydrostatic=0) if nh else {}
    c = HostCase(nx=nx, ny=nx, npz=npz, n_split=n_split, k_split=k_split, dt=dt, backend="host", nq=nq, seed=seed, **kw)
    T, P = H.step_state(c)
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
    for n in names:
        c.dy.put(n, T[n][None], 0); c.dy.put(n, P[n][None], 1)
    c.dy.state_save()
    t = {}
    for leg, fn in (("t_tl", c.dy.step_tl), ("t_nl", c.dy.step_nl), ("t_ad", c.dy.step_ad)):
        if leg != "t_ad":
            c.dy.state_restore()
        t0 = time.time(); fn(); t[leg] = time.time() - t0
    t["columns"] = nx * nx
    print(json.dumps(t))


if __name__ == "__main__":
    main()
