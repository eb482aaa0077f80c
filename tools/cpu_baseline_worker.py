"""One worker of bench.py's cpu_baseline leg: the oracle port (oracle/liboracle.so, TEST INFRASTRUCTURE) on one host core, one
tile of the same workload shape.  Prints a JSON line {"t_tl": s, "t_ad": s, "columns": n}.  BASELINE.md §5.3: the reference has no
threading, so P independent single-core workers each owning a slab of columns are the zero-communication upper bound of its MPI
decomposition; bench.py starts P of these at once and takes the slowest."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    nx, npz, n_split, k_split, nq = (int(a) for a in sys.argv[1:6])
    dt = float(sys.argv[6]); nh = int(sys.argv[7]); seed = int(sys.argv[8])
    import numpy as np
    from common import Case
    from oracle import TL, AD
    kw = dict(hydrostatic=0) if nh else {}
    c = Case(nx=nx, ny=nx, npz=npz, n_split=n_split, k_split=k_split, dt=dt, backend="none", nq=nq, seed=seed, **kw)
    if nh:
        from test_oracle_nh import nh_state_fv
        T, P = nh_state_fv(c)
        args = (c.nq, dt, n_split, k_split)
        run = c.oracle.fv_dynamics_nh
    else:
        from groups import step_state
        Td, Pd = step_state(c)
        ins = ["u", "v", "pt", "delp", "pe", "peln", "pk", "pkz"] + ["q%d" % (n + 1) for n in range(c.nq)]
        T = [Td[n] for n in ins]; P = [Pd[n] for n in ins]
        args = (c.nq, dt, n_split, k_split)
        run = c.oracle.fv_dynamics
    t0 = time.time(); run(TL, *args, T, P); t_tl = time.time() - t0
    nout = (6 if nh else 4) + c.nq
    seeds = [np.ones_like(T[0]) for _ in range(nout)]
    t0 = time.time(); run(AD, *args, T, None, seeds); t_ad = time.time() - t0
    print(json.dumps({"t_tl": t_tl, "t_ad": t_ad, "columns": nx * nx}))


if __name__ == "__main__":
    main()
