"""Worker of test_dist_cube.py: one rank of a world_size-N run of the six-face dycore over gloo (CPU): the host-emulation
build of the HIP sources with the faces dealt over ranks and a gloo transport behind fv3lm_set_transport_callback.
Every rank also runs the whole cube in-process (world = 1) and compares its own faces."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import CubeCase
    from groups import cube_step_state, masked
    from fv3_jedi_linearmodel_amd._lib import set_transport_callback, set_allreduce_callback
    nh = os.environ.get("FV3LM_DIST_NH", "0") == "1"      # non-hydrostatic: w, delz prognostic; w / heights / pressures join the exchanges
    layout = int(os.environ.get("FV3LM_DIST_LAYOUT", "1"))     # > 1: sub-face tiles dealt over the ranks (24 tiles of a 2 x 2 layout: 3 per rank at 8)
    kw = dict(n=8 * layout, npz=6 if layout == 1 else 4, n_split=2, k_split=2, backend="emul", nq=2 if layout == 1 else 1, layout=layout)
    if nh:
        kw.update(hydrostatic=0, dt=1200.0)
    ref = CubeCase(**kw)                       # whole cube in this process
    c = CubeCase(rank=rank, world=world, **kw)   # this rank's faces

    def transport(peers, sbufs, rbufs):
        reqs = []
        for p, r in zip(peers, rbufs):
            if r.size:
                reqs.append(dist.irecv(torch.from_numpy(r), src=p))
        for p, s in zip(peers, sbufs):
            if s.size:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(s)), dst=p))
        for q in reqs:
            q.wait()
    set_transport_callback(c.lib, transport)

    def allmax(buf):
        t = torch.from_numpy(buf.copy())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        buf[:] = t.numpy()
    set_allreduce_callback(c.lib, allmax)

    F = c.faces
    names = ["u", "v", "pt", "delp"] + (["w", "delz"] if nh else []) + ["q%d" % (n + 1) for n in range(c.nq)]
    T, P = cube_step_state(ref)
    if nh:
        import nh_checks
        Tn, Pn = nh_checks.cube_nh_state(ref)
        T.update(w=Tn[4], delz=Tn[5]); P.update(w=Pn[4], delz=Pn[5])
    rk = {"u": "U", "v": "V"}
    rng = np.random.default_rng(13)
    dy = {n: masked(ref, rng.standard_normal(T[n].shape), rk.get(n, "A")) for n in names}
    worst = 0.0
    # tangent-linear step
    for case, sl in ((ref, slice(None)), (c, F)):
        for n in names:
            case.dy.put(n, T[n][sl], 0); case.dy.put(n, masked(ref, P[n], rk.get(n, "A"))[sl], 1)
        case.dy.step_tl()
    for n in names:
        for w in (0, 1):
            a, b = c.dy.get(n, w), ref.dy.get(n, w)[F]
            r = ref.rect(1, ref.nx + (1 if n == "v" else 0), 1, ref.nx + (1 if n == "u" else 0))
            worst = max(worst, float(np.max(np.abs(a[r] - b[r])) / max(1e-300, np.max(np.abs(b[r])))))
    # adjoint step
    for case, sl in ((ref, slice(None)), (c, F)):
        for n in names:
            case.dy.put(n, T[n][sl], 0)
        case.dy.step_nl()
        for n in names:
            case.dy.put(n, dy[n][sl], 1)
        case.dy.step_ad()
    for n in names:
        a, b = c.dy.get(n, 1), ref.dy.get(n, 1)[F]
        r = ref.rect(1, ref.nx + (1 if n == "v" else 0), 1, ref.nx + (1 if n == "u" else 0))
        worst = max(worst, float(np.max(np.abs(a[r] - b[r])) / max(1e-300, np.max(np.abs(b[r])))))
    t = torch.tensor([worst], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print("DIST_WORST %.3e" % float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
