#!/usr/bin/env python
"""bench.py — column-updates/s per TL+AD dynamics step (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the resident synthetic state: one step_tl (tangent linear)
plus one step_ad (nonlinear forward sweep with stage checkpoints + backward sweep), i.e. what the
reference's %step_tl and %step_ad do for the dynamics (DYN/fv3jedi_lm_dynamics_mod.F90:347,460).
Workload at N=1 (BASELINE.json configs[3] held by one GPU): the six faces of a C192 L127 hydrostatic cubed
sphere (221,184 columns), k_split=2, n_split=6, dt=450 s, 4 tracers, all faces resident on the GPU with
the table-driven face exchange (--tiles cube, default).  --tiles periodic runs one doubly-periodic C192
tile (36,864 columns) per rank instead (the kernel-level workload of round 1's first measurements).
Extra objects on the JSON line: "roofline" (dominant kernel, HIP events on the library's own stream),
"contract" (whole step against BASELINE.md §3's algorithmic bytes) and "cpu_baseline" (the oracle
port timed on all host cores, one single-core worker per core, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)


def algorithmic_bytes(cells, k_split, n_split, nq, nonhydrostatic=False):
    """BASELINE.md §3 contract: B_TL = cells*k_split*(n_split*B_ac + (4nq+10)*8 + B_map); B_TL+AD = 2.75 B_TL.
    B_ac = 880 B (hydrostatic) / 1280 B (non-hydrostatic) per cell and acoustic step; B_map = ((3+nq)*4+12)*8 B hydrostatic
    (320 B at nq = 4), + 64 B non-hydrostatic (w and delz: 384 B at nq = 4)."""
    b_ac = 1280.0 if nonhydrostatic else 880.0
    b_map = ((3 + nq) * 4 + 12) * 8.0 + (64.0 if nonhydrostatic else 0.0)
    b_tl = cells * k_split * (n_split * b_ac + (4 * nq + 10) * 8.0 + b_map)
    return 2.75 * b_tl


def pmc_traffic(kernel, args, cube_mode, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same workload (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, gfx950 correction 2*FETCH+WRITE; tools/rocprof_summary.py), or None."""
    if not (cube_mode and world == 1 and args.nx == 192 and args.npz == 127 and args.nq == 4 and args.k_split == 2 and args.n_split == 6):
        return None, None
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)):
        if f.endswith("_pmc_traffic.csv") and "cube_c192l127" in f:
            best = os.path.join(pdir, f)
    if best is None:
        return None, None
    for line in open(best):
        w = line.strip().split(",")
        if w[0] == kernel:
            return float(w[4]), os.path.relpath(best, ROOT)
    return None, None


def cpu_baseline(args, opt):
    """Oracle port (oracle/liboracle.so) on ALL host cores: P = min(host cores, 16) single-core workers at once (the reference TL/AD have
    no threading; P independent workers, each with its own slab of columns, are the zero-communication upper bound of its MPI
    decomposition).  Bounded sample: one 8x8-column tile per worker with the same npz / k_split / n_split / nq,
    10-30 s of CPU work.  The port's adjoint is a generic operation tape (oracle/scalar.hpp), several times slower than a
    source-transformed adjoint: the tangent-only rate is reported beside it."""
    import subprocess
    P = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        P = min(P, len(os.sched_getaffinity(0)))
    # the GPU box gives one GPU a share of 16 host cores; a worker holds the port's operation tape (about 3 GB for an 8x8-column
    # L127 tile): the pool is sized to both
    P = min(P, 16)
    try:
        avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1048576.0
        P = max(1, min(P, int(avail_gb / 8.0)))
    except Exception:
        pass
    nx = 8
    env = dict(os.environ); env["HIP_VISIBLE_DEVICES"] = ""; env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), str(nx), str(args.npz), str(args.n_split), str(args.k_split),
           str(args.nq), str(args.dt), "1" if args.nonhydrostatic else "0"]
    t0 = time.time()
    procs = [subprocess.Popen(cmd + [str(20250114 + w)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True) for w in range(P)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=900)
        if p.returncode != 0:
            raise RuntimeError("cpu baseline worker failed: " + err[-400:])
        res.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.time() - t0
    t_tl = max(r["t_tl"] for r in res); t_ad = max(r["t_ad"] for r in res)
    cols = sum(r["columns"] for r in res)
    return {"value": cols / (t_tl + t_ad), "unit": "column-updates/s", "cores": P, "kind": "port",
            "tangent_only_value": cols / t_tl,
            "sample": "oracle C++ port, %d single-core workers at once (the host cores of one GPU's share), each one %dx%d-column periodic tile, L%d, k_split=%d "
                      "n_split=%d nq=%d %s: slowest worker TL (dual numbers) %.2f s + AD (taped forward + reverse sweep, %.0fx the tangent) "
                      "%.2f s; wall %.1f s incl. start-up" % (P, nx, nx, args.npz, args.k_split, args.n_split, args.nq,
                                                              "non-hydrostatic" if args.nonhydrostatic else "hydrostatic", t_tl,
                                                              t_ad / max(t_tl, 1e-9), t_ad, wall)}


def host_emulation_library():
    """The product's stage code compiled for the host (tests/_emul/, built by __graft_entry__.build(); -O3 with AVX2 + FMA where the host has
    them, else the tests' -O2 build)."""
    flags = ""
    try:
        flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][0]
    except Exception:
        pass
    fast = os.path.join(ROOT, "tests", "_emul", "libfv3lm_emul_o3.so")
    if os.path.exists(fast) and " avx2" in flags and " fma" in flags:
        return fast, "-O3 -mavx2 -mfma"
    slow = os.path.join(ROOT, "tests", "_emul", "libfv3lm_emul.so")
    return (slow, "-O2") if os.path.exists(slow) else (None, "")


def cpu_baseline_emul(args):
    """Second CPU comparator: the product's own algorithm with its hand-written adjoints on the host cores -- P single-core
    workers at once, each one 32 x 32-column tile of the workload's depth (1.5 MB per field, ~150 fields live per acoustic step: out of the core's caches), step_tl + step_nl +
    step_ad like the timed region.  Trajectory slots as memory allows (a worker with all of them recomputes nothing in its backward
    sweep, like the reference's tape)."""
    import subprocess
    so, how = host_emulation_library()
    if so is None:
        return None
    P = os.cpu_count() or 1
    if hasattr(os, "sched_getaffinity"):
        P = min(P, len(os.sched_getaffinity(0)))
    P = min(P, 16)
    nx = 32
    per_field = (nx + 7) * (nx + 7) * args.npz * 8 / 2.0 ** 30
    base_gb, slot_gb = 420 * per_field, 125 * per_field        # arenas (trajectory + perturbation sides) / one trajectory slot, generously
    try:
        avail_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1048576.0
    except Exception:
        avail_gb = 32.0
    P = max(1, min(P, int(0.8 * avail_gb / (base_gb + 2 * slot_gb))))
    slots = max(0, min(args.n_split * args.k_split, int((0.8 * avail_gb / P - base_gb) / slot_gb)))
    env = dict(os.environ); env["HIP_VISIBLE_DEVICES"] = ""; env["OMP_NUM_THREADS"] = "1"; env["FV3LM_TRAJ_SLOTS"] = str(slots)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_emul_worker.py"), so, str(nx), str(args.npz), str(args.n_split), str(args.k_split), str(args.nq),
           str(args.dt), "1" if args.nonhydrostatic else "0"]
    t0 = time.time()
    procs = [subprocess.Popen(cmd + [str(20250114 + w)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True) for w in range(P)]
    res = []
    for p in procs:
        out, err = p.communicate(timeout=1500)
        if p.returncode != 0:
            raise RuntimeError("host-emulation worker failed: " + err[-400:])
        res.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.time() - t0
    tt = {k: max(r[k] for r in res) for k in ("t_tl", "t_nl", "t_ad")}
    cols = sum(r["columns"] for r in res)
    return {"value": cols / (tt["t_tl"] + tt["t_nl"] + tt["t_ad"]), "unit": "column-updates/s", "cores": P, "kind": "product host-emulation, hand-written adjoints",
            "sample": "the product's stage code built for the host (g++ %s -DFV3LM_HOST_EMUL, single-threaded loops), %d single-core workers at once, each one %dx%d-column "
                      "periodic tile, L%d, k_split=%d n_split=%d nq=%d, %d of %d trajectory slots: slowest worker step_tl %.1f s + step_nl %.1f s + step_ad %.1f s; wall %.0f s "
                      "incl. start-up.  The reference Fortran is unbuildable here (FMS)" % (how, P, nx, nx, args.npz, args.k_split, args.n_split, args.nq, slots,
                                                                                           args.n_split * args.k_split, tt["t_tl"], tt["t_nl"], tt["t_ad"], wall)}


def scheme_string(o, nh):
    """the scheme flags actually in force (fv3lm_options of the instance that was timed)"""
    sponge = ("%d/%d/%d/%d below level %d" % (o.hord_mt_ks_pert, o.hord_vt_ks_pert, o.hord_tm_ks_pert, o.hord_dp_ks_pert, o.n_sponge_pert)) if o.hord_ks_pert else "off"
    s = "hord mt/vt/tm/dp/tr=%d/%d/%d/%d/%d (sponge: %s), kord=%d, nord=%d" % (o.hord_mt_pert, o.hord_vt_pert, o.hord_tm_pert, o.hord_dp_pert, o.hord_tr_pert, sponge, abs(o.kord_tm), o.nord)
    if (o.hord_mt, o.hord_vt, o.hord_tm, o.hord_dp, o.hord_tr) != (o.hord_mt_pert, o.hord_vt_pert, o.hord_tm_pert, o.hord_dp_pert, o.hord_tr_pert):
        s += ", split_hord: trajectory %d/%d/%d/%d/%d" % (o.hord_mt, o.hord_vt, o.hord_tm, o.hord_dp, o.hord_tr)
    if abs(o.kord_tm) != abs(o.kord_tm_pert):
        s = s.replace("kord=%d" % abs(o.kord_tm), "kord=%d (split_kord: trajectory %d)" % (abs(o.kord_tm_pert), abs(o.kord_tm)))
    if o.split_damp:
        s += ", split_damp: trajectory nord=%d dddmp=%g d4_bg=%g / perturbation nord=%d dddmp=%g d4_bg=%g" % (o.nord, o.dddmp, o.d4_bg, o.nord_pert, o.dddmp_pert, o.d4_bg_pert)
    if nh:
        s += ", a_imp=%g (%s)" % (o.a_imp, "SIM1" if o.a_imp > 0.999 else "SIM")
    return s


def default_layout(world):
    """Whole faces while they deal out evenly (1, 2, 3, 6 GPUs); otherwise the 24 sub-face tiles of a 2 x 2 layout, dealt in position-major
    order: 4 GPUs hold the six tiles of one position each (one tile class, balanced -- six faces on four GPUs would be 2 + 2 + 1 + 1),
    8 GPUs three tiles each."""
    return 1 if 6 % world == 0 else 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nx", type=int, default=192)
    ap.add_argument("--npz", type=int, default=127)
    ap.add_argument("--k_split", type=int, default=2)
    ap.add_argument("--n_split", type=int, default=6)
    ap.add_argument("--dt", type=float, default=450.0)
    ap.add_argument("--nq", type=int, default=4)
    ap.add_argument("--tiles", choices=["cube", "periodic"], default="cube")
    ap.add_argument("--nonhydrostatic", action="store_true",
                    help="BASELINE config 3: w, delz prognostic, nh_core active (hydrostatic = 0); not the headline workload")
    ap.add_argument("--hord-traj", type=int, default=0,
                    help="trajectory advection scheme (3 .. 13) with the perturbation schemes left at their defaults: split_hord (not the headline configuration)")
    ap.add_argument("--kord-traj", type=int, default=0, help="trajectory remap profile (8 .. 15) with the linear perturbation profile: split_kord, hydrostatic only")
    ap.add_argument("--nord-traj", type=int, default=0, help="trajectory divergence-damping order (2 or 3) beside nord_pert = 1: split_damp (not the headline configuration)")
    ap.add_argument("--split-damp", action="store_true", help="split_damp = .true. (the reference's default) with equal namelist values: the perturbation sponge rules differ")
    ap.add_argument("--layout", type=int, default=0,
                    help="tiles per face edge (fv_flags_type%%layout): 1 = whole faces, 2 = 24 sub-face tiles ... Default: 1 where six faces deal out evenly over the GPUs (1, 2, 3, 6), "
                         "else 2 (24 tiles: six per GPU at 4, three at 8 -- six whole faces would leave GPUs idle or half loaded)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-host-transport", action="store_true",
                    help="multi-rank rehearsal on ONE GPU (RCCL refuses two ranks per device): gloo process group, halo messages staged "
                         "through host memory by a transport callback; exercises everything but the RCCL send/recv calls themselves")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="one GPU: every halo row between two resident tiles travels as a message through ncclSend / ncclRecv on a one-rank communicator "
                         "(the rank as its own peer) instead of the local gather -- the cost of the message path itself, not the headline configuration")
    ap.add_argument("--profile-out", default="")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process starts N fresh children, one rank per GPU, before anything here has
        # touched the GPU (no torch import, no HIP call), hands them the torch.distributed environment and passes rank 0's line through.
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        out0, _ = procs[0].communicate()
        rcs = [procs[0].returncode] + [p_.wait() for p_ in procs[1:]]
        sys.stdout.write(out0)
        sys.exit(max(abs(rc) for rc in rcs))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_host_transport:
            local = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import ctypes as C
    import fv3_jedi_linearmodel_amd as fv3
    from fv3_jedi_linearmodel_amd.harness import Case, CubeCase, step_state, cube_step_state, cube_nh_state
    lib = fv3.load_hip_library()
    lib.L.fv3lm_set_device(C.c_int(local))
    cube_mode = args.tiles == "cube"
    active = True
    if cube_mode:
        from fv3_jedi_linearmodel_amd import cube
        from fv3_jedi_linearmodel_amd._lib import comm_init_rccl
        if world > 1 and args.rehearse_host_transport:
            import numpy as np
            from fv3_jedi_linearmodel_amd._lib import set_transport_callback, set_allreduce_callback
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

            def transport(peers, sbufs, rbufs):      # sbufs / rbufs are views of DEVICE memory: stage through host
                reqs, stage = [], []
                for p, r in zip(peers, rbufs):
                    if r.size:
                        h = torch.empty(r.size, dtype=torch.float64); stage.append((r, h)); reqs.append(dist.irecv(h, src=p))
                for p, s_ in zip(peers, sbufs):
                    if s_.size:
                        h = torch.empty(s_.size, dtype=torch.float64)
                        assert hip.hipMemcpy(h.data_ptr(), s_.ctypes.data, s_.size * 8, 2) == 0
                        reqs.append(dist.isend(h, dst=p))
                for q in reqs:
                    q.wait()
                for r, h in stage:
                    assert hip.hipMemcpy(r.ctypes.data, h.data_ptr(), r.size * 8, 1) == 0
            set_transport_callback(lib, transport)
            layout_ = args.layout if args.layout > 0 else default_layout(world)
            grp = dist.new_group(ranks=list(range(min(world, 6 * layout_ * layout_))))
            if rank < min(world, 6 * layout_ * layout_):
                def allmax(buf):
                    t = torch.from_numpy(buf.copy())
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
                    buf[:] = t.numpy()
                set_allreduce_callback(lib, allmax)
        elif world > 1:    # one RCCL communicator for the face exchange; rank 0's unique id travels through torch.distributed
            def bcast(data):
                t = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if rank == 0:
                    t.copy_(torch.tensor(list(data), dtype=torch.uint8))
                dist.broadcast(t, src=0)
                return bytes(t.cpu().tolist())
            layout_ = args.layout if args.layout > 0 else default_layout(world)
            nact = min(world, 6 * layout_ * layout_)      # ranks without a tile stay out of the communicator
            if rank < nact:
                comm_init_rccl(lib, rank, nact, bcast)
            else:
                bcast(None)
            # tracer_2d's per-level max Courant number over all faces: ncclAllReduce(max) inside the library, on the same communicator
        if args.rccl_loopback:
            assert world == 1, "--rccl-loopback is a one-GPU measurement"
            comm_init_rccl(lib, 0, 1, lambda data: data)
        layout = args.layout if args.layout > 0 else default_layout(world)
        ntiles = 6 * layout * layout
        active = len(cube.faces_of(rank, world, ntiles)) > 0
        if active:
            nhkw = dict(hydrostatic=0) if args.nonhydrostatic else {}
            if args.kord_traj:
                nhkw.update(kord_tm=-args.kord_traj, kord_mt=args.kord_traj, kord_tr=args.kord_traj)
            if args.nord_traj:
                nhkw.update(split_damp=1, nord=args.nord_traj)
            if args.split_damp:
                nhkw.update(split_damp=1)
            if args.hord_traj:
                nhkw.update(hord_mt=args.hord_traj, hord_vt=args.hord_traj, hord_tm=args.hord_traj, hord_dp=args.hord_traj, hord_tr=args.hord_traj)
            c = CubeCase(n=args.nx, npz=args.npz, n_split=args.n_split, k_split=args.k_split, dt=args.dt, backend="hip", nq=args.nq,
                         rank=rank, world=world, layout=layout, loopback=args.rccl_loopback, **nhkw)
            T, P = cube_step_state(c)
            if args.nonhydrostatic:
                Tn, Pn = cube_nh_state(c)
                T.update(w=Tn[4], delz=Tn[5]); P.update(w=Pn[4], delz=Pn[5])
    else:
        c = Case(nx=args.nx, ny=args.nx, npz=args.npz, n_split=args.n_split, k_split=args.k_split, dt=args.dt, backend="hip",
                 oracle=False, nq=args.nq, seed=20250114 + rank)
        T, P = step_state(c)
        T = {k: v[None] for k, v in T.items()}; P = {k: v[None] for k, v in P.items()}
    if active:
        names = ["u", "v", "pt", "delp"] + (["w", "delz"] if args.nonhydrostatic else []) + ["q%d" % (n + 1) for n in range(c.nq)]
        for n in names:
            c.dy.put(n, T[n], 0); c.dy.put(n, P[n], 1)
        c.dy.state_save()

    def one_step():
        if not active:
            return
        c.dy.state_restore(); c.dy.step_tl()
        c.dy.state_restore(); c.dy.step_nl(); c.dy.step_ad()

    def barrier():
        if active:
            c.dy.sync()
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if active:
            c.dy.sync()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_host_transport else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    if cube_mode:      # the cube is fixed: strong scaling, faces dealt over ranks
        cols_rank = (args.nx // layout) ** 2 * len(cube.faces_of(0, world, ntiles))
        value = 6 * args.nx * args.nx / (elapsed / args.steps)
    else:
        cols_rank = args.nx * args.nx
        value = world * cols_rank / (elapsed / args.steps)

    # roofline leg: per-kernel HIP-event durations on the library's stream over one more step
    if active:
        c.dy.profile_begin()
        one_step()
        prof = c.dy.profile_end()
    if rank == 0:
        dom = max(prof.items(), key=lambda kv: kv[1][1])
        cnt, ms, by = dom[1]
        achieved = (by / cnt) / (ms / cnt * 1e-3) / 1e9 if ms > 0 and by > 0 else 0.0
        cells = cols_rank * args.npz
        b_step = algorithmic_bytes(cells, args.k_split, args.n_split, args.nq, args.nonhydrostatic)
        contract_gbps = b_step / (ms_per_step * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom[0], args, cube_mode, world)
        out = {
            "metric": "column-updates/s per TL+AD dyn step", "value": value, "unit": "column-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if cube_mode else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C%dL%d %s TL+AD, %s (%d columns per GPU), k_split=%d n_split=%d dt=%gs nq=%d, %s"
                                   % (args.nx, args.npz, "non-hydrostatic" if args.nonhydrostatic else "hydrostatic", ("six cube faces%s dealt over %d GPU(s) (%s per rank), table-driven exchange%s"
                                                           % ("" if layout == 1 else " cut into %d sub-face tiles (layout %d x %d)" % (ntiles, layout, layout), world,
                                                              "/".join(str(len(cube.faces_of(r, world, ntiles))) for r in range(world)),
                                                              (", every row between two tiles as an ncclSend / ncclRecv message of the rank to itself (--rccl-loopback)" if args.rccl_loopback else "") if world == 1 else (", halo messages staged through host memory over gloo (rehearsal on one GPU, NOT RCCL)"
                                                                                    if args.rehearse_host_transport else ", RCCL point-to-point between ranks"))) if cube_mode
                                      else "1 doubly-periodic tile per GPU", cols_rank, args.k_split, args.n_split, args.dt, args.nq, scheme_string(c.opt, args.nonhydrostatic)),
                       "columns_per_gpu": cols_rank, "launches_per_step": sum(v[0] for v in prof.values()),
                       "trajectory_slots": "%d of %d acoustic steps keep their intermediates in HBM (no recompute in the backward sweep)"
                                           % (lib.L.fv3lm_traj_slots(c.dy.h), args.n_split * args.k_split)},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches_per_step": cnt, "avg_launch_ms": ms / cnt, "algorithmic_bytes_per_launch": by / cnt,
                         "share_of_step_time": ms / sum(v[1] for v in prof.values()),
                         # the next kernels by summed time, same accounting (the dominant kernel changed from the tangent to the adjoint
                         # form of fv_tp_2d when the adjoint became one launch: its fraction is not comparable with earlier rounds' line)
                         "next": [{"kernel": k, "launches_per_step": v[0], "avg_launch_ms": v[1] / v[0],
                                   "achieved": (v[2] / 1e9) / (v[1] * 1e-3) if v[1] > 0 else 0.0,
                                   "frac": ((v[2] / 1e9) / (v[1] * 1e-3) / HBM_PEAK_GBPS) if v[1] > 0 else 0.0}
                                  for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[1:4]]},
            "contract": {"algorithmic_bytes_per_step": b_step, "achieved_GBps": contract_gbps, "frac": contract_gbps / HBM_PEAK_GBPS,
                         "note": "whole TL+AD step against BASELINE.md §3 (2.75 x B_TL, %d B per cell and acoustic step); state resident in HBM: the host<->device "
                                 "copies of a drop-in step_tl/step_ad (DESIGN.md §6) are outside the timed region" % (1280 if args.nonhydrostatic else 880)},
        }
        if args.profile_out:
            with open(args.profile_out, "w") as f:
                f.write("# per-kernel HIP-event profile of one TL+AD step (bench.py roofline leg)\n")
                f.write("# kernel launches total_ms avg_ms algorithmic_GB achieved_GBps\n")
                for k, (n_, m_, b_) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                    f.write("%-22s %5d %10.3f %8.4f %9.3f %9.1f\n" % (k, n_, m_, m_ / n_, b_ / 1e9, (b_ / 1e9) / (m_ * 1e-3) if m_ > 0 else 0))
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args, c.opt)
            except Exception as e:   # the baseline leg must never hide the measurement
                out["cpu_baseline"] = {"value": None, "unit": "column-updates/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
            try:
                if not args.nonhydrostatic:
                    em = cpu_baseline_emul(args)
                    if em is not None:
                        out["cpu_baseline"]["host_emulation"] = em
            except Exception as e:
                out["cpu_baseline"]["host_emulation"] = {"value": None, "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
