"""Kernel-level timing on the MI355X: the headline workload (six C192 L127 faces) or a smaller one, one step_tl + step_nl + step_ad under the
library's HIP-event profiler, per-kernel table filtered by name.  Usage: python tools/kbench.py [--lib path.so] [--nx 192] [--npz 127] [--grep Tp]
Variants of one translation unit are compared by linking them into differently named libraries (FV3LM_LIB)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default="")
    ap.add_argument("--nx", type=int, default=192)
    ap.add_argument("--npz", type=int, default=127)
    ap.add_argument("--nq", type=int, default=4)
    ap.add_argument("--grep", default="")
    ap.add_argument("--top", type=int, default=25)
    ap.add_argument("--nh", action="store_true")
    ap.add_argument("--group", default="", help="after one ordinary step: time this kernel group alone (TL and NL), e.g. d_sw")
    ap.add_argument("--ad", action="store_true", help="group mode: time the adjoint of the group instead of TL + NL")
    ap.add_argument("--experiment", default="", help="comma list of FV3LM_TP2_EXPERIMENT values to time the group with (experiment builds only)")
    args = ap.parse_args()
    if args.lib:
        os.environ["FV3LM_LIB"] = os.path.abspath(args.lib)
    from fv3_jedi_linearmodel_amd.harness import CubeCase, cube_step_state, cube_nh_state
    kw = dict(hydrostatic=0) if args.nh else {}
    c = CubeCase(n=args.nx, npz=args.npz, n_split=6, k_split=2, dt=450.0, backend="hip", nq=args.nq, **kw)
    T, P = cube_step_state(c)
    if args.nh:
        Tn, Pn = cube_nh_state(c)
        T.update(w=Tn[4], delz=Tn[5]); P.update(w=Pn[4], delz=Pn[5])
    names = ["u", "v", "pt", "delp"] + (["w", "delz"] if args.nh else []) + ["q%d" % (n + 1) for n in range(c.nq)]
    for n in names:
        c.dy.put(n, T[n], 0); c.dy.put(n, P[n], 1)
    c.dy.state_save()

    def step():
        c.dy.state_restore(); c.dy.step_tl(); c.dy.state_restore(); c.dy.step_nl(); c.dy.step_ad()
    step(); c.dy.sync()
    if args.group:
        for ex_ in [e for e in (args.experiment.split(",") if args.experiment else ["0"])]:
            os.environ["FV3LM_TP2_EXPERIMENT"] = ex_
            modes = (2,) if args.ad else (1, 0)
            for m_ in modes:
                c.dy.run_group(args.group, m_)
            c.dy.sync()
            c.dy.profile_begin()
            for _ in range(3):
                for m_ in modes:
                    c.dy.run_group(args.group, m_)
            prof = c.dy.profile_end()
            print("group %s, experiment %s:" % (args.group, ex_))
            for k, (n_, m_, b_) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
                if not args.grep or any(g in k for g in args.grep.split(",")):
                    print("  %-22s %5d %10.3f ms %8.4f ms/launch %9.1f GB/s" % (k, n_, m_, m_ / n_, (b_ / 1e9) / (m_ * 1e-3) if m_ > 0 else 0))
        return
    t0 = time.perf_counter(); step(); step(); c.dy.sync(); dt = (time.perf_counter() - t0) / 2
    c.dy.profile_begin(); step(); prof = c.dy.profile_end()
    tot = sum(v[1] for v in prof.values())
    print("lib %s: %.1f ms per TL+AD step (kernel time %.1f ms, %d launches)" % (args.lib or "default", 1e3 * dt, tot, sum(v[0] for v in prof.values())))
    rows = sorted(prof.items(), key=lambda kv: -kv[1][1])
    if args.grep:
        rows = [r for r in rows if any(g in r[0] for g in args.grep.split(","))]
    for k, (n_, m_, b_) in rows[:args.top]:
        print("  %-22s %5d %10.3f ms %8.4f ms/launch %9.1f GB/s" % (k, n_, m_, m_ / n_, (b_ / 1e9) / (m_ * 1e-3) if m_ > 0 else 0))


if __name__ == "__main__":
    main()
