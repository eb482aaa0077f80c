"""Ad-hoc timing of dyn_core TL / NL+AD on the GPU (development aid, not the bench contract)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import Case
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 48
npz = int(sys.argv[2]) if len(sys.argv) > 2 else 72
c = Case(nx=nx, ny=nx, npz=npz, n_split=6, dt=900.0, backend="hip", oracle=False)
c.put_state(pert=c.pert)
for mode, name in ((1, "TL"), (0, "NL")):
    c.dy.dyn_core(mode); c.dy.sync()
    l0 = c.dy.launch_count(); t0 = time.time()
    for _ in range(3):
        c.put_state(pert=c.pert) if False else None
        c.dy.dyn_core(mode)
    c.dy.sync(); dt = (time.time() - t0) / 3
    print("%s dyn_core C%dL%d: %.3f ms  (%d launches)" % (name, nx, npz, dt * 1e3, (c.dy.launch_count() - l0) // 3)); sys.stdout.flush()
c.put_state()
c.dy.dyn_core(0)
c.dy.dyn_core(2); c.dy.sync()
t0 = time.time()
for _ in range(3):
    c.dy.dyn_core(2)
c.dy.sync()
print("AD dyn_core (recompute + backward): %.3f ms" % ((time.time() - t0) / 3 * 1e3))
