"""Ad-hoc timing of step_tl / step_nl+step_ad on the GPU (development aid, not the bench contract)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from common import Case
from groups import step_state
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 48
npz = int(sys.argv[2]) if len(sys.argv) > 2 else 72
ks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = Case(nx=nx, ny=nx, npz=npz, n_split=6, k_split=ks, dt=900.0, backend="hip", oracle=False, nq=4)
T, P = step_state(c)
names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(c.nq)]
def put():
    for n in names:
        c.dy.put(n, T[n][None], 0); c.dy.put(n, P[n][None], 1)
for name, fn in (("step_tl", c.dy.step_tl), ("step_nl", c.dy.step_nl)):
    put(); fn(); c.dy.sync()
    ts = []
    for _ in range(3):
        put(); c.dy.sync(); l0 = c.dy.launch_count(); t0 = time.time(); fn(); c.dy.sync(); ts.append(time.time() - t0)
    print("%s C%dL%d k_split=%d: %.3f ms (%d launches)" % (name, nx, npz, ks, min(ts) * 1e3, c.dy.launch_count() - l0)); sys.stdout.flush()
ts = []
for _ in range(3):
    put(); c.dy.step_nl(); c.dy.sync(); t0 = time.time(); c.dy.step_ad(); c.dy.sync(); ts.append(time.time() - t0)
print("step_ad (backward sweep only): %.3f ms" % (min(ts) * 1e3))
