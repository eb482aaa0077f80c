"""CPU (-m "not gpu") test of the N > 1 path: the six faces dealt over world_size ranks (one process per GPU in
production), halo exchange through per-peer send/receive lists.  Here: gloo, host-emulation build, transport callback;
each rank's faces must reproduce the single-process six-face run (TL step and AD step) to round-off."""
import os
import re
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,nh,layout", [(2, 0, 1), (4, 0, 1), (6, 0, 1), (2, 1, 1), (8, 0, 2), (5, 1, 2)])
def test_faces_over_ranks(world, nh, layout):
    """6 / layout 1: one face per rank, the partition of BASELINE config 4; 8 / layout 2: 24 sub-face tiles, three per rank, the partition
    of BASELINE config 5 (here C16, not C384); 5 ranks: an uneven deal (5 + 5 + 5 + 5 + 4 tiles), non-hydrostatic"""
    from common import build_emul
    build_emul()
    port = 29610 + world + 10 * nh + 20 * layout
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   FV3LM_DIST_NH=str(nh), FV3LM_DIST_LAYOUT=str(layout))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    m = re.search(r"DIST_WORST ([0-9.e+-]+)", outs[0])
    assert m, outs[0]
    assert float(m.group(1)) < (1e-10 if nh else 1e-12), outs[0]
