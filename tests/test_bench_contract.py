"""bench.py's own arithmetic and helpers (CPU): the contract byte count against BASELINE.md §3's worked example, the workload
string derived from the options in force, and one worker of the cpu_baseline leg end to end on a tiny tile."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_contract_bytes_match_baseline_md():
    cells = 6 * 192 * 192 * 127
    # C192L127 hydrostatic, k_split=2, n_split=6, nq=4 -> 326 GB (TL), 897 GB (TL+AD)
    b = bench.algorithmic_bytes(cells, 2, 6, 4)
    assert abs(b / 2.75 - 326e9) < 1e9 and abs(b - 897e9) < 2e9
    # non-hydrostatic: 1280 B per cell and acoustic step, 384 B remap
    bn = bench.algorithmic_bytes(cells, 2, 6, 4, nonhydrostatic=True)
    assert abs(bn / 2.75 / cells / 2 - (6 * 1280 + 208 + 384)) < 1e-6


def test_workload_string_follows_the_options():
    import fv3_jedi_linearmodel_amd as fv3
    o = fv3.default_options()
    s = bench.scheme_string(o, False)
    assert "2/2/2/2/2" in s and "sponge: 1/1/1/1 below level 9" in s and "a_imp" not in s
    o2 = fv3.default_options(hord_ks_pert=0, hord_ks_traj=0, hydrostatic=0, a_imp=1.0)
    s2 = bench.scheme_string(o2, True)
    assert "sponge: off" in s2 and "a_imp=1 (SIM1)" in s2


def test_cpu_baseline_worker_runs():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), "8", "6", "2", "1", "1", "450", "0", "3"],
                                  text=True)
    r = json.loads(out.strip().splitlines()[-1])
    assert r["columns"] == 64 and r["t_tl"] > 0 and r["t_ad"] > 0
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline_worker.py"), "8", "6", "2", "1", "1", "450", "1", "3"],
                                  text=True)
    assert json.loads(out.strip().splitlines()[-1])["columns"] == 64


def test_default_layout_balances_the_gpus():
    """whole faces where six deal out evenly, the 24 sub-face tiles elsewhere; every GPU then holds the same number of tiles at 4 and 8"""
    from fv3_jedi_linearmodel_amd import cube
    assert [bench.default_layout(w) for w in range(1, 9)] == [1, 1, 1, 2, 2, 1, 2, 2]
    for w in (1, 2, 3, 6):
        assert len({len(cube.faces_of(r, w, 6)) for r in range(w)}) == 1
    for w in (4, 8):
        assert len({len(cube.faces_of(r, w, 24)) for r in range(w)}) == 1
    # position-major order: the six tiles of a 4-GPU rank share one window of their faces (one tile class)
    tl = cube.tiles(48, 2)
    for r in range(4):
        assert len({(tl[t][1], tl[t][2]) for t in cube.faces_of(r, 4, 24)}) == 1
