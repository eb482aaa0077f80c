#!/usr/bin/env python
"""Summaries of rocprofv3 (ROCm 7.2 rocpd sqlite output) runs of bench.py for profiles/:

    python tools/rocprof_summary.py stats  gpurun_out/prof_stats/cube_results.db  profiles/<name>_kernel_stats.csv
    python tools/rocprof_summary.py pmc    gpurun_out/prof_fetch/cube_results.db gpurun_out/prof_write/cube_results.db profiles/<name>_pmc_traffic.csv

stats: per-kernel calls / total / average duration (the `--kernel-trace --stats` summary).
pmc:   per-kernel average FETCH_SIZE and WRITE_SIZE (separate --pmc passes, kilobytes per dispatch) and the HBM bytes per
       launch as MI355X_MICROARCH.md prescribes for gfx950: 2 x FETCH_SIZE (128-B requests tallied at 64 B) + WRITE_SIZE.
"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"k_stage_(nl|tl|ad_alias|ad)<fv3::([A-Za-z0-9_]+(?:<[a-z]+>)?)", name)
    if m:
        return "%s.%s" % (m.group(2).replace("_<true>", "e").replace("_<false>", ""), m.group(1))
    m = re.search(r"k_stage_fw_lds<fv3::(?:Edged<fv3::)?([A-Za-z0-9_]+?)D?(?:_?<|,)[^>]*?(true|false)\s*>*\s*,\s*(true|false)>", name)
    if m:       # LDS-staged forward launch of a bulk stage: last template argument true = tangent, false = nonlinear
        return "%s.%s(lds)" % (m.group(1).rstrip("_"), "tl" if m.group(3) == "true" else "nl")
    m = re.search(r"k_stage_multi_(fw|ad)<fv3::(Edged<fv3::)?([A-Za-z0-9]+?)(D, true>|_<true>|_)?\s*(?:,\s*(true|false))?\s*>\(", name)
    if m:       # the edge strips of a face in one launch
        mode = "ad" if m.group(1) == "ad" else ("tl" if m.group(5) == "true" else "nl")
        return "%s.%s(strips)" % (m.group(3) + ("" if m.group(3).endswith("E") else "e"), mode)
    m = re.search(r"k_stage_ad_lds<fv3::(?:Edged<fv3::)?([A-Za-z0-9_]+?)D?(?:_?<|,)", name)
    if m:
        return "%s.ad(lds)" % m.group(1).rstrip("_")
    m = re.search(r"k_tp_(fused|march)<(fv3::Dual|double)", name)
    if m:       # fv_tp_2d as one launch (tpfused.h): tiled or marching form; Dual = tangent, double = nonlinear (stores the intermediates)
        return "%s.%s" % ("TpFused" if m.group(1) == "fused" else "TpMarch", "tl" if "Dual" in m.group(2) else "nl")
    m = re.search(r"k_tp2<(fv3::Dual|double)", name)
    if m:       # fv_tp_2d as one launch, wavefront layout (tp2.h): the same profile name as the first form
        return "TpFused.%s" % ("tl" if "Dual" in m.group(1) else "nl")
    if "k_zero16" in name:         # field-sized clears (dycore.h dev_zero)
        return "clear"
    if "k_tp_ad_corner" in name:   # corner-alias contributions of the fused adjoint (tpad.h)
        return "TpAd.ad_corner"
    if "k_tp_ad" in name:          # the whole adjoint of fv_tp_2d as one launch (tpad.h)
        return "TpAd.ad"
    if "k_tp_outer_ad" in name:    # hand-written adjoint of the outer sweeps + flux assembly of fv_tp_2d (tpfused.h)
        return "TpOuter.ad"
    m = re.search(r"k_points<fv3::([A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    if m:
        return "points<%s>" % m.group(1)
    return name.split("(")[0][:60]


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("kernel,calls,total_ms,avg_us,min_us,max_us,percent,full_name\n")
        for n, c, s, a, lo, hi in rows:
            f.write('%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f,"%s"\n' % (short(n), c, s / 1e6, a / 1e3, lo / 1e3, hi / 1e3, 100.0 * s / tot, n))
    print("wrote", out, "kernels:", len(rows), "total GPU ms:", tot / 1e6)


def pmc(dbf, dbw, out):
    def load(db, ctr):
        cur = sqlite3.connect(db).cursor()
        return {n: (c, v) for n, c, v in cur.execute(
            "select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? group by kernel_name", (ctr,))}
    F, W = load(dbf, "FETCH_SIZE"), load(dbw, "WRITE_SIZE")
    rows = []
    for n in sorted(set(F) | set(W)):
        cf, vf = F.get(n, (0, 0.0)); cw, vw = W.get(n, (0, 0.0))
        rows.append((short(n), cf, vf, vw, (2.0 * vf + vw) * 1024.0))
    rows.sort(key=lambda r: -r[1] * r[4])
    with open(out, "w") as f:
        f.write("kernel,dispatches,avg_FETCH_SIZE_KB,avg_WRITE_SIZE_KB,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
        for r in rows:
            f.write("%s,%d,%.1f,%.1f,%.0f\n" % r)
    print("wrote", out, "kernels:", len(rows))


def counters(db, out):
    """any --pmc pass: per kernel, dispatch count and the average of every collected counter per dispatch"""
    cur = sqlite3.connect(db).cursor()
    names = [r[0] for r in cur.execute("select distinct counter_name from counters_collection order by 1")]
    tab = {}
    for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        tab.setdefault(k, {})[c] = (n, v)
    with open(out, "w") as f:
        f.write("kernel,dispatches," + ",".join(names) + "\n")
        for k in sorted(tab, key=lambda k: -sum(n * v for n, v in tab[k].values())):
            f.write("%s,%d,%s\n" % (short(k), max(n for n, _ in tab[k].values()), ",".join("%.4g" % tab[k].get(c, (0, 0.0))[1] for c in names)))
    print("wrote", out, "kernels:", len(tab))


if __name__ == "__main__":
    if sys.argv[1] == "counters":
        counters(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
