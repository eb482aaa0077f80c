"""The Fortran host in miniature (fortran/shim_driver.F90 + fortran/fv3lm_hip_mod.F90, linked by amdflang against the product
library on the MI355X, or against the host-emulation build for the CPU run) against the same calls made through ctypes:
step_tl outputs (trajectory, perturbation) and step_ad outputs, bit for bit, on the reference's own array shapes
(u(isd:ied, jsd:jed+1, npz), v(isd:ied+1, jsd:jed, npz), pt/delp/q(isd:ied, jsd:jed, npz)); and the error path (status ->
fv3lm_last_error -> fatal exit, the reference's failure mode src/fv3jedi_lm_mod.F90:93)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "fortran")


def build_driver(libdir, libname, out):
    """amdflang: shim module + driver -> executable linked against lib<libname>.so in libdir (rpath set)"""
    srcs = [os.path.join(FDIR, "fv3lm_hip_mod.F90"), os.path.join(FDIR, "shim_driver.F90")]
    lib = os.path.join(libdir, "lib%s.so" % libname)
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs + [lib]):
        return out
    mod = os.path.join(os.path.dirname(out), "mod_" + libname)
    os.makedirs(mod, exist_ok=True)
    # compiled from copies inside the build directory: a stale fv3lm_hip_mod.mod next to the sources must not be picked up
    import shutil
    for s_ in srcs:
        shutil.copy(s_, mod)
    subprocess.check_call(["amdflang", "-cpp", "-fPIC", "-c", "fv3lm_hip_mod.F90", "-o", "fv3lm_hip_mod.o"], cwd=mod)
    subprocess.check_call(["amdflang", "-cpp", "shim_driver.F90", "fv3lm_hip_mod.o", "-o", out, "-L", libdir, "-l" + libname, "-Wl,-rpath," + libdir], cwd=mod)
    return out


def _fa(a):
    """padded plane [nk, pj, pi] -> Fortran (i, j, k) order bytes"""
    return np.asfortranarray(np.transpose(a, (2, 1, 0)))


def run_shim_check(c, driver, tmpdir):
    from groups import step_state
    T, P = step_state(c)
    nx, ny, npz, nq = c.nx, c.ny, c.npz, c.nq
    names = ["u", "v", "pt", "delp"] + ["q%d" % (n + 1) for n in range(nq)]
    # the reference's array shapes as slices of the padded plane (isd:ied+1, jsd:jed+1)
    sl = {"u": (slice(None), slice(0, ny + 7), slice(0, nx + 6)), "v": (slice(None), slice(0, ny + 6), slice(0, nx + 7))}
    cell = (slice(None), slice(0, ny + 6), slice(0, nx + 6))

    def cut(n, a):
        return _fa(a[sl.get(n, cell)])
    fin, fout = os.path.join(tmpdir, "shim_in.bin"), os.path.join(tmpdir, "shim_out.bin")

    def write_input(bad):
        with open(fin, "wb") as f:
            for st in (c.dims, c.opt):
                raw = bytes(st)
                f.write(np.int32(len(raw)).tobytes()); f.write(raw)
            f.write(np.array([c.da_min, c.da_min_c]).tobytes())
            mnames = c.lib.metric_names()
            f.write(_fa(np.stack([c.metrics[n][0] for n in mnames], axis=0)).tobytes(order="F"))
            f.write(_fa(c.phis).tobytes(order="F")); f.write(np.asarray(c.ak, dtype=np.float64).tobytes()); f.write(np.asarray(c.bk, dtype=np.float64).tobytes())
            for D in (T, P):
                for n in ["u", "v", "pt", "delp"]:
                    f.write(cut(n, D[n]).tobytes(order="F"))
                for n in range(nq):       # q(isd:ied, jsd:jed, npz, nq)
                    f.write(cut("q", D["q%d" % (n + 1)]).tobytes(order="F"))
            f.write(np.int32(bad).tobytes())
    # ---- the error path first: a refused option must end the Fortran host with the library's message
    write_input(1)
    r = subprocess.run([driver, fin, fout], capture_output=True, text=True)
    assert r.returncode != 0 and "FATAL fv3lm_hip create" in r.stdout and "nord" in r.stdout, (r.returncode, r.stdout, r.stderr)
    # ---- the real run
    write_input(0)
    r = subprocess.run([driver, fin, fout], capture_output=True, text=True)
    assert r.returncode == 0 and "shim_driver OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    raw = np.fromfile(fout, dtype=np.float64)
    shapes = {"u": (nx + 6, ny + 7, npz), "v": (nx + 7, ny + 6, npz)}
    pos = 0
    got = {}
    for tag in ("tl0", "tl1", "ad1"):
        for n in names:
            shp = shapes.get(n, (nx + 6, ny + 6, npz)); cnt = int(np.prod(shp))
            got[(tag, n)] = raw[pos:pos + cnt].reshape(shp, order="F"); pos += cnt
    nb = nx * ny * npz
    bnd = {}
    for n in names:
        bnd[n] = raw[pos:pos + nb].reshape((nx, ny, npz), order="F"); pos += nb
    assert pos == raw.size
    # ---- the same calls through ctypes; the shim zero-fills what lies outside the reference's array bounds
    def put(D, which):
        for n in names:
            a = np.zeros_like(D[n]); s_ = sl.get(n, cell); a[s_] = D[n][s_]
            c.dy.put(n, a[None], which)
    put(T, 0); put(P, 1); c.dy.step_tl()
    ref = {("tl0", n): c.dy.get(n, 0)[0] for n in names}; ref.update({("tl1", n): c.dy.get(n, 1)[0] for n in names})
    put(T, 0); put(P, 1); c.dy.step_nl(); c.dy.step_ad()
    ref.update({("ad1", n): c.dy.get(n, 1)[0] for n in names})
    for key, a in got.items():
        b = np.transpose(ref[key][sl.get(key[1], cell)], (2, 1, 0))
        assert np.array_equal(a, b), key
        assert np.isfinite(a).all() and np.abs(a).max() > 0, key
    # ---- the boundary entry points through the shim (compact arrays, no halo) against the same calls through ctypes
    I = (slice(None), slice(None), slice(3, 3 + ny), slice(3, 3 + nx))
    c.dy.traj_to_fv3({**{n: T[n][None][I] for n in names}, "phis": np.ascontiguousarray(c.phis[:, 3:3 + ny, 3:3 + nx])})
    c.dy.pert_to_fv3({n: P[n][None][I] for n in names})
    c.dy.step_tl()
    out = c.dy.fv3_to_pert(names)
    for n in names:
        assert np.array_equal(bnd[n], np.transpose(out[n][0], (2, 1, 0))), ("boundary", n)
