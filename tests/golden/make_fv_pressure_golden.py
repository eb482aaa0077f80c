"""Generates tests/golden/fv_pressure_ref.npz by running the reference's own
compute_fv3_pressures_tlm / _bwd (oracle/_ref/libfvpressure_ref.so, built by oracle/ref/Makefile from
/root/reference/src/dynamics/atmos_cubed_sphere/model_tlmadm/fv_pressure.F90) on seeded inputs.
The fixture holds data only (inputs + reference outputs) in the padded-plane layout."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libfvpressure_ref.so")
_dp = C.POINTER(C.c_double)


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def run_reference(delp_pad, delp_tl_pad, nx, ny, npz, kappa, ptop, seeds_pad):
    L = C.CDLL(REF_SO)
    is_, ie, js, je, isd, ied, jsd, jed = 1, nx, 1, ny, -2, nx + 3, -2, ny + 3
    # reference shapes (Fortran order): delp(isd:ied,jsd:jed,npz), pe(is-1:ie+1,npz+1,js-1:je+1),
    # pk(is:ie,js:je,npz+1), peln(is:ie,npz+1,js:je), pkz(is:ie,js:je,npz)
    def from_pad(a, i0, i1, j0, j1):
        return _f(a[:, j0 + 2:j1 + 3, i0 + 2:i1 + 3].transpose(2, 1, 0))
    delp, delp_tl = from_pad(delp_pad, isd, ied, jsd, jed), from_pad(delp_tl_pad, isd, ied, jsd, jed)
    pe = _f(np.zeros((nx + 2, npz + 1, ny + 2))); pe_tl = _f(pe.copy())
    pk = _f(np.zeros((nx, ny, npz + 1))); pk_tl = _f(pk.copy())
    peln = _f(np.zeros((nx, npz + 1, ny))); peln_tl = _f(peln.copy())
    pkz = _f(np.zeros((nx, ny, npz))); pkz_tl = _f(pkz.copy())
    ints = [C.c_int(v) for v in (is_, ie, js, je, isd, ied, jsd, jed, npz)]
    P = lambda a: a.ctypes.data_as(_dp)
    L.ref_pressures_tlm(*ints, C.c_double(kappa), C.c_double(ptop), P(delp), P(delp_tl), P(pe), P(pe_tl), P(pk), P(pk_tl),
                        P(pkz), P(pkz_tl), P(peln), P(peln_tl))

    def to_pad(a_ijk, i0, j0):
        ni, nj, nk = a_ijk.shape
        out = np.zeros((nk, ny + 7, nx + 7))
        out[:, j0 + 2:j0 + 2 + nj, i0 + 2:i0 + 2 + ni] = a_ijk.transpose(2, 1, 0)
        return out
    res = {}
    res["pe"], res["pe_tl"] = to_pad(pe.transpose(0, 2, 1), 0, 0), to_pad(pe_tl.transpose(0, 2, 1), 0, 0)
    res["pk"], res["pk_tl"] = to_pad(pk, 1, 1), to_pad(pk_tl, 1, 1)
    res["peln"], res["peln_tl"] = to_pad(peln.transpose(0, 2, 1), 1, 1), to_pad(peln_tl.transpose(0, 2, 1), 1, 1)
    res["pkz"], res["pkz_tl"] = to_pad(pkz, 1, 1), to_pad(pkz_tl, 1, 1)
    # adjoint
    delp_ad = _f(np.zeros_like(delp))
    pe_ad = _f(from_pad(seeds_pad["pe"], 0, nx + 1, 0, ny + 1).transpose(0, 2, 1))
    pk_ad = _f(from_pad(seeds_pad["pk"], 1, nx, 1, ny))
    peln_ad = _f(from_pad(seeds_pad["peln"], 1, nx, 1, ny).transpose(0, 2, 1))
    pkz_ad = _f(from_pad(seeds_pad["pkz"], 1, nx, 1, ny))
    L.ref_pressures_bwd(*ints, C.c_double(kappa), C.c_double(ptop), P(delp), P(delp_ad), P(pe), P(pe_ad), P(pk), P(pk_ad),
                        P(pkz), P(pkz_ad), P(peln), P(peln_ad))
    res["delp_ad"] = to_pad(delp_ad, isd, jsd)
    return res


def main():
    nx, ny, npz, kappa, ptop = 9, 7, 11, 2.0 / 7.0, 1.0
    rng = np.random.default_rng(20250114)
    pj, pi = ny + 7, nx + 7
    delp = 50.0 + 900.0 * rng.random((npz, pj, pi)) * np.linspace(0.05, 1.0, npz)[:, None, None]
    delp_tl = 10.0 * rng.standard_normal((npz, pj, pi))
    seeds = {}
    for n, nk in (("pe", npz + 1), ("pk", npz + 1), ("pkz", npz), ("peln", npz + 1)):
        s = np.zeros((nk, pj, pi))
        s[:, 3:3 + ny, 3:3 + nx] = rng.standard_normal((nk, ny, nx))   # the reference's bwd touches is..ie only
        seeds[n] = s
    out = run_reference(delp, delp_tl, nx, ny, npz, kappa, ptop, seeds)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fv_pressure_ref.npz"), nx=nx, ny=ny, npz=npz, kappa=kappa,
                        ptop=ptop, delp=delp, delp_tl=delp_tl, **{k + "_adseed": v for k, v in seeds.items()}, **out)
    print("wrote fv_pressure_ref.npz")


if __name__ == "__main__":
    main()
