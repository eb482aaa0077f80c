// fv3lm-hip: column (k-sequential) operators and data-motion kernels that do not fit the stencil
// stage pattern: geopk (hydrostatic integrals), doubly-periodic halo fill and its adjoint, flux
// accumulators.  One thread per (i,j) column, i fastest → coalesced at every level.
#pragma once
#include "exec.h"

namespace fv3 {

// ---------------------------------------------------------------- geopk (dyn_core_tlm.F90:4763-4905)
// p = ptop + cumsum(delp); logp = ln p; pk = exp(akap*logp); gz = hs + reverse-cumsum(cp_air*pt*dpk);
// pkz = dpk / (akap * dlogp) (D-grid call only).  pe/peln/pkz are written on the whole computed
// rectangle (a superset of the reference's is-1..ie+1 / is..ie ranges).
template <class T> struct FIO;
template <> struct FIO<double> {
  HD static double ld(const Fld& f, size_t n) { return f.t[n]; }
  HD static void st(const Fld& f, size_t n, double x) { f.t[n] = x; }
};
template <> struct FIO<Dual> {
  HD static Dual ld(const Fld& f, size_t n) { return Dual(f.t[n], f.p[n]); }
  HD static void st(const Fld& f, size_t n, const Dual& x) { f.t[n] = x.v; f.p[n] = x.d; }
};
struct GeopkArgs {
  Geom g; Rect R;
  Fld delp, pt;          // in  (npz)
  Fld pe, peln, pk, gz;  // out (npz+1)
  Fld pkz;               // out (npz), D-grid call only
  const double* hs;      // [ntile][plane]
  double ptop, akap, cp_air; int cg;
};
template <class T>
HD void geopk_col(const GeopkArgs& a, int i, int j, int tile) {
  const Geom& g = a.g; const int km = g.npz; const size_t pl = g.plane, o = g.idx(i, j);
  const size_t b0 = (size_t)tile * km * pl + o, b1 = (size_t)tile * (km + 1) * pl + o;
  auto ld = [&](const Fld& f, size_t n) -> T { return FIO<T>::ld(f, n); };
  auto st = [&](const Fld& f, size_t n, const T& x) { FIO<T>::st(f, n, x); };
  T p1d = T(a.ptop);
  const double ptk = pow(a.ptop, a.akap), peln1 = log(a.ptop);
  st(a.pe, b1, p1d); st(a.peln, b1, T(peln1)); st(a.pk, b1, T(ptk));
  T pkm = T(ptk), lnm = T(peln1);
#pragma unroll 4
  for (int k = 2; k <= km + 1; ++k) {
    p1d = p1d + ld(a.delp, b0 + (size_t)(k - 2) * pl);
    T lp = dlog(p1d), pkk = dexp(a.akap * lp);
    st(a.pe, b1 + (size_t)(k - 1) * pl, p1d); st(a.peln, b1 + (size_t)(k - 1) * pl, lp); st(a.pk, b1 + (size_t)(k - 1) * pl, pkk);
    if (!a.cg) st(a.pkz, b0 + (size_t)(k - 2) * pl, (pkk - pkm) / (a.akap * (lp - lnm)));
    pkm = pkk; lnm = lp;
  }
  T g1d = T(a.hs[(size_t)tile * pl + o]);
  st(a.gz, b1 + (size_t)km * pl, g1d);
  T pk1 = pkm;                                     // pk(km+1), still in a register; pk(k) is loaded once per level
#pragma unroll 4
  for (int k = km; k >= 1; --k) {
    const T pk0 = ld(a.pk, b1 + (size_t)(k - 1) * pl);
    g1d = g1d + a.cp_air * ld(a.pt, b0 + (size_t)(k - 1) * pl) * (pk1 - pk0);
    st(a.gz, b1 + (size_t)(k - 1) * pl, g1d);
    pk1 = pk0;
  }
}
// adjoint: consumes gz.p, pk.p, pe.p, peln.p, pkz.p; accumulates into delp.p, pt.p
HD void geopk_col_ad(const GeopkArgs& a, int i, int j, int tile) {
  const Geom& g = a.g; const int km = g.npz; const size_t pl = g.plane, o = g.idx(i, j);
  const size_t b0 = (size_t)tile * km * pl + o, b1 = (size_t)tile * (km + 1) * pl + o;
  double g_ad = 0.0;
  for (int k = 1; k <= km; ++k) {
    const size_t n0 = b0 + (size_t)(k - 1) * pl, m0 = b1 + (size_t)(k - 1) * pl, m1 = b1 + (size_t)k * pl;
    g_ad += a.gz.p[m0];
    const double ptk = a.pt.t[n0], dpk = a.pk.t[m1] - a.pk.t[m0];
    a.pt.p[n0] += a.cp_air * dpk * g_ad;
    a.pk.p[m1] += a.cp_air * ptk * g_ad;
    a.pk.p[m0] -= a.cp_air * ptk * g_ad;
    if (!a.cg) {
      const double den = a.akap * (a.peln.t[m1] - a.peln.t[m0]);
      const double za = a.pkz.p[n0] / den;
      a.pk.p[m1] += za; a.pk.p[m0] -= za;
      const double zb = -a.pkz.t[n0] * a.akap * za;
      a.peln.p[m1] += zb; a.peln.p[m0] -= zb;
    }
  }
  double carry = 0.0;
  for (int k = km + 1; k >= 2; --k) {
    const size_t m = b1 + (size_t)(k - 1) * pl;
    const double l_ad = a.peln.p[m] + a.akap * a.pk.t[m] * a.pk.p[m];
    const double p_ad = a.pe.p[m] + l_ad / a.pe.t[m] + carry;
    a.delp.p[b0 + (size_t)(k - 2) * pl] += p_ad;
    carry = p_ad;
  }
}
template <int MODE>      // one kernel per mode: separate register allocations
struct GeopkFn {
  GeopkArgs a;
  HD void operator()(int i, int j, int z) const {
    // the corner-halo cells of a cube face hold no exchanged data (the reference integrates whatever the halo
    // buffers contain there and never reads the result)
    if (a.g.face && (i < 1 || i > a.g.nx) && (j < 1 || j > a.g.ny)) return;
    if (MODE == MODE_NL) geopk_col<double>(a, i, j, z);
    else if (MODE == MODE_TL) geopk_col<Dual>(a, i, j, z);
    else geopk_col_ad(a, i, j, z);
  }
};
inline void run_geopk(Exec& ex, int mode, const GeopkArgs& a0) {
  GeopkArgs a = a0;
  if (a.hs) a.hs += ex.cls_off;      // surface geopotential of the class's first tile (one plane per tile)
  a.delp = ex.sh(a.delp); a.pt = ex.sh(a.pt); a.pe = ex.sh(a.pe); a.peln = ex.sh(a.peln); a.pk = ex.sh(a.pk); a.gz = ex.sh(a.gz); a.pkz = ex.sh(a.pkz);
  // algorithmic bytes: delp, pt in; pe, peln, pk, gz (+pkz) out; x2 for TL; adjoint reads/updates the same set
  const double cells = double(a.R.i1 - a.R.i0 + 1) * (a.R.j1 - a.R.j0 + 1) * a.g.ntile * a.g.npz;
  const double per = (a.cg ? 6. : 7.) * (mode == MODE_NL ? 1. : mode == MODE_TL ? 2. : 3.);
  if (mode == MODE_NL) for_points(ex, a.R, a.g.ntile, GeopkFn<MODE_NL>{a}, "geopk.nl", 8. * per * cells);
  else if (mode == MODE_TL) for_points(ex, a.R, a.g.ntile, GeopkFn<MODE_TL>{a}, "geopk.tl", 8. * per * cells);
  else for_points(ex, a.R, a.g.ntile, GeopkFn<MODE_AD>{a}, "geopk.ad", 8. * per * cells);
}

// ---------------------------------------------------------------- doubly-periodic halo fill
// Every point of the padded plane outside 1..nx x 1..ny takes the value of its periodic image
// (single-tile mode; the 6-face cube exchange replaces this in multi-tile mode).
HD int wrap1(int i, int n) { int r = (i - 1) % n; if (r < 0) r += n; return r + 1; }
struct HaloFn {
  Geom g; Fld f; int mode;   // MODE_NL: traj only, MODE_TL: traj+pert
  HD void operator()(int i, int j, int z) const {
    if (i >= 1 && i <= g.nx && j >= 1 && j <= g.ny) return;
    const size_t b = (size_t)z * g.plane, d = b + g.idx(i, j), s = b + g.idx(wrap1(i, g.nx), wrap1(j, g.ny));
    f.t[d] = f.t[s];
    if (mode == MODE_TL) f.p[d] = f.p[s];
  }
};
// adjoint: interior point gathers the adjoints of all its halo images, then the halo is zeroed
struct HaloAdGatherFn {
  Geom g; Fld f;
  HD void operator()(int i, int j, int z) const {   // launched over the interior
    const size_t b = (size_t)z * g.plane;
    double acc = 0.0;
    for (int bj = -1; bj <= 1; ++bj) {
      const int jj = j + bj * g.ny;
      if (jj < g.jsd() || jj > g.jed() + 1) continue;
      for (int bi = -1; bi <= 1; ++bi) {
        if (bi == 0 && bj == 0) continue;
        const int ii = i + bi * g.nx;
        if (ii < g.isd() || ii > g.ied() + 1) continue;
        acc += f.p[b + g.idx(ii, jj)];
      }
    }
    f.p[b + g.idx(i, j)] += acc;
  }
};
struct HaloAdZeroFn {
  Geom g; Fld f;
  HD void operator()(int i, int j, int z) const {
    if (i >= 1 && i <= g.nx && j >= 1 && j <= g.ny) return;
    f.p[(size_t)z * g.plane + g.idx(i, j)] = 0.0;
  }
};
inline void run_halo(Exec& ex, int mode, const Geom& g, const Fld& f0) {
  const Fld f = ex.sh(f0);
  const Rect full{g.isd(), g.ied() + 1, g.jsd(), g.jed() + 1}, inner{1, g.nx, 1, g.ny};
  const int nz = g.ntile * f.nk;
  if (mode == MODE_AD) {
    for_points(ex, inner, nz, HaloAdGatherFn{g, f}, "halo.ad");
    for_points(ex, full, nz, HaloAdZeroFn{g, f}, "halo.ad");
  } else {
    for_points(ex, full, nz, HaloFn{g, f, mode}, "halo");
  }
}

// ---------------------------------------------------------------- acc += x on a rectangle
struct AccumFn {
  Geom g; Fld acc, x; int mode;
  HD void operator()(int i, int j, int z) const {
    const size_t n = (size_t)z * g.plane + g.idx(i, j);
    if (mode == MODE_AD) { x.p[n] += acc.p[n]; return; }
    acc.t[n] += x.t[n];
    if (mode == MODE_TL) acc.p[n] += x.p[n];
  }
};
inline void run_accum(Exec& ex, int mode, const Geom& g, const Fld& acc, const Fld& x, const Rect& R) {
  for_points(ex, R, g.ntile * acc.nk, AccumFn{g, ex.sh(acc), ex.sh(x), mode}, "accum");
}

// ---------------------------------------------------------------- plane copy of the interior+everything (state hand-over)
struct CopyFn {
  Geom g; Fld dst, src; int what;   // 1 traj, 2 pert, 3 both
  HD void operator()(int i, int j, int z) const {
    const size_t n = (size_t)z * g.plane + g.idx(i, j);
    if (what & 1) dst.t[n] = src.t[n];
    if (what & 2) dst.p[n] = src.p[n];
  }
};
inline void run_copy(Exec& ex, const Geom& g, const Fld& dst, const Fld& src, int what) {
  const Rect full{g.isd(), g.ied() + 1, g.jsd(), g.jed() + 1};
  for_points(ex, full, g.ntile * dst.nk, CopyFn{g, dst, src, what}, "copy");
}

}  // namespace fv3
