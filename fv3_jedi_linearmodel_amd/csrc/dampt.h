// fv3lm-hip: the TRAJECTORY side of split_damp (fv_arrays_tlmadm.F90:76, the reference's default) -- values only.
//
// With split_damp the reference evaluates the divergence damping twice (sw_core_tlm.F90:2350-2369): COMPUTE_DIVERGENCE_DAMPING_TLM
// with the perturbation's coefficients on copies, whose tangents are kept, then the nonlinear compute_divergence_damping with the
// trajectory's coefficients, whose values are kept.  The differentiated chain is the stage list of stages.h (DdA, DdB, DdC reading
// LevelParams::nord_p / d2_divg_p / dddmp_p / d4_bg_p); this file is the second call: plain double kernels that no tangent or adjoint is
// ever taken of, which is what allows the trajectory's nord to be 2 or 3 (sw_core_tlm.F90:8463-8530: nord passes, the valid region
// shrinking by one per pass, fill_corners before each difference).  Likewise the trajectory's vorticity damping with nord_v = 2
// (`call del6_vt_flux(nord_v, ...)`, sw_core_tlm.F90:2436-2441).
//
// fill_corners without mutation: the reference fills the corner squares of divg_d in place, X-direction before the x-difference,
// Y-direction before the y-difference, and then the corner squares of the difference pair (vc, uc) as a rotated D-grid vector
// (tools/fv_mp_nlm_mod.F90:1055-1084, :1271-1303).  Every source of those fills lies outside the corner squares, so each fill is an index
// map; one pass of the loop is then ONE kernel from the old divergence to the new one.
#pragma once
#include "stages.h"

namespace fv3 {

// B-grid (corner-point) scalar: where does corner-square point (i,j) come from (fill_corners_2d BGRID, XDir / YDir).  No-op elsewhere.
HD void bgrid_fill_map(const Geom& g, int dir, int& i, int& j) {
  const int npx = g.nx + 1, npy = g.ny + 1;
  const bool w = i < 1, e = i > npx, s = j < 1, n = j > npy;
  if (!((w || e) && (s || n))) return;
  const int a = i, b = j;
  if (dir == 1) {
    if (w && s) { i = b; j = 2 - a; }
    else if (w && n) { i = 1 - b + npy; j = npy - 1 + a; }
    else if (e && s) { i = npx + 1 - b; j = a - npx + 1; }
    else { i = npx + b - npy; j = npy + npx - a; }
  } else {
    if (w && s) { i = 2 - b; j = a; }
    else if (w && n) { i = b - npy + 1; j = npy + 1 - a; }
    else if (e && s) { i = npx - 1 + b; j = 1 - a + npx; }
    else { i = npx - (b - npy); j = npy + a - npx; }
  }
}

struct DampTArgs {
  Geom g; Metrics m; const LevelParams* lev;
  const double *u, *v, *ua, *va, *uc, *vc;      // winds (nord_k = 0: the divergence is rebuilt from them, sw_core_tlm.F90:8255-8400)
  const double *divgd, *vortb, *ke;              // divg_d of c_sw, a2b_ord4(wk), KE before damping
  double *s1, *s2;                               // ping-pong scratch for the iterated divergence
  double* ke2;                                   // out: KE + damping, values
  double dt;
};

// one pass n of the n-loop (sw_core_tlm.F90:8463-8530) for the levels with nord_k >= pass: src -> dst on is-nt .. ie+1+nt, nt = nord_k - pass
struct DampTPass {
  DampTArgs a; int pass; const double* src; double* dst;
  HD void operator()(int i, int j, int z) const {
    const Geom& g = a.g;
    const int tile = z / g.npz, k = 1 + z % g.npz;
    const int nord = a.lev[k - 1].nord, nt = nord - pass;
    if (nt < 0) return;
    if (i < g.is() - nt || i > g.ie() + 1 + nt || j < g.js() - nt || j > g.je() + 1 + nt) return;
    const bool fill = nt != 0 && g.face;
    const size_t base = (size_t)z * g.plane, mb = (size_t)tile * g.plane;
    const int npx = g.nx + 1, npy = g.ny + 1;
    auto dX = [&](int ii, int jj) { if (fill) bgrid_fill_map(g, 1, ii, jj); return src[base + g.idx(ii, jj)]; };
    auto dY = [&](int ii, int jj) { if (fill) bgrid_fill_map(g, 2, ii, jj); return src[base + g.idx(ii, jj)]; };
    auto vcf = [&](int p, int q) { return (dX(p + 1, q) - dX(p, q)) * a.m.divg_u[mb + g.idx(p, q)]; };
    auto ucf = [&](int p, int q) { return (dY(p, q + 1) - dY(p, q)) * a.m.divg_v[mb + g.idx(p, q)]; };
    // the pair as seen after fill_corners(vc, uc, DGRID, VECTOR): x = vc on (isd:ied, jsd:jed+1), y = uc on (isd:ied+1, jsd:jed)
    auto VC = [&](int p, int q) {
      if (fill) {
        const bool w = p < 1, e = p > npx - 1, s = q < 1, n = q > npy;
        if (w && s) return -ucf(q, 1 - p);
        if (w && n) return ucf(1 - q + npy, npy - 1 + p);
        if (e && s) return ucf(npx + 1 - q, p - npx + 1);
        if (e && n) return -ucf(npx + q - npy, npy + npx - 1 - p);
      }
      return vcf(p, q);
    };
    auto UC = [&](int p, int q) {
      if (fill) {
        const bool w = p < 1, e = p > npx, s = q < 1, n = q > npy - 1;
        if (w && s) return -vcf(1 - q, p);
        if (w && n) return vcf(q - npy + 1, npy + 1 - p);
        if (e && s) return vcf(npx - 1 + q, 1 - p + npx);
        if (e && n) return -vcf(npx + npy - 1 - q, npy + p - npx);
      }
      return ucf(p, q);
    };
    const bool cx = g.face && (i == 1 || i == npx);
    double d = VC(i - 1, j) - VC(i, j);
    if (!(cx && j == 1)) d += UC(i, j - 1);            // "remove the extra term at the corners" (:8494-8509)
    if (!(cx && j == npy)) d -= UC(i, j);
    dst[base + g.idx(i, j)] = d * a.m.rarea_c[mb + g.idx(i, j)];
  }
};

// the damping term with the trajectory's coefficients added to KE (:8531-8596; nord_k = 0: :8255-8461)
struct DampTFinal {
  DampTArgs a;
  HD void operator()(int i, int j, int z) const {
    const Geom& g = a.g;
    const int tile = z / g.npz, k = 1 + z % g.npz;
    const LevelParams& l = a.lev[k - 1];
    const size_t base = (size_t)z * g.plane, mb = (size_t)tile * g.plane, n0 = base + g.idx(i, j);
    const int npx = g.nx + 1, npy = g.ny + 1;
    const bool F = g.face != 0;
    const double absdt = a.dt >= 0. ? a.dt : -a.dt, da_min_c = a.m.da_min_c;
    auto M = [&](const double* p, int ii, int jj) { return p[mb + g.idx(ii, jj)]; };
    auto X = [&](const double* p, int ii, int jj) { return p[base + g.idx(ii, jj)]; };
    if (l.nord == 0) {
      auto ptc = [&](int ii, int jj) {
        if (F && (jj == 1 || jj == npy)) return (X(a.vc, ii, jj) > 0) ? X(a.u, ii, jj) * M(a.m.dyc, ii, jj) * M(a.m.sin_sg[4], ii, jj - 1) : X(a.u, ii, jj) * M(a.m.dyc, ii, jj) * M(a.m.sin_sg[2], ii, jj);
        return (X(a.u, ii, jj) - 0.5 * (X(a.va, ii, jj - 1) + X(a.va, ii, jj)) * M(a.m.cosa_v, ii, jj)) * M(a.m.dyc, ii, jj) * M(a.m.sina_v, ii, jj);
      };
      auto vrt = [&](int ii, int jj) {
        if (F && (ii == 1 || ii == npx)) return (X(a.uc, ii, jj) > 0) ? X(a.v, ii, jj) * M(a.m.dxc, ii, jj) * M(a.m.sin_sg[3], ii - 1, jj) : X(a.v, ii, jj) * M(a.m.dxc, ii, jj) * M(a.m.sin_sg[1], ii, jj);
        return (X(a.v, ii, jj) - 0.5 * (X(a.ua, ii - 1, jj) + X(a.ua, ii, jj)) * M(a.m.cosa_u, ii, jj)) * M(a.m.dxc, ii, jj) * M(a.m.sina_u, ii, jj);
      };
      const bool cx = F && (i == 1 || i == npx);
      double d = ptc(i - 1, j) - ptc(i, j);
      if (!(cx && j == 1)) d += vrt(i, j - 1);
      if (!(cx && j == npy)) d -= vrt(i, j);
      const double delpc = M(a.m.rarea_c, i, j) * d;
      const double x = delpc * a.dt, abs2 = x >= 0. ? x : -x, y3 = l.dddmp * abs2, y1 = (0.20 > y3) ? y3 : 0.20, mx = (l.d2_divg < y1) ? y1 : l.d2_divg;
      a.ke2[n0] = a.ke[n0] + (da_min_c * mx) * delpc;
    } else {
      const double delpc = a.divgd[n0];
      double vs = 0.;
      if (!(l.dddmp < 1.e-5)) { const double vb = a.vortb[n0]; vs = absdt * sqrt(delpc * delpc + vb * vb); }
      const double y2 = (0.20 > l.dddmp * vs) ? l.dddmp * vs : 0.20, mx = (l.d2_divg < y2) ? y2 : l.d2_divg;
      const double dd8 = damp_pow(da_min_c * l.d4_bg, l.nord);
      const double* dfin = (l.nord & 1) ? a.s1 : a.s2;
      a.ke2[n0] = a.ke[n0] + ((da_min_c * mx) * delpc + dd8 * dfin[n0]);
    }
  }
};

inline void run_damp_t(Exec& ex, const DampTArgs& a, int max_nord) {
  const Geom& g = a.g;
  const int nz = g.ntile * g.npz;
  for (int pass = 1; pass <= max_nord; ++pass) {
    const int nt = max_nord - pass;
    DampTPass p{a, pass, pass == 1 ? a.divgd : (pass & 1) ? a.s2 : a.s1, (pass & 1) ? a.s1 : a.s2};
    for_points(ex, Rect{g.is() - nt, g.ie() + 1 + nt, g.js() - nt, g.je() + 1 + nt}, nz, p, "DampT.pass", 0.);
  }
  DampTFinal f{a};
  for_points(ex, Rect{g.is(), g.ie() + 1, g.js(), g.je() + 1}, nz, f, "DampT.final", 0.);
}

// ---- del6_vt_flux of the trajectory with nord_v = 2 (sw_core_tlm.F90:3719-3801), values only: the two inner Laplacians, corner halo
// through the copy_corners views of the difference direction; the outer difference and the factor damp4 stay in DswUpdateUV
struct LapTArgs { Geom g; Metrics m; const LevelParams* lev; const double* src; double* dst; int pass; };
struct LapTPass {
  LapTArgs a;
  HD void operator()(int i, int j, int z) const {
    const Geom& g = a.g;
    const int tile = z / g.npz, k = 1 + z % g.npz;
    const LevelParams& l = a.lev[k - 1];
    if (!(l.nord_v == 2 && l.damp_vt > 1.e-5)) return;
    const size_t base = (size_t)z * g.plane, mb = (size_t)tile * g.plane;
    auto rx = [&](int ii, int jj) { corner_map(g, 1, ii, jj); return a.src[base + g.idx(ii, jj)]; };
    auto ry = [&](int ii, int jj) { corner_map(g, 2, ii, jj); return a.src[base + g.idx(ii, jj)]; };
    const double fxa = a.m.del6_v[mb + g.idx(i, j)] * (rx(i - 1, j) - rx(i, j)), fxb = a.m.del6_v[mb + g.idx(i + 1, j)] * (rx(i, j) - rx(i + 1, j));
    const double fya = a.m.del6_u[mb + g.idx(i, j)] * (ry(i, j - 1) - ry(i, j)), fyb = a.m.del6_u[mb + g.idx(i, j + 1)] * (ry(i, j) - ry(i, j + 1));
    const double lap = (fxa - fxb + (fya - fyb)) * a.m.rarea[mb + g.idx(i, j)];
    a.dst[base + g.idx(i, j)] = a.pass == 1 ? lap : -lap;       // second pass: the differences change sign (:3786-3799)
  }
};

}  // namespace fv3
