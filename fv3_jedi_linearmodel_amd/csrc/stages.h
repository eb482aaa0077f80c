// fv3lm-hip: stage functors of the hydrostatic acoustic step (c_sw, p_grad_c, d_sw, fv_tp_2d,
// divergence/vorticity damping, a2b_ord4, one_grad_p).  Each stage restates the nonlinear formula of
// the cited reference lines on a generic scalar T; tangent and adjoint come from exec.h.
// Interior-rank path: tiles that touch no cube-face edge (edge/corner branches: csrc/edges.h, next).
#pragma once
#include "exec.h"

namespace fv3 {

#define STAGE_BASE(NAME, NI, NO)                                     \
  static constexpr int NIN = NI, NOUT = NO;                          \
  Fld in[NI];                                                        \
  Fld out[NO];                                                       \
  Rect orect[NO];                                                    \
  int k0 = 1, k1 = 1;                                                \
  static const char* name() { return NAME; }
#define STAGE_COMMON(NAME, NI, NO) STAGE_BASE(NAME, NI, NO) STAGE_DEFAULTS_ON
// Adjoint refinements a stage may override: uses(M, di, dj, dk) = does any output read input M at that
// offset (exact stencil inside the box); wants(M) = bit mask of the outputs that depend on input M.
#define STAGE_DEFAULTS_ON                                             \
  HD static constexpr bool uses(int, int, int, int) { return true; }           \
  HD static constexpr unsigned wants(int) { return ~0u; }
#define MET(nm, i, j) c.m.nm[c.mi(tile, (i), (j))]
#define SSG(n, i, j) c.m.sin_sg[n][c.mi(tile, (i), (j))]
#define CSG(n, i, j) c.m.cos_sg[n][c.mi(tile, (i), (j))]

constexpr double A1 = 0.5625, A2 = -0.0625;            // sw_core_tlm.F90:56-57
constexpr double P1 = 7. / 12., P2 = -1. / 12.;        // tp_core_tlm.F90 p1,p2
constexpr double B1 = 7. / 12., B2 = -1. / 12.;        // a2b_edge_tlm.F90 b1,b2

// 1-D PPM flux at the interface between cells -1 and 0 of the line q(d), d = -3..2, Courant number
// cc (xppm/yppm iord in {1,2,333}: tp_core_tlm.F90:2397-2487).
template <class T, class Q>
HD T ppm_flux(int iord, const Q& q, T cc) {
  if (iord == 1) return (val(cc) > 0.) ? q(-1) : q(0);
  if (iord == 2) {
    T al0 = P1 * (q(-1) + q(0)) + P2 * (q(-2) + q(1));
    if (val(cc) > 0.) {
      T alm = P1 * (q(-2) + q(-1)) + P2 * (q(-3) + q(0));
      T qt = q(-1);
      return qt + (1. - cc) * (al0 - qt - cc * (alm + al0 - (qt + qt)));
    } else {
      T alp = P1 * (q(0) + q(1)) + P2 * (q(-1) + q(2));
      T qt = q(0);
      return qt + (1. + cc) * (al0 - qt + cc * (al0 + alp - (qt + qt)));
    }
  }
  // iord == 333
  if (val(cc) > 0.)
    return (2.0 * q(0) + 5.0 * q(-1) - q(-2)) / 6.0 - 0.5 * cc * (q(0) - q(-1)) + cc * cc / 6.0 * (q(0) - 2.0 * q(-1) + q(-2));
  return (2.0 * q(-1) + 5.0 * q(0) - q(1)) / 6.0 - 0.5 * cc * (q(0) - q(-1)) + cc * cc / 6.0 * (q(1) - 2.0 * q(0) + q(-1));
}

// xtp_u / ytp_v flux (sw_core_tlm.F90:7272-7486, :7490-7759): same stencils, cfl = c * rd(upwind cell).
template <class T, class Q>
HD T tp_uv_flux(int iord, const Q& q, T cc, double rd_m, double rd_0) {
  if (iord == 1) return (val(cc) > 0.) ? q(-1) : q(0);
  if (iord == 333) {
    if (val(cc) > 0.)
      return (2.0 * q(0) + 5.0 * q(-1) - q(-2)) / 6.0 - 0.5 * cc * rd_m * (q(0) - q(-1)) +
             cc * rd_m * cc * rd_m / 6.0 * (q(0) - 2.0 * q(-1) + q(-2));
    return (2.0 * q(-1) + 5.0 * q(0) - q(1)) / 6.0 - 0.5 * cc * rd_0 * (q(0) - q(-1)) +
           cc * rd_0 * cc * rd_0 / 6.0 * (q(1) - 2.0 * q(0) + q(-1));
  }
  T al0 = P1 * (q(-1) + q(0)) + P2 * (q(-2) + q(1));
  if (val(cc) > 0.) {
    T alm = P1 * (q(-2) + q(-1)) + P2 * (q(-3) + q(0));
    T bl = alm - q(-1), br = al0 - q(-1), b0 = bl + br;
    T cfl = cc * rd_m;
    return q(-1) + (1. - cfl) * (br - cfl * b0);
  }
  T alp = P1 * (q(0) + q(1)) + P2 * (q(-1) + q(2));
  T bl = al0 - q(0), br = alp - q(0), b0 = bl + br;
  T cfl = cc * rd_0;
  return q(0) + (1. + cfl) * (bl + cfl * b0);
}

// ===================================================================== c_sw
// d2a2c_vect A: D-grid winds -> A-grid (sw_core_tlm.F90:6505-6544, :6605-6611)
struct CswInterpA {
  STAGE_COMMON("CswInterpA", 2, 4)   // in: u v   out: utmp vtmp ua va
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -1, 2, 0, 0} : Box{-1, 2, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T ut = T(0.), vt = T(0.);
    const bool hu = orect[0].has(i, j), hv = orect[1].has(i, j);
    if (hu) ut = A2 * (a.template in<0>(i, j - 1) + a.template in<0>(i, j + 2)) + A1 * (a.template in<0>(i, j) + a.template in<0>(i, j + 1));
    if (hv) vt = A2 * (a.template in<1>(i - 1, j) + a.template in<1>(i + 2, j)) + A1 * (a.template in<1>(i, j) + a.template in<1>(i + 1, j));
    o[0] = ut; o[1] = vt;
    const double cs = MET(cosa_s, i, j), r2 = MET(rsin2, i, j);
    o[2] = (ut - vt * cs) * r2;
    o[3] = (vt - ut * cs) * r2;
  }
};

// d2a2c_vect C: A-grid -> C-grid + contravariant flux-form winds (:6655-6661, :6786-6792, :713-733)
struct CswInterpC {
  STAGE_BASE("CswInterpC", 4, 4)   // in: utmp vtmp u v   out: uc0 utf vc0 vtf
  double dt2;
  HD static constexpr bool uses(int, int, int, int) { return true; }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x3u : M == 1 ? 0xCu : M == 2 ? 0x8u : 0x2u; }
  HD static constexpr Box box(int M) { return M == 0 ? Box{-2, 1, 0, 0, 0, 0} : M == 1 ? Box{0, 0, -2, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = o[2] = o[3] = T(0.);
    if ((a.want & 0x3u) && orect[0].has(i, j)) {
      T uc = A2 * (a.template in<0>(i - 2, j) + a.template in<0>(i + 1, j)) + A1 * (a.template in<0>(i - 1, j) + a.template in<0>(i, j));
      T ut = (uc - a.template in<3>(i, j) * MET(cosa_u, i, j)) * MET(rsin_u, i, j);
      o[0] = uc;
      o[1] = (val(ut) > 0.) ? dt2 * ut * MET(dy, i, j) * SSG(3, i - 1, j) : dt2 * ut * MET(dy, i, j) * SSG(1, i, j);
    }
    if ((a.want & 0xCu) && orect[2].has(i, j)) {
      T vc = A2 * (a.template in<1>(i, j - 2) + a.template in<1>(i, j + 1)) + A1 * (a.template in<1>(i, j - 1) + a.template in<1>(i, j));
      T vt = (vc - a.template in<2>(i, j) * MET(cosa_v, i, j)) * MET(rsin_v, i, j);
      o[2] = vc;
      o[3] = (val(vt) > 0.) ? dt2 * vt * MET(dx, i, j) * SSG(4, i, j - 1) : dt2 * vt * MET(dx, i, j) * SSG(2, i, j);
    }
  }
};

// divergence_corner (sw_core_tlm.F90:4044-4081)
struct CswDivg {
  STAGE_COMMON("CswDivg", 4, 1)   // in: u v ua va   out: divgd
  HD static constexpr Box box(int M) { return M == 0 ? Box{-1, 0, 0, 0, 0, 0} : M == 1 ? Box{0, 0, -1, 0, 0, 0} : Box{-1, 0, -1, 0, 0, 0}; }
  template <class T, class A>
  HD T uf(const A& a, const Ctx& c, int tile, int i, int j) const {
    return (a.template in<0>(i, j) - 0.25 * (a.template in<3>(i, j - 1) + a.template in<3>(i, j)) * (CSG(4, i, j - 1) + CSG(2, i, j))) *
           MET(dyc, i, j) * 0.5 * (SSG(4, i, j - 1) + SSG(2, i, j));
  }
  template <class T, class A>
  HD T vf(const A& a, const Ctx& c, int tile, int i, int j) const {
    return (a.template in<1>(i, j) - 0.25 * (a.template in<2>(i - 1, j) + a.template in<2>(i, j)) * (CSG(3, i - 1, j) + CSG(1, i, j))) *
           MET(dxc, i, j) * 0.5 * (SSG(3, i - 1, j) + SSG(1, i, j));
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T d = vf<T>(a, c, tile, i, j - 1) - vf<T>(a, c, tile, i, j) + (uf<T>(a, c, tile, i - 1, j) - uf<T>(a, c, tile, i, j));
    o[0] = MET(rarea_c, i, j) * d;
  }
};

// first-order upwind transport of delp, pt on the C grid (sw_core_tlm.F90:744-808)
struct CswTransport {
  STAGE_BASE("CswTransport", 4, 2)   // in: delp pt utf vtf   out: delpc ptc
  HD static constexpr bool uses(int M, int di, int dj, int) { return M >= 2 || di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int M) { return M == 1 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 1, -1, 1, 0, 0} : M == 2 ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T fx1[2], fx[2], fy1[2], fy[2];
    for (int d = 0; d < 2; ++d) {
      T ut = a.template in<2>(i + d, j);
      const int iu = (val(ut) > 0.) ? i + d - 1 : i + d;
      fx1[d] = ut * a.template in<0>(iu, j);
      fx[d] = fx1[d] * a.template in<1>(iu, j);
      T vt = a.template in<3>(i, j + d);
      const int ju = (val(vt) > 0.) ? j + d - 1 : j + d;
      fy1[d] = vt * a.template in<0>(i, ju);
      fy[d] = fy1[d] * a.template in<1>(i, ju);
    }
    const double ra = MET(rarea, i, j);
    T dp = a.template in<0>(i, j), p = a.template in<1>(i, j);
    T dpc = dp + (fx1[0] - fx1[1] + (fy1[0] - fy1[1])) * ra;
    o[0] = dpc;
    o[1] = (p * dp + (fx[0] - fx[1] + (fy[0] - fy[1])) * ra) / dpc;
  }
};

// kinetic energy at cell centres and absolute vorticity at corners (sw_core_tlm.F90:870-957)
struct CswKeVort {
  STAGE_BASE("CswKeVort", 4, 2)   // in: ua va uc0 vc0   out: ke vort
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 2 || (M == 2 ? !(di == 1 && dj == -1) : !(di == -1 && dj == 1)); }
  HD static constexpr unsigned wants(int M) { return M < 2 ? 0x1u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : M == 2 ? Box{0, 1, -1, 0, 0, 0} : Box{-1, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T ua = a.template in<0>(i, j), va = a.template in<1>(i, j);
      T ku = (val(ua) > 0.) ? a.template in<2>(i, j) : a.template in<2>(i + 1, j);
      T kv = (val(va) > 0.) ? a.template in<3>(i, j) : a.template in<3>(i, j + 1);
      o[0] = (0.5 * dt2) * (ua * ku + va * kv);
    }
    if (orect[1].has(i, j)) {
      T v = a.template in<2>(i, j - 1) * MET(dxc, i, j - 1) - a.template in<2>(i, j) * MET(dxc, i, j) +
            (a.template in<3>(i, j) * MET(dyc, i, j) - a.template in<3>(i - 1, j) * MET(dyc, i - 1, j));
      o[1] = MET(fC, i, j) + MET(rarea_c, i, j) * v;
    }
  }
};

// time-centred C-grid winds (sw_core_tlm.F90:991-1037)
struct CswUpdate {
  STAGE_BASE("CswUpdate", 6, 2)   // in: uc0 vc0 u v vort ke   out: uc1 vc1
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 4 || (M == 4 ? !(di == 1 && dj == 1) : !(di == -1 && dj == -1)); }
  HD static constexpr unsigned wants(int M) { return (M == 0 || M == 3) ? 0x1u : (M == 1 || M == 2) ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 4 ? Box{0, 0, 0, 0, 0, 0} : M == 4 ? Box{0, 1, 0, 1, 0, 0} : Box{-1, 0, -1, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T uc = a.template in<0>(i, j);
      T fy1 = dt2 * (a.template in<3>(i, j) - uc * MET(cosa_u, i, j)) / MET(sina_u, i, j);
      T fy = (val(fy1) > 0.) ? a.template in<4>(i, j) : a.template in<4>(i, j + 1);
      o[0] = uc + fy1 * fy + MET(rdxc, i, j) * (a.template in<5>(i - 1, j) - a.template in<5>(i, j));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T vc = a.template in<1>(i, j);
      T fx1 = dt2 * (a.template in<2>(i, j) - vc * MET(cosa_v, i, j)) / MET(sina_v, i, j);
      T fx = (val(fx1) > 0.) ? a.template in<4>(i, j) : a.template in<4>(i + 1, j);
      o[1] = vc - fx1 * fx + MET(rdyc, i, j) * (a.template in<5>(i, j - 1) - a.template in<5>(i, j));
    }
  }
};

// p_grad_c, hydrostatic (dyn_core_tlm.F90:3310-3334)
struct PGradC {
  STAGE_BASE("PGradC", 4, 2)   // in: pkc gz (npz+1) uc1 vc1   out: uc2 vc2
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M >= 2 || !(di == -1 && dj == -1); }
  HD static constexpr unsigned wants(int M) { return M == 2 ? 0x1u : M == 3 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 0, -1, 0, 0, 1} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    T pk00 = a.template in<0>(i, j, 0), pk01 = a.template in<0>(i, j, 1);
    T gz00 = a.template in<1>(i, j, 0), gz01 = a.template in<1>(i, j, 1);
    T wk0 = pk01 - pk00;
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T pkm0 = a.template in<0>(i - 1, j, 0), pkm1 = a.template in<0>(i - 1, j, 1);
      T gzm0 = a.template in<1>(i - 1, j, 0), gzm1 = a.template in<1>(i - 1, j, 1);
      o[0] = a.template in<2>(i, j) + dt2 * MET(rdxc, i, j) / ((pkm1 - pkm0) + wk0) *
             ((gzm1 - gz00) * (pk01 - pkm0) + (gzm0 - gz01) * (pkm1 - pk00));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T pkm0 = a.template in<0>(i, j - 1, 0), pkm1 = a.template in<0>(i, j - 1, 1);
      T gzm0 = a.template in<1>(i, j - 1, 0), gzm1 = a.template in<1>(i, j - 1, 1);
      o[1] = a.template in<3>(i, j) + dt2 * MET(rdyc, i, j) / ((pkm1 - pkm0) + wk0) *
             ((gzm1 - gz00) * (pk01 - pkm0) + (gzm0 - gz01) * (pkm1 - pk00));
    }
  }
};

// ===================================================================== d_sw
// contravariant winds, Courant numbers, area fluxes (sw_core_tlm.F90:2722-2738, :2932-2968)
struct DswWinds {
  STAGE_COMMON("DswWinds", 2, 6)   // in: uc vc   out: ut crx xfx vt cry yfx
  double dt;
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, -1, 0, 0, 0} : Box{-1, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    for (int n = 0; n < 6; ++n) o[n] = T(0.);
    if (orect[0].has(i, j)) {
      T ut = (a.template in<0>(i, j) - 0.25 * MET(cosa_u, i, j) * (a.template in<1>(i - 1, j) + a.template in<1>(i, j) +
              a.template in<1>(i - 1, j + 1) + a.template in<1>(i, j + 1))) * MET(rsin_u, i, j);
      o[0] = ut;
      if (orect[1].has(i, j)) {
        T x = dt * ut;
        if (val(x) > 0.) { o[1] = x * MET(rdxa, i - 1, j); o[2] = MET(dy, i, j) * x * SSG(3, i - 1, j); }
        else             { o[1] = x * MET(rdxa, i, j);     o[2] = MET(dy, i, j) * x * SSG(1, i, j); }
      }
    }
    if (orect[3].has(i, j)) {
      T vt = (a.template in<1>(i, j) - 0.25 * MET(cosa_v, i, j) * (a.template in<0>(i, j - 1) + a.template in<0>(i + 1, j - 1) +
              a.template in<0>(i, j) + a.template in<0>(i + 1, j))) * MET(rsin_v, i, j);
      o[3] = vt;
      if (orect[4].has(i, j)) {
        T y = dt * vt;
        if (val(y) > 0.) { o[4] = y * MET(rdya, i, j - 1); o[5] = MET(dx, i, j) * y * SSG(4, i, j - 1); }
        else             { o[4] = y * MET(rdya, i, j);     o[5] = MET(dx, i, j) * y * SSG(2, i, j); }
      }
    }
  }
};

struct DswRa {   // sw_core_tlm.F90:2969-2978
  STAGE_COMMON("DswRa", 2, 2)   // in: xfx yfx   out: ra_x ra_y
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    const double ar = MET(area, i, j);
    if (orect[0].has(i, j)) o[0] = ar + (a.template in<0>(i, j) - a.template in<0>(i + 1, j));
    if (orect[1].has(i, j)) o[1] = ar + (a.template in<1>(i, j) - a.template in<1>(i, j + 1));
  }
};

// ---- fv_tp_2d building blocks (tp_core_tlm.F90:83-236) ----
enum HordSel { HORD_MT = 0, HORD_VT, HORD_TM, HORD_DP, HORD_TR };
HD int hord_of(const LevelParams& l, int sel) {
  return sel == HORD_MT ? l.hord_mt : sel == HORD_VT ? l.hord_vt : sel == HORD_TM ? l.hord_tm : sel == HORD_DP ? l.hord_dp : l.hord_tr;
}
template <class A, int M>
struct LineX { const A& a; int i, j; HD auto operator()(int d) const { return a.template in<M>(i + d, j); } };
template <class A, int M>
struct LineY { const A& a; int i, j; HD auto operator()(int d) const { return a.template in<M>(i, j + d); } };

struct TpPpmX {
  STAGE_COMMON("TpPpmX", 2, 1)   // in: q crx   out: flux
  int hsel;
  HD static constexpr Box box(int M) { return M == 0 ? Box{-3, 2, 0, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    LineX<A, 0> q{a, i, j};
    o[0] = ppm_flux<T>(hord_of(c.lev[k - 1], hsel), q, a.template in<1>(i, j));
  }
};
struct TpPpmY {
  STAGE_COMMON("TpPpmY", 2, 1)   // in: q cry   out: flux
  int hsel;
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -3, 2, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    LineY<A, 0> q{a, i, j};
    o[0] = ppm_flux<T>(hord_of(c.lev[k - 1], hsel), q, a.template in<1>(i, j));
  }
};
struct TpQi {   // tp_core_tlm.F90:149-159
  STAGE_COMMON("TpQi", 4, 1)   // in: q fy2 yfx ra_y   out: q_i
  HD static constexpr Box box(int M) { return (M == 1 || M == 2) ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T f0 = a.template in<2>(i, j) * a.template in<1>(i, j), f1 = a.template in<2>(i, j + 1) * a.template in<1>(i, j + 1);
    o[0] = (a.template in<0>(i, j) * MET(area, i, j) + f0 - f1) / a.template in<3>(i, j);
  }
};
struct TpQj {   // tp_core_tlm.F90:173-181
  STAGE_COMMON("TpQj", 4, 1)   // in: q fx2 xfx ra_x   out: q_j
  HD static constexpr Box box(int M) { return (M == 1 || M == 2) ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T f0 = a.template in<2>(i, j) * a.template in<1>(i, j), f1 = a.template in<2>(i + 1, j) * a.template in<1>(i + 1, j);
    o[0] = (a.template in<0>(i, j) * MET(area, i, j) + f0 - f1) / a.template in<3>(i, j);
  }
};
// deln_flux second-order pass for nord=1: d2 after one Laplacian (tp_core_tlm.F90:1984-2008)
enum DampSel { DAMP_NONE = 0, DAMP_V = 1, DAMP_T = 2 };
HD void damp_of(const LevelParams& l, int sel, int& nord, double& damp_c) {
  if (sel == DAMP_V) { nord = l.nord_v; damp_c = l.damp_vt; }
  else if (sel == DAMP_T) { nord = l.nord_t; damp_c = l.damp_t; }
  else { nord = -1; damp_c = 0.; }
}
struct TpD2 {
  STAGE_BASE("TpD2", 1, 1)
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
    // in: q   out: d2b  (is-1..ie+1, js-1..je+1); zero where the level does not use nord=1
  int dsel; int use_mass;
  HD static constexpr Box box(int) { return Box{-1, 1, -1, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    int nord; double dc; damp_of(c.lev[k - 1], dsel, nord, dc);
    o[0] = T(0.);
    if (nord != 1 || !(dc > 1.e-4)) return;
    const double damp = use_mass ? 1.0 : (dc * c.m.da_min) * (dc * c.m.da_min);
    T q0 = a.template in<0>(i, j);
    T fxa = MET(del6_v, i, j) * (a.template in<0>(i - 1, j) - q0);
    T fxb = MET(del6_v, i + 1, j) * (q0 - a.template in<0>(i + 1, j));
    T fya = MET(del6_u, i, j) * (a.template in<0>(i, j - 1) - q0);
    T fyb = MET(del6_u, i, j + 1) * (q0 - a.template in<0>(i, j + 1));
    o[0] = damp * ((fxa - fxb + fya - fyb) * MET(rarea, i, j));
  }
};
// flux averaging + damping fluxes (tp_core_tlm.F90:187-234, deln_flux :1918-2043)
struct TpFlux {
  STAGE_BASE("TpFlux", 9, 2)   // in: fx_o fx2 mx fy_o fy2 my q d2b mass   out: fx fy
  int dsel; int use_mass;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 6 || !(di == -1 && dj == -1); }
  HD static constexpr unsigned wants(int M) { return M < 3 ? 0x1u : M < 6 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) {
    return (M == 6 || M == 7 || M == 8) ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0};
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    int nord; double dc; damp_of(c.lev[k - 1], dsel, nord, dc);
    const bool dmp = (dsel != DAMP_NONE) && (dc > 1.e-4);
    double damp = 0.;
    if (dmp) { damp = dc * c.m.da_min; if (nord == 1) damp = damp * damp; }
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T f = 0.5 * (a.template in<0>(i, j) + a.template in<1>(i, j)) * a.template in<2>(i, j);
      if (dmp) {
        T f2;
        if (nord == 0) {
          f2 = MET(del6_v, i, j) * (a.template in<6>(i - 1, j) - a.template in<6>(i, j));
          if (!use_mass) f2 = damp * f2;
        } else {
          f2 = MET(del6_v, i, j) * (a.template in<7>(i, j) - a.template in<7>(i - 1, j));
        }
        if (use_mass) f = f + (0.5 * damp) * (a.template in<8>(i - 1, j) + a.template in<8>(i, j)) * f2;
        else f = f + f2;
      }
      o[0] = f;
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T f = 0.5 * (a.template in<3>(i, j) + a.template in<4>(i, j)) * a.template in<5>(i, j);
      if (dmp) {
        T f2;
        if (nord == 0) {
          f2 = MET(del6_u, i, j) * (a.template in<6>(i, j - 1) - a.template in<6>(i, j));
          if (!use_mass) f2 = damp * f2;
        } else {
          f2 = MET(del6_u, i, j) * (a.template in<7>(i, j) - a.template in<7>(i, j - 1));
        }
        if (use_mass) f = f + (0.5 * damp) * (a.template in<8>(i, j - 1) + a.template in<8>(i, j)) * f2;
        else f = f + f2;
      }
      o[1] = f;
    }
  }
};

// forward-in-time update of delp and pt (sw_core_tlm.F90:3107-3116)
struct DswUpdateDp {
  STAGE_COMMON("DswUpdateDp", 6, 2)   // in: delp pt fx fy gx gy   out: delp_n pt_n
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : (M == 2 || M == 4) ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const double ra = MET(rarea, i, j);
    T dp = a.template in<0>(i, j);
    T p = a.template in<1>(i, j) * dp + (a.template in<4>(i, j) - a.template in<4>(i + 1, j) + (a.template in<5>(i, j) - a.template in<5>(i, j + 1))) * ra;
    T dpn = dp + (a.template in<2>(i, j) - a.template in<2>(i + 1, j) + (a.template in<3>(i, j) - a.template in<3>(i, j + 1))) * ra;
    o[0] = dpn;
    o[1] = p / dpn;
  }
};

// B-grid advective winds for the KE fluxes (sw_core_tlm.F90:3164-3170, :3218-3224)
struct DswKeWinds {
  STAGE_COMMON("DswKeWinds", 2, 2)   // in: uc vc   out: vb ub
  double dt;
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -1, 0, 0, 0} : Box{-1, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const double dt5 = 0.5 * dt, cs = MET(cosa, i, j), rs = MET(rsina, i, j);
    T su = a.template in<0>(i, j - 1) + a.template in<0>(i, j), sv = a.template in<1>(i - 1, j) + a.template in<1>(i, j);
    o[0] = dt5 * (sv - su * cs) * rs;
    o[1] = dt5 * (su - sv * cs) * rs;
  }
};
// KE = 0.5*(vb*ytp_v + ub*xtp_u) (sw_core_tlm.F90:3197-3254)
struct DswKe {
  STAGE_COMMON("DswKe", 4, 1)   // in: vb ub u v   out: ke
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : M == 2 ? Box{-3, 2, 0, 0, 0, 0} : Box{0, 0, -3, 2, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const int iord = c.lev[k - 1].hord_mt;
    T vb = a.template in<0>(i, j), ub = a.template in<1>(i, j);
    LineY<A, 3> qv{a, i, j};
    T fv = tp_uv_flux<T>(iord, qv, vb, MET(rdy, i, j - 1), MET(rdy, i, j));
    LineX<A, 2> qu{a, i, j};
    T fu = tp_uv_flux<T>(iord, qu, ub, MET(rdx, i - 1, j), MET(rdx, i, j));
    o[0] = 0.5 * (vb * fv + ub * fu);
  }
};
// relative and absolute vorticity (sw_core_tlm.F90:3275-3293, :3535-3540)
struct DswVort {
  STAGE_COMMON("DswVort", 2, 2)   // in: u v   out: wk vort_abs
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 1, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T w = MET(rarea, i, j) * (a.template in<0>(i, j) * MET(dx, i, j) - a.template in<0>(i, j + 1) * MET(dx, i, j + 1) +
                              (a.template in<1>(i + 1, j) * MET(dy, i + 1, j) - a.template in<1>(i, j) * MET(dy, i, j)));
    o[0] = w;
    o[1] = w + MET(f0, i, j);
  }
};

// ---- divergence damping (compute_divergence_damping, sw_core_tlm.F90:7760-8072), nord in {0,1} ----
struct DdA {
  STAGE_BASE("DdA", 5, 2)   // in: divgd u v ua va   out: da db
  HD static constexpr bool uses(int M, int di, int dj, int) { return M == 0 ? !(di == 1 && dj == 1) : M == 3 ? dj == 0 : M == 4 ? di == 0 : true; }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x3u : (M == 1 || M == 4) ? 0x1u : 0x2u; }
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, 0, 1, 0, 0} : (M == 3 || M == 4) ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const int nord = c.lev[k - 1].nord;
    o[0] = o[1] = T(0.);
    if (nord == 0) {   // :7874-7886 (interior)
      if (orect[0].has(i, j))
        o[0] = (a.template in<1>(i, j) - 0.5 * (a.template in<4>(i, j - 1) + a.template in<4>(i, j)) * MET(cosa_v, i, j)) * MET(dyc, i, j) * MET(sina_v, i, j);
      if (orect[1].has(i, j))
        o[1] = (a.template in<2>(i, j) - 0.5 * (a.template in<3>(i - 1, j) + a.template in<3>(i, j)) * MET(cosa_u, i, j)) * MET(dxc, i, j) * MET(sina_u, i, j);
    } else {           // :7976-7987, nt = 0
      if (orect[0].has(i, j)) o[0] = (a.template in<0>(i + 1, j) - a.template in<0>(i, j)) * MET(divg_u, i, j);
      if (orect[1].has(i, j)) o[1] = (a.template in<0>(i, j + 1) - a.template in<0>(i, j)) * MET(divg_v, i, j);
    }
  }
};
struct DdB {   // :7924-7937 (nord=0: delpc) / :7990-8006 (nord>0: new divg_d)
  STAGE_COMMON("DdB", 2, 1)   // in: da db   out: dc
  HD static constexpr Box box(int M) { return M == 0 ? Box{-1, 0, 0, 0, 0, 0} : Box{0, 0, -1, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = MET(rarea_c, i, j) * (a.template in<1>(i, j - 1) - a.template in<1>(i, j) + a.template in<0>(i - 1, j) - a.template in<0>(i, j));
  }
};
// a2b_ord4 interior (a2b_edge_tlm.F90:163-176, :268-291, :365-420, :441-505)
struct A2bA {
  STAGE_BASE("A2bA", 1, 2)   // in: q   out: qx qy
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x3u; }
  HD static constexpr Box box(int) { return Box{-2, 1, -2, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if (orect[0].has(i, j)) o[0] = B2 * (a.template in<0>(i - 2, j) + a.template in<0>(i + 1, j)) + B1 * (a.template in<0>(i - 1, j) + a.template in<0>(i, j));
    if (orect[1].has(i, j)) o[1] = B2 * (a.template in<0>(i, j - 2) + a.template in<0>(i, j + 1)) + B1 * (a.template in<0>(i, j - 1) + a.template in<0>(i, j));
  }
};
struct A2bB {
  STAGE_COMMON("A2bB", 2, 1)   // in: qx qy   out: qout
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -2, 1, 0, 0} : Box{-2, 1, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T qxx = A2 * (a.template in<0>(i, j - 2) + a.template in<0>(i, j + 1)) + A1 * (a.template in<0>(i, j - 1) + a.template in<0>(i, j));
    T qyy = A2 * (a.template in<1>(i - 2, j) + a.template in<1>(i + 1, j)) + A1 * (a.template in<1>(i - 1, j) + a.template in<1>(i, j));
    o[0] = 0.5 * (qxx + qyy);
  }
};
struct DdC {   // Smagorinsky-type coefficient and damping term added to KE (:7938-7956, :8023-8070)
  STAGE_COMMON("DdC", 4, 1)   // in: ke dc divgd vort_b   out: ke2
  double dt, dddmp, d4_bg;
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const LevelParams& l = c.lev[k - 1];
    const double absdt = dt >= 0. ? dt : -dt;
    T ke = a.template in<0>(i, j), dcv = a.template in<1>(i, j);
    if (l.nord == 0) {
      T x = dcv * dt;
      T abs2 = (val(x) >= 0.) ? x : -x;
      T y3 = dddmp * abs2;
      T y1 = (0.20 > val(y3)) ? y3 : T(0.20);
      T mx = (l.d2_divg < val(y1)) ? y1 : T(l.d2_divg);
      o[0] = ke + (c.m.da_min_c * mx) * dcv;
    } else {
      T delpc = a.template in<2>(i, j);
      T vs = T(0.);
      if (!(dddmp < 1.e-5)) {
        T vb = a.template in<3>(i, j);
        vs = absdt * dsqrt(delpc * delpc + vb * vb);
      }
      T y2 = (0.20 > dddmp * val(vs)) ? dddmp * vs : T(0.20);
      T mx = (l.d2_divg < val(y2)) ? y2 : T(l.d2_divg);
      const double pw = c.m.da_min_c * d4_bg;
      double dd8 = pw;
      for (int n = 0; n < l.nord; ++n) dd8 *= pw;
      o[0] = ke + ((c.m.da_min_c * mx) * delpc + dd8 * dcv);
    }
  }
};
// del6_vt_flux inner Laplacian without the damp factor (sw_core_tlm.F90:3747-3776, nord_v = 1)
struct Del6A {
  STAGE_BASE("Del6A", 1, 1)   // in: wk   out: d2b
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  HD static constexpr Box box(int) { return Box{-1, 1, -1, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T q0 = a.template in<0>(i, j);
    T fxa = MET(del6_v, i, j) * (a.template in<0>(i - 1, j) - q0);
    T fxb = MET(del6_v, i + 1, j) * (q0 - a.template in<0>(i + 1, j));
    T fya = MET(del6_u, i, j) * (a.template in<0>(i, j - 1) - q0);
    T fyb = MET(del6_u, i, j + 1) * (q0 - a.template in<0>(i, j + 1));
    o[0] = (fxa - fxb + (fya - fyb)) * MET(rarea, i, j);
  }
};
// momentum update (sw_core_tlm.F90:3555-3564) + vorticity-damping fluxes, trajectory and
// perturbation coefficients kept apart (sw_core_tlm.F90:2436-2452, :2502-2530)
struct DswUpdateUV {
  STAGE_BASE("DswUpdateUV", 7, 2)   // in: u v ke2 fxv fyv wk d2b   out: u_n v_n
  HD static constexpr bool uses(int M, int di, int dj, int) { return M == 2 ? !(di == 1 && dj == 1) : (M == 5 || M == 6) ? !(di == -1 && dj == -1) : true; }
  HD static constexpr unsigned wants(int M) { return (M == 0 || M == 4) ? 0x1u : (M == 1 || M == 3) ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) {
    return M == 2 ? Box{0, 1, 0, 1, 0, 0} : (M == 5 || M == 6) ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0};
  }
  HD static double pw(double x, int n) { double r = x; for (int m = 0; m < n; ++m) r *= x; return r; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const LevelParams& l = c.lev[k - 1];
    const bool dt_ = l.damp_vt > 1.e-5, dp_ = l.damp_vt_pert > 1.e-5;
    const double d4t = dt_ ? pw(l.damp_vt * c.m.da_min_c, l.nord_v) : 0., d4p = dp_ ? pw(l.damp_vt_pert * c.m.da_min_c, l.nord_v_pert) : 0.;
    T ke = a.template in<2>(i, j);
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T e0 = T(0.), e1 = T(0.);
      if ((dt_ && l.nord_v == 0) || (dp_ && l.nord_v_pert == 0)) e0 = MET(del6_u, i, j) * (a.template in<5>(i, j - 1) - a.template in<5>(i, j));
      if ((dt_ && l.nord_v == 1) || (dp_ && l.nord_v_pert == 1)) e1 = MET(del6_u, i, j) * (a.template in<6>(i, j) - a.template in<6>(i, j - 1));
      T vt_t = d4t * (l.nord_v == 0 ? e0 : e1), vt_p = d4p * (l.nord_v_pert == 0 ? e0 : e1);
      o[0] = a.template in<0>(i, j) * MET(dx, i, j) + (ke - a.template in<2>(i + 1, j)) + a.template in<4>(i, j) + combine(vt_t, vt_p);
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T e0 = T(0.), e1 = T(0.);
      if ((dt_ && l.nord_v == 0) || (dp_ && l.nord_v_pert == 0)) e0 = MET(del6_v, i, j) * (a.template in<5>(i - 1, j) - a.template in<5>(i, j));
      if ((dt_ && l.nord_v == 1) || (dp_ && l.nord_v_pert == 1)) e1 = MET(del6_v, i, j) * (a.template in<6>(i, j) - a.template in<6>(i - 1, j));
      T ut_t = d4t * (l.nord_v == 0 ? e0 : e1), ut_p = d4p * (l.nord_v_pert == 0 ? e0 : e1);
      o[1] = a.template in<1>(i, j) * MET(dy, i, j) + (ke - a.template in<2>(i, j + 1)) - a.template in<3>(i, j) - combine(ut_t, ut_p);
    }
  }
};

// ===================================================================== tracer_2d (fv_tracer2d_tlm.F90:1148-1446)
// accumulated Courant numbers -> area fluxes (:1226-1247)
struct TrFlux {
  STAGE_COMMON("TrFlux", 2, 2)   // in: cx cy   out: xfx yfx
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if (orect[0].has(i, j)) {
      T cx = a.template in<0>(i, j);
      o[0] = (val(cx) > 0.) ? cx * MET(dxa, i - 1, j) * MET(dy, i, j) * SSG(3, i - 1, j) : cx * MET(dxa, i, j) * MET(dy, i, j) * SSG(1, i, j);
    }
    if (orect[1].has(i, j)) {
      T cy = a.template in<1>(i, j);
      o[1] = (val(cy) > 0.) ? cy * MET(dya, i, j - 1) * MET(dx, i, j) * SSG(4, i, j - 1) : cy * MET(dya, i, j) * MET(dx, i, j) * SSG(2, i, j);
    }
  }
};
// dp2 and the flux-form areas (:1375-1392)
struct TrDp2Ra {
  STAGE_COMMON("TrDp2Ra", 5, 3)   // in: dp1 mfx mfy xfx yfx   out: dp2 ra_x ra_y
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, 0, 0, 0, 0} : (M == 1 || M == 3) ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = o[2] = T(0.);
    const double ar = MET(area, i, j);
    if (orect[0].has(i, j))
      o[0] = a.template in<0>(i, j) + (a.template in<1>(i, j) - a.template in<1>(i + 1, j) + (a.template in<2>(i, j) - a.template in<2>(i, j + 1))) * MET(rarea, i, j);
    if (orect[1].has(i, j)) o[1] = ar + (a.template in<3>(i, j) - a.template in<3>(i + 1, j));
    if (orect[2].has(i, j)) o[2] = ar + (a.template in<4>(i, j) - a.template in<4>(i, j + 1));
  }
};
// q update (:1423-1430)
struct TrUpdate {
  STAGE_COMMON("TrUpdate", 5, 1)   // in: q dp1 dp2 fx fy   out: q_o
  HD static constexpr Box box(int M) { return M == 3 ? Box{0, 1, 0, 0, 0, 0} : M == 4 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = (a.template in<0>(i, j) * a.template in<1>(i, j) +
            (a.template in<3>(i, j) - a.template in<3>(i + 1, j) + (a.template in<4>(i, j) - a.template in<4>(i, j + 1))) * MET(rarea, i, j)) /
           a.template in<2>(i, j);
  }
};
// pt <-> virtual potential temperature at the ends of fv_dynamics (fv_dynamics_tlm.F90:1395-1403)
struct DynPtIn {
  STAGE_COMMON("DynPtIn", 3, 1)   // in: pt(T) qv pkz   out: pt(theta_v)
  double zvir; int has_q;
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T d = has_q ? zvir * a.template in<1>(i, j) : T(0.);
    o[0] = a.template in<0>(i, j) * (1. + d) / a.template in<2>(i, j);
  }
};

// one_grad_p wind update from corner pk, gz (dyn_core_tlm.F90:4128-4157); level 1 of pk is the
// constant top value (:4068-4072).
struct OneGradP {
  STAGE_BASE("OneGradP", 4, 2)   // in: u v pk_b gz_b (npz+1)   out: u_n v_n
  double dt, ptk;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 2 || !(di == 1 && dj == 1); }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x1u : M == 1 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : Box{0, 1, 0, 1, 0, 1}; }
  template <class T, class A>
  HD T pk(const A& a, int i, int j, int k, int dk) const { return (k + dk == 1) ? T(ptk) : a.template in<2>(i, j, dk); }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T p00 = pk<T>(a, i, j, k, 0), p01 = pk<T>(a, i, j, k, 1);
    T g00 = a.template in<3>(i, j, 0), g01 = a.template in<3>(i, j, 1);
    T wk0 = p01 - p00;
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T p10 = pk<T>(a, i + 1, j, k, 0), p11 = pk<T>(a, i + 1, j, k, 1);
      T g10 = a.template in<3>(i + 1, j, 0), g11 = a.template in<3>(i + 1, j, 1);
      o[0] = MET(rdx, i, j) * (a.template in<0>(i, j) + dt / (wk0 + (p11 - p10)) * ((g01 - g10) * (p11 - p00) + (g00 - g11) * (p01 - p10)));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T p10 = pk<T>(a, i, j + 1, k, 0), p11 = pk<T>(a, i, j + 1, k, 1);
      T g10 = a.template in<3>(i, j + 1, 0), g11 = a.template in<3>(i, j + 1, 1);
      o[1] = MET(rdy, i, j) * (a.template in<1>(i, j) + dt / (wk0 + (p11 - p10)) * ((g01 - g10) * (p11 - p00) + (g00 - g11) * (p01 - p10)));
    }
  }
};

}  // namespace fv3
