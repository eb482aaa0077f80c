// fv3lm-hip: stage functors of the hydrostatic acoustic step (c_sw, p_grad_c, d_sw, fv_tp_2d,
// divergence/vorticity damping, a2b_ord4, one_grad_p) and of tracer_2d.  Each stage restates the
// nonlinear formula of the cited reference lines on a generic scalar T; tangent and adjoint come
// from exec.h.  Two tile kinds (Geom::face): 0 = no cube edge in the tile (interior MPI rank of the
// reference; used with the doubly-periodic single tile), 1 = whole cube face with all the
// `is .EQ. 1` / `j .EQ. npy` / corner branches of the reference (helpers in edges.h).
#pragma once
#include <type_traits>
#include "exec.h"
#include "edges.h"

namespace fv3 {

#define STAGE_BASE(NAME, NI, NO)                                     \
  static constexpr int NIN = NI, NOUT = NO;                          \
  Fld in[NI];                                                        \
  Fld out[NO];                                                       \
  Rect orect[NO];                                                    \
  int k0 = 1, k1 = 1;                                                \
  unsigned wmask = 0;   /* adjoint: inputs whose adjoint this instance stores instead of accumulating (exec.h body_ad_joint) */ \
  Rect wrect{1, 0, 1, 0};   /* ... and the rectangle those stores must cover besides the stage's own reach (dycore.h plan_adjoint) */ \
  static const char* name() { return NAME; }                        \
  static const char* ename() { return NAME "e"; }
// Adjoint refinements a stage may override: uses(M, di, dj, dk) = does any output read input M at that
// offset (exact stencil inside the box); wants(M) = bit mask of the outputs that depend on input M
// (0: the input only steers branches).  Stages that read corner-halo points through an index map
// (edges.h) also declare the inverse map: alias(c, M, i, j, n, ai, aj) = n-th virtual location at which
// input point (i,j) of input M is read, as seen through the stencil box of input alias_box(M).
#define STAGE_DEFAULTS_ON                                                      \
  HD static constexpr bool uses(int, int, int, int) { return true; }          \
  HD static constexpr unsigned wants(int) { return ~0u; }
#define STAGE_NO_ALIAS                                                         \
  static constexpr int NALIAS = 0;                                             \
  HD static constexpr int alias_box(int M) { return M; }                       \
  HD bool alias(const Ctx&, int, int, int, int, int&, int&) const { return false; }
#define STAGE_COMMON(NAME, NI, NO) STAGE_BASE(NAME, NI, NO) STAGE_DEFAULTS_ON STAGE_NO_ALIAS
#define MET(nm, i, j) c.m.nm[c.mi(tile, (i), (j))]
#define SSG(n, i, j) c.m.sin_sg[n][c.mi(tile, (i), (j))]
#define CSG(n, i, j) c.m.cos_sg[n][c.mi(tile, (i), (j))]
#define IN(M, ...) a.template in<M>(__VA_ARGS__)

// A stage body written with `template <bool EDGE, ...> eval_e` becomes two stage types: Edged<D,false> for the bulk launch
// (no face edge within reach of its outputs: every edge branch and corner view compiled out) and Edged<D,true> for the
// strips next to the face edges (dycore.h add_face).
template <class D, class = void> struct d_lds_fw { static constexpr bool v = false; };
template <class D> struct d_lds_fw<D, typename std::enable_if<D::LDS_FW_OK>::type> { static constexpr bool v = true; };
template <class D, class = void> struct d_merge_ad { static constexpr bool v = false; };
template <class D> struct d_merge_ad<D, typename std::enable_if<D::MERGE_AD_OK>::type> { static constexpr bool v = true; };
template <class D, class = void> struct d_lds_ad { static constexpr bool v = false; };
template <class D> struct d_lds_ad<D, typename std::enable_if<D::LDS_AD_OK>::type> { static constexpr bool v = true; };
template <class D, bool EDGE>
struct Edged : D {
  // bulk launch: tiles + halos staged through LDS for the stages and modes where that measured faster (exec.h)
  static constexpr bool LDS_FW = !EDGE && d_lds_fw<D>::v, LDS_AD = !EDGE && d_lds_ad<D>::v;
  static constexpr bool MERGE_AD_STRIPS = EDGE && d_merge_ad<D>::v;     // adjoint of the four strips in one launch (exec.h run_multi)
  static constexpr int LDS_BY = 8;
  Edged() = default;
  Edged(const D& d) : D(d) {}
  static constexpr int NALIAS = EDGE ? D::NALIAS : 0;
  static const char* name() { return EDGE ? D::ename() : D::name(); }
  HD bool alias(const Ctx& c, int M, int i, int j, int n, int& ai, int& aj) const { return EDGE && D::alias(c, M, i, j, n, ai, aj); }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const { this->template eval_e<EDGE, T, A>(a, c, tile, i, j, k, o); }
};

constexpr double A1 = 0.5625, A2 = -0.0625;            // sw_core_tlm.F90:56-57
constexpr double P1 = 7. / 12., P2 = -1. / 12.;        // tp_core_tlm.F90 p1,p2
constexpr double B1 = 7. / 12., B2 = -1. / 12.;        // a2b_edge_tlm.F90 b1,b2

// 1-D PPM flux at interface m (between cells m-1 and m) of a line: q(k), da(k) by absolute cell index
// (xppm/yppm iord in {1,2,333}: tp_core_tlm.F90:2397-2487; one-sided edge values :2402-2429).
template <class T, class Q, class D>
HD T ppm_flux(int iord, bool face, int m, int n1, const Q& q, const D& da, T cc) {
  if (iord == 1) return (val(cc) > 0.) ? q(m - 1) : q(m);
  if (iord == 2) {
    T al0 = ppm_al<T>(face, m, n1, q, da);
    if (val(cc) > 0.) {
      T alm = ppm_al<T>(face, m - 1, n1, q, da);
      T qt = q(m - 1);
      return qt + (1. - cc) * (al0 - qt - cc * (alm + al0 - (qt + qt)));
    } else {
      T alp = ppm_al<T>(face, m + 1, n1, q, da);
      T qt = q(m);
      return qt + (1. + cc) * (al0 - qt + cc * (al0 + alp - (qt + qt)));
    }
  }
  // iord == 333
  if (val(cc) > 0.)
    return (2.0 * q(m) + 5.0 * q(m - 1) - q(m - 2)) / 6.0 - 0.5 * cc * (q(m) - q(m - 1)) + cc * cc / 6.0 * (q(m) - 2.0 * q(m - 1) + q(m - 2));
  return (2.0 * q(m - 1) + 5.0 * q(m) - q(m + 1)) / 6.0 - 0.5 * cc * (q(m) - q(m - 1)) + cc * cc / 6.0 * (q(m + 1) - 2.0 * q(m) + q(m - 1));
}

// d flux(m) / d q(k) of the same schemes in closed form (they are linear in q for a given Courant number): the transposed sweep of
// the hand-written outer adjoint (tpfused.h).  al(x) = sum_e w_e(x) q(x - 2 + e) with the edge-aware weights of edges.h ppm_w.
template <class D>
HD double ppm_al_coef(bool face, int x, int n1, const D& da, int k) {
  const int e = k - x + 2;
  if (e < 0 || e > 3) return 0.;
  double w[4]; ppm_w(face, x, n1, da, w);
  return w[e];
}
template <class D>
HD double ppm_dq(int iord, bool face, int m, int n1, int k, const D& da, double c) {
  if (iord == 1) return (c > 0.) ? (k == m - 1 ? 1. : 0.) : (k == m ? 1. : 0.);
  if (iord == 2) {   // c > 0: qt c(3-2c) + al(m) (1-c)^2 - al(m-1) c(1-c);  else: qt (-c)(3+2c) + al(m) (1+c)^2 + al(m+1) c(1+c)
    if (c > 0.) return (1. - c) * (1. - c) * ppm_al_coef(face, m, n1, da, k) - c * (1. - c) * ppm_al_coef(face, m - 1, n1, da, k) + (k == m - 1 ? c * (3. - 2. * c) : 0.);
    return (1. + c) * (1. + c) * ppm_al_coef(face, m, n1, da, k) + c * (1. + c) * ppm_al_coef(face, m + 1, n1, da, k) + (k == m ? -c * (3. + 2. * c) : 0.);
  }
  const double c2 = c * c / 6.;      // iord == 333
  if (c > 0.) return k == m ? 2. / 6. - 0.5 * c + c2 : k == m - 1 ? 5. / 6. + 0.5 * c - 2. * c2 : k == m - 2 ? -1. / 6. + c2 : 0.;
  return k == m - 1 ? 2. / 6. + 0.5 * c + c2 : k == m ? 5. / 6. - 0.5 * c - 2. * c2 : k == m + 1 ? -1. / 6. + c2 : 0.;
}

// ---- monotone PPM of the NONLINEAR xppm / yppm, iord 8 and 10 (tp_core_tlm.F90:592-955, :1339-1723; pert_ppm :1853-1915) -- values
// only: the reference never differentiates them; with split schemes they give the trajectory values of a transport whose tangent
// / adjoint is taken of the perturbation scheme (sw_core_tlm.F90:1664-1682).  Point-wise form: the slopes bl, br of ONE cell from
// the cells within three of it, so the flux at interface m reads q(m-3 .. m+2), inside the three-cell halo like the reference.
constexpr double MONO_R3 = 1. / 3., MONO_S11 = 11. / 14., MONO_S14 = 4. / 7., MONO_S15 = 3. / 14., MONO_NEAR_ZERO = 1.e-25;
HD double mono_sign(double a, double b) { return b >= 0. ? fabs(a) : -fabs(a); }     // Fortran SIGN(a, b)
HD double mono_dm3(double a, double b, double c_) {                                   // dm of the middle cell (:596-640)
  const double xt = 0.25 * (c_ - a);
  const double hi = fmax(fmax(a, b), c_) - b, lo = b - fmin(fmin(a, b), c_);
  return mono_sign(fmin(fmin(fabs(xt), hi), lo), xt);
}
HD void mono_pert_ppm(double& al, double& ar) {                                       // pert_ppm, iv = 1 (:1893-1913)
  if (al * ar < 0.) {
    const double da1 = al - ar, da2 = da1 * da1, a6da = 3. * (al + ar) * da1;
    if (a6da < -da2) ar = -(2. * al);
    else if (a6da > da2) al = -(2. * ar);
  } else { al = 0.; ar = 0.; }
}
// the two-sided value on a cube edge between cells e-1 and e: w = q(e-2 .. e+1), metric by absolute index (:833-835)
template <class D>
HD double mono_two_sided(const D& da, int e, double w0, double w1, double w2, double w3) {
  return 0.5 * (((2. * da(e - 1) + da(e - 2)) * w1 - da(e - 1) * w0) / (da(e - 2) + da(e - 1)) + ((2. * da(e) + da(e + 1)) * w2 - da(e) * w3) / (da(e) + da(e + 1)));
}
// Slopes of cell i from its five-cell window v[0..4] = q(i-2 .. i+2) (registers; all indices static).
template <class D>
HD void mono_blbr(int iord, bool face, int i, int n1, const double* v, const D& da, double& bl, double& br) {
  const double dm_m = mono_dm3(v[0], v[1], v[2]), dm_0 = mono_dm3(v[1], v[2], v[3]), dm_p = mono_dm3(v[2], v[3], v[4]);
  if (face && (i <= 2 || i >= n1 - 2)) {       // the three cells each side of a cube edge (:828-942)
    auto clamp4 = [](double xt, double a, double b, double c_, double d) { xt = fmax(xt, fmin(fmin(a, b), fmin(c_, d))); return fmin(xt, fmax(fmax(a, b), fmax(c_, d))); };
    if (i == 0) { bl = MONO_S14 * dm_m + MONO_S11 * (v[1] - v[2]); br = clamp4(mono_two_sided(da, 1, v[1], v[2], v[3], v[4]), v[1], v[2], v[3], v[4]) - v[2]; }
    else if (i == 1) { bl = clamp4(mono_two_sided(da, 1, v[0], v[1], v[2], v[3]), v[0], v[1], v[2], v[3]) - v[2]; br = MONO_S15 * v[2] + MONO_S11 * v[3] - MONO_S14 * dm_p - v[2]; }
    else if (i == 2) { bl = MONO_S15 * v[1] + MONO_S11 * v[2] - MONO_S14 * dm_0 - v[2]; br = 0.5 * (v[2] + v[3]) + MONO_R3 * (dm_0 - dm_p) - v[2]; }
    else if (i == n1 - 2) { bl = 0.5 * (v[1] + v[2]) + MONO_R3 * (dm_m - dm_0) - v[2]; br = MONO_S15 * v[3] + MONO_S11 * v[2] + MONO_S14 * dm_0 - v[2]; }
    else if (i == n1 - 1) { bl = MONO_S15 * v[2] + MONO_S11 * v[1] + MONO_S14 * dm_m - v[2]; br = clamp4(mono_two_sided(da, n1, v[1], v[2], v[3], v[4]), v[1], v[2], v[3], v[4]) - v[2]; }
    else { bl = clamp4(mono_two_sided(da, n1, v[0], v[1], v[2], v[3]), v[0], v[1], v[2], v[3]) - v[2]; br = MONO_S11 * (v[3] - v[2]) - MONO_S14 * dm_p; }
    mono_pert_ppm(bl, br);
    return;
  }
  const double qi = v[2], al0 = 0.5 * (v[1] + v[2]) + MONO_R3 * (dm_m - dm_0), al1 = 0.5 * (v[2] + v[3]) + MONO_R3 * (dm_0 - dm_p);   // :641-642
  if (iord == 8 || iord == 11) {               // Lin's fast monotone constraint (:643-678); 11 (:679-715): the same with ppm_fac = 1.5 (:35) in place of 2
    const double xt = (iord == 8 ? 2. : 1.5) * dm_0;
    bl = -mono_sign(fmin(fabs(xt), fabs(al0 - qi)), xt);
    br = mono_sign(fmin(fabs(xt), fabs(al1 - qi)), xt);
    return;
  }
  bl = al0 - qi; br = al1 - qi;                // iord = 9, 10, 12, 13: Huynh's second constraint (:716-823)
  if (fabs(dm_m) + fabs(dm_0) + fabs(dm_p) < MONO_NEAR_ZERO) { bl = 0.; br = 0.; }
  else if (fabs(3. * (bl + br)) > fabs(bl - br)) {
    const double pmp_2 = 2. * (v[2] - v[1]), lac_2 = pmp_2 - 0.75 * (2. * (v[1] - v[0]));
    br = fmin(fmax(0., fmax(pmp_2, lac_2)), fmax(br, fmin(0., fmin(pmp_2, lac_2))));
    const double pmp_1 = -(2. * (v[3] - v[2])), lac_1 = pmp_1 + 0.75 * (2. * (v[4] - v[3]));
    bl = fmin(fmax(0., fmax(pmp_1, lac_1)), fmax(bl, fmin(0., fmin(pmp_1, lac_1))));
  }
  if (iord == 9 || iord == 13) {               // positive definite constraint: pert_ppm with iv = 0 on the cells away from a cube edge (:826-828, :1867-1892)
    if (qi <= 0.) { bl = 0.; br = 0.; return; }
    const double a4 = -3. * (br + bl), da1 = br - bl;
    if (fabs(da1) < -a4) {
      const double fmn = qi + 0.25 / a4 * da1 * da1 + a4 * (1. / 12.);
      if (fmn < 0.) {
        if (br > 0. && bl > 0.) { br = 0.; bl = 0.; }
        else if (da1 > 0.) br = -2. * bl;
        else bl = -2. * br;
      }
    }
  }
}
HD constexpr bool hord_mono(int iord) { return iord >= 8 && iord <= 13; }
// flux at interface m with the trajectory scheme: the differentiable schemes through ppm_flux, 8 / 10 through the slopes of the upwind
// cell (:944-952).  The six cells q(m-3 .. m+2) are read once; the upwind cell's window is picked by selects, never by a dynamic index.
// The limited low-order schemes of the NONLINEAR routines, iord 3 .. 7 (xppm tp_core_tlm.F90:442-590, yppm :1060-1336; xtp_u
// sw_core_tlm.F90:4483-4612, ytp_v :5147-5360) -- values only, like 8 / 10: the edge values and slopes of the linear scheme 2, and the tests
// smt5 / smt6 on the slopes of the two cells beside the interface decide which of them keeps its parabola.  (blm, brm): cell m-1, (bl0, br0): cell m;
// x = the Courant number of the flux formula (c, or c * rd(upwind) for the momentum fluxes).
HD double low_flux(int iord, double qm, double q0, double blm, double brm, double bl0, double br0, bool up, double x) {
  const double b0m = blm + brm, b00 = bl0 + br0;
  if (iord == 3 || iord == 4) {
    const double xm = fabs(blm - brm), x0 = fabs(bl0 - br0);
    const bool s5m = fabs(b0m) < xm, s6m = 3. * fabs(b0m) < xm, s50 = fabs(b00) < x0, s60 = 3. * fabs(b00) < x0;
    if (iord == 4) {
      if (up) return (s6m || s50) ? qm + (1. - x) * (brm - x * b0m) : qm;
      return (s60 || s5m) ? q0 + (1. + x) * (bl0 + x * b00) : q0;
    }
    double fx1 = 0.;
    if (up) {
      if (s6m || s50) fx1 = brm - x * b0m;
      else if (s5m) fx1 = mono_sign(fmin(fabs(blm), fabs(brm)), brm);
      return qm + (1. - x) * fx1;
    }
    if (s60 || s5m) fx1 = bl0 + x * b00;
    else if (s50) fx1 = mono_sign(fmin(fabs(bl0), fabs(br0)), bl0);
    return q0 + (1. + x) * fx1;
  }
  const bool s5m = iord == 5 ? blm * brm < 0. : fabs(3. * b0m) < fabs(blm - brm), s50 = iord == 5 ? bl0 * br0 < 0. : fabs(3. * b00) < fabs(bl0 - br0);
  const double fx1 = up ? (1. - x) * (brm - x * b0m) : (1. + x) * (bl0 + x * b00);
  const double f = up ? qm : q0;
  return (s5m || s50) ? f + fx1 : f;
}
HD constexpr bool hord_low(int iord) { return iord >= 3 && iord <= 7; }
template <class Q, class D>
HD double ppm_al7(int iord, bool face, int x, int n1, const Q& q, const D& da) {      // iord 7: edge values kept non-negative (tp_core_tlm.F90:345-349, :357-373, :382-405)
  const double al = ppm_al<double>(face, x, n1, q, da);
  if (iord != 7) return al;
  if (face && (x <= 2 || x >= n1 - 1)) return fmax(0., al);
  return al < 0. ? 0.5 * (q(x - 1) + q(x)) : al;
}
template <class Q, class D>
HD double ppm_flux_traj(int iord, bool face, int m, int n1, const Q& q, const D& da, double cc) {
  if (hord_low(iord)) {
    const double alm = ppm_al7(iord, face, m - 1, n1, q, da), al0 = ppm_al7(iord, face, m, n1, q, da), alp = ppm_al7(iord, face, m + 1, n1, q, da);
    const double qm = q(m - 1), q0 = q(m);
    return low_flux(iord, qm, q0, alm - qm, al0 - qm, al0 - q0, alp - q0, cc > 0., cc);
  }
  if (!hord_mono(iord)) return ppm_flux<double>(iord, face, m, n1, q, da, cc);
  const double w0 = q(m - 3), w1 = q(m - 2), w2 = q(m - 1), w3 = q(m), w4 = q(m + 1), w5 = q(m + 2);
  const bool up = cc > 0.;
  const double v[5] = {up ? w0 : w1, up ? w1 : w2, up ? w2 : w3, up ? w3 : w4, up ? w4 : w5};
  double bl, br;
  mono_blbr(iord, face, up ? m - 1 : m, n1, v, da, bl, br);
  return up ? v[2] + (1. - cc) * (br - cc * (bl + br)) : v[2] + (1. + cc) * (bl + cc * (bl + br));
}

// xtp_u / ytp_v flux at interface m (sw_core_tlm.F90:7272-7486, :7490-7759): cfl = c * rd(upwind cell).
template <class T, class Q, class D>
HD T tp_uv_flux(int iord, bool face, int m, int n1, bool row_edge, const Q& q, const D& dd, T cc, double rd_m, double rd_0) {
  if (iord == 1) return (val(cc) > 0.) ? q(m - 1) : q(m);
  if (iord == 333) {
    if (val(cc) > 0.)
      return (2.0 * q(m) + 5.0 * q(m - 1) - q(m - 2)) / 6.0 - 0.5 * cc * rd_m * (q(m) - q(m - 1)) +
             cc * rd_m * cc * rd_m / 6.0 * (q(m) - 2.0 * q(m - 1) + q(m - 2));
    return (2.0 * q(m - 1) + 5.0 * q(m) - q(m + 1)) / 6.0 - 0.5 * cc * rd_0 * (q(m) - q(m - 1)) +
           cc * rd_0 * cc * rd_0 / 6.0 * (q(m + 1) - 2.0 * q(m) + q(m - 1));
  }
  T bl, br;
  if (val(cc) > 0.) {
    uv_blbr<T>(face, m - 1, n1, row_edge, q, dd, bl, br);
    T cfl = cc * rd_m;
    return q(m - 1) + (1. - cfl) * (br - cfl * (bl + br));
  }
  uv_blbr<T>(face, m, n1, row_edge, q, dd, bl, br);
  T cfl = cc * rd_0;
  return q(m) + (1. + cfl) * (bl + cfl * (bl + br));
}

// The monotone slopes of the NONLINEAR xtp_u / ytp_v, iord 8 / 10 (sw_core_tlm.F90:4620-5041, :5361-5865): as mono_blbr with the
// momentum fluxes' own edge treatment (D-grid metric, two-sided edge value unclamped, zero slopes next to a face corner, pert_ppm
// on cells 2 and n1-2 only) and their own 2-delta-x test.  Values only (split_hord, :1987-2002).
template <class D>
HD void uv_mono_blbr(int iord, bool face, int i, int n1, bool row_edge, const double* v, const D& dd, double& bl, double& br) {
  const double dm_m = mono_dm3(v[0], v[1], v[2]), dm_0 = mono_dm3(v[1], v[2], v[3]), dm_p = mono_dm3(v[2], v[3], v[4]);
  if (face && (i <= 2 || i >= n1 - 2)) {
    auto two_sided = [&](int e, double w0, double w1, double w2, double w3) {    // x0l + x0r (:4907-4912)
      return 0.5 * ((2. * dd(e - 1) + dd(e - 2)) * w1 - dd(e - 1) * w0) / (dd(e - 1) + dd(e - 2)) + 0.5 * ((2. * dd(e) + dd(e + 1)) * w2 - dd(e) * w3) / (dd(e) + dd(e + 1));
    };
    if (row_edge && (i <= 1 || i >= n1 - 1)) { bl = 0.; br = 0.; return; }
    if (i == 0) { bl = MONO_S14 * dm_m - MONO_S11 * (v[2] - v[1]); br = two_sided(1, v[1], v[2], v[3], v[4]) - v[2]; }
    else if (i == 1) { bl = two_sided(1, v[0], v[1], v[2], v[3]) - v[2]; br = MONO_S15 * v[2] + MONO_S11 * v[3] - MONO_S14 * dm_p - v[2]; }
    else if (i == 2) { bl = MONO_S15 * v[1] + MONO_S11 * v[2] - MONO_S14 * dm_0 - v[2]; br = 0.5 * (v[2] + v[3]) + MONO_R3 * (dm_0 - dm_p) - v[2]; mono_pert_ppm(bl, br); }
    else if (i == n1 - 2) { bl = 0.5 * (v[1] + v[2]) + MONO_R3 * (dm_m - dm_0) - v[2]; br = MONO_S15 * v[3] + MONO_S11 * v[2] + MONO_S14 * dm_0 - v[2]; mono_pert_ppm(bl, br); }
    else if (i == n1 - 1) { bl = MONO_S15 * v[2] + MONO_S11 * v[1] + MONO_S14 * dm_m - v[2]; br = two_sided(n1, v[1], v[2], v[3], v[4]) - v[2]; }
    else { bl = two_sided(n1, v[0], v[1], v[2], v[3]) - v[2]; br = MONO_S11 * (v[3] - v[2]) - MONO_S14 * dm_p; }
    return;
  }
  const double qi = v[2], al0 = 0.5 * (v[1] + v[2]) + MONO_R3 * (dm_m - dm_0), al1 = 0.5 * (v[2] + v[3]) + MONO_R3 * (dm_0 - dm_p);
  if (iord == 8) {
    const double xt = 2. * dm_0;
    bl = -mono_sign(fmin(fabs(xt), fabs(al0 - qi)), xt);
    br = mono_sign(fmin(fabs(xt), fabs(al1 - qi)), xt);
    return;
  }
  bl = al0 - qi; br = al1 - qi;
  if (iord >= 11) return;                      // "un-limited: 11" -- the ELSE of the chain (sw_core_tlm.F90:4890-4896): 11, 12, 13
  if (iord == 9) {                             // the constraint everywhere, no 2-delta-x test (:4710-4780)
    const double pmp_1 = -(2. * (v[3] - v[2])), lac_1 = pmp_1 + 1.5 * (v[4] - v[3]);
    bl = fmin(fmax(0., fmax(pmp_1, lac_1)), fmax(bl, fmin(0., fmin(pmp_1, lac_1))));
    const double pmp_2 = 2. * (v[2] - v[1]), lac_2 = pmp_2 - 1.5 * (v[1] - v[0]);
    br = fmin(fmax(0., fmax(pmp_2, lac_2)), fmax(br, fmin(0., fmin(pmp_2, lac_2))));
    return;
  }
  if (fabs(dm_0) < MONO_NEAR_ZERO) {
    if (fabs(dm_m) + fabs(dm_p) < MONO_NEAR_ZERO) { bl = 0.; br = 0.; }
  } else if (fabs(3. * (bl + br)) > fabs(bl - br)) {
    const double pmp_1 = -(2. * (v[3] - v[2])), lac_1 = pmp_1 + 1.5 * (v[4] - v[3]);
    bl = fmin(fmax(0., fmax(pmp_1, lac_1)), fmax(bl, fmin(0., fmin(pmp_1, lac_1))));
    const double pmp_2 = 2. * (v[2] - v[1]), lac_2 = pmp_2 - 1.5 * (v[1] - v[0]);
    br = fmin(fmax(0., fmax(pmp_2, lac_2)), fmax(br, fmin(0., fmin(pmp_2, lac_2))));
  }
}
template <class Q, class D>
HD double tp_uv_flux_traj(int iord, bool face, int m, int n1, bool row_edge, const Q& q, const D& dd, double cc, double rd_m, double rd_0) {
  if (hord_low(iord)) {
    double blm, brm, bl0, br0;
    uv_blbr<double>(face, m - 1, n1, row_edge, q, dd, blm, brm);
    uv_blbr<double>(face, m, n1, row_edge, q, dd, bl0, br0);
    const bool up = cc > 0.;
    return low_flux(iord, q(m - 1), q(m), blm, brm, bl0, br0, up, cc * (up ? rd_m : rd_0));
  }
  if (!hord_mono(iord)) return tp_uv_flux<double>(iord, face, m, n1, row_edge, q, dd, cc, rd_m, rd_0);
  const double w0 = q(m - 3), w1 = q(m - 2), w2 = q(m - 1), w3 = q(m), w4 = q(m + 1), w5 = q(m + 2);
  const bool up = cc > 0.;
  const double v[5] = {up ? w0 : w1, up ? w1 : w2, up ? w2 : w3, up ? w3 : w4, up ? w4 : w5};
  double bl, br;
  uv_mono_blbr(iord, face, up ? m - 1 : m, n1, row_edge, v, dd, bl, br);
  const double cfl = cc * (up ? rd_m : rd_0);
  return up ? v[2] + (1. - cfl) * (br - cfl * (bl + br)) : v[2] + (1. + cfl) * (bl + cfl * (bl + br));
}

// ===================================================================== c_sw
// d2a2c_vect A: D-grid winds -> A-grid (sw_core_tlm.F90:6505-6611); within 3 cells of a face edge the
// 4-point Lagrange interpolation gives way to 2-point averages (:6548-6602, npt = 4).
struct CswInterpAD {
  STAGE_COMMON("CswInterpA", 2, 4)   // in: u v   out: utmp vtmp ua va
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -1, 2, 0, 0} : Box{-1, 2, 0, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T ut = T(0.), vt = T(0.);
    const bool band = EDGE && (i <= 3 || i >= c.g.nx - 2 || j <= 3 || j >= c.g.ny - 2);
    if (orect[0].has(i, j)) ut = band ? 0.5 * (IN(0, i, j) + IN(0, i, j + 1)) : A2 * (IN(0, i, j - 1) + IN(0, i, j + 2)) + A1 * (IN(0, i, j) + IN(0, i, j + 1));
    if (orect[1].has(i, j)) vt = band ? 0.5 * (IN(1, i, j) + IN(1, i + 1, j)) : A2 * (IN(1, i - 1, j) + IN(1, i + 2, j)) + A1 * (IN(1, i, j) + IN(1, i + 1, j));
    o[0] = ut; o[1] = vt;
    const double cs = MET(cosa_s, i, j), r2 = MET(rsin2, i, j);
    o[2] = (ut - vt * cs) * r2;
    o[3] = (vt - ut * cs) * r2;
  }
};
typedef Edged<CswInterpAD, false> CswInterpA;
typedef Edged<CswInterpAD, true> CswInterpAE;

// d2a2c_vect C: A-grid -> C-grid + contravariant flux-form winds (:6617-6803, :713-733).  The corner
// fixes of utmp/vtmp/ua/va (:6617-6640, :6662-6677, :6726-6760) are read through d2a2c_xview/yview.
template <bool EDGE>
struct CswInterpC_ {
  STAGE_BASE(EDGE ? "CswInterpCe" : "CswInterpC", 6, 4)   // in: utmp vtmp u v ua va   out: uc0 utf vc0 vtf
  double dt2;
  HD static constexpr bool uses(int, int, int, int) { return true; }
  HD static constexpr unsigned wants(int M) { return (M == 0 || M == 4) ? 0x3u : (M == 1 || M == 5) ? 0xCu : M == 2 ? 0x8u : 0x2u; }
  HD static constexpr Box box(int M) { return (M == 0 || M == 4) ? Box{-2, 1, 0, 0, 0, 0} : (M == 1 || M == 5) ? Box{0, 0, -2, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  static constexpr int NALIAS = EDGE ? 1 : 0;
  HD static constexpr int alias_box(int M) { return M == 0 ? 1 : M == 1 ? 0 : M == 4 ? 5 : M == 5 ? 4 : M; }
  HD bool alias(const Ctx& c, int M, int i, int j, int, int& ai, int& aj) const {
    if (!EDGE) return false;
    if (M == 1 || M == 5) return d2a2c_xalias(c.g, i, j, ai, aj);
    if (M == 0 || M == 4) return d2a2c_yalias(c.g, i, j, ai, aj);
    return false;
  }
  template <int MU, int MV, class T, class A>
  HD T rx(const A& a, const Ctx& c, int i, int j) const {   // utmp / ua as the x-direction pass sees them
    int oi, oj; double sg;
    if (EDGE && d2a2c_xview(c.g, i, j, oi, oj, sg)) return sg * IN(MV, oi, oj);
    return IN(MU, i, j);
  }
  template <int MV, int MU, class T, class A>
  HD T ry(const A& a, const Ctx& c, int i, int j) const {   // vtmp / va as the y-direction pass sees them
    int oi, oj; double sg;
    if (EDGE && d2a2c_yview(c.g, i, j, oi, oj, sg)) return sg * IN(MU, oi, oj);
    return IN(MV, i, j);
  }
  // Hand-written gather adjoint of the bulk form: away from the cube edges the interpolation is linear with the constant weights A1, A2 and the
  // only trajectory dependence is the sign of ut / vt that picks the sin_sg factor.  The generic gather evaluated the stage seven times per
  // point on DualV<6> (2.19 ms per launch against 0.94 for the tangent); the store rules are those of exec.h body_ad_joint.
  static constexpr bool HAND_AD = !EDGE;
  HD void hand_ad(const Ctx& c, const Rect& R, int i, int j, int z) const {
    const int nk = in[0].nk;
    if (z >= c.g.ntile * nk) return;
    const int tile = z / nk, k = 1 + z % nk;
    if (!wmask && (i < R.i0 - 2 || i > R.i1 + 1 || j < R.j0 - 2 || j > R.j1 + 1)) return;
    const size_t pb = (size_t)(tile * nk + k - 1) * c.g.plane;
    auto tr = [&](int m, int ii, int jj) { return in[m].t[pb + c.g.idx(ii, jj)]; };
    auto oa = [&](int n, int ii, int jj) { return out[n].p[pb + c.g.idx(ii, jj)]; };
    double acc[4] = {0., 0., 0., 0.};
    const double wgt[4] = {A2, A1, A1, A2};      // output at i-1+d reads the input at i with this weight
    if (k >= k0 && k <= k1) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {                // uc0, utf at (oi, j)
        const int oi = i - 1 + d;
        if (!orect[0].has(oi, j)) continue;
        const double uc = A2 * (tr(0, oi - 2, j) + tr(0, oi + 1, j)) + A1 * (tr(0, oi - 1, j) + tr(0, oi, j));
        const double rs = MET(rsin_u, oi, j), ut = (uc - tr(3, oi, j) * MET(cosa_u, oi, j)) * rs;
        const double f = orect[1].has(oi, j) ? dt2 * MET(dy, oi, j) * ((ut > 0.) ? SSG(3, oi - 1, j) : SSG(1, oi, j)) * oa(1, oi, j) : 0.;      // d(utf)/d(ut) utf_ad
        acc[0] += wgt[d] * (oa(0, oi, j) + rs * f);
        if (d == 1) acc[3] = -MET(cosa_u, oi, j) * rs * f;
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {                // vc0, vtf at (i, oj)
        const int oj = j - 1 + d;
        if (!orect[2].has(i, oj)) continue;
        const double vc = A2 * (tr(1, i, oj - 2) + tr(1, i, oj + 1)) + A1 * (tr(1, i, oj - 1) + tr(1, i, oj));
        const double rs = MET(rsin_v, i, oj), vt = (vc - tr(2, i, oj) * MET(cosa_v, i, oj)) * rs;
        const double f = orect[3].has(i, oj) ? dt2 * MET(dx, i, oj) * ((vt > 0.) ? SSG(4, i, oj - 1) : SSG(2, i, oj)) * oa(3, i, oj) : 0.;
        acc[1] += wgt[d] * (oa(2, i, oj) + rs * f);
        if (d == 1) acc[2] = -MET(cosa_v, i, oj) * rs * f;
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
      if (in[m].p) {
        const Box b = box(m);
        const bool in_m = !(i < R.i0 + b.di0 || i > R.i1 + b.di1 || j < R.j0 + b.dj0 || j > R.j1 + b.dj1);
        double* q = &in[m].p[pb + c.g.idx(i, j)];
        if ((wmask >> m) & 1u) *q = in_m ? acc[m] : 0.0;
        else if (in_m) *q += acc[m];
      }
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = o[2] = o[3] = T(0.);
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if ((a.want & 0x3u) && orect[0].has(i, j)) {
      T uc, ut;
      if (F && (i == 1 || i == npx)) {
        ut = edge_interp4<T>(rx<4, 5, T>(a, c, i - 2, j), rx<4, 5, T>(a, c, i - 1, j), rx<4, 5, T>(a, c, i, j), rx<4, 5, T>(a, c, i + 1, j),
                             MET(dxa, i - 2, j), MET(dxa, i - 1, j), MET(dxa, i, j), MET(dxa, i + 1, j));
        uc = (val(ut) > 0.) ? ut * SSG(3, i - 1, j) : ut * SSG(1, i, j);
      } else {
        if (F && (i == 0 || i == npx - 1)) uc = EC1 * rx<0, 1, T>(a, c, i - 2, j) + EC2 * rx<0, 1, T>(a, c, i - 1, j) + EC3 * rx<0, 1, T>(a, c, i, j);
        else if (F && (i == 2 || i == npx + 1)) uc = EC1 * rx<0, 1, T>(a, c, i + 1, j) + EC2 * rx<0, 1, T>(a, c, i, j) + EC3 * rx<0, 1, T>(a, c, i - 1, j);
        else uc = A2 * (rx<0, 1, T>(a, c, i - 2, j) + rx<0, 1, T>(a, c, i + 1, j)) + A1 * (rx<0, 1, T>(a, c, i - 1, j) + rx<0, 1, T>(a, c, i, j));
        ut = (uc - IN(3, i, j) * MET(cosa_u, i, j)) * MET(rsin_u, i, j);
      }
      o[0] = uc;
      o[1] = (val(ut) > 0.) ? dt2 * ut * MET(dy, i, j) * SSG(3, i - 1, j) : dt2 * ut * MET(dy, i, j) * SSG(1, i, j);
    }
    if ((a.want & 0xCu) && orect[2].has(i, j)) {
      T vc, vt;
      if (F && (j == 1 || j == npy)) {
        vt = edge_interp4<T>(ry<5, 4, T>(a, c, i, j - 2), ry<5, 4, T>(a, c, i, j - 1), ry<5, 4, T>(a, c, i, j), ry<5, 4, T>(a, c, i, j + 1),
                             MET(dya, i, j - 2), MET(dya, i, j - 1), MET(dya, i, j), MET(dya, i, j + 1));
        vc = (val(vt) > 0.) ? vt * SSG(4, i, j - 1) : vt * SSG(2, i, j);
      } else {
        if (F && (j == 0 || j == npy - 1)) vc = EC1 * ry<1, 0, T>(a, c, i, j - 2) + EC2 * ry<1, 0, T>(a, c, i, j - 1) + EC3 * ry<1, 0, T>(a, c, i, j);
        else if (F && (j == 2 || j == npy + 1)) vc = EC1 * ry<1, 0, T>(a, c, i, j + 1) + EC2 * ry<1, 0, T>(a, c, i, j) + EC3 * ry<1, 0, T>(a, c, i, j - 1);
        else vc = A2 * (ry<1, 0, T>(a, c, i, j - 2) + ry<1, 0, T>(a, c, i, j + 1)) + A1 * (ry<1, 0, T>(a, c, i, j - 1) + ry<1, 0, T>(a, c, i, j));
        vt = (vc - IN(2, i, j) * MET(cosa_v, i, j)) * MET(rsin_v, i, j);
      }
      o[2] = vc;
      o[3] = (val(vt) > 0.) ? dt2 * vt * MET(dx, i, j) * SSG(4, i, j - 1) : dt2 * vt * MET(dx, i, j) * SSG(2, i, j);
    }
  }
};

// divergence_corner (sw_core_tlm.F90:4036-4081)
struct CswDivgD {
  STAGE_COMMON("CswDivg", 4, 1)   // in: u v ua va   out: divgd
  HD static constexpr Box box(int M) { return M == 0 ? Box{-1, 0, 0, 0, 0, 0} : M == 1 ? Box{0, 0, -1, 0, 0, 0} : Box{-1, 0, -1, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD T uf(const A& a, const Ctx& c, int tile, int i, int j) const {
    const double w = MET(dyc, i, j) * 0.5 * (SSG(4, i, j - 1) + SSG(2, i, j));
    if (EDGE && (j == 1 || j == c.g.ny + 1)) return IN(0, i, j) * w;
    return (IN(0, i, j) - 0.25 * (IN(3, i, j - 1) + IN(3, i, j)) * (CSG(4, i, j - 1) + CSG(2, i, j))) * w;
  }
  template <bool EDGE, class T, class A>
  HD T vf(const A& a, const Ctx& c, int tile, int i, int j) const {
    const double w = MET(dxc, i, j) * 0.5 * (SSG(3, i - 1, j) + SSG(1, i, j));
    if (EDGE && (i == 1 || i == c.g.nx + 1)) return IN(1, i, j) * w;
    return (IN(1, i, j) - 0.25 * (IN(2, i - 1, j) + IN(2, i, j)) * (CSG(3, i - 1, j) + CSG(1, i, j))) * w;
  }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    constexpr bool F = EDGE;
    const bool drop_lo = F && j == 1 && (i == 1 || i == npx), drop_hi = F && j == npy && (i == 1 || i == npx);   // :4071-4078
    T d = uf<EDGE, T>(a, c, tile, i - 1, j) - uf<EDGE, T>(a, c, tile, i, j);
    if (!drop_lo) d = d + vf<EDGE, T>(a, c, tile, i, j - 1);
    if (!drop_hi) d = d - vf<EDGE, T>(a, c, tile, i, j);
    o[0] = MET(rarea_c, i, j) * d;
  }
};
typedef Edged<CswDivgD, false> CswDivg;
typedef Edged<CswDivgD, true> CswDivgE;

// first-order upwind transport of delp, pt on the C grid (sw_core_tlm.F90:739-808); the corner halo of
// delp/pt is read through fill2_4corners' x-view for the x-fluxes and y-view otherwise.
struct CswTransportD {
  static constexpr bool LDS_AD_OK = true;
  STAGE_BASE("CswTransport", 4, 2)   // in: delp pt utf vtf   out: delpc ptc
  HD static constexpr bool uses(int M, int di, int dj, int) { return M >= 2 || di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int M) { return M == 1 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 1, -1, 1, 0, 0} : M == 2 ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  static constexpr int NALIAS = 2;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int M, int i, int j, int n, int& ai, int& aj) const {
    return M < 2 && fill2_alias(c.g, n + 1, i, j, ai, aj);
  }
  template <bool EDGE, int M, class T, class A>
  HD T rd(const A& a, const Ctx& c, int dir, int i, int j) const { if (EDGE) fill2_map(c.g, dir, i, j); return IN(M, i, j); }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T fx1[2], fx[2], fy1[2], fy[2];
    for (int d = 0; d < 2; ++d) {
      T ut = IN(2, i + d, j);
      const int iu = (val(ut) > 0.) ? i + d - 1 : i + d;
      fx1[d] = ut * rd<EDGE, 0, T>(a, c, 1, iu, j);
      fx[d] = fx1[d] * rd<EDGE, 1, T>(a, c, 1, iu, j);
      T vt = IN(3, i, j + d);
      const int ju = (val(vt) > 0.) ? j + d - 1 : j + d;
      fy1[d] = vt * rd<EDGE, 0, T>(a, c, 2, i, ju);
      fy[d] = fy1[d] * rd<EDGE, 1, T>(a, c, 2, i, ju);
    }
    const double ra = MET(rarea, i, j);
    T dp = rd<EDGE, 0, T>(a, c, 2, i, j), p = rd<EDGE, 1, T>(a, c, 2, i, j);
    T dpc = dp + (fx1[0] - fx1[1] + (fy1[0] - fy1[1])) * ra;
    o[0] = dpc;
    o[1] = (p * dp + (fx[0] - fx[1] + (fy[0] - fy[1])) * ra) / dpc;
  }
};
typedef Edged<CswTransportD, false> CswTransport;
typedef Edged<CswTransportD, true> CswTransportE;

// kinetic energy at cell centres and absolute vorticity at corners (sw_core_tlm.F90:853-957)
struct CswKeVortD {
  STAGE_BASE("CswKeVort", 6, 2)   // in: ua va uc0 vc0 u v   out: ke vort
  STAGE_NO_ALIAS
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M == 2 ? !(di == 1 && dj == -1) : M == 3 ? !(di == -1 && dj == 1) : true; }
  HD static constexpr unsigned wants(int M) { return (M < 2 || M > 3) ? 0x1u : 0x3u; }
  HD static constexpr Box box(int M) {
    return M < 2 ? Box{0, 0, 0, 0, 0, 0} : M == 2 ? Box{0, 1, -1, 0, 0, 0} : M == 3 ? Box{-1, 0, 0, 1, 0, 0} : M == 4 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 1, 0, 0, 0, 0};
  }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T ua = IN(0, i, j), va = IN(1, i, j), ku, kv;
      if (val(ua) > 0.) {
        if (F && (i == 1 || i == npx)) ku = IN(2, i, j) * SSG(1, i, j) + IN(5, i, j) * CSG(1, i, j);
        else ku = IN(2, i, j);
      } else {
        if (F && (i == 0 || i == npx - 1)) ku = IN(2, i + 1, j) * SSG(3, i, j) + IN(5, i + 1, j) * CSG(3, i, j);
        else ku = IN(2, i + 1, j);
      }
      if (val(va) > 0.) {
        if (F && (j == 1 || j == npy)) kv = IN(3, i, j) * SSG(2, i, j) + IN(4, i, j) * CSG(2, i, j);
        else kv = IN(3, i, j);
      } else {
        if (F && (j == 0 || j == npy - 1)) kv = IN(3, i, j + 1) * SSG(4, i, j) + IN(4, i, j + 1) * CSG(4, i, j);
        else kv = IN(3, i, j + 1);
      }
      o[0] = (0.5 * dt2) * (ua * ku + va * kv);
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      const bool drop_m = F && i == 1 && (j == 1 || j == npy), drop_p = F && i == npx && (j == 1 || j == npy);   // :945-948
      T v = IN(2, i, j - 1) * MET(dxc, i, j - 1) - IN(2, i, j) * MET(dxc, i, j);
      if (!drop_p) v = v + IN(3, i, j) * MET(dyc, i, j);
      if (!drop_m) v = v - IN(3, i - 1, j) * MET(dyc, i - 1, j);
      o[1] = MET(fC, i, j) + MET(rarea_c, i, j) * v;
    }
  }
};
typedef Edged<CswKeVortD, false> CswKeVort;
typedef Edged<CswKeVortD, true> CswKeVortE;

// time-centred C-grid winds (sw_core_tlm.F90:985-1037)
struct CswUpdateD {
  STAGE_BASE("CswUpdate", 6, 2)   // in: uc0 vc0 u v vort ke   out: uc1 vc1
  STAGE_NO_ALIAS
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 4 || (M == 4 ? !(di == 1 && dj == 1) : !(di == -1 && dj == -1)); }
  HD static constexpr unsigned wants(int M) { return (M == 0 || M == 3) ? 0x1u : (M == 1 || M == 2) ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 4 ? Box{0, 0, 0, 0, 0, 0} : M == 4 ? Box{0, 1, 0, 1, 0, 0} : Box{-1, 0, -1, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T uc = IN(0, i, j);
      T fy1 = (F && (i == 1 || i == npx)) ? dt2 * IN(3, i, j) : dt2 * (IN(3, i, j) - uc * MET(cosa_u, i, j)) / MET(sina_u, i, j);
      T fy = (val(fy1) > 0.) ? IN(4, i, j) : IN(4, i, j + 1);
      o[0] = uc + fy1 * fy + MET(rdxc, i, j) * (IN(5, i - 1, j) - IN(5, i, j));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T vc = IN(1, i, j);
      T fx1 = (F && (j == 1 || j == npy)) ? dt2 * IN(2, i, j) : dt2 * (IN(2, i, j) - vc * MET(cosa_v, i, j)) / MET(sina_v, i, j);
      T fx = (val(fx1) > 0.) ? IN(4, i, j) : IN(4, i + 1, j);
      o[1] = vc - fx1 * fx + MET(rdyc, i, j) * (IN(5, i, j - 1) - IN(5, i, j));
    }
  }
};
typedef Edged<CswUpdateD, false> CswUpdate;
typedef Edged<CswUpdateD, true> CswUpdateE;

// p_grad_c, hydrostatic (dyn_core_tlm.F90:3310-3334)
struct PGradC {
  STAGE_BASE("PGradC", 4, 2)   // in: pkc gz (npz+1) uc1 vc1   out: uc2 vc2
  STAGE_NO_ALIAS
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M >= 2 || !(di == -1 && dj == -1); }
  HD static constexpr unsigned wants(int M) { return M == 2 ? 0x1u : M == 3 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 0, -1, 0, 0, 1} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    T pk00 = IN(0, i, j, 0), pk01 = IN(0, i, j, 1);
    T gz00 = IN(1, i, j, 0), gz01 = IN(1, i, j, 1);
    T wk0 = pk01 - pk00;
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T pkm0 = IN(0, i - 1, j, 0), pkm1 = IN(0, i - 1, j, 1);
      T gzm0 = IN(1, i - 1, j, 0), gzm1 = IN(1, i - 1, j, 1);
      o[0] = IN(2, i, j) + dt2 * MET(rdxc, i, j) / ((pkm1 - pkm0) + wk0) * ((gzm1 - gz00) * (pk01 - pkm0) + (gzm0 - gz01) * (pkm1 - pk00));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T pkm0 = IN(0, i, j - 1, 0), pkm1 = IN(0, i, j - 1, 1);
      T gzm0 = IN(1, i, j - 1, 0), gzm1 = IN(1, i, j - 1, 1);
      o[1] = IN(3, i, j) + dt2 * MET(rdyc, i, j) / ((pkm1 - pkm0) + wk0) * ((gzm1 - gz00) * (pk01 - pkm0) + (gzm0 - gz01) * (pkm1 - pk00));
    }
  }
};

// ===================================================================== d_sw
// Courant numbers and area fluxes from the contravariant winds (sw_core_tlm.F90:2932-2968)
template <class T>
HD void winds_to_flux_x(const Ctx& c, int tile, int i, int j, double dt, const T& ut, T& crx, T& xfx) {
  T x = dt * ut;
  if (val(x) > 0.) { crx = x * MET(rdxa, i - 1, j); xfx = MET(dy, i, j) * x * SSG(3, i - 1, j); }
  else             { crx = x * MET(rdxa, i, j);     xfx = MET(dy, i, j) * x * SSG(1, i, j); }
}
template <class T>
HD void winds_to_flux_y(const Ctx& c, int tile, int i, int j, double dt, const T& vt, T& cry, T& yfx) {
  T y = dt * vt;
  if (val(y) > 0.) { cry = y * MET(rdya, i, j - 1); yfx = MET(dx, i, j) * y * SSG(4, i, j - 1); }
  else             { cry = y * MET(rdya, i, j);     yfx = MET(dx, i, j) * y * SSG(2, i, j); }
}
// contravariant winds, Courant numbers, area fluxes; no face edge in the tile (:2722-2738, :2932-2968)
struct DswWinds {
  STAGE_COMMON("DswWinds", 2, 6)   // in: uc vc   out: ut crx xfx vt cry yfx
  double dt;
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, -1, 0, 0, 0} : Box{-1, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    for (int n = 0; n < 6; ++n) o[n] = T(0.);
    if (orect[0].has(i, j)) {
      T ut = (IN(0, i, j) - 0.25 * MET(cosa_u, i, j) * (IN(1, i - 1, j) + IN(1, i, j) + IN(1, i - 1, j + 1) + IN(1, i, j + 1))) * MET(rsin_u, i, j);
      o[0] = ut;
      if (orect[1].has(i, j)) winds_to_flux_x<T>(c, tile, i, j, dt, ut, o[1], o[2]);
    }
    if (orect[3].has(i, j)) {
      T vt = (IN(1, i, j) - 0.25 * MET(cosa_v, i, j) * (IN(0, i, j - 1) + IN(0, i + 1, j - 1) + IN(0, i, j) + IN(0, i + 1, j))) * MET(rsin_v, i, j);
      o[3] = vt;
      if (orect[4].has(i, j)) winds_to_flux_y<T>(c, tile, i, j, dt, vt, o[4], o[5]);
    }
  }
};
// Face tiles, pass A: the winds away from the edges and the edge-normal values on the edges
// (:2717-2751, :2768-2776, :2795-2803, :2821-2830).  ut on the two rows next to a south/north edge is
// left 0 here (pass B).
struct DswWindsAD {
  static constexpr bool LDS_FW_OK = true;
  STAGE_COMMON("DswWindsA", 2, 2)   // in: uc vc   out: ut_a vt_a
  double dt;
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, -1, 0, 0, 0} : Box{-1, 0, 0, 1, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if (orect[0].has(i, j)) {
      if (i == 1 || i == npx) { T uc = IN(0, i, j); o[0] = (val(uc) * dt > 0.) ? uc / SSG(3, i - 1, j) : uc / SSG(1, i, j); }
      else if (!(j == 0 || j == 1 || j == npy - 1 || j == npy))
        o[0] = (IN(0, i, j) - 0.25 * MET(cosa_u, i, j) * (IN(1, i - 1, j) + IN(1, i, j) + IN(1, i - 1, j + 1) + IN(1, i, j + 1))) * MET(rsin_u, i, j);
    }
    if (orect[1].has(i, j)) {
      if (j == 1 || j == npy) { T vc = IN(1, i, j); o[1] = (val(vc) * dt > 0.) ? vc / SSG(4, i, j - 1) : vc / SSG(2, i, j); }
      else o[1] = (IN(1, i, j) - 0.25 * MET(cosa_v, i, j) * (IN(0, i, j - 1) + IN(0, i + 1, j - 1) + IN(0, i, j) + IN(0, i + 1, j))) * MET(rsin_v, i, j);
    }
  }
};
typedef Edged<DswWindsAD, false> DswWindsA;
typedef Edged<DswWindsAD, true> DswWindsAE;
// Pass B, on the strips next to the face edges only: the two ut rows next to a south/north edge and the two vt
// columns next to a west/east edge from the pass-A winds (:2752-2766, :2777-2793, :2804-2819, :2831-2846), and the
// 2x2 systems at the four corners (:2856-2919), which also only read pass-A values.  Everything else is pass A.
struct DswWindsE {
  STAGE_COMMON("DswWindsE", 4, 2)   // in: ut_a vt_a uc vc   out: ut_e (row strips) vt_e (column strips)
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 1, -1, 1, 0, 0} : M == 2 ? Box{0, 1, -1, 0, 0, 0} : Box{-1, 0, 0, 1, 0, 0}; }
  // the four vt around ut(i,j) / the four ut around vt(i,j), one of them (ei,ej) left out
  template <class T, class A>
  HD T vt4(const A& a, int i, int j, int ei, int ej) const {
    T s = T(0.);
    for (int dj = 0; dj <= 1; ++dj) for (int di = -1; di <= 0; ++di) if (!(i + di == ei && j + dj == ej)) s = s + IN(1, i + di, j + dj);
    return s;
  }
  template <class T, class A>
  HD T ut4(const A& a, int i, int j, int ei, int ej) const {
    T s = T(0.);
    for (int dj = -1; dj <= 0; ++dj) for (int di = 0; di <= 1; ++di) if (!(i + di == ei && j + dj == ej)) s = s + IN(0, i + di, j + dj);
    return s;
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    constexpr int NONE = -1000;
    if (orect[0].has(i, j)) {      // rows j in {0,1} or {npy-1,npy}
      if (i == 2 || i == npx - 1) {          // corner system: the unknown vt(qi,qj) eliminated
        const int qi = (i == 2) ? 1 : npx - 1, qj = (j == 0) ? 0 : (j == 1) ? 2 : (j == npy - 1) ? npy - 1 : npy + 1;
        const double cu = MET(cosa_u, i, j), cv = MET(cosa_v, qi, qj);
        o[0] = (IN(2, i, j) - 0.25 * cu * (vt4<T>(a, i, j, qi, qj) + IN(3, qi, qj) - 0.25 * cv * ut4<T>(a, qi, qj, i, j))) * (1. / (1. - 0.0625 * cu * cv));
      } else if (i >= 3 && i <= npx - 2) o[0] = IN(2, i, j) - 0.25 * MET(cosa_u, i, j) * vt4<T>(a, i, j, NONE, NONE);
      else o[0] = IN(0, i, j);
    }
    if (orect[1].has(i, j)) {      // columns i in {0,1} or {npx-1,npx}
      if (j == 2 || j == npy - 1) {
        const int qj = (j == 2) ? 1 : npy - 1, qi = (i == 0) ? 0 : (i == 1) ? 2 : (i == npx - 1) ? npx - 1 : npx + 1;
        const double cv = MET(cosa_v, i, j), cu = MET(cosa_u, qi, qj);
        o[1] = (IN(3, i, j) - 0.25 * cv * (ut4<T>(a, i, j, qi, qj) + IN(2, qi, qj) - 0.25 * cu * vt4<T>(a, qi, qj, i, j))) * (1. / (1. - 0.0625 * cu * cv));
      } else if (j >= 3 && j <= npy - 2) o[1] = IN(3, i, j) - 0.25 * MET(cosa_v, i, j) * ut4<T>(a, i, j, NONE, NONE);
      else o[1] = IN(1, i, j);
    }
  }
};
// Pass C (pointwise): final winds = strip values where there are any, pass A elsewhere; Courant numbers and area
// fluxes (:2932-2968).
struct DswWindsCD {
  STAGE_COMMON("DswWindsC", 4, 6)   // in: ut_a vt_a ut_e vt_e   out: ut crx xfx vt cry yfx
  double dt;
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    for (int n = 0; n < 6; ++n) o[n] = T(0.);
    const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if (orect[0].has(i, j)) {
      T ut = (j == 0 || j == 1 || j == npy - 1 || j == npy) ? IN(2, i, j) : IN(0, i, j);
      o[0] = ut;
      if (orect[1].has(i, j)) winds_to_flux_x<T>(c, tile, i, j, dt, ut, o[1], o[2]);
    }
    if (orect[3].has(i, j)) {
      T vt = (i == 0 || i == 1 || i == npx - 1 || i == npx) ? IN(3, i, j) : IN(1, i, j);
      o[3] = vt;
      if (orect[4].has(i, j)) winds_to_flux_y<T>(c, tile, i, j, dt, vt, o[4], o[5]);
    }
  }
};
typedef Edged<DswWindsCD, false> DswWindsC;
typedef Edged<DswWindsCD, true> DswWindsCE;

struct DswRa {   // sw_core_tlm.F90:2969-2978
  STAGE_COMMON("DswRa", 2, 2)   // in: xfx yfx   out: ra_x ra_y
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    const double ar = MET(area, i, j);
    if (orect[0].has(i, j)) o[0] = ar + (IN(0, i, j) - IN(0, i + 1, j));
    if (orect[1].has(i, j)) o[1] = ar + (IN(1, i, j) - IN(1, i, j + 1));
  }
};

// ---- fv_tp_2d building blocks (tp_core_tlm.F90:83-236) ----
enum HordSel { HORD_MT = 0, HORD_VT, HORD_TM, HORD_DP, HORD_TR, HORD_TM_G };
// the trajectory scheme of the same transport (== hord_of unless the schemes are split)
HD int hord_traj_of(const LevelParams& l, int sel) {
  return sel == HORD_MT ? l.hord_mt_t : sel == HORD_VT ? l.hord_vt_t : sel == HORD_TM ? l.hord_tm_t : sel == HORD_DP ? l.hord_dp_t : sel == HORD_TM_G ? l.hord_tm_g_t : l.hord_tr_t;
}
HD int hord_of(const LevelParams& l, int sel) {
  return sel == HORD_MT ? l.hord_mt : sel == HORD_VT ? l.hord_vt : sel == HORD_TM ? l.hord_tm : sel == HORD_DP ? l.hord_dp : sel == HORD_TM_G ? l.hord_tm_g : l.hord_tr;
}
// Line accessors by absolute cell index along x / y; corner-halo points are read through the
// copy_corners view of the sweep direction (cdir = 1 for x sweeps, 2 for y sweeps, 0: as stored).
template <class A, int M>
struct LineX { const A& a; const Geom& g; int j, cdir; HD auto operator()(int i) const { int ii = i, jj = j; if (cdir) corner_map(g, cdir, ii, jj); return a.template in<M>(ii, jj); } };
template <class A, int M>
struct LineY { const A& a; const Geom& g; int i, cdir; HD auto operator()(int j) const { int ii = i, jj = j; if (cdir) corner_map(g, cdir, ii, jj); return a.template in<M>(ii, jj); } };
struct MetX { const double* p; const Ctx& c; int tile, j; HD double operator()(int i) const { return p[c.mi(tile, i, j)]; } };
struct MetY { const double* p; const Ctx& c; int tile, i; HD double operator()(int j) const { return p[c.mi(tile, i, j)]; } };

// EDGE = false: no face edge within reach of the outputs (standard 4-point edge values, no corner views) — the bulk
// launch; EDGE = true: the strips next to the face edges (and the single-tile periodic case never uses it).
template <bool EDGE>
struct TpPpmX_ {
  static constexpr bool LDS_FW = !EDGE, LDS_AD = !EDGE;     // bulk launch: tile + halo staged through LDS (exec.h)
  static constexpr int LDS_BY = 4;
  STAGE_BASE(EDGE ? "TpPpmXe" : "TpPpmX", 2, 1)   // in: q crx   out: flux
  STAGE_DEFAULTS_ON
  int hsel; int cdir = 0;        // cdir = 1: the inner sweep, which covers the halo rows (copy_corners(q,1), :162-171)
  HD static constexpr Box box(int M) { return M == 0 ? Box{-3, 2, 0, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  static constexpr int NALIAS = EDGE ? 1 : 0;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int M, int i, int j, int, int& ai, int& aj) const { return EDGE && M == 0 && cdir && corner_alias(c.g, cdir, i, j, ai, aj); }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    LineX<A, 0> q{a, c.g, j, EDGE ? cdir : 0};
    MetX da{c.m.dxa, c, tile, j};
    o[0] = ppm_flux<T>(hord_of(c.lev[k - 1], hsel), EDGE, i, c.g.nx + 1, q, da, IN(1, i, j));
  }
};
template <bool EDGE>
struct TpPpmY_ {
  static constexpr bool LDS_FW = !EDGE, LDS_AD = !EDGE;     // bulk launch: tile + halo staged through LDS (exec.h)
  static constexpr int LDS_BY = 16;
  STAGE_BASE(EDGE ? "TpPpmYe" : "TpPpmY", 2, 1)   // in: q cry   out: flux
  STAGE_DEFAULTS_ON
  int hsel; int cdir = 0;        // cdir = 2: the inner sweep, which covers the halo columns (copy_corners(q,2), :138-147)
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -3, 2, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  static constexpr int NALIAS = EDGE ? 1 : 0;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int M, int i, int j, int, int& ai, int& aj) const { return EDGE && M == 0 && cdir && corner_alias(c.g, cdir, i, j, ai, aj); }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    LineY<A, 0> q{a, c.g, i, EDGE ? cdir : 0};
    MetY da{c.m.dya, c, tile, i};
    o[0] = ppm_flux<T>(hord_of(c.lev[k - 1], hsel), EDGE, j, c.g.ny + 1, q, da, IN(1, i, j));
  }
};
struct TpQi {   // tp_core_tlm.F90:149-159
  STAGE_COMMON("TpQi", 4, 1)   // in: q fy2 yfx ra_y   out: q_i
  HD static constexpr Box box(int M) { return (M == 1 || M == 2) ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T f0 = IN(2, i, j) * IN(1, i, j), f1 = IN(2, i, j + 1) * IN(1, i, j + 1);
    o[0] = (IN(0, i, j) * MET(area, i, j) + f0 - f1) / IN(3, i, j);
  }
};
struct TpQj {   // tp_core_tlm.F90:173-181
  STAGE_COMMON("TpQj", 4, 1)   // in: q fx2 xfx ra_x   out: q_j
  HD static constexpr Box box(int M) { return (M == 1 || M == 2) ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T f0 = IN(2, i, j) * IN(1, i, j), f1 = IN(2, i + 1, j) * IN(1, i + 1, j);
    o[0] = (IN(0, i, j) * MET(area, i, j) + f0 - f1) / IN(3, i, j);
  }
};
// del-4 inner Laplacian of deln_flux / del6_vt_flux (nord = 1): d2 after one pass, the corner halo of
// the damped field read through copy_corners' x-view for the x-differences and y-view for the
// y-differences (tp_core_tlm.F90:1970-2008, sw_core_tlm.F90:3747-3776).
template <bool EDGE, int M, class T, class A>
HD T lap_corner(const A& a, const Ctx& c, int tile, int i, int j) {
  auto rx = [&](int ii, int jj) -> T { if (EDGE) corner_map(c.g, 1, ii, jj); return a.template in<M>(ii, jj); };
  auto ry = [&](int ii, int jj) -> T { if (EDGE) corner_map(c.g, 2, ii, jj); return a.template in<M>(ii, jj); };
  T fxa = MET(del6_v, i, j) * (rx(i - 1, j) - rx(i, j));
  T fxb = MET(del6_v, i + 1, j) * (rx(i, j) - rx(i + 1, j));
  T fya = MET(del6_u, i, j) * (ry(i, j - 1) - ry(i, j));
  T fyb = MET(del6_u, i, j + 1) * (ry(i, j) - ry(i, j + 1));
  return (fxa - fxb + (fya - fyb)) * MET(rarea, i, j);
}
enum DampSel { DAMP_NONE = 0, DAMP_V = 1, DAMP_T = 2 };
// pert: the level's transport runs with split schemes and this is the perturbation chain: nord_v_pert / damp_vt_pert for the mass
// (sw_core_tlm.F90:1672-1679), nord_t_pert / damp_t_pert for the heat transport (:1795-1800) -- the latter pair is the former as it
// stands BEFORE the perturbation sponge rules override it (dyn_core_tlm.F90:856-859 vs :913-917)
HD void damp_of(const LevelParams& l, int sel, int& nord, double& damp_c, bool pert = false) {
  if (sel != DAMP_V && sel != DAMP_T) { nord = -1; damp_c = 0.; }
  else if (pert) { if (sel == DAMP_V) { nord = l.nord_v_pert; damp_c = l.damp_vt_pert; } else { nord = l.nord_t_pert; damp_c = l.damp_t_pert; } }
  else if (sel == DAMP_V) { nord = l.nord_v; damp_c = l.damp_vt; }
  else { nord = l.nord_t; damp_c = l.damp_t; }
}
// (damp_c da_min)^(nord+1), tp_core_tlm.F90:206, :228
HD double damp_pow(double x, int nord) { double r = x; for (int n = 0; n < nord; ++n) r *= x; return r; }
// Does the level run this transport twice -- the tangent / adjoint with the perturbation's scheme and damping, the values again with the
// trajectory's?  split_hord: the schemes differ (sw_core_tlm.F90:1664-1682); split_damp: the reference takes the two-call branch for the
// mass and heat transports whatever the schemes (:1664, :1787) -- the second call changes the values only where the damping pairs differ,
// so only then is it made here.
HD bool level_split(const LevelParams& l, int hsel, int dsel = DAMP_NONE) {
  if (hord_traj_of(l, hsel) != hord_of(l, hsel)) return true;
  if (!l.split_damp || (dsel != DAMP_V && dsel != DAMP_T)) return false;
  int n0, n1; double d0, d1; damp_of(l, dsel, n0, d0, false); damp_of(l, dsel, n1, d1, true);
  const bool on0 = d0 > 1.e-4, on1 = d1 > 1.e-4;
  return on0 != on1 || (on0 && (n0 != n1 || d0 != d1));
}
struct TpD2D {
  static constexpr bool LDS_FW_OK = true;
  STAGE_BASE("TpD2", 1, 1)   // in: q   out: d2b  (is-1..ie+1, js-1..je+1); zero where the level does not use nord=1
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  static constexpr int NALIAS = 2;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int, int i, int j, int n, int& ai, int& aj) const { return corner_alias(c.g, n + 1, i, j, ai, aj); }
  int dsel; int use_mass;
  int hsel = -1, traj = 0;      // the transport's scheme selector; traj: the Laplacian of the values-only trajectory pass of a split level
  int pass = 0;                 // 1: the first of the two Laplacians of a nord = 2 level (trajectory pass), written to a scratch array
  HD static constexpr Box box(int) { return Box{-1, 1, -1, 1, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const bool split = hsel >= 0 && level_split(c.lev[k - 1], hsel, dsel);
    int nord; double dc; damp_of(c.lev[k - 1], dsel, nord, dc, split && !traj);
    o[0] = T(0.);
    if (traj && !split) return;
    if (nord < 1 || !(dc > 1.e-4)) return;
    // nord = 2 (a trajectory pass only): this is the second Laplacian, of the first one's result (in[0] = TpD2D with pass = 1), sign as
    // in tp_core_tlm.F90:2019-2031
    if (pass == 1) { if (nord == 2) o[0] = lap_corner<EDGE, 0, T>(a, c, tile, i, j); return; }
    const double damp = use_mass ? 1.0 : damp_pow(dc * c.m.da_min, nord);
    o[0] = (nord == 2 ? -damp : damp) * lap_corner<EDGE, 0, T>(a, c, tile, i, j);
  }
};
typedef Edged<TpD2D, false> TpD2;
typedef Edged<TpD2D, true> TpD2E;
// flux averaging + damping fluxes (tp_core_tlm.F90:187-234, deln_flux :1918-2043)
struct TpFlux {
  STAGE_BASE("TpFlux", 9, 2)   // in: fx_o fx2 mx fy_o fy2 my q d2b mass   out: fx fy
  STAGE_NO_ALIAS
  int dsel; int use_mass; int hsel = -1;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 6 || !(di == -1 && dj == -1); }
  HD static constexpr unsigned wants(int M) { return M < 3 ? 0x1u : M < 6 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return (M == 6 || M == 7 || M == 8) ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    int nord; double dc; damp_of(c.lev[k - 1], dsel, nord, dc, hsel >= 0 && level_split(c.lev[k - 1], hsel, dsel));
    const bool dmp = (dsel != DAMP_NONE) && (dc > 1.e-4);
    double damp = 0.;
    if (dmp) damp = damp_pow(dc * c.m.da_min, nord);
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T f = 0.5 * (IN(0, i, j) + IN(1, i, j)) * IN(2, i, j);
      if (dmp) {
        T f2;
        if (nord == 0) { f2 = MET(del6_v, i, j) * (IN(6, i - 1, j) - IN(6, i, j)); if (!use_mass) f2 = damp * f2; }
        else f2 = MET(del6_v, i, j) * (IN(7, i, j) - IN(7, i - 1, j));
        if (use_mass) f = f + (0.5 * damp) * (IN(8, i - 1, j) + IN(8, i, j)) * f2;
        else f = f + f2;
      }
      o[0] = f;
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T f = 0.5 * (IN(3, i, j) + IN(4, i, j)) * IN(5, i, j);
      if (dmp) {
        T f2;
        if (nord == 0) { f2 = MET(del6_u, i, j) * (IN(6, i, j - 1) - IN(6, i, j)); if (!use_mass) f2 = damp * f2; }
        else f2 = MET(del6_u, i, j) * (IN(7, i, j) - IN(7, i, j - 1));
        if (use_mass) f = f + (0.5 * damp) * (IN(8, i, j - 1) + IN(8, i, j)) * f2;
        else f = f + f2;
      }
      o[1] = f;
    }
  }
};

// The damping part of the flux assembly alone (deln_flux, tp_core_tlm.F90:1918-2043): fx, fy <- the del-2 / del-4 fluxes of q.  Used in
// the adjoint mode only, beside the fused outer adjoint of tpfused.h (which handles 0.5 (f_outer + f_inner) m): the flux is the sum of the
// two parts, so the adjoints with respect to q, d2b and mass are exactly this stage's.
struct TpDamp {
  STAGE_BASE("TpDamp", 3, 2)   // in: q d2b mass   out: fx fy (adjoints read; never run forward)
  STAGE_NO_ALIAS
  int dsel; int use_mass; int hsel = -1;
  HD static constexpr bool uses(int, int di, int dj, int) { return !(di == -1 && dj == -1); }
  HD static constexpr unsigned wants(int) { return 0x3u; }
  HD static constexpr Box box(int) { return Box{-1, 0, -1, 0, 0, 0}; }
  // Hand-written gather adjoint (the stage only ever runs in the adjoint mode): the damping flux at a face is a metric factor times a
  // two-point difference of q (nord 0) or of the damping Laplacian, optionally times the mean mass of the two cells -- each input adjoint
  // is a sum over the four faces of its cell.  Store rules of exec.h body_ad_joint.
  static constexpr bool HAND_AD = true;
  HD void hand_ad(const Ctx& c, const Rect& R, int i, int j, int z) const {
    const int nk = in[0].nk;
    if (z >= c.g.ntile * nk) return;
    const int tile = z / nk, k = 1 + z % nk;
    if (!wmask && (i < R.i0 - 1 || i > R.i1 || j < R.j0 - 1 || j > R.j1)) return;
    const size_t pb = (size_t)(tile * nk + k - 1) * c.g.plane;
    double acc[3] = {0., 0., 0.};
    if (k >= k0 && k <= k1) {
      int nord; double dc; damp_of(c.lev[k - 1], dsel, nord, dc, hsel >= 0 && level_split(c.lev[k - 1], hsel, dsel));
      if ((dsel != DAMP_NONE) && (dc > 1.e-4)) {
        const double damp = damp_pow(dc * c.m.da_min, nord);
        auto tr = [&](int m, int ii, int jj) { return in[m].t[pb + c.g.idx(ii, jj)]; };
        // face (ii, jj) of direction dir (0: x-face between cells ii-1, ii; 1: y-face between jj-1, jj): sgn = +1 if the point is the face's
        // upper cell (ii / jj), -1 if it is the lower one
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int dir = f >> 1, up = f & 1;                 // up: the face on the high side of cell (i, j), where the cell is the LOWER one
          const int ii = i + (dir == 0 ? up : 0), jj = j + (dir == 1 ? up : 0);
          if (!orect[dir].has(ii, jj)) continue;
          const double fa = out[dir].p[pb + c.g.idx(ii, jj)];
          const int li = ii - (dir == 0 ? 1 : 0), lj = jj - (dir == 1 ? 1 : 0);      // the face's lower cell
          const double d6 = dir == 0 ? MET(del6_v, ii, jj) : MET(del6_u, ii, jj);
          const double mm = use_mass ? 0.5 * damp * (tr(2, li, lj) + tr(2, ii, jj)) : 1.;
          const double sgn = up ? -1. : 1.;                   // the cell is the face's upper cell (+) or its lower cell (-) in  (upper - lower)
          if (nord == 0) {           // f2 = d6 (q_lower - q_upper) [damp]; flux = mm f2 (use_mass) | f2
            const double kq = d6 * (use_mass ? mm : damp);
            acc[0] += -sgn * kq * fa;
            if (use_mass) acc[2] += 0.5 * damp * (d6 * (tr(0, li, lj) - tr(0, ii, jj))) * fa;
          } else {                   // f2 = d6 (d2_upper - d2_lower); flux = mm f2 | f2
            acc[1] += sgn * d6 * mm * fa;
            if (use_mass) acc[2] += 0.5 * damp * (d6 * (tr(1, ii, jj) - tr(1, li, lj))) * fa;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
      if (in[m].p) {
        const bool in_m = !(i < R.i0 - 1 || i > R.i1 || j < R.j0 - 1 || j > R.j1);
        double* q = &in[m].p[pb + c.g.idx(i, j)];
        if ((wmask >> m) & 1u) *q = in_m ? acc[m] : 0.0;
        else if (in_m) *q += acc[m];
      }
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    int nord; double dc; damp_of(c.lev[k - 1], dsel, nord, dc, hsel >= 0 && level_split(c.lev[k - 1], hsel, dsel));
    const bool dmp = (dsel != DAMP_NONE) && (dc > 1.e-4);
    o[0] = o[1] = T(0.);
    if (!dmp) return;
    const double damp = damp_pow(dc * c.m.da_min, nord);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T f2;
      if (nord == 0) { f2 = MET(del6_v, i, j) * (IN(0, i - 1, j) - IN(0, i, j)); if (!use_mass) f2 = damp * f2; }
      else f2 = MET(del6_v, i, j) * (IN(1, i, j) - IN(1, i - 1, j));
      o[0] = use_mass ? (0.5 * damp) * (IN(2, i - 1, j) + IN(2, i, j)) * f2 : f2;
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T f2;
      if (nord == 0) { f2 = MET(del6_u, i, j) * (IN(0, i, j - 1) - IN(0, i, j)); if (!use_mass) f2 = damp * f2; }
      else f2 = MET(del6_u, i, j) * (IN(1, i, j) - IN(1, i, j - 1));
      o[1] = use_mass ? (0.5 * damp) * (IN(2, i, j - 1) + IN(2, i, j)) * f2 : f2;
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
    T dp = IN(0, i, j);
    T p = IN(1, i, j) * dp + (IN(4, i, j) - IN(4, i + 1, j) + (IN(5, i, j) - IN(5, i, j + 1))) * ra;
    T dpn = dp + (IN(2, i, j) - IN(2, i + 1, j) + (IN(3, i, j) - IN(3, i, j + 1))) * ra;
    o[0] = dpn;
    o[1] = p / dpn;
  }
};

// B-grid advective winds for the KE fluxes (sw_core_tlm.F90:3126-3254): standard formula, mean of the
// edge-normal winds along a face edge, 2-point extrapolation from both sides across it.
struct DswKeWindsD {
  STAGE_COMMON("DswKeWinds", 4, 2)   // in: uc vc ut vt   out: vb ub
  double dt;
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, -1, 0, 0, 0} : M == 1 ? Box{-1, 0, 0, 0, 0, 0} : M == 2 ? Box{0, 0, -2, 1, 0, 0} : Box{-2, 1, 0, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const double dt5 = 0.5 * dt, dt4 = 0.25 * dt;
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    const bool ie_ = F && (i == 1 || i == npx), je_ = F && (j == 1 || j == npy);
    if (!ie_ && !je_) {
      const double cs = MET(cosa, i, j), rs = MET(rsina, i, j);
      T su = IN(0, i, j - 1) + IN(0, i, j), sv = IN(1, i - 1, j) + IN(1, i, j);
      o[0] = dt5 * (sv - su * cs) * rs;
      o[1] = dt5 * (su - sv * cs) * rs;
      return;
    }
    if (je_) o[0] = dt5 * (IN(3, i - 1, j) + IN(3, i, j));
    else o[0] = dt4 * (-IN(3, i - 2, j) + 3. * (IN(3, i - 1, j) + IN(3, i, j)) - IN(3, i + 1, j));
    if (ie_) o[1] = dt5 * (IN(2, i, j - 1) + IN(2, i, j));
    else o[1] = dt4 * (-IN(2, i, j - 2) + 3. * (IN(2, i, j - 1) + IN(2, i, j)) - IN(2, i, j + 1));
  }
};
typedef Edged<DswKeWindsD, false> DswKeWinds;
typedef Edged<DswKeWindsD, true> DswKeWindsE;
// KE = 0.5*(vb*ytp_v + ub*xtp_u), face corners from the edge-normal winds (sw_core_tlm.F90:3197-3273)
// SPLIT: the build with a trajectory hord_mt that differs from the perturbation's on some level (split_hord) -- a stage type of its
// own, so that the kernels of the unsplit configuration carry none of the monotone code
template <bool SPLIT>
struct DswKeT {
  static constexpr bool MERGE_AD_OK = !SPLIT;
  static constexpr bool LDS_FW_OK = true;
  STAGE_COMMON(SPLIT ? "DswKeS" : "DswKe", 6, 1)   // in: vb ub u v ut vt   out: ke
  double dt;
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : M == 2 ? Box{-3, 2, 0, 0, 0, 0} : M == 3 ? Box{0, 0, -3, 2, 0, 0} : M == 4 ? Box{0, 0, -1, 0, 0, 0} : Box{-1, 0, 0, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if (F && (i == 1 || i == npx) && (j == 1 || j == npy)) {   // :3258-3273
      const double dt6 = dt / 6.;
      if (i == 1 && j == 1)
        o[0] = dt6 * ((IN(4, 1, 1) + IN(4, 1, 0)) * IN(2, 1, 1) + (IN(5, 1, 1) + IN(5, 0, 1)) * IN(3, 1, 1) + (IN(4, 1, 1) + IN(5, 1, 1)) * IN(2, 0, 1));
      else if (i == npx && j == 1)
        o[0] = dt6 * ((IN(4, npx, 1) + IN(4, npx, 0)) * IN(2, npx - 1, 1) + (IN(5, npx, 1) + IN(5, npx - 1, 1)) * IN(3, npx, 1) + (IN(4, npx, 1) - IN(5, npx - 1, 1)) * IN(2, npx, 1));
      else if (i == npx)
        o[0] = dt6 * ((IN(4, npx, npy) + IN(4, npx, npy - 1)) * IN(2, npx - 1, npy) + (IN(5, npx, npy) + IN(5, npx - 1, npy)) * IN(3, npx, npy - 1) +
                      (IN(4, npx, npy - 1) + IN(5, npx - 1, npy)) * IN(2, npx, npy));
      else
        o[0] = dt6 * ((IN(4, 1, npy) + IN(4, 1, npy - 1)) * IN(2, 1, npy) + (IN(5, 1, npy) + IN(5, 0, npy)) * IN(3, 1, npy - 1) + (IN(4, 1, npy - 1) - IN(5, 1, npy)) * IN(2, 0, npy));
      return;
    }
    const int iord = c.lev[k - 1].hord_mt;
    T vb = IN(0, i, j), ub = IN(1, i, j);
    LineY<A, 3> qv{a, c.g, i, 0}; MetY ddy{c.m.dy, c, tile, i};
    T fv = tp_uv_flux<T>(iord, F, j, npy, i == 1 || i == npx, qv, ddy, vb, MET(rdy, i, j - 1), MET(rdy, i, j));
    LineX<A, 2> qu{a, c.g, j, 0}; MetX ddx{c.m.dx, c, tile, j};
    T fu = tp_uv_flux<T>(iord, F, i, npx, j == 1 || j == npy, qu, ddx, ub, MET(rdx, i - 1, j), MET(rdx, i, j));
    const int iord_t = SPLIT ? c.lev[k - 1].hord_mt_t : iord;
    if (SPLIT && iord_t != iord) {      // split_hord: the nonlinear fluxes of the trajectory scheme for the values (sw_core_tlm.F90:1987-2002, :2059-2072)
      auto qvd = [&](int jj) { return val(qv(jj)); };
      auto qud = [&](int ii) { return val(qu(ii)); };
      set_val(fv, tp_uv_flux_traj(iord_t, F, j, npy, i == 1 || i == npx, qvd, ddy, val(vb), MET(rdy, i, j - 1), MET(rdy, i, j)));
      set_val(fu, tp_uv_flux_traj(iord_t, F, i, npx, j == 1 || j == npy, qud, ddx, val(ub), MET(rdx, i - 1, j), MET(rdx, i, j)));
    }
    o[0] = 0.5 * (vb * fv + ub * fu);
  }
};
typedef DswKeT<false> DswKeD;
typedef DswKeT<true> DswKeSD;
typedef Edged<DswKeD, false> DswKe;
typedef Edged<DswKeD, true> DswKeE;
// relative and absolute vorticity (sw_core_tlm.F90:3275-3293, :3535-3540)
struct DswVort {
  STAGE_COMMON("DswVort", 2, 2)   // in: u v   out: wk vort_abs
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 1, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T w = MET(rarea, i, j) * (IN(0, i, j) * MET(dx, i, j) - IN(0, i, j + 1) * MET(dx, i, j + 1) + (IN(1, i + 1, j) * MET(dy, i + 1, j) - IN(1, i, j) * MET(dy, i, j)));
    o[0] = w;
    o[1] = w + MET(f0, i, j);
  }
};

// ---- divergence damping (compute_divergence_damping, sw_core_tlm.F90:7760-8072), nord in {0,1} ----
struct DdAD {
  STAGE_BASE("DdA", 7, 2)   // in: divgd u v ua va uc vc (uc, vc: upwind side on a face edge only)   out: da db
  STAGE_NO_ALIAS
  HD static constexpr bool uses(int M, int di, int dj, int) { return M == 0 ? !(di == 1 && dj == 1) : M == 3 ? dj == 0 : M == 4 ? di == 0 : true; }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x3u : (M == 1 || M == 4) ? 0x1u : M >= 5 ? 0x0u : 0x2u; }
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 1, 0, 1, 0, 0} : (M == 3 || M == 4) ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const int nord = c.lev[k - 1].nord_p;      // the damping whose tangent / adjoint is taken (split_damp: the perturbation's)
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    o[0] = o[1] = T(0.);
    if (nord == 0) {   // :7874-7922
      if ((a.want & 0x1u) && orect[0].has(i, j)) {
        if (F && (j == 1 || j == npy)) o[0] = (val(IN(6, i, j)) > 0) ? IN(1, i, j) * MET(dyc, i, j) * SSG(4, i, j - 1) : IN(1, i, j) * MET(dyc, i, j) * SSG(2, i, j);
        else o[0] = (IN(1, i, j) - 0.5 * (IN(4, i, j - 1) + IN(4, i, j)) * MET(cosa_v, i, j)) * MET(dyc, i, j) * MET(sina_v, i, j);
      }
      if ((a.want & 0x2u) && orect[1].has(i, j)) {
        if (F && (i == 1 || i == npx)) o[1] = (val(IN(5, i, j)) > 0) ? IN(2, i, j) * MET(dxc, i, j) * SSG(3, i - 1, j) : IN(2, i, j) * MET(dxc, i, j) * SSG(1, i, j);
        else o[1] = (IN(2, i, j) - 0.5 * (IN(3, i - 1, j) + IN(3, i, j)) * MET(cosa_u, i, j)) * MET(dxc, i, j) * MET(sina_u, i, j);
      }
    } else {           // :7976-7987, nt = 0
      if ((a.want & 0x1u) && orect[0].has(i, j)) o[0] = (IN(0, i + 1, j) - IN(0, i, j)) * MET(divg_u, i, j);
      if ((a.want & 0x2u) && orect[1].has(i, j)) o[1] = (IN(0, i, j + 1) - IN(0, i, j)) * MET(divg_v, i, j);
    }
  }
};
typedef Edged<DdAD, false> DdA;
typedef Edged<DdAD, true> DdAE;
struct DdBD {   // :7924-7937 (nord=0: delpc) / :7990-8006 (nord>0: new divg_d); one term dropped at the face corners
  STAGE_COMMON("DdB", 2, 1)   // in: da db   out: dc
  HD static constexpr Box box(int M) { return M == 0 ? Box{-1, 0, 0, 0, 0, 0} : Box{0, 0, -1, 0, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    const bool drop_lo = F && j == 1 && (i == 1 || i == npx), drop_hi = F && j == npy && (i == 1 || i == npx);
    T d = IN(0, i - 1, j) - IN(0, i, j);
    if (!drop_lo) d = d + IN(1, i, j - 1);
    if (!drop_hi) d = d - IN(1, i, j);
    o[0] = MET(rarea_c, i, j) * d;
  }
};
typedef Edged<DdBD, false> DdB;
typedef Edged<DdBD, true> DdBE;
// a2b_ord4 (a2b_edge_tlm.F90:48-542): (A) the x- and y-interpolated edge-centred values qx, qy
template <bool EDGE>
struct A2bA_ {
  static constexpr bool LDS_FW = !EDGE, LDS_AD = false; static constexpr int LDS_BY = 8;
  STAGE_BASE(EDGE ? "A2bAe" : "A2bA", 1, 2)   // in: q   out: qx qy
  STAGE_NO_ALIAS
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x3u; }
  HD static constexpr Box box(int) { return EDGE ? Box{-3, 2, -3, 2, 0, 0} : Box{-2, 1, -2, 1, 0, 0}; }
  template <class T, class Q, class D>
  HD static T edge1(const Q& q, const D& d, int e, int s) {   // qx(1) / qx(npx): e = face cell at the edge, s = +1 (west) / -1 (east); :197-203
    const double g_in = d(e + s) / d(e), g_ou = d(e - 2 * s) / d(e - s);
    return 0.5 * (((2. + g_in) * q(e) - q(e + s)) / (1. + g_in) + ((2. + g_ou) * q(e - s) - q(e - 2 * s)) / (1. + g_ou));
  }
  template <class T, class Q, class D>
  HD static T line(bool F, int m, int n1, const Q& q, const D& d) {
    auto std4 = [&](int k) -> T { return B2 * (q(k - 2) + q(k + 1)) + B1 * (q(k - 1) + q(k)); };
    if (F) {
      if (m == 1) return edge1<T>(q, d, 1, 1);
      if (m == n1) return edge1<T>(q, d, n1 - 1, -1);
      if (m == 2) { const double g_in = d(2) / d(1); return (3. * (g_in * q(1) + q(2)) - (g_in * edge1<T>(q, d, 1, 1) + std4(3))) / (2. + 2. * g_in); }
      if (m == n1 - 1) { const double g_in = d(n1 - 2) / d(n1 - 1); return (3. * (q(n1 - 2) + g_in * q(n1 - 1)) - (g_in * edge1<T>(q, d, n1 - 1, -1) + std4(n1 - 2))) / (2. + 2. * g_in); }
    }
    return std4(m);
  }
  // Hand-written gather adjoint of the bulk form: two 4-point interpolations with the constant weights B1, B2 -- q_ad is the weighted sum
  // of the four qx adjoints along i and the four qy adjoints along j (the generic gather evaluated the stage seven times per point).
  static constexpr bool HAND_AD = !EDGE;
  HD void hand_ad(const Ctx& c, const Rect& R, int i, int j, int z) const {
    const int nk = in[0].nk;
    if (z >= c.g.ntile * nk || !in[0].p) return;
    const int tile = z / nk, k = 1 + z % nk;
    const bool in_m = !(i < R.i0 - 2 || i > R.i1 + 1 || j < R.j0 - 2 || j > R.j1 + 1);
    if (!in_m && !wmask) return;
    const size_t pb = (size_t)(tile * nk + k - 1) * c.g.plane;
    double acc = 0.;
    if (in_m && k >= k0 && k <= k1) {
      const double w[4] = {B2, B1, B1, B2};        // output at offset -1, 0, +1, +2 from the point
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int m = i - 1 + d, n = j - 1 + d;
        if (orect[0].has(m, j)) acc += w[d] * out[0].p[pb + c.g.idx(m, j)];
        if (orect[1].has(i, n)) acc += w[d] * out[1].p[pb + c.g.idx(i, n)];
      }
    }
    double* q = &in[0].p[pb + c.g.idx(i, j)];
    if (wmask & 1u) *q = in_m ? acc : 0.0; else if (in_m) *q += acc;
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    constexpr bool F = EDGE;
    if (orect[0].has(i, j)) { LineX<A, 0> q{a, c.g, j, 0}; MetX d{c.m.dxa, c, tile, j}; o[0] = line<T>(F, i, c.g.nx + 1, q, d); }
    if (orect[1].has(i, j)) { LineY<A, 0> q{a, c.g, i, 0}; MetY d{c.m.dya, c, tile, i}; o[1] = line<T>(F, j, c.g.ny + 1, q, d); }
  }
};
// (B) corner values: 4-point average of qx/qy, the face-edge values and the 3-way extrapolated face corners
template <bool EDGE>
struct A2bB_ {
  STAGE_COMMON(EDGE ? "A2bBe" : "A2bB", 3, 1)   // in: qx qy q   out: qout
  HD static constexpr Box box(int M) {
    return EDGE ? (M == 0 ? Box{0, 0, -3, 2, 0, 0} : M == 1 ? Box{-3, 2, 0, 0, 0, 0} : Box{-2, 1, -2, 1, 0, 0})
                : (M == 0 ? Box{0, 0, -2, 1, 0, 0} : M == 1 ? Box{-2, 1, 0, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0});
  }
  template <class T, class A>
  HD T edge_x(const A& a, const Ctx& c, int tile, int i, int j) const {   // qout(1|npx, j): :179-187, :211-219
    const int ii = i - 1;     // cells ii, ii+1 straddle the edge
    auto q2 = [&](int jj) -> T { return (IN(2, ii, jj) * MET(dxa, ii + 1, jj) + IN(2, ii + 1, jj) * MET(dxa, ii, jj)) / (MET(dxa, ii, jj) + MET(dxa, ii + 1, jj)); };
    const double w = c.m.edge[((size_t)tile * 4 + (i == 1 ? 0 : 1)) * c.g.pj + (j - c.g.j0 + c.g.ng)];
    return w * q2(j - 1) + (1. - w) * q2(j);
  }
  template <class T, class A>
  HD T edge_y(const A& a, const Ctx& c, int tile, int i, int j) const {   // qout(i, 1|npy): :294-302, :326-334
    const int jj = j - 1;
    auto q1 = [&](int ii) -> T { return (IN(2, ii, jj) * MET(dya, ii, jj + 1) + IN(2, ii, jj + 1) * MET(dya, ii, jj)) / (MET(dya, ii, jj) + MET(dya, ii, jj + 1)); };
    const double w = c.m.edge[((size_t)tile * 4 + (j == 1 ? 2 : 3)) * c.g.pj + (i - c.g.i0 + c.g.ng)];
    return w * q1(i - 1) + (1. - w) * q1(i);
  }
  template <class T, class A>
  HD T qxx(const A& a, const Ctx& c, int tile, int i, int j) const {
    const int npy = c.g.ny + 1; const double c1 = 2. / 3., c2 = -(1. / 6.);
    auto s = [&](int jj) -> T { return A2 * (IN(0, i, jj - 2) + IN(0, i, jj + 1)) + A1 * (IN(0, i, jj - 1) + IN(0, i, jj)); };
    if (EDGE && j == 2) return c1 * (IN(0, i, 1) + IN(0, i, 2)) + c2 * (edge_y<T>(a, c, tile, i, 1) + s(3));
    if (EDGE && j == npy - 1) return c1 * (IN(0, i, npy - 2) + IN(0, i, npy - 1)) + c2 * (edge_y<T>(a, c, tile, i, npy) + s(npy - 2));
    return s(j);
  }
  template <class T, class A>
  HD T qyy(const A& a, const Ctx& c, int tile, int i, int j) const {
    const int npx = c.g.nx + 1; const double c1 = 2. / 3., c2 = -(1. / 6.);
    auto s = [&](int ii) -> T { return A2 * (IN(1, ii - 2, j) + IN(1, ii + 1, j)) + A1 * (IN(1, ii - 1, j) + IN(1, ii, j)); };
    if (EDGE && i == 2) return c1 * (IN(1, 1, j) + IN(1, 2, j)) + c2 * (edge_x<T>(a, c, tile, 1, j) + s(3));
    if (EDGE && i == npx - 1) return c1 * (IN(1, npx - 2, j) + IN(1, npx - 1, j)) + c2 * (edge_x<T>(a, c, tile, npx, j) + s(npx - 2));
    return s(i);
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    constexpr bool F = EDGE; const int npx = c.g.nx + 1, npy = c.g.ny + 1;
    if (F && (i == 1 || i == npx) && (j == 1 || j == npy)) {   // :101-139: mean of three extrapolations
      const int cn = (j == 1) ? (i == 1 ? 0 : 1) : (i == 1 ? 3 : 2);
      const double* ec = c.m.ecorner + ((size_t)tile * 4 + cn) * 3;
      const int si = (i == 1) ? 1 : -1, sj = (j == 1) ? 1 : -1;           // direction into the face
      const int ci = (i == 1) ? 1 : npx - 1, cj = (j == 1) ? 1 : npy - 1;   // the face cell at the corner
      T r1 = IN(2, ci, cj) + ec[0] * (IN(2, ci, cj) - IN(2, ci + si, cj + sj));
      T q2a, q2b, q3a, q3b;   // the cells across the two edges and their diagonal successors (order of the reference)
      if (cn == 0)      { q2a = IN(2, 0, 1); q2b = IN(2, -1, 2); q3a = IN(2, 1, 0); q3b = IN(2, 2, -1); }
      else if (cn == 1) { q2a = IN(2, npx - 1, 0); q2b = IN(2, npx - 2, -1); q3a = IN(2, npx, 1); q3b = IN(2, npx + 1, 2); }
      else if (cn == 2) { q2a = IN(2, npx, npy - 1); q2b = IN(2, npx + 1, npy - 2); q3a = IN(2, npx - 1, npy); q3b = IN(2, npx - 2, npy + 1); }
      else              { q2a = IN(2, 0, npy - 1); q2b = IN(2, -1, npy - 2); q3a = IN(2, 1, npy); q3b = IN(2, 2, npy + 1); }
      T r2 = q2a + ec[1] * (q2a - q2b), r3 = q3a + ec[2] * (q3a - q3b);
      o[0] = (r1 + r2 + r3) * (1. / 3.);
      return;
    }
    if (F && (i == 1 || i == npx)) { o[0] = edge_x<T>(a, c, tile, i, j); return; }
    if (F && (j == 1 || j == npy)) { o[0] = edge_y<T>(a, c, tile, i, j); return; }
    o[0] = 0.5 * (qxx<T>(a, c, tile, i, j) + qyy<T>(a, c, tile, i, j));
  }
};
struct DdC {   // Smagorinsky-type coefficient and damping term added to KE (:7938-7956, :8023-8070)
  STAGE_COMMON("DdC", 4, 1)   // in: ke dc divgd vort_b   out: ke2
  double dt;
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const LevelParams& l = c.lev[k - 1];
    const double absdt = dt >= 0. ? dt : -dt, dddmp = l.dddmp_p, d4_bg = l.d4_bg_p, d2_divg = l.d2_divg_p;
    T ke = IN(0, i, j), dcv = IN(1, i, j);
    if (l.nord_p == 0) {
      T x = dcv * dt;
      T abs2 = (val(x) >= 0.) ? x : -x;
      T y3 = dddmp * abs2;
      T y1 = (0.20 > val(y3)) ? y3 : T(0.20);
      T mx = (d2_divg < val(y1)) ? y1 : T(d2_divg);
      o[0] = ke + (c.m.da_min_c * mx) * dcv;
    } else {
      T delpc = IN(2, i, j);
      T vs = T(0.);
      if (!(dddmp < 1.e-5)) { T vb = IN(3, i, j); vs = absdt * dsqrt(delpc * delpc + vb * vb); }
      T y2 = (0.20 > dddmp * val(vs)) ? dddmp * vs : T(0.20);
      T mx = (d2_divg < val(y2)) ? y2 : T(d2_divg);
      const double dd8 = damp_pow(c.m.da_min_c * d4_bg, l.nord_p);
      o[0] = ke + ((c.m.da_min_c * mx) * delpc + dd8 * dcv);
    }
  }
};
// del6_vt_flux inner Laplacian without the damp factor (sw_core_tlm.F90:3747-3776, nord_v = 1)
struct Del6AD {
  static constexpr bool LDS_FW_OK = true;
  STAGE_BASE("Del6A", 1, 1)   // in: wk   out: d2b
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  static constexpr int NALIAS = 2;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int, int i, int j, int n, int& ai, int& aj) const { return corner_alias(c.g, n + 1, i, j, ai, aj); }
  HD static constexpr Box box(int) { return Box{-1, 1, -1, 1, 0, 0}; }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const { o[0] = lap_corner<EDGE, 0, T>(a, c, tile, i, j); }
};
typedef Edged<Del6AD, false> Del6A;
typedef Edged<Del6AD, true> Del6AE;
// momentum update (sw_core_tlm.F90:3555-3564) + vorticity-damping fluxes, trajectory and
// perturbation coefficients kept apart (sw_core_tlm.F90:2436-2452, :2502-2530)
// N2: the build whose trajectory runs its vorticity damping with nord_v = 2 on some level (split_damp with nord >= 2): the twice-applied
// Laplacian comes in as one more, values-only, input (dampt.h LapTPass)
template <bool N2>
struct DswUpdateUVT {
  STAGE_BASE(N2 ? "DswUpdateUV2" : "DswUpdateUV", N2 ? 8 : 7, 2)   // in: u v ke2 fxv fyv wk d2b [d2bb]   out: u_n v_n
  STAGE_NO_ALIAS
  HD static constexpr bool uses(int M, int di, int dj, int) { return M == 2 ? !(di == 1 && dj == 1) : (M >= 5) ? !(di == -1 && dj == -1) : true; }
  HD static constexpr unsigned wants(int M) { return M == 7 ? 0x0u : (M == 0 || M == 4) ? 0x1u : (M == 1 || M == 3) ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M == 2 ? Box{0, 1, 0, 1, 0, 0} : (M >= 5) ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  HD static double pw(double x, int n) { double r = x; for (int m = 0; m < n; ++m) r *= x; return r; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const LevelParams& l = c.lev[k - 1];
    const bool dt_ = l.damp_vt > 1.e-5, dp_ = l.damp_vt_pert > 1.e-5;
    const double d4t = dt_ ? pw(l.damp_vt * c.m.da_min_c, l.nord_v) : 0., d4p = dp_ ? pw(l.damp_vt_pert * c.m.da_min_c, l.nord_v_pert) : 0.;
    T ke = IN(2, i, j);
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T e0 = T(0.), e1 = T(0.);
      if ((dt_ && l.nord_v == 0) || (dp_ && l.nord_v_pert == 0)) e0 = MET(del6_u, i, j) * (IN(5, i, j - 1) - IN(5, i, j));
      if ((dt_ && l.nord_v == 1) || (dp_ && l.nord_v_pert == 1)) e1 = MET(del6_u, i, j) * (IN(6, i, j) - IN(6, i, j - 1));
      T vt_t = d4t * (l.nord_v == 0 ? e0 : e1), vt_p = d4p * (l.nord_v_pert == 0 ? e0 : e1);
      if constexpr (N2) if (dt_ && l.nord_v == 2) vt_t = T(d4t * (MET(del6_u, i, j) * (val(IN(7, i, j)) - val(IN(7, i, j - 1)))));
      o[0] = IN(0, i, j) * MET(dx, i, j) + (ke - IN(2, i + 1, j)) + IN(4, i, j) + combine(vt_t, vt_p);
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T e0 = T(0.), e1 = T(0.);
      if ((dt_ && l.nord_v == 0) || (dp_ && l.nord_v_pert == 0)) e0 = MET(del6_v, i, j) * (IN(5, i - 1, j) - IN(5, i, j));
      if ((dt_ && l.nord_v == 1) || (dp_ && l.nord_v_pert == 1)) e1 = MET(del6_v, i, j) * (IN(6, i, j) - IN(6, i - 1, j));
      T ut_t = d4t * (l.nord_v == 0 ? e0 : e1), ut_p = d4p * (l.nord_v_pert == 0 ? e0 : e1);
      if constexpr (N2) if (dt_ && l.nord_v == 2) ut_t = T(d4t * (MET(del6_v, i, j) * (val(IN(7, i, j)) - val(IN(7, i - 1, j)))));
      o[1] = IN(1, i, j) * MET(dy, i, j) + (ke - IN(2, i, j + 1)) - IN(3, i, j) - combine(ut_t, ut_p);
    }
  }
};
typedef DswUpdateUVT<false> DswUpdateUV;

// ===================================================================== tracer_2d (fv_tracer2d_tlm.F90:1148-1446)
// Sub-cycling: Courant numbers and mass fluxes of a level are divided by the number of sub-steps it takes
// (fv_tracer2d_tlm.F90:1318-1345; the count comes from the trajectory only).  Scaled copies: the accumulators stay intact.
struct TrScale {
  STAGE_COMMON("TrScale", 4, 4)   // in: cx cy mfx mfy   out: cxs cys mfxs mfys
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const double f = c.lev[k - 1].tr_frac;
    o[0] = o[1] = o[2] = o[3] = T(0.);
    if (orect[0].has(i, j)) o[0] = IN(0, i, j) * f;
    if (orect[1].has(i, j)) o[1] = IN(1, i, j) * f;
    if (orect[2].has(i, j)) o[2] = IN(2, i, j) * f;
    if (orect[3].has(i, j)) o[3] = IN(3, i, j) * f;
  }
};
struct TrFlux {   // accumulated Courant numbers -> area fluxes (:1226-1247)
  STAGE_COMMON("TrFlux", 2, 2)   // in: cx cy   out: xfx yfx
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if (orect[0].has(i, j)) {
      T cx = IN(0, i, j);
      o[0] = (val(cx) > 0.) ? cx * MET(dxa, i - 1, j) * MET(dy, i, j) * SSG(3, i - 1, j) : cx * MET(dxa, i, j) * MET(dy, i, j) * SSG(1, i, j);
    }
    if (orect[1].has(i, j)) {
      T cy = IN(1, i, j);
      o[1] = (val(cy) > 0.) ? cy * MET(dya, i, j - 1) * MET(dx, i, j) * SSG(4, i, j - 1) : cy * MET(dya, i, j) * MET(dx, i, j) * SSG(2, i, j);
    }
  }
};
struct TrDp2Ra {   // dp2 and the flux-form areas (:1375-1392)
  STAGE_COMMON("TrDp2Ra", 5, 3)   // in: dp1 mfx mfy xfx yfx   out: dp2 ra_x ra_y
  HD static constexpr Box box(int M) { return M == 0 ? Box{0, 0, 0, 0, 0, 0} : (M == 1 || M == 3) ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = o[2] = T(0.);
    const double ar = MET(area, i, j);
    if (orect[0].has(i, j)) o[0] = IN(0, i, j) + (IN(1, i, j) - IN(1, i + 1, j) + (IN(2, i, j) - IN(2, i, j + 1))) * MET(rarea, i, j);
    if (orect[1].has(i, j)) o[1] = ar + (IN(3, i, j) - IN(3, i + 1, j));
    if (orect[2].has(i, j)) o[2] = ar + (IN(4, i, j) - IN(4, i, j + 1));
  }
};
struct TrUpdate {   // q update (:1423-1430)
  STAGE_COMMON("TrUpdate", 5, 1)   // in: q dp1 dp2 fx fy   out: q_o
  HD static constexpr Box box(int M) { return M == 3 ? Box{0, 1, 0, 0, 0, 0} : M == 4 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    if (c.tr_it > c.lev[k - 1].tr_ksplt) { o[0] = IN(0, i, j); return; }   // this level has finished its sub-steps (:1361)
    o[0] = (IN(0, i, j) * IN(1, i, j) + (IN(3, i, j) - IN(3, i + 1, j) + (IN(4, i, j) - IN(4, i, j + 1))) * MET(rarea, i, j)) / IN(2, i, j);
  }
};
// pt <-> virtual potential temperature at the start of fv_dynamics (fv_dynamics_tlm.F90:1395-1403)
struct DynPtIn {
  STAGE_COMMON("DynPtIn", 3, 1)   // in: pt(T) qv pkz   out: pt(theta_v)
  double zvir; int has_q;
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T d = has_q ? zvir * IN(1, i, j) : T(0.);
    o[0] = IN(0, i, j) * (1. + d) / IN(2, i, j);
  }
};
// non-hydrostatic: pkz from the equation of state first (fv_dynamics_tlm.F90:443-466, moist_phys = .true.)
struct DynPtInNh {
  STAGE_COMMON("DynPtInNh", 4, 2)   // in: pt(T) qv delp delz   out: pt(theta_v) pkz
  double zvir, akap, rdg; int has_q;
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T d = has_q ? zvir * IN(1, i, j) : T(0.);
    T tv = IN(0, i, j) * (1. + d);
    T pkz = dexp(akap * dlog(rdg * IN(2, i, j) * tv / IN(3, i, j)));
    o[0] = tv / pkz; o[1] = pkz;
  }
};

// one_grad_p wind update from corner pk, gz (dyn_core_tlm.F90:4128-4157); level 1 of pk is the
// constant top value (:4068-4072).
struct OneGradP {
  STAGE_BASE("OneGradP", 4, 2)   // in: u v pk_b gz_b (npz+1)   out: u_n v_n
  STAGE_NO_ALIAS
  double dt, ptk;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 2 || !(di == 1 && dj == 1); }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x1u : M == 1 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : Box{0, 1, 0, 1, 0, 1}; }
  template <class T, class A>
  HD T pk(const A& a, int i, int j, int k, int dk) const { return (k + dk == 1) ? T(ptk) : a.template in<2>(i, j, dk); }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T p00 = pk<T>(a, i, j, k, 0), p01 = pk<T>(a, i, j, k, 1);
    T g00 = IN(3, i, j, 0), g01 = IN(3, i, j, 1);
    T wk0 = p01 - p00;
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T p10 = pk<T>(a, i + 1, j, k, 0), p11 = pk<T>(a, i + 1, j, k, 1);
      T g10 = IN(3, i + 1, j, 0), g11 = IN(3, i + 1, j, 1);
      o[0] = MET(rdx, i, j) * (IN(0, i, j) + dt / (wk0 + (p11 - p10)) * ((g01 - g10) * (p11 - p00) + (g00 - g11) * (p01 - p10)));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T p10 = pk<T>(a, i, j + 1, k, 0), p11 = pk<T>(a, i, j + 1, k, 1);
      T g10 = IN(3, i, j + 1, 0), g11 = IN(3, i, j + 1, 1);
      o[1] = MET(rdy, i, j) * (IN(1, i, j) + dt / (wk0 + (p11 - p10)) * ((g01 - g10) * (p11 - p00) + (g00 - g11) * (p01 - p10)));
    }
  }
};

// ===================================================================== non-hydrostatic stages
// c_sw: w carried by the upwind mass fluxes of delp (sw_core_tlm.F90:758-778, :812-838); corner halo of delp and w through
// fill_4corners' views like CswTransport.
struct CswTransportWD {
  STAGE_BASE("CswTransportW", 5, 1)   // in: delp w utf vtf delpc   out: wc
  HD static constexpr bool uses(int M, int di, int dj, int) { return M >= 2 || di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 1, -1, 1, 0, 0} : M == 2 ? Box{0, 1, 0, 0, 0, 0} : M == 3 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  static constexpr int NALIAS = 2;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int M, int i, int j, int n, int& ai, int& aj) const { return M < 2 && fill2_alias(c.g, n + 1, i, j, ai, aj); }
  template <bool EDGE, int M, class T, class A>
  HD T rd(const A& a, const Ctx& c, int dir, int i, int j) const { if (EDGE) fill2_map(c.g, dir, i, j); return IN(M, i, j); }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    T fx2[2], fy2[2];
    for (int d = 0; d < 2; ++d) {
      T ut = IN(2, i + d, j);
      const int iu = (val(ut) > 0.) ? i + d - 1 : i + d;
      fx2[d] = ut * rd<EDGE, 0, T>(a, c, 1, iu, j) * rd<EDGE, 1, T>(a, c, 1, iu, j);
      T vt = IN(3, i, j + d);
      const int ju = (val(vt) > 0.) ? j + d - 1 : j + d;
      fy2[d] = vt * rd<EDGE, 0, T>(a, c, 2, i, ju) * rd<EDGE, 1, T>(a, c, 2, i, ju);
    }
    o[0] = (rd<EDGE, 1, T>(a, c, 2, i, j) * rd<EDGE, 0, T>(a, c, 2, i, j) + (fx2[0] - fx2[1] + (fy2[0] - fy2[1])) * MET(rarea, i, j)) / IN(4, i, j);
  }
};
typedef Edged<CswTransportWD, false> CswTransportW;
// UPDATE_DZ_C, advection part (nh_utils_tlm.F90:247-366), in two stages: the C-grid area fluxes interpolated to the npz+1
// interfaces (vertical stencil only), then the upwind update of the interface heights (horizontal stencil only).
struct DzFluxC {
  STAGE_BASE("DzFluxC", 2, 2)   // in: utf vtf (npz)   out: xfz yfz (npz+1)
  STAGE_NO_ALIAS
  HD static constexpr bool uses(int, int, int, int) { return true; }
  HD static constexpr Box box(int) { return Box{0, 0, 0, 0, -2, 1}; }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x1u : 0x2u; }
  template <int M, class T, class A>
  HD T flux(const A& a, const Ctx& c, int i, int j, int k) const {   // area flux at interface k
    const int km = c.g.npz;
    auto dp0 = [&](int kk) { return c.lev[kk - 1].dp_ref; };
    if (k == 1) return IN(M, i, j, 0) + (IN(M, i, j, 0) - IN(M, i, j, 1)) * (dp0(1) / (dp0(1) + dp0(2)));
    if (k == km + 1) return IN(M, i, j, -1) + (IN(M, i, j, -1) - IN(M, i, j, -2)) * (dp0(km) / (dp0(km - 1) + dp0(km)));
    return (dp0(k) * IN(M, i, j, -1) + dp0(k - 1) * IN(M, i, j, 0)) * (1. / (dp0(k - 1) + dp0(k)));
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) o[0] = flux<0, T>(a, c, i, j, k);
    if ((a.want & 0x2u) && orect[1].has(i, j)) o[1] = flux<1, T>(a, c, i, j, k);
  }
};
struct UpdateDzCD {
  STAGE_BASE("UpdateDzC", 3, 1)   // in: gz xfz yfz (npz+1)   out: gz_a (npz+1)
  HD static constexpr bool uses(int M, int di, int dj, int) { return M > 0 || di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  HD static constexpr Box box(int M) { return M == 0 ? Box{-1, 1, -1, 1, 0, 0} : M == 1 ? Box{0, 1, 0, 0, 0, 0} : Box{0, 0, 0, 1, 0, 0}; }
  static constexpr int NALIAS = 2;
  HD static constexpr int alias_box(int M) { return M; }
  HD bool alias(const Ctx& c, int M, int i, int j, int n, int& ai, int& aj) const { return M == 0 && fill2_alias(c.g, n + 1, i, j, ai, aj); }
  template <bool EDGE, class T, class A>
  HD void eval_e(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    auto gx = [&](int ii, int jj) -> T { if (EDGE) fill2_map(c.g, 1, ii, jj); return IN(0, ii, jj); };
    auto gy = [&](int ii, int jj) -> T { if (EDGE) fill2_map(c.g, 2, ii, jj); return IN(0, ii, jj); };
    T xf[2], yf[2], fx[2], fy[2];
    for (int d = 0; d < 2; ++d) {
      xf[d] = IN(1, i + d, j);
      fx[d] = xf[d] * ((val(xf[d]) > 0.) ? gx(i + d - 1, j) : gx(i + d, j));
      yf[d] = IN(2, i, j + d);
      fy[d] = yf[d] * ((val(yf[d]) > 0.) ? gy(i, j + d - 1) : gy(i, j + d));
    }
    const double ar = MET(area, i, j);
    o[0] = (gy(i, j) * ar + (fx[0] - fx[1]) + (fy[0] - fy[1])) / (ar + (xf[0] - xf[1]) + (yf[0] - yf[1]));
  }
};
typedef Edged<UpdateDzCD, false> UpdateDzC;
// P_GRAD_C, non-hydrostatic (wk = delpc), dyn_core_tlm.F90:3295-3334
struct PGradCNh {
  STAGE_BASE("PGradCNh", 5, 2)   // in: pkc gz (npz+1) uc1 vc1 delpc   out: uc2 vc2
  STAGE_NO_ALIAS
  double dt2;
  HD static constexpr bool uses(int M, int di, int dj, int) { return (M >= 2 && M <= 3) || !(di == -1 && dj == -1); }
  HD static constexpr unsigned wants(int M) { return M == 2 ? 0x1u : M == 3 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 0, -1, 0, 0, 1} : M == 4 ? Box{-1, 0, -1, 0, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    T pk00 = IN(0, i, j, 0), pk01 = IN(0, i, j, 1), gz00 = IN(1, i, j, 0), gz01 = IN(1, i, j, 1), wk0 = IN(4, i, j);
    if ((a.want & 0x1u) && orect[0].has(i, j)) {
      T pkm0 = IN(0, i - 1, j, 0), pkm1 = IN(0, i - 1, j, 1), gzm0 = IN(1, i - 1, j, 0), gzm1 = IN(1, i - 1, j, 1);
      o[0] = IN(2, i, j) + dt2 * MET(rdxc, i, j) / (IN(4, i - 1, j) + wk0) * ((gzm1 - gz00) * (pk01 - pkm0) + (gzm0 - gz01) * (pkm1 - pk00));
    }
    if ((a.want & 0x2u) && orect[1].has(i, j)) {
      T pkm0 = IN(0, i, j - 1, 0), pkm1 = IN(0, i, j - 1, 1), gzm0 = IN(1, i, j - 1, 0), gzm1 = IN(1, i, j - 1, 1);
      o[1] = IN(3, i, j) + dt2 * MET(rdyc, i, j) / (IN(4, i, j - 1) + wk0) * ((gzm1 - gz00) * (pk01 - pkm0) + (gzm0 - gz01) * (pkm1 - pk00));
    }
  }
};
// del-2 / del-4 damping increment of a cell scalar (DEL6_VT_FLUX with nord <= 1 + its divergence): dw for w in d_sw
// (sw_core_tlm.F90:3020-3040, damp = (damp_w da_min_c)^(nord_w+1)), and the damping term of UPDATE_DZ_D (:655-667, damp raw).
template <class T, class A>
HD T del6_increment(const A& a, const Ctx& c, int tile, int i, int j, int nord, double damp) {
  // inputs: 0 = q, 1 = inner Laplacian of q (Del6A)
  T fxa, fxb, fya, fyb;
  if (nord == 0) {
    fxa = MET(del6_v, i, j) * (IN(0, i - 1, j) - IN(0, i, j)); fxb = MET(del6_v, i + 1, j) * (IN(0, i, j) - IN(0, i + 1, j));
    fya = MET(del6_u, i, j) * (IN(0, i, j - 1) - IN(0, i, j)); fyb = MET(del6_u, i, j + 1) * (IN(0, i, j) - IN(0, i, j + 1));
  } else {
    fxa = MET(del6_v, i, j) * (IN(1, i, j) - IN(1, i - 1, j)); fxb = MET(del6_v, i + 1, j) * (IN(1, i + 1, j) - IN(1, i, j));
    fya = MET(del6_u, i, j) * (IN(1, i, j) - IN(1, i, j - 1)); fyb = MET(del6_u, i, j + 1) * (IN(1, i, j + 1) - IN(1, i, j));
  }
  return damp * ((fxa - fxb + (fya - fyb)) * MET(rarea, i, j));
}
struct DswDw {
  STAGE_BASE("DswDw", 2, 1)   // in: w d6w   out: dw
  STAGE_NO_ALIAS
  HD static constexpr bool uses(int, int di, int dj, int) { return di == 0 || dj == 0; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  HD static constexpr Box box(int) { return Box{-1, 1, -1, 1, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const LevelParams& l = c.lev[k - 1];
    o[0] = T(0.);
    if (!(l.damp_w > 1.e-5)) return;
    double d4 = l.damp_w * c.m.da_min_c;
    if (l.nord_w == 1) d4 = d4 * d4;
    o[0] = del6_increment<T>(a, c, tile, i, j, l.nord_w, d4);
  }
};
// w after the transport: back to velocity with the new delp, plus the damping increment (sw_core_tlm.F90:3050-3055, :3305-3330)
struct DswUpdateW {
  STAGE_COMMON("DswUpdateW", 6, 1)   // in: w delp delp_n gx gy dw   out: w_n
  HD static constexpr Box box(int M) { return M == 3 ? Box{0, 1, 0, 0, 0, 0} : M == 4 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = (IN(1, i, j) * IN(0, i, j) + (IN(3, i, j) - IN(3, i + 1, j) + (IN(4, i, j) - IN(4, i, j + 1))) * MET(rarea, i, j)) / IN(2, i, j) + IN(5, i, j);
  }
};
// UPDATE_DZ_D, update of the interface heights from the transport fluxes + damping (nh_utils_tlm.F90:655-700)
struct UpdateDzD {
  STAGE_BASE("UpdateDzD", 6, 1)   // in: zh d6z fx fy ra_x ra_y (all npz+1)   out: zh_a
  STAGE_NO_ALIAS
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 2 ? (di == 0 || dj == 0) : true; }
  HD static constexpr unsigned wants(int) { return 0x1u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{-1, 1, -1, 1, 0, 0} : M == 2 ? Box{0, 1, 0, 0, 0, 0} : M == 3 ? Box{0, 0, 0, 1, 0, 0} : Box{0, 0, 0, 0, 0, 0}; }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    const LevelParams& l = c.lev[k - 1];      // entry npz+1 repeats entry npz (ndif(km+1) = ndif(km))
    const double ar = MET(area, i, j);
    T z = (IN(0, i, j) * ar + (IN(2, i, j) - IN(2, i + 1, j)) + (IN(3, i, j) - IN(3, i, j + 1))) / (IN(4, i, j) + IN(5, i, j) - ar);
    if (l.damp_vt > 1.e-5) z = z + del6_increment<T>(a, c, tile, i, j, l.nord_v, l.damp_vt);
    o[0] = z;
  }
};
// NH_P_GRAD (use_logp = .false.), dyn_core_tlm.F90:3491-3588, from the corner values of pp, pk, zh (a2b_ord4) and delp
struct NhPGrad {
  STAGE_BASE("NhPGrad", 6, 2)   // in: u v dp_b (npz)  pp_b pk_b zh_b (npz+1)   out: u_n v_n
  STAGE_NO_ALIAS
  double dt, ptk, grav;
  HD static constexpr bool uses(int M, int di, int dj, int) { return M < 2 || !(di == 1 && dj == 1); }
  HD static constexpr unsigned wants(int M) { return M == 0 ? 0x1u : M == 1 ? 0x2u : 0x3u; }
  HD static constexpr Box box(int M) { return M < 2 ? Box{0, 0, 0, 0, 0, 0} : M == 2 ? Box{0, 1, 0, 1, 0, 0} : Box{0, 1, 0, 1, 0, 1}; }
  template <class T, class A> HD T pp(const A& a, int i, int j, int k, int dk) const { return (k + dk == 1) ? T(0.) : a.template in<3>(i, j, dk); }
  template <class T, class A> HD T pk(const A& a, int i, int j, int k, int dk) const { return (k + dk == 1) ? T(ptk) : a.template in<4>(i, j, dk); }
  template <class T, class A>
  HD T grad(const A& a, int i, int j, int i2, int j2, int k) const {   // between corner (i,j) and (i2,j2)
    T g00 = grav * IN(5, i, j, 0), g01 = grav * IN(5, i, j, 1), g10 = grav * IN(5, i2, j2, 0), g11 = grav * IN(5, i2, j2, 1);
    T k00 = pk<T>(a, i, j, k, 0), k01 = pk<T>(a, i, j, k, 1), k10 = pk<T>(a, i2, j2, k, 0), k11 = pk<T>(a, i2, j2, k, 1);
    T p00 = pp<T>(a, i, j, k, 0), p01 = pp<T>(a, i, j, k, 1), p10 = pp<T>(a, i2, j2, k, 0), p11 = pp<T>(a, i2, j2, k, 1);
    T du = dt / ((k01 - k00) + (k11 - k10)) * ((g01 - g10) * (k11 - k00) + (g00 - g11) * (k01 - k10));
    return du + dt / (IN(2, i, j) + IN(2, i2, j2)) * ((g01 - g10) * (p11 - p00) + (g00 - g11) * (p01 - p10));
  }
  template <class T, class A>
  HD void eval(const A& a, const Ctx& c, int tile, int i, int j, int k, T* o) const {
    o[0] = o[1] = T(0.);
    if ((a.want & 0x1u) && orect[0].has(i, j)) o[0] = (IN(0, i, j) + grad<T>(a, i, j, i + 1, j, k)) * MET(rdx, i, j);
    if ((a.want & 0x2u) && orect[1].has(i, j)) o[1] = (IN(1, i, j) + grad<T>(a, i, j, i, j + 1, k)) * MET(rdy, i, j);
  }
};

// level classes for the joint adjoint (exec.h): pkc/gz and pk_b/gz_b carry npz+1 levels, the winds npz
constexpr int kclass_of(const PGradC*, int M) { return M < 2 ? 1 : 0; }
constexpr int kclass_of(const OneGradP*, int M) { return M < 2 ? 0 : 1; }
constexpr int kclass_of(const PGradCNh*, int M) { return M < 2 ? 1 : 0; }
constexpr int kclass_of(const NhPGrad*, int M) { return M < 3 ? 0 : 1; }

// inputs that only the face-edge formulas of a stage read (bit m = input m): not given to the bulk launch
template <class D> constexpr unsigned edge_only_inputs(const D*) { return 0u; }
constexpr unsigned edge_only_inputs(const CswKeVortD*) { return 0x30u; }
constexpr unsigned edge_only_inputs(const DswKeWindsD*) { return 0xCu; }
constexpr unsigned edge_only_inputs(const DswKeD*) { return 0x30u; }
constexpr unsigned edge_only_inputs(const DswKeSD*) { return 0x30u; }
constexpr unsigned edge_only_inputs(const DdAD*) { return 0x60u; }
constexpr unsigned edge_only_inputs(const DswWindsCD*) { return 0xCu; }

}  // namespace fv3
