// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Restatement of model_tlmadm/tp_core_tlm.F90: fv_tp_2d (:83-236 / _TLM :2123-2324), xppm
// (_TLM :2328-2492), yppm (_TLM :2496-2669), deln_flux (:1918-2043 / _TLM :2673-2837),
// copy_corners (:2046-2118).  Only the schemes the TL/AD code implements are restated:
// iord/jord in {1, 2, 333} (tp_core_tlm.F90:2393,2431,2441,2467); on double (the nonlinear routine) also the limited
// iord 3 .. 7 and the monotone 8 / 10 of tp_mono.hpp, which fv_tp_2d_split below runs for the trajectory values when the schemes are split.
#pragma once
#include "arrays.hpp"
#include "tp_mono.hpp"

namespace orc {

static const double ppm_p1 = 7. / 12., ppm_p2 = -1. / 12.;   // tp_core_tlm.F90:57-58 region (p1,p2)
static const double ppm_c1 = -2. / 14., ppm_c2 = 11. / 14., ppm_c3 = 5. / 14.;

// the monotone schemes exist for the nonlinear routine only (never differentiated by the reference)
template <class T> struct Mono {
  static bool x(Arr2<T>&, const Arr2<T>&, const Arr2<T>&, int, int, int, int, int, const Bounds&, const Grid&) { std::fprintf(stderr, "oracle: monotone xppm on a differentiated scalar\n"); std::abort(); }
  static bool y(Arr2<T>&, const Arr2<T>&, const Arr2<T>&, int, int, int, int, int, const Bounds&, const Grid&) { std::fprintf(stderr, "oracle: monotone yppm on a differentiated scalar\n"); std::abort(); }
};
template <> struct Mono<double> {
  static bool x(Arr2<double>& f, const Arr2<double>& q, const Arr2<double>& c, int iord, int is, int ie, int j0, int j1, const Bounds& bd, const Grid& g) { xppm_mono(f, q, c, iord, is, ie, j0, j1, bd, g); return true; }
  static bool y(Arr2<double>& f, const Arr2<double>& q, const Arr2<double>& c, int jord, int i0, int i1, int js, int je, const Bounds& bd, const Grid& g) { yppm_mono(f, q, c, jord, i0, i1, js, je, bd, g); return true; }
};

// xppm: 1-D flux in x along rows jfirst..jlast.  c and flux live on is..ie+1.
// tp_core_tlm.F90:2328-2492, with the one-sided edge values at cube-face edges (:2402-2429).
// copy_corners, tp_core_tlm.F90:2046-2118 (no-op unless the tile owns a cube corner).
template <class T>
void copy_corners(Arr2<T>& q, int dir, const Bounds& bd) {
  const int ng = bd.ng, npx = bd.npx, npy = bd.npy;
  if (dir == 1) {
    if (bd.sw_corner) for (int j = 1 - ng; j <= 0; ++j) for (int i = 1 - ng; i <= 0; ++i) q(i, j) = q(j, 1 - i);
    if (bd.se_corner) for (int j = 1 - ng; j <= 0; ++j) for (int i = npx; i <= npx + ng - 1; ++i) q(i, j) = q(npy - j, i - npx + 1);
    if (bd.ne_corner) for (int j = npy; j <= npy + ng - 1; ++j) for (int i = npx; i <= npx + ng - 1; ++i) q(i, j) = q(j, 2 * npx - 1 - i);
    if (bd.nw_corner) for (int j = npy; j <= npy + ng - 1; ++j) for (int i = 1 - ng; i <= 0; ++i) q(i, j) = q(npy - j, i - 1 + npx);
  } else {
    if (bd.sw_corner) for (int j = 1 - ng; j <= 0; ++j) for (int i = 1 - ng; i <= 0; ++i) q(i, j) = q(1 - j, i);
    if (bd.se_corner) for (int j = 1 - ng; j <= 0; ++j) for (int i = npx; i <= npx + ng - 1; ++i) q(i, j) = q(npy + j - 1, npx - i);
    if (bd.ne_corner) for (int j = npy; j <= npy + ng - 1; ++j) for (int i = npx; i <= npx + ng - 1; ++i) q(i, j) = q(2 * npy - 1 - j, i);
    if (bd.nw_corner) for (int j = npy; j <= npy + ng - 1; ++j) for (int i = 1 - ng; i <= 0; ++i) q(i, j) = q(j + 1 - npx, npy - i);
  }
}

template <class T>
void xppm(Arr2<T>& flux, const Arr2<T>& q, const Arr2<T>& c, int iord, int is, int ie, int jfirst, int jlast,
          const Bounds& bd, const Grid& g) {
  if (iord >= 3 && iord <= 13) { Mono<T>::x(flux, q, c, iord, is, ie, jfirst, jlast, bd, g); return; }
  assert(iord == 1 || iord == 2 || iord == 333);
  const int npx = bd.npx;
  int is1 = is - 1, ie3 = ie + 2;
  if (bd.any_edge()) { is1 = std::max(3, is - 1); ie3 = std::min(npx - 2, ie + 2); }   // :2363-2378
  std::vector<T> al(ie + 2 - (is - 1) + 2);
  auto AL = [&](int i) -> T& { return al[i - (is - 1)]; };
  for (int j = jfirst; j <= jlast; ++j) {
    for (int i = is1; i <= ie3; ++i)   // :2397-2399
      AL(i) = ppm_p1 * (q(i - 1, j) + q(i, j)) + ppm_p2 * (q(i - 2, j) + q(i + 1, j));
    if (bd.edge_w) {                   // is == 1, :2402-2413
      AL(0) = ppm_c1 * q(-2, j) + ppm_c2 * q(-1, j) + ppm_c3 * q(0, j);
      AL(1) = 0.5 * (((2. * g.dxa(0, j) + g.dxa(-1, j)) * q(0, j) - g.dxa(0, j) * q(-1, j)) / (g.dxa(-1, j) + g.dxa(0, j)) +
                     ((2. * g.dxa(1, j) + g.dxa(2, j)) * q(1, j) - g.dxa(1, j) * q(2, j)) / (g.dxa(1, j) + g.dxa(2, j)));
      AL(2) = ppm_c3 * q(1, j) + ppm_c2 * q(2, j) + ppm_c1 * q(3, j);
    }
    if (bd.edge_e) {                   // ie+1 == npx, :2414-2429
      AL(npx - 1) = ppm_c1 * q(npx - 3, j) + ppm_c2 * q(npx - 2, j) + ppm_c3 * q(npx - 1, j);
      AL(npx) = 0.5 * (((2. * g.dxa(npx - 1, j) + g.dxa(npx - 2, j)) * q(npx - 1, j) - g.dxa(npx - 1, j) * q(npx - 2, j)) /
                           (g.dxa(npx - 2, j) + g.dxa(npx - 1, j)) +
                       ((2. * g.dxa(npx, j) + g.dxa(npx + 1, j)) * q(npx, j) - g.dxa(npx, j) * q(npx + 1, j)) /
                           (g.dxa(npx, j) + g.dxa(npx + 1, j)));
      AL(npx + 1) = ppm_c3 * q(npx, j) + ppm_c2 * q(npx + 1, j) + ppm_c1 * q(npx + 2, j);
    }
    if (iord == 1) {                   // :2431-2440
      for (int i = is; i <= ie + 1; ++i) flux(i, j) = (val(c(i, j)) > 0.) ? q(i - 1, j) : q(i, j);
    } else if (iord == 2) {            // :2441-2466
      for (int i = is; i <= ie + 1; ++i) {
        T xt = c(i, j);
        if (val(xt) > 0.) {
          T qtmp = q(i - 1, j);
          flux(i, j) = qtmp + (1. - xt) * (AL(i) - qtmp - xt * (AL(i - 1) + AL(i) - (qtmp + qtmp)));
        } else {
          T qtmp = q(i, j);
          flux(i, j) = qtmp + (1. + xt) * (AL(i) - qtmp + xt * (AL(i) + AL(i + 1) - (qtmp + qtmp)));
        }
      }
    } else {                           // iord == 333, :2467-2487
      for (int i = is; i <= ie + 1; ++i) {
        T xt = c(i, j);
        if (val(xt) > 0.)
          flux(i, j) = (2.0 * q(i, j) + 5.0 * q(i - 1, j) - q(i - 2, j)) / 6.0 - 0.5 * xt * (q(i, j) - q(i - 1, j)) +
                       xt * xt / 6.0 * (q(i, j) - 2.0 * q(i - 1, j) + q(i - 2, j));
        else
          flux(i, j) = (2.0 * q(i - 1, j) + 5.0 * q(i, j) - q(i + 1, j)) / 6.0 - 0.5 * xt * (q(i, j) - q(i - 1, j)) +
                       xt * xt / 6.0 * (q(i + 1, j) - 2.0 * q(i, j) + q(i - 1, j));
      }
    }
  }
}

// yppm: 1-D flux in y for columns ifirst..ilast; c and flux on js..je+1.  tp_core_tlm.F90:2496-2669.
template <class T>
void yppm(Arr2<T>& flux, const Arr2<T>& q, const Arr2<T>& c, int jord, int ifirst, int ilast, int js, int je,
          const Bounds& bd, const Grid& g) {
  if (jord >= 3 && jord <= 13) { Mono<T>::y(flux, q, c, jord, ifirst, ilast, js, je, bd, g); return; }
  assert(jord == 1 || jord == 2 || jord == 333);
  const int npy = bd.npy;
  int js1 = js - 1, je3 = je + 2;
  if (bd.any_edge()) { js1 = std::max(3, js - 1); je3 = std::min(npy - 2, je + 2); }
  if (jord == 1) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = ifirst; i <= ilast; ++i) flux(i, j) = (val(c(i, j)) > 0.) ? q(i, j - 1) : q(i, j);
    return;
  }
  Arr2<T> al(bd);
  for (int j = js1; j <= je3; ++j)
    for (int i = ifirst; i <= ilast; ++i)
      al(i, j) = ppm_p1 * (q(i, j - 1) + q(i, j)) + ppm_p2 * (q(i, j - 2) + q(i, j + 1));
  if (bd.edge_s)     // js == 1, :2559-2573
    for (int i = ifirst; i <= ilast; ++i) {
      al(i, 0) = ppm_c1 * q(i, -2) + ppm_c2 * q(i, -1) + ppm_c3 * q(i, 0);
      al(i, 1) = 0.5 * (((2. * g.dya(i, 0) + g.dya(i, -1)) * q(i, 0) - g.dya(i, 0) * q(i, -1)) / (g.dya(i, -1) + g.dya(i, 0)) +
                        ((2. * g.dya(i, 1) + g.dya(i, 2)) * q(i, 1) - g.dya(i, 1) * q(i, 2)) / (g.dya(i, 1) + g.dya(i, 2)));
      al(i, 2) = ppm_c3 * q(i, 1) + ppm_c2 * q(i, 2) + ppm_c1 * q(i, 3);
    }
  if (bd.edge_n)     // je+1 == npy
    for (int i = ifirst; i <= ilast; ++i) {
      al(i, npy - 1) = ppm_c1 * q(i, npy - 3) + ppm_c2 * q(i, npy - 2) + ppm_c3 * q(i, npy - 1);
      al(i, npy) = 0.5 * (((2. * g.dya(i, npy - 1) + g.dya(i, npy - 2)) * q(i, npy - 1) - g.dya(i, npy - 1) * q(i, npy - 2)) /
                              (g.dya(i, npy - 2) + g.dya(i, npy - 1)) +
                          ((2. * g.dya(i, npy) + g.dya(i, npy + 1)) * q(i, npy) - g.dya(i, npy) * q(i, npy + 1)) /
                              (g.dya(i, npy) + g.dya(i, npy + 1)));
      al(i, npy + 1) = ppm_c3 * q(i, npy) + ppm_c2 * q(i, npy + 1) + ppm_c1 * q(i, npy + 2);
    }
  if (jord == 2) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = ifirst; i <= ilast; ++i) {
        T xt = c(i, j);
        if (val(xt) > 0.) {
          T qtmp = q(i, j - 1);
          flux(i, j) = qtmp + (1. - xt) * (al(i, j) - qtmp - xt * (al(i, j - 1) + al(i, j) - (qtmp + qtmp)));
        } else {
          T qtmp = q(i, j);
          flux(i, j) = qtmp + (1. + xt) * (al(i, j) - qtmp + xt * (al(i, j) + al(i, j + 1) - (qtmp + qtmp)));
        }
      }
  } else {
    for (int j = js; j <= je + 1; ++j)
      for (int i = ifirst; i <= ilast; ++i) {
        T xt = c(i, j);
        if (val(xt) > 0.)
          flux(i, j) = (2.0 * q(i, j) + 5.0 * q(i, j - 1) - q(i, j - 2)) / 6.0 - 0.5 * xt * (q(i, j) - q(i, j - 1)) +
                       xt * xt / 6.0 * (q(i, j) - 2.0 * q(i, j - 1) + q(i, j - 2));
        else
          flux(i, j) = (2.0 * q(i, j - 1) + 5.0 * q(i, j) - q(i, j + 1)) / 6.0 - 0.5 * xt * (q(i, j) - q(i, j - 1)) +
                       xt * xt / 6.0 * (q(i, j + 1) - 2.0 * q(i, j) + q(i, j - 1));
      }
  }
}

// deln_flux: del-(2*nord+2) damping fluxes added to fx,fy.  tp_core_tlm.F90:1918-2043.
// mass==nullptr: d2 = damp*q, fluxes added unweighted; else mass-weighted with damp2=0.5*damp.
template <class T>
void deln_flux(int nord, const Bounds& bd, double damp, const Arr2<T>& q, Arr2<T>& fx, Arr2<T>& fy, const Grid& g,
               const Arr2<T>* mass) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  Arr2<T> fx2(bd), fy2(bd), d2(bd);
  const int i1 = is - 1 - nord, i2 = ie + 1 + nord, j1 = js - 1 - nord, j2 = je + 1 + nord;
  for (int j = j1; j <= j2; ++j)
    for (int i = i1; i <= i2; ++i) d2(i, j) = mass ? q(i, j) : damp * q(i, j);
  if (nord > 0) copy_corners(d2, 1, bd);
  for (int j = js - nord; j <= je + nord; ++j)
    for (int i = is - nord; i <= ie + nord + 1; ++i) fx2(i, j) = g.del6_v(i, j) * (d2(i - 1, j) - d2(i, j));
  if (nord > 0) copy_corners(d2, 2, bd);
  for (int j = js - nord; j <= je + nord + 1; ++j)
    for (int i = is - nord; i <= ie + nord; ++i) fy2(i, j) = g.del6_u(i, j) * (d2(i, j - 1) - d2(i, j));
  for (int n = 1; n <= nord; ++n) {
    const int nt = nord - n;
    for (int j = js - nt - 1; j <= je + nt + 1; ++j)
      for (int i = is - nt - 1; i <= ie + nt + 1; ++i)
        d2(i, j) = (fx2(i, j) - fx2(i + 1, j) + fy2(i, j) - fy2(i, j + 1)) * g.rarea(i, j);
    copy_corners(d2, 1, bd);
    for (int j = js - nt; j <= je + nt; ++j)
      for (int i = is - nt; i <= ie + nt + 1; ++i) fx2(i, j) = g.del6_v(i, j) * (d2(i, j) - d2(i - 1, j));
    copy_corners(d2, 2, bd);
    for (int j = js - nt; j <= je + nt + 1; ++j)
      for (int i = is - nt; i <= ie + nt; ++i) fy2(i, j) = g.del6_u(i, j) * (d2(i, j) - d2(i, j - 1));
  }
  if (mass) {
    const double damp2 = 0.5 * damp;
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i) fx(i, j) = fx(i, j) + damp2 * ((*mass)(i - 1, j) + (*mass)(i, j)) * fx2(i, j);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i) fy(i, j) = fy(i, j) + damp2 * ((*mass)(i, j - 1) + (*mass)(i, j)) * fy2(i, j);
  } else {
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i) fx(i, j) = fx(i, j) + fx2(i, j);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i) fy(i, j) = fy(i, j) + fy2(i, j);
  }
}

// fv_tp_2d: Lin–Rood 2-D flux-form transport operator.  tp_core_tlm.F90:83-236 (_TLM :2123-2324).
// mfx/mfy non-null → "transport of pt and tracers" branch; mass+nord+damp_c → mass-weighted damping.
// nord<0 means "nord/damp_c not present".
template <class T>
void fv_tp_2d(const Arr2<T>& q_in, const Arr2<T>& crx, const Arr2<T>& cry, int hord, Arr2<T>& fx, Arr2<T>& fy,
              const Arr2<T>& xfx, const Arr2<T>& yfx, const Grid& g, const Bounds& bd, const Arr2<T>& ra_x,
              const Arr2<T>& ra_y, const Arr2<T>* mfx, const Arr2<T>* mfy, const Arr2<T>* mass, int nord,
              double damp_c) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  const int ord_in = (hord == 10) ? 8 : hord, ord_ou = hord;
  Arr2<T> q_i(bd), q_j(bd), fx2(bd), fy2(bd), fyy(bd);
  Arr2<T> q = q_in;                 // the reference fills q's corner halo in place (intent(inout))
  copy_corners(q, 2, bd);           // :138-143
  yppm(fy2, q, cry, ord_in, isd, ied, js, je, bd, g);
  for (int j = js; j <= je + 1; ++j)
    for (int i = isd; i <= ied; ++i) fyy(i, j) = yfx(i, j) * fy2(i, j);
  for (int j = js; j <= je; ++j)
    for (int i = isd; i <= ied; ++i) q_i(i, j) = (q(i, j) * g.area(i, j) + fyy(i, j) - fyy(i, j + 1)) / ra_y(i, j);
  xppm(fx, q_i, crx, ord_ou, is, ie, js, je, bd, g);
  copy_corners(q, 1, bd);           // :162-167
  xppm(fx2, q, crx, ord_in, is, ie, jsd, jed, bd, g);
  for (int j = jsd; j <= jed; ++j) {
    std::vector<T> fx1(ie + 2 - is + 1);
    for (int i = is; i <= ie + 1; ++i) fx1[i - is] = xfx(i, j) * fx2(i, j);
    for (int i = is; i <= ie; ++i) q_j(i, j) = (q(i, j) * g.area(i, j) + fx1[i - is] - fx1[i + 1 - is]) / ra_x(i, j);
  }
  yppm(fy, q_j, cry, ord_ou, is, ie, js, je, bd, g);
  if (mfx && mfy) {
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i) fx(i, j) = 0.5 * (fx(i, j) + fx2(i, j)) * (*mfx)(i, j);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i) fy(i, j) = 0.5 * (fy(i, j) + fy2(i, j)) * (*mfy)(i, j);
    if (nord >= 0 && mass && damp_c > 1.e-4) {
      double damp = std::pow(damp_c * g.da_min, nord + 1);
      deln_flux(nord, bd, damp, q, fx, fy, g, mass);
    }
  } else {
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i) fx(i, j) = 0.5 * (fx(i, j) + fx2(i, j)) * xfx(i, j);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i) fy(i, j) = 0.5 * (fy(i, j) + fy2(i, j)) * yfx(i, j);
    if (nord >= 0 && damp_c > 1.e-4) {
      double damp = std::pow(damp_c * g.da_min, nord + 1);
      deln_flux<T>(nord, bd, damp, q, fx, fy, g, nullptr);
    }
  }
}


// Trajectory scheme != perturbation scheme (split_hord): the reference calls FV_TP_2D_TLM with the perturbation scheme and
// damping and then the nonlinear fv_tp_2d with the trajectory scheme and damping, whose fluxes replace the VALUES
// (sw_core_tlm.F90:1664-1682 delp, :1746-1763 w, :1787-1803 pt, :2407-2420 vorticity; fv_tracer2d_tlm.F90, nh_utils_tlm.F90 likewise).
// The tangents stay those of the perturbation chain.  Equal schemes: one call with the trajectory damping, as before.
// trajectory / perturbation scheme pair of a transport call (an int converts: both the same)
struct Hord { int traj, pert; Hord(int t) : traj(t), pert(t) {} Hord(int t, int p) : traj(t), pert(p) {} };
inline Arr2<double> values_of(const Arr2<double>& a) { return a; }
template <class T> Arr2<double> values_of(const Arr2<T>& a) {
  Arr2<double> r; r.isd = a.isd; r.jsd = a.jsd; r.pi = a.pi; r.pj = a.pj; r.d.resize(a.d.size());
  for (size_t n = 0; n < a.d.size(); ++n) r.d[n] = val(a.d[n]);
  return r;
}
template <class T>
void fv_tp_2d_split(const Arr2<T>& q, const Arr2<T>& crx, const Arr2<T>& cry, int hord, int hord_pert, Arr2<T>& fx, Arr2<T>& fy,
                    const Arr2<T>& xfx, const Arr2<T>& yfx, const Grid& g, const Bounds& bd, const Arr2<T>& ra_x, const Arr2<T>& ra_y,
                    const Arr2<T>* mfx, const Arr2<T>* mfy, const Arr2<T>* mass, int nord, double damp_c, int nord_pert, double damp_c_pert,
                    bool split_damp = false) {      // split_damp: the two-call branch whatever the schemes (sw_core_tlm.F90:1664, :1787)
  if (hord == hord_pert && !split_damp) { fv_tp_2d<T>(q, crx, cry, hord, fx, fy, xfx, yfx, g, bd, ra_x, ra_y, mfx, mfy, mass, nord, damp_c); return; }
  fv_tp_2d<T>(q, crx, cry, hord_pert, fx, fy, xfx, yfx, g, bd, ra_x, ra_y, mfx, mfy, mass, nord_pert, damp_c_pert);
  const Arr2<double> qd = values_of(q), cxd = values_of(crx), cyd = values_of(cry), xd = values_of(xfx), yd = values_of(yfx), rxd = values_of(ra_x), ryd = values_of(ra_y);
  Arr2<double> mxd, myd, msd, fxd(bd), fyd(bd);
  if (mfx) mxd = values_of(*mfx);
  if (mfy) myd = values_of(*mfy);
  if (mass) msd = values_of(*mass);
  fv_tp_2d<double>(qd, cxd, cyd, hord, fxd, fyd, xd, yd, g, bd, rxd, ryd, mfx ? &mxd : nullptr, mfy ? &myd : nullptr, mass ? &msd : nullptr, nord, damp_c);
  for (int j = bd.js; j <= bd.je; ++j)
    for (int i = bd.is; i <= bd.ie + 1; ++i) set_val(fx(i, j), fxd(i, j));
  for (int j = bd.js; j <= bd.je + 1; ++j)
    for (int i = bd.is; i <= bd.ie; ++i) set_val(fy(i, j), fyd(i, j));
}

}  // namespace orc
