// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// The monotone 1-D PPM fluxes of the NONLINEAR xppm / yppm, iord = 8 and 10 (model_tlmadm/tp_core_tlm.F90:592-955 for
// x, :1339-1723 for y; pert_ppm :1853-1915).  The tangent-linear and adjoint code never differentiates them: with a
// trajectory scheme that differs from the perturbation scheme the reference runs the _TLM routine with the perturbation
// scheme and then the nonlinear routine with the trajectory scheme for the values (sw_core_tlm.F90:1664-1682) — so
// these are written on double only.  x and y share one line routine here (the two reference copies are the same
// statements with the indices swapped; the cube-edge values use dxa along x, dya along y).
#pragma once
#include "arrays.hpp"
#include <algorithm>

namespace orc {

static const double mono_r3 = 1. / 3., mono_near_zero = 1.e-25;
static const double mono_s11 = 11. / 14., mono_s14 = 4. / 7., mono_s15 = 3. / 14.;   // tp_core_tlm.F90:57

inline double f_sign(double a, double b) { return b >= 0. ? std::fabs(a) : -std::fabs(a); }   // Fortran SIGN(a, b)

// pert_ppm with iv = 1 ("standard PPM constraint"), tp_core_tlm.F90:1893-1913; iv = 0 is only reached for iord 9 / 13
inline void pert_ppm_std(int im, double* al, double* ar) {
  for (int i = 0; i < im; ++i) {
    if (al[i] * ar[i] < 0.) {
      const double da1 = al[i] - ar[i], da2 = da1 * da1, a6da = 3. * (al[i] + ar[i]) * da1;
      if (a6da < -da2) ar[i] = -(2. * al[i]);
      else if (a6da > da2) al[i] = -(2. * ar[i]);
    } else { al[i] = 0.; ar[i] = 0.; }
  }
}

// pert_ppm with iv = 0 (positive definite constraint, tp_core_tlm.F90:1867-1892): iord 9 and 13 of xppm / yppm, interior cells
inline void pert_ppm_pos(int im, const double* a0, double* al, double* ar) {
  for (int i = 0; i < im; ++i) {
    if (a0[i] <= 0.) { al[i] = 0.; ar[i] = 0.; continue; }
    const double a4 = -3. * (ar[i] + al[i]), da1 = ar[i] - al[i];
    if (std::fabs(da1) < -a4) {
      const double fmin = a0[i] + 0.25 / a4 * da1 * da1 + a4 * (1. / 12.);
      if (fmin < 0.) {
        if (ar[i] > 0. && al[i] > 0.) { ar[i] = 0.; al[i] = 0.; }
        else if (da1 > 0.) ar[i] = -2. * al[i];
        else al[i] = -2. * ar[i];
      }
    }
  }
}

// One line.  Cells are numbered as in the reference (first..last = is..ie or js..je, three halo cells each side);
// q(n), c(n) (n = first..last+1), da(n) = dxa or dya along the line; edge_lo: first == 1 on a cube edge,
// edge_hi: last + 1 == np on a cube edge; any_edge: the tile is part of a cubed-sphere face (is1 / ie1 clipping, :314-330).
template <class FQ, class FC, class FD, class FOut>
void ppm_line_mono(int iord, int first, int last, int np, bool any_edge, bool edge_lo, bool edge_hi, const FQ& q, const FC& c, const FD& da,
                   const FOut& flux) {
  assert(iord >= 8 && iord <= 13);
  int is1 = first - 1, ie1 = last + 1;
  if (any_edge) { is1 = std::max(3, first - 1); ie1 = std::min(np - 3, last + 1); }
  const int lo = first - 4, n = last - first + 10;
  std::vector<double> dm_(n, 0.), al_(n, 0.), bl_(n, 0.), br_(n, 0.), dq_(n, 0.);
  auto dm = [&](int i) -> double& { return dm_[i - lo]; };
  auto al = [&](int i) -> double& { return al_[i - lo]; };
  auto bl = [&](int i) -> double& { return bl_[i - lo]; };
  auto br = [&](int i) -> double& { return br_[i - lo]; };
  auto dq = [&](int i) -> double& { return dq_[i - lo]; };
  for (int i = first - 2; i <= last + 2; ++i) {      // :596-640
    const double xt = 0.25 * (q(i + 1) - q(i - 1));
    const double hi = std::max(std::max(q(i - 1), q(i)), q(i + 1)) - q(i), lw = q(i) - std::min(std::min(q(i - 1), q(i)), q(i + 1));
    dm(i) = f_sign(std::min(std::min(std::fabs(xt), hi), lw), xt);
  }
  for (int i = is1; i <= ie1 + 1; ++i) al(i) = 0.5 * (q(i - 1) + q(i)) + mono_r3 * (dm(i - 1) - dm(i));     // :641-642
  if (iord == 8 || iord == 11) {                     // :643-678; 11 ("2nd van Leer scheme using PPM codes", :679-715): ppm_fac = 1.5 (:35) in place of 2
    for (int i = is1; i <= ie1; ++i) {
      const double xt = (iord == 8 ? 2. : 1.5) * dm(i);
      bl(i) = -f_sign(std::min(std::fabs(xt), std::fabs(al(i) - q(i))), xt);
      br(i) = f_sign(std::min(std::fabs(xt), std::fabs(al(i + 1) - q(i))), xt);
    }
  } else {                                           // iord = 9, 10, 12, 13: Huynh's second constraint, :716-823
    for (int i = is1 - 2; i <= ie1 + 1; ++i) dq(i) = 2. * (q(i + 1) - q(i));
    for (int i = is1; i <= ie1; ++i) {
      bl(i) = al(i) - q(i);
      br(i) = al(i + 1) - q(i);
      if (std::fabs(dm(i - 1)) + std::fabs(dm(i)) + std::fabs(dm(i + 1)) < mono_near_zero) { bl(i) = 0.; br(i) = 0.; }
      else if (std::fabs(3. * (bl(i) + br(i))) > std::fabs(bl(i) - br(i))) {
        const double pmp_2 = dq(i - 1), lac_2 = pmp_2 - 0.75 * dq(i - 2);
        br(i) = std::min(std::max(0., std::max(pmp_2, lac_2)), std::max(br(i), std::min(0., std::min(pmp_2, lac_2))));
        const double pmp_1 = -dq(i), lac_1 = pmp_1 + 0.75 * dq(i + 1);
        bl(i) = std::min(std::max(0., std::max(pmp_1, lac_1)), std::max(bl(i), std::min(0., std::min(pmp_1, lac_1))));
      }
    }
  }
  if (iord == 9 || iord == 13) {                     // positive definite constraint on the cells is1 .. ie1, :826-828
    std::vector<double> a0(ie1 - is1 + 1);
    for (int i = is1; i <= ie1; ++i) a0[i - is1] = q(i);
    pert_ppm_pos(ie1 - is1 + 1, a0.data(), &bl(is1), &br(is1));
  }
  auto two_sided = [&](int e) {     // the dxa-weighted value on a cube edge between cells e-1 and e (:833-835, :892-895)
    return 0.5 * (((2. * da(e - 1) + da(e - 2)) * q(e - 1) - da(e - 1) * q(e - 2)) / (da(e - 2) + da(e - 1)) +
                  ((2. * da(e) + da(e + 1)) * q(e) - da(e) * q(e + 1)) / (da(e) + da(e + 1)));
  };
  if (any_edge) {
    if (edge_lo) {                  // :830-884
      bl(0) = mono_s14 * dm(-1) + mono_s11 * (q(-1) - q(0));
      double xt = two_sided(1);
      xt = std::max(xt, std::min(std::min(q(-1), q(0)), std::min(q(1), q(2))));
      xt = std::min(xt, std::max(std::max(q(-1), q(0)), std::max(q(1), q(2))));
      br(0) = xt - q(0);
      bl(1) = xt - q(1);
      xt = mono_s15 * q(1) + mono_s11 * q(2) - mono_s14 * dm(2);
      br(1) = xt - q(1);
      bl(2) = xt - q(2);
      br(2) = al(3) - q(2);
      pert_ppm_std(3, &bl(0), &br(0));
    }
    if (edge_hi) {                  // :886-942
      bl(np - 2) = al(np - 2) - q(np - 2);
      double xt = mono_s15 * q(np - 1) + mono_s11 * q(np - 2) + mono_s14 * dm(np - 2);
      br(np - 2) = xt - q(np - 2);
      bl(np - 1) = xt - q(np - 1);
      xt = two_sided(np);
      xt = std::max(xt, std::min(std::min(q(np - 2), q(np - 1)), std::min(q(np), q(np + 1))));
      xt = std::min(xt, std::max(std::max(q(np - 2), q(np - 1)), std::max(q(np), q(np + 1))));
      br(np - 1) = xt - q(np - 1);
      bl(np) = xt - q(np);
      br(np) = mono_s11 * (q(np + 1) - q(np)) - mono_s14 * dm(np + 1);
      pert_ppm_std(3, &bl(np - 2), &br(np - 2));
    }
  }
  for (int i = first; i <= last + 1; ++i) {          // :944-952
    const double cc = c(i);
    if (cc > 0.) flux(i, q(i - 1) + (1. - cc) * (br(i - 1) - cc * (bl(i - 1) + br(i - 1))));
    else flux(i, q(i) + (1. + cc) * (bl(i) + cc * (bl(i) + br(i))));
  }
}

template <class FQ, class FC, class FD, class FOut>
void ppm_line_low(int iord, int first, int last, int np, bool any_edge, bool edge_lo, bool edge_hi, const FQ& q, const FC& c, const FD& da, const FOut& flux);

inline void xppm_mono(Arr2<double>& flux, const Arr2<double>& q, const Arr2<double>& c, int iord, int is, int ie, int jfirst, int jlast,
                      const Bounds& bd, const Grid& g) {
  for (int j = jfirst; j <= jlast; ++j)
    if (iord >= 3 && iord <= 7)
      ppm_line_low(iord, is, ie, bd.npx, bd.any_edge(), bd.edge_w, bd.edge_e, [&](int i) { return q(i, j); }, [&](int i) { return c(i, j); },
                   [&](int i) { return g.dxa(i, j); }, [&](int i, double f) { flux(i, j) = f; });
    else
    ppm_line_mono(iord, is, ie, bd.npx, bd.any_edge(), bd.edge_w, bd.edge_e, [&](int i) { return q(i, j); }, [&](int i) { return c(i, j); },
                  [&](int i) { return g.dxa(i, j); }, [&](int i, double f) { flux(i, j) = f; });
}
inline void yppm_mono(Arr2<double>& flux, const Arr2<double>& q, const Arr2<double>& c, int jord, int ifirst, int ilast, int js, int je,
                      const Bounds& bd, const Grid& g) {
  for (int i = ifirst; i <= ilast; ++i)
    if (jord >= 3 && jord <= 7)
      ppm_line_low(jord, js, je, bd.npy, bd.any_edge(), bd.edge_s, bd.edge_n, [&](int j) { return q(i, j); }, [&](int j) { return c(i, j); },
                   [&](int j) { return g.dya(i, j); }, [&](int j, double f) { flux(i, j) = f; });
    else
    ppm_line_mono(jord, js, je, bd.npy, bd.any_edge(), bd.edge_s, bd.edge_n, [&](int j) { return q(i, j); }, [&](int j) { return c(i, j); },
                  [&](int j) { return g.dya(i, j); }, [&](int j, double f) { flux(i, j) = f; });
}

// ---- xtp_u / ytp_v, iord = 8 and 10 of the NONLINEAR routines (sw_core_tlm.F90:4620-5041 for x, :5361-5865 for y): the same
// monotone slopes with their own edge treatment (no clamping of the two-sided edge value, zero slopes next to a corner of the
// face, pert_ppm on cells 2 and np-2 only) and their own form of the 2-delta-x test of iord 10.  One line routine for both
// directions; dd = dx or dy, rd = rdx or rdy along the line, c = the displacement (not a Courant number: cfl = c * rd(upwind)).
template <class FQ, class FC, class FD, class FR, class FOut>
void uv_line_mono(int iord, int first, int last, int np, bool any_edge, bool edge_lo, bool edge_hi, bool row_edge, const FQ& q, const FC& c,
                  const FD& dd, const FR& rd, const FOut& flux) {
  assert(iord >= 8 && iord <= 13);
  int is3 = first - 1, ie3 = last + 1;
  if (any_edge) { is3 = std::max(3, first - 1); ie3 = std::min(np - 3, last + 1); }
  const int lo = first - 4, n = last - first + 10;
  std::vector<double> dm_(n, 0.), al_(n, 0.), bl_(n, 0.), br_(n, 0.), dq_(n, 0.);
  auto dm = [&](int i) -> double& { return dm_[i - lo]; };
  auto al = [&](int i) -> double& { return al_[i - lo]; };
  auto bl = [&](int i) -> double& { return bl_[i - lo]; };
  auto br = [&](int i) -> double& { return br_[i - lo]; };
  auto dq = [&](int i) -> double& { return dq_[i - lo]; };
  for (int i = first - 2; i <= last + 2; ++i) {      // :4622-4666
    const double xt = 0.25 * (q(i + 1) - q(i - 1));
    const double hi = std::max(std::max(q(i - 1), q(i)), q(i + 1)) - q(i), lw = q(i) - std::min(std::min(q(i - 1), q(i)), q(i + 1));
    dm(i) = f_sign(std::min(std::min(std::fabs(xt), hi), lw), xt);
  }
  for (int i = first - 3; i <= last + 2; ++i) dq(i) = q(i + 1) - q(i);
  for (int i = is3; i <= ie3 + 1; ++i) al(i) = 0.5 * (q(i - 1) + q(i)) + mono_r3 * (dm(i - 1) - dm(i));
  if (iord == 8) {                                   // :4674-4709
    for (int i = is3; i <= ie3; ++i) {
      const double xt = 2. * dm(i);
      bl(i) = -f_sign(std::min(std::fabs(xt), std::fabs(al(i) - q(i))), xt);
      br(i) = f_sign(std::min(std::fabs(xt), std::fabs(al(i + 1) - q(i))), xt);
    }
  } else if (iord == 9) {                            // :4710-4780: the constraint applied everywhere, no 2-delta-x test
    for (int i = is3; i <= ie3; ++i) {
      const double pmp_1 = -(2. * dq(i)), lac_1 = pmp_1 + 1.5 * dq(i + 1);
      bl(i) = std::min(std::max(0., std::max(pmp_1, lac_1)), std::max(al(i) - q(i), std::min(0., std::min(pmp_1, lac_1))));
      const double pmp_2 = 2. * dq(i - 1), lac_2 = pmp_2 - 1.5 * dq(i - 2);
      br(i) = std::min(std::max(0., std::max(pmp_2, lac_2)), std::max(al(i + 1) - q(i), std::min(0., std::min(pmp_2, lac_2))));
    }
  } else if (iord >= 11) {                           // :4890-4896 "un-limited: 11" (the ELSE of the chain: 11, 12, 13)
    for (int i = is3; i <= ie3; ++i) { bl(i) = al(i) - q(i); br(i) = al(i + 1) - q(i); }
  } else {                                           // iord = 10, :4781-4890
    for (int i = is3; i <= ie3; ++i) {
      bl(i) = al(i) - q(i);
      br(i) = al(i + 1) - q(i);
      if (std::fabs(dm(i)) < mono_near_zero) {
        if (std::fabs(dm(i - 1)) + std::fabs(dm(i + 1)) < mono_near_zero) { bl(i) = 0.; br(i) = 0.; }
      } else if (std::fabs(3. * (bl(i) + br(i))) > std::fabs(bl(i) - br(i))) {
        const double pmp_1 = -(2. * dq(i)), lac_1 = pmp_1 + 1.5 * dq(i + 1);
        bl(i) = std::min(std::max(0., std::max(pmp_1, lac_1)), std::max(bl(i), std::min(0., std::min(pmp_1, lac_1))));
        const double pmp_2 = 2. * dq(i - 1), lac_2 = pmp_2 - 1.5 * dq(i - 2);
        br(i) = std::min(std::max(0., std::max(pmp_2, lac_2)), std::max(br(i), std::min(0., std::min(pmp_2, lac_2))));
      }
    }
  }
  auto two_sided = [&](int e) {     // x0l + x0r between cells e-1 and e (:4907-4912)
    return 0.5 * ((2. * dd(e - 1) + dd(e - 2)) * q(e - 1) - dd(e - 1) * q(e - 2)) / (dd(e - 1) + dd(e - 2)) +
           0.5 * ((2. * dd(e) + dd(e + 1)) * q(e) - dd(e) * q(e + 1)) / (dd(e) + dd(e + 1));
  };
  if (any_edge && edge_lo) {        // :4893-4917
    br(2) = al(3) - q(2);
    double xt = mono_s15 * q(1) + mono_s11 * q(2) - mono_s14 * dm(2);
    bl(2) = xt - q(2);
    br(1) = xt - q(1);
    if (row_edge) { bl(0) = 0.; br(0) = 0.; bl(1) = 0.; br(1) = 0.; }
    else {
      bl(0) = mono_s14 * dm(-1) - mono_s11 * dq(-1);
      xt = two_sided(1);
      br(0) = xt - q(0);
      bl(1) = xt - q(1);
    }
    pert_ppm_std(1, &bl(2), &br(2));
  }
  if (any_edge && edge_hi) {        // :4918-4946
    bl(np - 2) = al(np - 2) - q(np - 2);
    double xt = mono_s15 * q(np - 1) + mono_s11 * q(np - 2) + mono_s14 * dm(np - 2);
    br(np - 2) = xt - q(np - 2);
    bl(np - 1) = xt - q(np - 1);
    if (row_edge) { bl(np - 1) = 0.; br(np - 1) = 0.; bl(np) = 0.; br(np) = 0.; }
    else {
      br(np) = mono_s11 * dq(np) - mono_s14 * dm(np + 1);
      xt = two_sided(np);
      br(np - 1) = xt - q(np - 1);
      bl(np) = xt - q(np);
    }
    pert_ppm_std(1, &bl(np - 2), &br(np - 2));
  }
  for (int i = first; i <= last + 1; ++i) {          // :5028-5038
    if (c(i) > 0.) { const double cfl = c(i) * rd(i - 1); flux(i, q(i - 1) + (1. - cfl) * (br(i - 1) - cfl * (bl(i - 1) + br(i - 1)))); }
    else { const double cfl = c(i) * rd(i); flux(i, q(i) + (1. + cfl) * (bl(i) + cfl * (bl(i) + br(i)))); }
  }
}

// ---- the limited low-order schemes of the NONLINEAR routines, iord = 3 .. 7 (xppm: tp_core_tlm.F90:339-408 edge values, :442-590 fluxes;
// yppm :1060-1336; xtp_u: sw_core_tlm.F90:4406-4470 slopes, :4483-4612 fluxes; ytp_v :5147-5360).  Like 8 / 10 they are never differentiated:
// a trajectory scheme in 3 .. 7 beside a perturbation scheme in {1, 2, 333} gives the values only (split_hord).  Same edge values al as the
// linear scheme 2; what differs is which cells keep their parabola: smt5 / smt6 tests on (bl, br) of the two cells beside the interface.
static const double low_p1 = 7. / 12., low_p2 = -1. / 12., low_c1 = -2. / 14., low_c2 = 11. / 14., low_c3 = 5. / 14.;

// fluxes of one line from the slopes bl, br (cells first-1 .. last+1); cf(i, upwind cell) = the Courant number used in the flux formula
template <class FQ, class FC, class FBL, class FBR, class FCfl, class FOut>
void low_fluxes(int iord, int first, int last, const FQ& q, const FC& c, const FBL& bl, const FBR& br, const FCfl& cf, const FOut& flux) {
  const int lo = first - 1, n = last - first + 3;
  std::vector<double> b0_(n); std::vector<char> s5_(n), s6_(n);
  auto b0 = [&](int i) -> double& { return b0_[i - lo]; };
  auto smt5 = [&](int i) -> char& { return s5_[i - lo]; };
  auto smt6 = [&](int i) -> char& { return s6_[i - lo]; };
  for (int i = first - 1; i <= last + 1; ++i) {
    b0(i) = bl(i) + br(i);
    const double x0 = std::fabs(b0(i)), xt = std::fabs(bl(i) - br(i));
    if (iord == 3 || iord == 4) { smt5(i) = x0 < xt; smt6(i) = 3. * x0 < xt; }
    else if (iord == 5) smt5(i) = bl(i) * br(i) < 0.;
    else smt5(i) = std::fabs(3. * b0(i)) < xt;                 // 6 and 7
  }
  for (int i = first; i <= last + 1; ++i) {
    const bool up = c(i) > 0.;
    const double xt = up ? cf(i, i - 1) : cf(i, i);
    if (iord == 3) {                                           // tp_core :442-517, sw_core :4483-4549
      double fx1 = 0.;
      if (up) {
        if (smt6(i - 1) || smt5(i)) fx1 = br(i - 1) - xt * b0(i - 1);
        else if (smt5(i - 1)) fx1 = f_sign(std::min(std::fabs(bl(i - 1)), std::fabs(br(i - 1))), br(i - 1));
        flux(i, q(i - 1) + (1. - xt) * fx1);
      } else {
        if (smt6(i) || smt5(i - 1)) fx1 = bl(i) + xt * b0(i);
        else if (smt5(i)) fx1 = f_sign(std::min(std::fabs(bl(i)), std::fabs(br(i))), bl(i));
        flux(i, q(i) + (1. + xt) * fx1);
      }
    } else if (iord == 4) {                                    // tp_core :518-552, sw_core :4550-4582
      if (up) flux(i, (smt6(i - 1) || smt5(i)) ? q(i - 1) + (1. - xt) * (br(i - 1) - xt * b0(i - 1)) : q(i - 1));
      else flux(i, (smt6(i) || smt5(i - 1)) ? q(i) + (1. + xt) * (bl(i) + xt * b0(i)) : q(i));
    } else {                                                   // 5, 6, 7: tp_core :553-590, sw_core :4583-4612
      const double fx1 = up ? (1. - xt) * (br(i - 1) - xt * b0(i - 1)) : (1. + xt) * (bl(i) + xt * b0(i));
      double f = up ? q(i - 1) : q(i);
      if (smt5(i - 1) || smt5(i)) f = f + fx1;
      flux(i, f);
    }
  }
}

template <class FQ, class FC, class FD, class FOut>
void ppm_line_low(int iord, int first, int last, int np, bool any_edge, bool edge_lo, bool edge_hi, const FQ& q, const FC& c, const FD& da,
                  const FOut& flux) {
  assert(iord >= 3 && iord <= 7);
  int is1 = first - 1, ie3 = last + 2;
  if (any_edge) { is1 = std::max(3, first - 1); ie3 = std::min(np - 2, last + 2); }      // :314-330
  const int lo = first - 4, n = last - first + 10;
  std::vector<double> al_(n, 0.);
  auto al = [&](int i) -> double& { return al_[i - lo]; };
  for (int i = is1; i <= ie3; ++i) al(i) = low_p1 * (q(i - 1) + q(i)) + low_p2 * (q(i - 2) + q(i + 1));      // :342-344
  if (iord == 7) for (int i = is1; i <= ie3; ++i) if (al(i) < 0.) al(i) = 0.5 * (q(i - 1) + q(i));           // :345-349
  auto two_sided = [&](int e) {
    return 0.5 * (((2. * da(e - 1) + da(e - 2)) * q(e - 1) - da(e - 1) * q(e - 2)) / (da(e - 2) + da(e - 1)) +
                  ((2. * da(e) + da(e + 1)) * q(e) - da(e) * q(e + 1)) / (da(e) + da(e + 1)));
  };
  if (any_edge && edge_lo) {        // :351-374
    al(0) = low_c1 * q(-2) + low_c2 * q(-1) + low_c3 * q(0);
    al(1) = two_sided(1);
    al(2) = low_c3 * q(1) + low_c2 * q(2) + low_c1 * q(3);
    if (iord == 7) for (int i = 0; i <= 2; ++i) al(i) = std::max(0., al(i));
  }
  if (any_edge && edge_hi) {        // :376-406
    al(np - 1) = low_c1 * q(np - 3) + low_c2 * q(np - 2) + low_c3 * q(np - 1);
    al(np) = two_sided(np);
    al(np + 1) = low_c3 * q(np) + low_c2 * q(np + 1) + low_c1 * q(np + 2);
    if (iord == 7) for (int i = np - 1; i <= np + 1; ++i) al(i) = std::max(0., al(i));
  }
  low_fluxes(iord, first, last, q, c, [&](int i) { return al(i) - q(i); }, [&](int i) { return al(i + 1) - q(i); }, [&](int i, int) { return c(i); }, flux);
}

template <class FQ, class FC, class FD, class FR, class FOut>
void uv_line_low(int iord, int first, int last, int np, bool any_edge, bool edge_lo, bool edge_hi, bool row_edge, const FQ& q, const FC& c,
                 const FD& dd, const FR& rd, const FOut& flux) {
  assert(iord >= 3 && iord <= 7);
  int is3 = first - 1, ie3 = last + 1;
  if (any_edge) { is3 = std::max(3, first - 1); ie3 = std::min(np - 3, last + 1); }      // :4384-4395
  const int lo = first - 4, n = last - first + 10;
  std::vector<double> al_(n, 0.), bl_(n, 0.), br_(n, 0.);
  auto al = [&](int i) -> double& { return al_[i - lo]; };
  auto bl = [&](int i) -> double& { return bl_[i - lo]; };
  auto br = [&](int i) -> double& { return br_[i - lo]; };
  for (int i = is3; i <= ie3 + 1; ++i) al(i) = low_p1 * (q(i - 1) + q(i)) + low_p2 * (q(i - 2) + q(i + 1));      // :4408-4411
  for (int i = is3; i <= ie3; ++i) { bl(i) = al(i) - q(i); br(i) = al(i + 1) - q(i); }
  auto two_sided = [&](int e) {
    return 0.5 * (((2. * dd(e - 1) + dd(e - 2)) * q(e - 1) - dd(e - 1) * q(e - 2)) / (dd(e - 1) + dd(e - 2)) +
                  ((2. * dd(e) + dd(e + 1)) * q(e) - dd(e) * q(e + 1)) / (dd(e) + dd(e + 1)));
  };
  if (any_edge && edge_lo) {        // :4417-4440
    double xt = low_c3 * q(1) + low_c2 * q(2) + low_c1 * q(3);
    br(1) = xt - q(1); bl(2) = xt - q(2); br(2) = al(3) - q(2);
    if (row_edge) { bl(0) = 0.; br(0) = 0.; bl(1) = 0.; br(1) = 0.; }
    else {
      bl(0) = low_c1 * q(-2) + low_c2 * q(-1) + low_c3 * q(0) - q(0);
      xt = two_sided(1);
      br(0) = xt - q(0); bl(1) = xt - q(1);
    }
  }
  if (any_edge && edge_hi) {        // :4441-4466
    bl(np - 2) = al(np - 2) - q(np - 2);
    double xt = low_c1 * q(np - 3) + low_c2 * q(np - 2) + low_c3 * q(np - 1);
    br(np - 2) = xt - q(np - 2); bl(np - 1) = xt - q(np - 1);
    if (row_edge) { bl(np - 1) = 0.; br(np - 1) = 0.; bl(np) = 0.; br(np) = 0.; }
    else {
      xt = two_sided(np);
      br(np - 1) = xt - q(np - 1); bl(np) = xt - q(np);
      br(np) = low_c3 * q(np) + low_c2 * q(np + 1) + low_c1 * q(np + 2) - q(np);
    }
  }
  low_fluxes(iord, first, last, q, c, [&](int i) { return bl(i); }, [&](int i) { return br(i); }, [&](int i, int cell) { return c(i) * rd(cell); }, flux);
}

}  // namespace orc
