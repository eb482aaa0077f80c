// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Restatement of
//   fv_tracer2d_tlm.F90  TRACER_2D (:1148-1446 / _TLM :757-1147), q_split=0, trdm=0
//   fv_mapz_tlm.F90      LAGRANGIAN_TO_EULERIAN (:1361-2286 / _TLM :69-1360), hydrostatic,
//                        remap_option=0 (T in log p), consv=0, no sat_adj, do_omega diagnostics
//                        not restated; map_scalar (_TLM :7765), map1_ppm (_TLM :7909), map1_q2
//                        (_TLM :8222), scalar_profile/cs_profile with |kord|>16 — the only profile
//                        the TL/AD reference implements (_TLM :8356-8667)
//   fv_dynamics_tlm.F90  FV_DYNAMICS (:999-1745 / _TLM :87-995), hydrostatic, adiabatic=.false.,
//                        consv_te=0, tau=0, nwat<=1 (no fill2d), omega diagnostics not restated.
#pragma once
#include <functional>
#include "dyn_core.hpp"

namespace orc {
struct RemapOpts { int kord_tm = -17, kord_mt = 17, kord_wz = 17, kord_tr = 17, kord_tm_pert = -17, kord_mt_pert = 17, kord_wz_pert = 17, kord_tr_pert = 17; };
// trajectory / perturbation profile of one map (split_kord): equal -> one map; else the _TLM map with the perturbation profile on a copy
// and the nonlinear map with the trajectory profile for the values (fv_mapz_tlm.F90:494-523, :596-637, :780-827)
// the options of the instance a C-ABI call runs on (set at every entry point of capi.cpp)
inline RemapOpts& remap_opts() { static thread_local RemapOpts r; return r; }
struct Kord { int traj = 17, pert = 17; double qmin = 0.; bool scalar = false; Kord() {} Kord(int t, int p, bool sc = false, double qm = 0.) : traj(t), pert(p), qmin(qm), scalar(sc) {} };

// scalar_profile / cs_profile, iv != -2, |kord| > 16 (fv_mapz_tlm.F90:8424-8509 == :8592-8666)
template <class T>
void cs_profile_linear(std::vector<T>& a1, std::vector<T>& a2, std::vector<T>& a3, std::vector<T>& a4,
                       const std::vector<T>& delp, int km, int iv = 1, T qs = T(0.)) {
  std::vector<T> q(km + 2), gam(km + 2);
  if (iv == -2) {      // vertical velocity: lower boundary value qs given (fv_mapz_tlm.F90:8549-8590)
    gam[2] = T(0.5);
    q[1] = 1.5 * a1[1];
    for (int k = 2; k <= km - 1; ++k) {
      T grat = delp[k - 1] / delp[k];
      T bet = 2. + grat + grat - gam[k];
      q[k] = (3. * (a1[k - 1] + a1[k]) - q[k - 1]) / bet;
      gam[k + 1] = grat / bet;
    }
    T grat = delp[km - 1] / delp[km];
    q[km] = (3. * (a1[km - 1] + a1[km]) - grat * qs - q[km - 1]) / (2. + grat + grat - gam[km]);
    q[km + 1] = qs;
    for (int k = km - 1; k >= 1; --k) q[k] = q[k] - gam[k + 1] * q[k + 1];
    for (int k = 1; k <= km; ++k) { a2[k] = q[k]; a3[k] = q[k + 1]; a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k])); }
    return;
  }
  T grat = delp[2] / delp[1];
  T bet = grat * (grat + 0.5);
  q[1] = ((grat + grat) * (grat + 1.) * a1[1] + a1[2]) / bet;
  gam[1] = (1. + grat * (grat + 1.5)) / bet;
  T d4 = grat;
  for (int k = 2; k <= km; ++k) {
    d4 = delp[k - 1] / delp[k];
    bet = 2. + d4 + d4 - gam[k - 1];
    q[k] = (3. * (a1[k - 1] + d4 * a1[k]) - q[k - 1]) / bet;
    gam[k] = d4 / bet;
  }
  T a_bot = 1. + d4 * (d4 + 1.5);
  q[km + 1] = (2. * d4 * (d4 + 1.) * a1[km] + a1[km - 1] - a_bot * q[km]) / (d4 * (d4 + 0.5) - a_bot * gam[km]);
  for (int k = km; k >= 1; --k) q[k] = q[k] - gam[k] * q[k + 1];
  for (int k = 1; k <= km; ++k) {
    a2[k] = q[k];
    a3[k] = q[k + 1];
    a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
  }
}

// cs_limiters, fv_mapz_tlm.F90:6434-6519 (model/fv_mapz_nlm.F90:2467-2542): one layer
inline void cs_limiters(bool extm, double a1, double& a2, double& a3, double& a4, int iv) {
  const double r12 = 1. / 12.;
  if (iv == 0) {          // positive definite constraint
    if (a1 <= 0.) { a2 = a1; a3 = a1; a4 = 0.; }
    else if (std::fabs(a3 - a2) < -a4) {
      if (a1 + 0.25 * (a3 - a2) * (a3 - a2) / a4 + a4 * r12 < 0.) {
        if (a1 < a3 && a1 < a2) { a3 = a1; a2 = a1; a4 = 0.; }
        else if (a3 > a2) { a4 = 3. * (a2 - a1); a3 = a2 - a4; }
        else { a4 = 3. * (a3 - a1); a2 = a3 - a4; }
      }
    }
  } else {
    const bool flat = (iv == 1) ? ((a1 - a2) * (a1 - a3) >= 0.) : extm;
    if (flat) { a2 = a1; a3 = a1; a4 = 0.; }
    else {
      const double da1 = a3 - a2, da2 = da1 * da1, a6da = a4 * da1;
      if (a6da < -da2) { a4 = 3. * (a2 - a1); a3 = a2 - a4; }
      else if (a6da > da2) { a4 = 3. * (a3 - a1); a2 = a3 - a4; }
    }
  }
}
// The limited profiles of the NONLINEAR remap, |kord| in {9, 10, 11}: cs_profile (fv_mapz_tlm.F90:5566-6433) and scalar_profile
// (:3240-4228; = cs_profile + the q < qmin tests), model/fv_mapz_nlm.F90:2113-2464 / :1730-2110.  Values only: the reference
// differentiates the linear profile alone (:8653-8666).  a2, a3 enter as the unlimited edge values of the tridiagonal solve.
inline void cs_profile_limited(const std::vector<double>& a1, std::vector<double>& a2, std::vector<double>& a3, std::vector<double>& a4, int km, int iv,
                               int kord, bool scalar, double qmin) {
  const int ak = std::abs(kord);
  assert(ak >= 8 && ak <= 15);      // 16: scalar_profile has its own edge values (:1878); <= 7: ppm_profile
  std::vector<double> q(km + 2), gam(km + 3, 0.);
  std::vector<char> extm(km + 2, 0);
  for (int k = 1; k <= km; ++k) q[k] = a2[k];
  q[km + 1] = a3[km];
  auto mx = [](double a, double b) { return a > b ? a : b; };
  auto mn = [](double a, double b) { return a < b ? a : b; };
  q[2] = mn(q[2], mx(a1[1], a1[2])); q[2] = mx(q[2], mn(a1[1], a1[2]));       // large-scale constraints
  for (int k = 2; k <= km; ++k) gam[k] = a1[k] - a1[k - 1];
  for (int k = 3; k <= km - 1; ++k) {
    if (gam[k - 1] * gam[k + 1] > 0.) { q[k] = mn(q[k], mx(a1[k - 1], a1[k])); q[k] = mx(q[k], mn(a1[k - 1], a1[k])); }
    else if (gam[k - 1] > 0.) q[k] = mx(q[k], mn(a1[k - 1], a1[k]));
    else { q[k] = mn(q[k], mx(a1[k - 1], a1[k])); if (iv == 0) q[k] = mx(0., q[k]); }
  }
  q[km] = mn(q[km], mx(a1[km - 1], a1[km])); q[km] = mx(q[km], mn(a1[km - 1], a1[km]));
  for (int k = 1; k <= km; ++k) { a2[k] = q[k]; a3[k] = q[k + 1]; }
  for (int k = 1; k <= km; ++k)
    extm[k] = (k == 1 || k == km) ? ((a2[k] - a1[k]) * (a3[k] - a1[k]) > 0.) : (gam[k] * gam[k + 1] < 0.);
  // top two layers
  if (iv == 0) a2[1] = mx(0., a2[1]);
  else if (iv == -1) { if (a2[1] * a1[1] <= 0.) a2[1] = 0.; }
  else if (iv == 2) { a2[1] = a1[1]; a3[1] = a1[1]; a4[1] = 0.; }
  if (iv != 2) { a4[1] = 3. * (2. * a1[1] - (a2[1] + a3[1])); cs_limiters(extm[1], a1[1], a2[1], a3[1], a4[1], 1); }
  a4[2] = 3. * (2. * a1[2] - (a2[2] + a3[2]));
  cs_limiters(extm[2], a1[2], a2[2], a3[2], a4[2], 2);
  auto huynh = [&](int k) {
    const double pmp_1 = a1[k] - 2. * gam[k + 1], lac_1 = pmp_1 + 1.5 * gam[k + 2];
    a2[k] = mn(mx(a2[k], mn(a1[k], mn(pmp_1, lac_1))), mx(a1[k], mx(pmp_1, lac_1)));
    const double pmp_2 = a1[k] + 2. * gam[k], lac_2 = pmp_2 - 1.5 * gam[k - 1];
    a3[k] = mn(mx(a3[k], mn(a1[k], mn(pmp_2, lac_2))), mx(a1[k], mx(pmp_2, lac_2)));
  };
  for (int k = 3; k <= km - 2; ++k) {      // Huynh's second constraint in the interior
    const bool small = scalar && a1[k] < qmin;
    auto flat = [&]() { a2[k] = a1[k]; a3[k] = a1[k]; a4[k] = 0.; };
    if (ak == 9) {
      if (extm[k] && (extm[k - 1] || extm[k + 1] || small)) flat();
      else {
        a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
        if (std::fabs(a4[k]) > std::fabs(a2[k] - a3[k])) { huynh(k); a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k])); }
      }
    } else if (ak == 10) {
      if (extm[k]) {
        if (small || extm[k - 1] || extm[k + 1]) flat();
        else a4[k] = 6. * a1[k] - 3. * (a2[k] + a3[k]);
      } else {
        a4[k] = 6. * a1[k] - 3. * (a2[k] + a3[k]);
        if (std::fabs(a4[k]) > std::fabs(a2[k] - a3[k])) { huynh(k); a4[k] = 6. * a1[k] - 3. * (a2[k] + a3[k]); }
      }
    } else if (ak < 9) {                   // 8 (7 and below never get here: ppm_profile), model/fv_mapz_nlm.F90:2301-2315 == :1925-1939
      huynh(k);
      a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
    } else if (ak == 12) {                 // :2370-2390 == :2002-2023
      if (extm[k]) flat();
      else {
        a4[k] = 6. * a1[k] - 3. * (a2[k] + a3[k]);
        if (std::fabs(a4[k]) > std::fabs(a2[k] - a3[k])) { huynh(k); a4[k] = 6. * a1[k] - 3. * (a2[k] + a3[k]); }
      }
    } else if (ak == 13) {                 // :2391-2420 == :2024-2048
      if (extm[k]) {
        if (extm[k - 1] && extm[k + 1]) flat();
        else { huynh(k); a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k])); }
      } else a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
    } else if (ak == 14) {                 // :2421-2424 == :2049-2052
      a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
    } else {                               // 11 and, through the same ELSE, 15 (:2425-2436; scalar_profile :2071-2082 with the q < qmin test)
      if (extm[k] && (extm[k - 1] || extm[k + 1] || small)) flat();
      else a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
    }
    if (iv == 0) cs_limiters(extm[k], a1[k], a2[k], a3[k], a4[k], 0);
  }
  // bottom two layers
  if (iv == 0) a3[km] = mx(0., a3[km]);
  else if (iv == -1) { if (a3[km] * a1[km] <= 0.) a3[km] = 0.; }
  for (int k = km - 1; k <= km; ++k) {
    a4[k] = 3. * (2. * a1[k] - (a2[k] + a3[k]));
    cs_limiters(extm[k], a1[k], a2[k], a3[k], a4[k], k == km - 1 ? 2 : 1);
  }
}

// One column of map_scalar / map1_ppm / map1_q2 (fv_mapz_tlm.F90:7812-7905): q1 on pe1 (km layers)
// -> q2 on pe2 (kn layers).  1-based vectors.
template <class T>
void map_col(int km, const std::vector<T>& pe1, const std::vector<T>& q1, int kn, const std::vector<T>& pe2,
             std::vector<T>& q2, int iv = 1, T qs = T(0.), int kord = 17, bool scalar = false, double qmin = 0.) {
  const double r3 = 1. / 3., r23 = 2. / 3.;
  std::vector<T> dp1(km + 1), a1(km + 1), a2(km + 1), a3(km + 1), a4(km + 1);
  for (int k = 1; k <= km; ++k) { dp1[k] = pe1[k + 1] - pe1[k]; a1[k] = q1[k]; }
  cs_profile_linear(a1, a2, a3, a4, dp1, km, iv, qs);
  if (std::abs(kord) <= 16) {      // the limited profiles: nonlinear routine only
    if constexpr (std::is_same<T, double>::value) cs_profile_limited(a1, a2, a3, a4, km, iv, kord, scalar, qmin);
    else { std::fprintf(stderr, "oracle: limited remap profile on a differentiated scalar\n"); std::abort(); }
  }
  int k0 = 1;
  T qsum = T(0.);
  for (int k = 1; k <= kn; ++k) {
    int l; bool found = false;
    for (l = k0; l <= km; ++l)
      if (val(pe2[k]) >= val(pe1[l]) && val(pe2[k]) <= val(pe1[l + 1])) { found = true; break; }
    if (found) {
      T pl = (pe2[k] - pe1[l]) / dp1[l];
      if (val(pe2[k + 1]) <= val(pe1[l + 1])) {
        T pr = (pe2[k + 1] - pe1[l]) / dp1[l];
        q2[k] = a2[l] + 0.5 * (a4[l] + a3[l] - a2[l]) * (pr + pl) - a4[l] * r3 * (pr * (pr + pl) + pl * pl);
        k0 = l;
        continue;
      }
      qsum = (pe1[l + 1] - pe2[k]) * (a2[l] + 0.5 * (a4[l] + a3[l] - a2[l]) * (1. + pl) - a4[l] * (r3 * (1. + pl * (1. + pl))));
      int m; bool bottom = false;
      for (m = l + 1; m <= km; ++m) {
        if (val(pe2[k + 1]) > val(pe1[m + 1])) qsum = qsum + dp1[m] * a1[m];
        else { bottom = true; break; }
      }
      if (bottom) {
        T dp = pe2[k + 1] - pe1[m];
        T esl = dp / dp1[m];
        qsum = qsum + dp * (a2[m] + 0.5 * esl * (a3[m] - a2[m] + a4[m] * (1. - r23 * esl)));
        k0 = m;
      }
    }
    q2[k] = qsum / (pe2[k + 1] - pe2[k]);
  }
}

template <class T>
void map_col_split(int km, const std::vector<T>& pe1, const std::vector<T>& q1, int kn, const std::vector<T>& pe2, std::vector<T>& q2, int iv, T qs, const Kord& kd) {
  if (std::abs(kd.traj) == std::abs(kd.pert)) { map_col(km, pe1, q1, kn, pe2, q2, iv, qs, kd.traj, kd.scalar, kd.qmin); return; }
  map_col(km, pe1, q1, kn, pe2, q2, iv, qs, kd.pert, kd.scalar, kd.qmin);
  std::vector<double> p1(pe1.size()), p2(pe2.size()), qd(q1.size()), od(q2.size());
  for (size_t n = 0; n < pe1.size(); ++n) p1[n] = val(pe1[n]);
  for (size_t n = 0; n < pe2.size(); ++n) p2[n] = val(pe2[n]);
  for (size_t n = 0; n < q1.size(); ++n) qd[n] = val(q1[n]);
  map_col<double>(km, p1, qd, kn, p2, od, iv, val(qs), kd.traj, kd.scalar, kd.qmin);
  for (int k = 1; k <= kn; ++k) set_val(q2[k], od[k]);
}

// tracer_2d, fv_tracer2d_tlm.F90:1148-1446 (q_split = 0, nord_tr/trdm = 0).  dp1 = delp before dyn_core.
template <class T>
void tracer_2d(std::vector<Arr3<T>>& q, Arr3<T>& dp1, Arr3<T>& mfx, Arr3<T>& mfy, Arr3<T>& cx, Arr3<T>& cy, int npz,
               Hord hord, const Grid& g, const Bounds& bd, int* nsplt_out = nullptr, const std::vector<double>* cmax_all = nullptr,
               std::vector<double>* cmax_out = nullptr, const std::function<void(std::vector<Arr3<T>>&)>* halo = nullptr) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  const int nq = (int)q.size();
  Arr3<T> xfx(bd, npz), yfx(bd, npz);
  std::vector<double> cmax(npz + 1, 0.);
  std::vector<int> ksplt(npz + 1, 1);
  for (int k = 1; k <= npz; ++k) {
    for (int j = jsd; j <= jed; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        if (val(cx(i, j, k)) > 0.) xfx(i, j, k) = cx(i, j, k) * g.dxa(i - 1, j) * g.dy(i, j) * g.sin_sg[3](i - 1, j);
        else                       xfx(i, j, k) = cx(i, j, k) * g.dxa(i, j) * g.dy(i, j) * g.sin_sg[1](i, j);
      }
    for (int j = js; j <= je + 1; ++j)
      for (int i = isd; i <= ied; ++i) {
        if (val(cy(i, j, k)) > 0.) yfx(i, j, k) = cy(i, j, k) * g.dya(i, j - 1) * g.dx(i, j) * g.sin_sg[4](i, j - 1);
        else                       yfx(i, j, k) = cy(i, j, k) * g.dya(i, j) * g.dx(i, j) * g.sin_sg[2](i, j);
      }
    cmax[k] = 0.;
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) {
        double ax = std::fabs(val(cx(i, j, k))), ay = std::fabs(val(cy(i, j, k)));
        double c = (k < npz / 6) ? std::max(ax, ay) : std::max(ax, ay) + 1. - g.sin_sg[5](i, j);
        if (cmax[k] < c) cmax[k] = c;
      }
  }
  if (cmax_out) { *cmax_out = cmax; return; }          // first pass of a multi-tile run: only the local maxima
  if (cmax_all) cmax = *cmax_all;                      // mp_reduce_max over all tiles / ranks (fv_tracer2d_tlm.F90:1306)
  double c_global = cmax[1];
  for (int k = 2; k <= npz; ++k) if (!(cmax[k] < c_global)) c_global = cmax[k];
  const int nsplt = int(1. + c_global);
  if (nsplt_out) *nsplt_out = nsplt;
  if (nsplt != 1)
    for (int k = 1; k <= npz; ++k) {
      ksplt[k] = int(1. + cmax[k]);
      const double frac = 1. / double(ksplt[k]);
      for (int j = jsd; j <= jed; ++j)
        for (int i = is; i <= ie + 1; ++i) { cx(i, j, k) = cx(i, j, k) * frac; xfx(i, j, k) = xfx(i, j, k) * frac; }
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie + 1; ++i) mfx(i, j, k) = mfx(i, j, k) * frac;
      for (int j = js; j <= je + 1; ++j)
        for (int i = isd; i <= ied; ++i) { cy(i, j, k) = cy(i, j, k) * frac; yfx(i, j, k) = yfx(i, j, k) * frac; }
      for (int j = js; j <= je + 1; ++j)
        for (int i = is; i <= ie; ++i) mfy(i, j, k) = mfy(i, j, k) * frac;
    }
  Arr2<T> dp2(bd), ra_x(bd), ra_y(bd), fx(bd), fy(bd);
  for (int it = 1; it <= nsplt; ++it) {
    for (int k = 1; k <= npz; ++k) {
      if (it > ksplt[k]) continue;
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i)
          dp2(i, j) = dp1(i, j, k) + (mfx(i, j, k) - mfx(i + 1, j, k) + (mfy(i, j, k) - mfy(i, j + 1, k))) * g.rarea(i, j);
      for (int j = jsd; j <= jed; ++j)
        for (int i = is; i <= ie; ++i) ra_x(i, j) = g.area(i, j) + (xfx(i, j, k) - xfx(i + 1, j, k));
      for (int j = js; j <= je; ++j)
        for (int i = isd; i <= ied; ++i) ra_y(i, j) = g.area(i, j) + (yfx(i, j, k) - yfx(i, j + 1, k));
      for (int iq = 0; iq < nq; ++iq) {
        fv_tp_2d_split<T>(q[iq].plane(k), cx.plane(k), cy.plane(k), hord.traj, hord.pert, fx, fy, xfx.plane(k), yfx.plane(k), g, bd, ra_x, ra_y,
                          &mfx.plane(k), &mfy.plane(k), nullptr, -1, 0.0, -1, 0.0);       // fv_tracer2d_tlm.F90:1047-1110
        for (int j = js; j <= je; ++j)
          for (int i = is; i <= ie; ++i)
            q[iq](i, j, k) = (q[iq](i, j, k) * dp1(i, j, k) + (fx(i, j) - fx(i + 1, j) + (fy(i, j) - fy(i, j + 1))) * g.rarea(i, j)) / dp2(i, j);
      }
      if (it != nsplt)
        for (int j = js; j <= je; ++j)
          for (int i = is; i <= ie; ++i) dp1(i, j, k) = dp2(i, j);
    }
    if (it != nsplt) { if (halo) (*halo)(q); else for (auto& qq : q) halo_periodic(qq, bd); }
  }
}

// winds of Lagrangian_to_Eulerian: u, v on the pressures averaged across the edge (fv_mapz_tlm.F90:1884-1934); pe is still the
// Lagrangian one here.
template <class T>
void l2e_winds(DynState<T>& s, int km, const std::vector<double>& ak, const std::vector<double>& bk, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  std::vector<T> q1(km + 1), q2(km + 1), pe0(km + 2), pe3(km + 2);
  for (int j = js; j <= je + 1; ++j) {
    // map u (:1884-1909)
    for (int i = is; i <= ie; ++i) {
      pe0[1] = s.pe(i, j, 1);
      for (int k = 2; k <= km + 1; ++k) pe0[k] = 0.5 * (s.pe(i, j - 1, k) + s.pe(i, j, k));
      for (int k = 1; k <= km + 1; ++k) pe3[k] = ak[k - 1] + (0.5 * bk[k - 1]) * (s.pe(i, j - 1, km + 1) + s.pe(i, j, km + 1));
      for (int k = 1; k <= km; ++k) q1[k] = s.u(i, j, k);
      map_col_split(km, pe0, q1, km, pe3, q2, -1, T(0.), Kord(remap_opts().kord_mt, remap_opts().kord_mt_pert));
      for (int k = 1; k <= km; ++k) s.u(i, j, k) = q2[k];
    }
    if (j < je + 1)     // map v (:1913-1934)
      for (int i = is; i <= ie + 1; ++i) {
        pe3[1] = T(ak[0]);
        pe0[1] = s.pe(i, j, 1);
        for (int k = 2; k <= km + 1; ++k) {
          pe0[k] = 0.5 * (s.pe(i - 1, j, k) + s.pe(i, j, k));
          pe3[k] = ak[k - 1] + (0.5 * bk[k - 1]) * (s.pe(i - 1, j, km + 1) + s.pe(i, j, km + 1));
        }
        for (int k = 1; k <= km; ++k) q1[k] = s.v(i, j, k);
        map_col_split(km, pe0, q1, km, pe3, q2, -1, T(0.), Kord(remap_opts().kord_mt, remap_opts().kord_mt_pert));
        for (int k = 1; k <= km; ++k) s.v(i, j, k) = q2[k];
      }
  }
}

// Lagrangian_to_Eulerian, hydrostatic, remap_t (fv_mapz_tlm.F90:1586-1951, :2203-2250).
// pe must be valid on is-1..ie+1, js-1..je+1 (old Lagrangian pressures); sphum = tracer 0.
template <class T>
void lagrangian_to_eulerian(bool last_step, DynState<T>& s, int km, double akap, double zvir, double ptop,
                            const std::vector<double>& ak, const std::vector<double>& bk, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  const int nq = (int)s.q.size();
  Arr3<T> pe2s(bd, km + 1);
  std::vector<T> pe1(km + 2), pe2(km + 2), pn1(km + 2), pn2(km + 2), pk2(km + 2), q1(km + 1), q2(km + 1), pe0(km + 2), pe3(km + 2);
  for (int j = js; j <= je + 1; ++j) {
    if (j != je + 1)
      for (int i = is; i <= ie; ++i) {
        for (int k = 1; k <= km + 1; ++k) { pe1[k] = s.pe(i, j, k); pn1[k] = s.peln(i, j, k); }
        pe2[1] = T(ptop); pe2[km + 1] = s.pe(i, j, km + 1);
        for (int k = 1; k <= km; ++k)    // virtual pt -> virtual T (:1600-1606)
          s.pt(i, j, k) = s.pt(i, j, k) * (s.pk(i, j, k + 1) - s.pk(i, j, k)) / (akap * (pn1[k + 1] - pn1[k]));
        for (int k = 2; k <= km; ++k) pe2[k] = ak[k - 1] + bk[k - 1] * s.pe(i, j, km + 1);
        for (int k = 1; k <= km; ++k) s.delp(i, j, k) = pe2[k + 1] - pe2[k];
        pn2[1] = pn1[1]; pn2[km + 1] = pn1[km + 1]; pk2[1] = s.pk(i, j, 1); pk2[km + 1] = s.pk(i, j, km + 1);
        for (int k = 2; k <= km; ++k) { pn2[k] = log(pe2[k]); pk2[k] = exp(akap * pn2[k]); }
        for (int k = 1; k <= km; ++k) q1[k] = s.pt(i, j, k);
        map_col_split(km, pn1, q1, km, pn2, q2, 1, T(0.), Kord(remap_opts().kord_tm, remap_opts().kord_tm_pert, true, 184.));       // map_scalar in log p (:1676-1688), t_min = 184
        for (int k = 1; k <= km; ++k) s.pt(i, j, k) = q2[k];
        for (int iq = 0; iq < nq; ++iq) {        // map1_q2 (:1746-1763)
          for (int k = 1; k <= km; ++k) q1[k] = s.q[iq](i, j, k);
          map_col_split(km, pe1, q1, km, pe2, q2, 0, T(0.), Kord(remap_opts().kord_tr, remap_opts().kord_tr_pert, true, 0.));
          for (int k = 1; k <= km; ++k) s.q[iq](i, j, k) = q2[k];
        }
        for (int k = 1; k <= km + 1; ++k) { s.pk(i, j, k) = pk2[k]; s.peln(i, j, k) = pn2[k]; pe2s(i, j, k) = pe2[k]; }
        for (int k = 1; k <= km; ++k) s.pkz(i, j, k) = (pk2[k + 1] - pk2[k]) / (akap * (pn2[k + 1] - pn2[k]));
      }
  }
  l2e_winds(s, km, ak, bk, bd);
  for (int k = 2; k <= km; ++k)      // :1944-1950
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) s.pe(i, j, k) = pe2s(i, j, k);
  for (int k = 1; k <= km; ++k)      // :2203-2250 (dtmp = 0)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) {
        if (last_step) s.pt(i, j, k) = s.pt(i, j, k) / (1. + zvir * (nq > 0 ? s.q[0](i, j, k) : T(0.)));
        else           s.pt(i, j, k) = s.pt(i, j, k) / s.pkz(i, j, k);
      }
}

// fv_dynamics, hydrostatic (fv_dynamics_tlm.F90:1255-1262, :1395-1403, :1430-1680).  On entry pt is
// temperature, pe/pk/peln/pkz consistent with delp (compute_fv3_pressures); on exit pt is temperature.
template <class T>
void fv_dynamics(DynState<T>& s, const Arr2<double>& phis, int npz, double bdt, int n_split, int k_split,
                 const DampOpts& o, const Consts& c, double ptop, const std::vector<double>& ak,
                 const std::vector<double>& bk, const Grid& g, const Bounds& bd) {
  const int nq = (int)s.q.size();
  Arr3<T> dp1(bd, npz);
  for (int k = 1; k <= npz; ++k)
    for (int j = bd.js; j <= bd.je; ++j)
      for (int i = bd.is; i <= bd.ie; ++i) {
        T d = (nq > 0) ? c.zvir * s.q[0](i, j, k) : T(0.);
        s.pt(i, j, k) = s.pt(i, j, k) * (1. + d) / s.pkz(i, j, k);
      }
  const double mdt = bdt / double(k_split);
  for (int n_map = 1; n_map <= k_split; ++n_map) {
    halo_periodic(s.delp, bd); halo_periodic(s.pt, bd); halo_periodic(s.u, bd); halo_periodic(s.v, bd);
    for (int k = 1; k <= npz; ++k) dp1.plane(k) = s.delp.plane(k);
    const bool last_step = (n_map == k_split);
    dyn_core(s, phis, npz, mdt, n_split, o, c, ptop, g, bd);
    if (nq > 0) {
      for (auto& qq : s.q) halo_periodic(qq, bd);     // dyn_core_tlm.F90:2452-2454 halo of q
      tracer_2d(s.q, dp1, s.mfx, s.mfy, s.cx, s.cy, npz, Hord(o.hord_tr, o.hord_tr_pert), g, bd);
    }
    if (npz > 4) lagrangian_to_eulerian(last_step, s, npz, c.akap, c.zvir, ptop, ak, bk, bd);
  }
}

}  // namespace orc
