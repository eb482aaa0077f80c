// fv3lm-hip: Lagrangian-to-Eulerian vertical remap (LAGRANGIAN_TO_EULERIAN_TLM / _FWD+_BWD,
// fv_mapz_tlm.F90:69-1360, fv_mapz_adm.F90:91/1614) as column kernels: one thread per column,
// i fastest so every level access of a wave is one contiguous row piece.  Hydrostatic,
// remap_option = 0 (T in log p), |kord| > 16: the "perfectly linear" profile, the only one the TL/AD
// reference implements (fv_mapz_tlm.F90:8653-8666).
//   forward (nonlinear / tangent):  templated on the scalar (double / Dual)
//   adjoint: hand-written reverse of the same statements (tridiagonal edge-value solve reversed,
//            the monotone layer search replayed on the trajectory), no tape.
// Per-thread work vectors live in a global workspace laid out [slot][k][column] (coalesced), not in
// private memory.
#pragma once
#include "column.h"

namespace fv3 {

// workspace view of one column: element (slot, k), k = 0..kw-1
struct ColWs {
  double* base; size_t stride; int kw;   // stride = number of columns in the launch
  HD double& at(int slot, int k) const { return base[((size_t)slot * kw + k) * stride]; }
};
template <class T> struct WsIO;
template <> struct WsIO<double> {
  static constexpr int W = 1;
  HD static double get(const ColWs& w, int s, int k) { return w.at(s, k); }
  HD static void set(const ColWs& w, int s, int k, double x) { w.at(s, k) = x; }
};
template <> struct WsIO<Dual> {
  static constexpr int W = 2;
  HD static Dual get(const ColWs& w, int s, int k) { return Dual(w.at(2 * s, k), w.at(2 * s + 1, k)); }
  HD static void set(const ColWs& w, int s, int k, const Dual& x) { w.at(2 * s, k) = x.v; w.at(2 * s + 1, k) = x.d; }
};

constexpr double R3 = 1. / 3., R23 = 2. / 3.;

// ---------------------------------------------------------------- forward column map
// cs_profile (|kord|>16, iv != -2; fv_mapz_tlm.F90:8592-8666) + map1_ppm/map_scalar/map1_q2 loop
// (:7980-8045).  pe1, q1, pe2: functors k -> T (1-based); out(k, value).  Workspace slots used
// (in units of T): SG = gam, SE = edge values qe.
template <class T, class FP1, class FQ1, class FP2, class FOut>
HD void map_col(int km, const FP1& pe1, const FQ1& q1, const FP2& pe2, const FOut& out, const ColWs& ws, int SG, int SE) {
  typedef WsIO<T> W;
  // edge values by the tridiagonal solve
  T dpa = pe1(2) - pe1(1), dpb = pe1(3) - pe1(2);
  T grat = dpb / dpa;
  T bet = grat * (grat + 0.5);
  T qf = ((grat + grat) * (grat + 1.) * q1(1) + q1(2)) / bet;
  T gam = (1. + grat * (grat + 1.5)) / bet;
  W::set(ws, SE, 1, qf); W::set(ws, SG, 1, gam);
  T d4 = grat, a_prev = q1(1), dp_prev = dpa;
  for (int k = 2; k <= km; ++k) {
    T dpk = pe1(k + 1) - pe1(k);
    T ak_ = q1(k);
    d4 = dp_prev / dpk;
    bet = 2. + d4 + d4 - gam;
    qf = (3. * (a_prev + d4 * ak_) - qf) / bet;
    gam = d4 / bet;
    W::set(ws, SE, k, qf); W::set(ws, SG, k, gam);
    a_prev = ak_; dp_prev = dpk;
  }
  T a_bot = 1. + d4 * (d4 + 1.5);
  T qe = (2. * d4 * (d4 + 1.) * q1(km) + q1(km - 1) - a_bot * qf) / (d4 * (d4 + 0.5) - a_bot * gam);
  W::set(ws, SE, km + 1, qe);
  for (int k = km; k >= 1; --k) {
    qe = W::get(ws, SE, k) - W::get(ws, SG, k) * qe;
    W::set(ws, SE, k, qe);
  }
  // conservative mapping
  int k0 = 1;
  T qsum = T(0.);
  for (int k = 1; k <= km; ++k) {
    const T p2t = pe2(k), p2b = pe2(k + 1);
    int l; bool found = false;
    for (l = k0; l <= km; ++l)
      if (val(p2t) >= val(pe1(l)) && val(p2t) <= val(pe1(l + 1))) { found = true; break; }
    if (found) {
      const T p1l = pe1(l), p1r = pe1(l + 1), dpl = p1r - p1l;
      const T a1 = q1(l), a2 = W::get(ws, SE, l), a3 = W::get(ws, SE, l + 1), a4 = 3. * (2. * a1 - (a2 + a3));
      const T pl = (p2t - p1l) / dpl;
      if (val(p2b) <= val(p1r)) {
        const T pr = (p2b - p1l) / dpl;
        out(k, a2 + 0.5 * (a4 + a3 - a2) * (pr + pl) - a4 * R3 * (pr * (pr + pl) + pl * pl));
        k0 = l;
        continue;
      }
      qsum = (p1r - p2t) * (a2 + 0.5 * (a4 + a3 - a2) * (1. + pl) - a4 * (R3 * (1. + pl * (1. + pl))));
      int m; bool bottom = false;
      for (m = l + 1; m <= km; ++m) {
        if (val(p2b) > val(pe1(m + 1))) qsum = qsum + (pe1(m + 1) - pe1(m)) * q1(m);
        else { bottom = true; break; }
      }
      if (bottom) {
        const T p1m = pe1(m), dpm = pe1(m + 1) - p1m;
        const T b1 = q1(m), b2 = W::get(ws, SE, m), b3 = W::get(ws, SE, m + 1), b4 = 3. * (2. * b1 - (b2 + b3));
        const T dp = p2b - p1m, esl = dp / dpm;
        qsum = qsum + dp * (b2 + 0.5 * esl * (b3 - b2 + b4 * (1. - R23 * esl)));
        k0 = m;
      }
    }
    out(k, qsum / (p2b - p2t));
  }
}

// ---------------------------------------------------------------- limited profiles of the nonlinear remap (split_kord)
// cs_profile / scalar_profile with |kord| in {9, 10, 11} (fv_mapz_tlm.F90:5566-6433, :3240-4228; readable form
// model/fv_mapz_nlm.F90:2113-2464, :1730-2110) and cs_limiters (:6434-6519).  Values only: the reference differentiates the linear
// profile alone (:8653-8666) and, when the trajectory's kord differs from the perturbation's, runs the nonlinear map for the values
// (:494-523, :596-637, :780-827).  Written per LAYER: after the tridiagonal solve every constraint is local in k, so the mapping loop
// asks for (a2, a3, a4) of the one or two source layers a target touches instead of a pass that stores three more vectors.
// a1(k): layer means, qe(k): the unlimited edge values (k = 1..km+1).  iv: -1 winds, 0 positive definite, 1 others.
HD void cs_limiters(bool extm, double a1, double& a2, double& a3, double& a4, int iv) {
  if (iv == 0) {
    if (a1 <= 0.) { a2 = a1; a3 = a1; a4 = 0.; }
    else if (fabs(a3 - a2) < -a4) {
      if (a1 + 0.25 * (a3 - a2) * (a3 - a2) / a4 + a4 * (1. / 12.) < 0.) {
        if (a1 < a3 && a1 < a2) { a3 = a1; a2 = a1; a4 = 0.; }
        else if (a3 > a2) { a4 = 3. * (a2 - a1); a3 = a2 - a4; }
        else { a4 = 3. * (a3 - a1); a2 = a3 - a4; }
      }
    }
    return;
  }
  const bool flat = (iv == 1) ? ((a1 - a2) * (a1 - a3) >= 0.) : extm;
  if (flat) { a2 = a1; a3 = a1; a4 = 0.; return; }
  const double da1 = a3 - a2, da2 = da1 * da1, a6da = a4 * da1;
  if (a6da < -da2) { a4 = 3. * (a2 - a1); a3 = a2 - a4; }
  else if (a6da > da2) { a4 = 3. * (a3 - a1); a2 = a3 - a4; }
}
struct LimProfile { int kord, iv; bool scalar; double qmin; };     // scalar: scalar_profile's q < qmin tests (map_scalar, map1_q2)
template <class FA, class FE>
HD void lim_layer(const LimProfile& P, int k, int km, const FA& a1, const FE& qe, double& a2, double& a3, double& a4) {
  const int ak = P.kord < 0 ? -P.kord : P.kord, iv = P.iv;
  // the edge value at interface j after the large-scale constraints (:2207-2245 of the nlm file)
  auto qlim = [&](int j) -> double {
    double q = qe(j);
    if (j <= 1 || j >= km + 1) return q;
    const double lo = fmin(a1(j - 1), a1(j)), hi = fmax(a1(j - 1), a1(j));
    if (j == 2 || j == km) return fmax(fmin(q, hi), lo);
    const double g1 = a1(j - 1) - a1(j - 2), g2 = a1(j + 1) - a1(j);
    if (g1 * g2 > 0.) return fmax(fmin(q, hi), lo);
    if (g1 > 0.) return fmax(q, lo);
    q = fmin(q, hi);
    return iv == 0 ? fmax(0., q) : q;
  };
  auto gam = [&](int j) { return a1(j) - a1(j - 1); };          // j = 2..km
  auto extm = [&](int j) -> bool {                               // interior layers only (2..km-1)
    return gam(j) * gam(j + 1) < 0.;
  };
  const double a = a1(k);
  a2 = qlim(k); a3 = qlim(k + 1);
  if (k == 1 || k == km) {
    const bool ex = (a2 - a) * (a3 - a) > 0.;
    if (k == 1) {
      if (iv == 0) a2 = fmax(0., a2);
      else if (iv == -1) { if (a2 * a <= 0.) a2 = 0.; }
    } else {
      if (iv == 0) a3 = fmax(0., a3);
      else if (iv == -1) { if (a3 * a <= 0.) a3 = 0.; }
    }
    a4 = 3. * (2. * a - (a2 + a3));
    cs_limiters(ex, a, a2, a3, a4, 1);
    return;
  }
  if (k == 2 || k == km - 1) { a4 = 3. * (2. * a - (a2 + a3)); cs_limiters(extm(k), a, a2, a3, a4, 2); return; }
  const bool ex = extm(k), small = P.scalar && a < P.qmin;
  auto huynh = [&]() {
    const double pmp_1 = a - 2. * gam(k + 1), lac_1 = pmp_1 + 1.5 * gam(k + 2);
    a2 = fmin(fmax(a2, fmin(a, fmin(pmp_1, lac_1))), fmax(a, fmax(pmp_1, lac_1)));
    const double pmp_2 = a + 2. * gam(k), lac_2 = pmp_2 - 1.5 * gam(k - 1);
    a3 = fmin(fmax(a3, fmin(a, fmin(pmp_2, lac_2))), fmax(a, fmax(pmp_2, lac_2)));
  };
  // extm(k-1), extm(k+1): k-1 >= 2 and k+1 <= km-1 here, interior form
  if (ak == 9) {
    if (ex && (extm(k - 1) || extm(k + 1) || small)) { a2 = a; a3 = a; a4 = 0.; }
    else {
      a4 = 3. * (2. * a - (a2 + a3));
      if (fabs(a4) > fabs(a2 - a3)) { huynh(); a4 = 3. * (2. * a - (a2 + a3)); }
    }
  } else if (ak == 10) {
    if (ex) {
      if (small || extm(k - 1) || extm(k + 1)) { a2 = a; a3 = a; a4 = 0.; }
      else a4 = 6. * a - 3. * (a2 + a3);
    } else {
      a4 = 6. * a - 3. * (a2 + a3);
      if (fabs(a4) > fabs(a2 - a3)) { huynh(); a4 = 6. * a - 3. * (a2 + a3); }
    }
  } else if (ak == 8) {      // |kord| < 9: the constraint on every interior layer (model/fv_mapz_nlm.F90:2301-2315; scalar_profile :1925-1939)
    huynh(); a4 = 3. * (2. * a - (a2 + a3));
  } else if (ak == 12) {     // :2370-2390 (:2002-2023): flat at every local extremum
    if (ex) { a2 = a; a3 = a; a4 = 0.; }
    else {
      a4 = 6. * a - 3. * (a2 + a3);
      if (fabs(a4) > fabs(a2 - a3)) { huynh(); a4 = 6. * a - 3. * (a2 + a3); }
    }
  } else if (ak == 13) {     // :2391-2420 (:2024-2048)
    if (ex) {
      if (extm(k - 1) && extm(k + 1)) { a2 = a; a3 = a; a4 = 0.; }
      else { huynh(); a4 = 3. * (2. * a - (a2 + a3)); }
    } else a4 = 3. * (2. * a - (a2 + a3));
  } else if (ak == 14) {     // :2421-2424 (:2049-2052): no constraint in the interior
    a4 = 3. * (2. * a - (a2 + a3));
  } else {      // 11, and 15 through the same ELSE (:2425-2436; scalar_profile :2071-2082)
    if (ex && (extm(k - 1) || extm(k + 1) || small)) { a2 = a; a3 = a; a4 = 0.; }
    else a4 = 3. * (2. * a - (a2 + a3));
  }
  if (iv == 0) cs_limiters(ex, a, a2, a3, a4, 0);
}
// the mapping loop of map_scalar / map1_ppm / map1_q2 (fv_mapz_tlm.F90:2832-3239) on the limited profile: qe(k) = the unlimited edge values
template <class FP1, class FQ1, class FP2, class FE, class FOut>
HD void map_loop_lim(const LimProfile& P, int km, const FP1& pe1, const FQ1& q1, const FP2& pe2, const FE& qe, const FOut& out) {
  int k0 = 1;
  double qsum = 0.;
  for (int k = 1; k <= km; ++k) {
    const double p2t = pe2(k), p2b = pe2(k + 1);
    int l; bool found = false;
    for (l = k0; l <= km; ++l)
      if (p2t >= pe1(l) && p2t <= pe1(l + 1)) { found = true; break; }
    if (found) {
      const double p1l = pe1(l), p1r = pe1(l + 1), dpl = p1r - p1l;
      double a2, a3, a4; lim_layer(P, l, km, q1, qe, a2, a3, a4);
      const double pl = (p2t - p1l) / dpl;
      if (p2b <= p1r) {
        const double pr = (p2b - p1l) / dpl;
        out(k, a2 + 0.5 * (a4 + a3 - a2) * (pr + pl) - a4 * R3 * (pr * (pr + pl) + pl * pl));
        k0 = l;
        continue;
      }
      qsum = (p1r - p2t) * (a2 + 0.5 * (a4 + a3 - a2) * (1. + pl) - a4 * (R3 * (1. + pl * (1. + pl))));
      int m; bool bottom = false;
      for (m = l + 1; m <= km; ++m) {
        if (p2b > pe1(m + 1)) qsum = qsum + (pe1(m + 1) - pe1(m)) * q1(m);
        else { bottom = true; break; }
      }
      if (bottom) {
        const double p1m = pe1(m), dpm = pe1(m + 1) - p1m;
        double b2, b3, b4; lim_layer(P, m, km, q1, qe, b2, b3, b4);
        const double dp = p2b - p1m, esl = dp / dpm;
        qsum = qsum + dp * (b2 + 0.5 * esl * (b3 - b2 + b4 * (1. - R23 * esl)));
        k0 = m;
      }
    }
    out(k, qsum / (p2b - p2t));
  }
}

// map_col on double with the limited profile: the same tridiagonal solve (edge values into slot SE), then the mapping loop with
// lim_layer's (a2, a3, a4).  Slots SG, SE in units of double.
template <class FP1, class FQ1, class FP2, class FOut>
HD void map_col_lim(const LimProfile& P, int km, const FP1& pe1, const FQ1& q1, const FP2& pe2, const FOut& out, const ColWs& ws, int SG, int SE) {
  {
    double dpa = pe1(2) - pe1(1), dpb = pe1(3) - pe1(2);
    double grat = dpb / dpa, bet = grat * (grat + 0.5);
    double qf = ((grat + grat) * (grat + 1.) * q1(1) + q1(2)) / bet, gam = (1. + grat * (grat + 1.5)) / bet;
    ws.at(SE, 1) = qf; ws.at(SG, 1) = gam;
    double d4 = grat, a_prev = q1(1), dp_prev = dpa;
    for (int k = 2; k <= km; ++k) {
      const double dpk = pe1(k + 1) - pe1(k), ak_ = q1(k);
      d4 = dp_prev / dpk;
      bet = 2. + d4 + d4 - gam;
      qf = (3. * (a_prev + d4 * ak_) - qf) / bet;
      gam = d4 / bet;
      ws.at(SE, k) = qf; ws.at(SG, k) = gam;
      a_prev = ak_; dp_prev = dpk;
    }
    const double a_bot = 1. + d4 * (d4 + 1.5);
    double qe = (2. * d4 * (d4 + 1.) * q1(km) + q1(km - 1) - a_bot * qf) / (d4 * (d4 + 0.5) - a_bot * gam);
    ws.at(SE, km + 1) = qe;
    for (int k = km; k >= 1; --k) { qe = ws.at(SE, k) - ws.at(SG, k) * qe; ws.at(SE, k) = qe; }
  }
  auto qe = [&](int k) { return ws.at(SE, k); };
  map_loop_lim(P, km, pe1, q1, pe2, qe, out);
}
HD bool kord_limited(int kord) { return (kord < 0 ? -kord : kord) <= 16; }

// ---------------------------------------------------------------- adjoint column map
// Trajectory functors pe1, q1, pe2 (double); q2_ad(k) functor; accumulates into workspace slots
// SP1 (pe1_ad), SQ1 (q1_ad), SP2 (pe2_ad), which the caller has zeroed or pre-loaded.  Scratch
// slots: SG gam, SB bet, SF qf (pre back-substitution), SE qe, SD d4, SDP dp1_ad, SGA gam_ad,
// SFA qf_ad, SEA qe_ad, SDA d4_ad.
struct MapAdSlots { int SP1, SQ1, SP2, SG, SB, SF, SE, SD, SDP, SGA, SFA, SEA, SDA; };
// IV = -2: the profile of the vertical velocity (lower boundary value qs given, fv_mapz_tlm.F90:8549-8590); qs_ad receives its adjoint.
template <int IV, class FP1, class FQ1, class FP2, class FAd>
HD void map_col_ad_iv(int km, const FP1& pe1, const FQ1& q1, const FP2& pe2, const FAd& q2_ad, const ColWs& ws,
                      const MapAdSlots& S, double qs, double* qs_ad) {
  // ---- forward replay with storage
  const double dp1_1 = pe1(2) - pe1(1), dp1_2 = pe1(3) - pe1(2);
  static_assert(IV == -2, "the general profile has the streaming form below (map_col_ad_s)");
  {
    double qf = 1.5 * q1(1), dp_prev = dp1_1;
    ws.at(S.SF, 1) = qf; ws.at(S.SG, 2) = 0.5;
    for (int k = 2; k <= km - 1; ++k) {
      const double dpk = pe1(k + 1) - pe1(k), gr = dp_prev / dpk, bet = 2. + gr + gr - ws.at(S.SG, k);
      qf = (3. * (q1(k - 1) + q1(k)) - qf) / bet;
      ws.at(S.SF, k) = qf; ws.at(S.SD, k) = gr; ws.at(S.SG, k + 1) = gr / bet;
      dp_prev = dpk;
    }
    const double gr = dp_prev / (pe1(km + 1) - pe1(km)), den = 2. + gr + gr - ws.at(S.SG, km);
    qf = (3. * (q1(km - 1) + q1(km)) - gr * qs - qf) / den;
    ws.at(S.SF, km) = qf; ws.at(S.SD, km) = gr; ws.at(S.SF, km + 1) = qs;
    double qe = qf;
    ws.at(S.SE, km + 1) = qs; ws.at(S.SE, km) = qe;
    for (int k = km - 1; k >= 1; --k) { qe = ws.at(S.SF, k) - ws.at(S.SG, k + 1) * qe; ws.at(S.SE, k) = qe; }
  }
  // accumulated by the mapping loop; SFA / SGA are written before they are read, the sweeps below carry their recurrences in registers
  for (int k = 0; k <= km + 1; ++k) { ws.at(S.SDP, k) = 0.; ws.at(S.SEA, k) = 0.; }
  // ---- reverse of the mapping loop (targets are independent; the search is replayed in order)
  auto addq = [&](int l, double a1_ad, double a2_ad, double a3_ad, double a4_ad) {   // a4 = 3(2 a1 - a2 - a3)
    ws.at(S.SQ1, l) += a1_ad + 6. * a4_ad;
    ws.at(S.SEA, l) += a2_ad - 3. * a4_ad;
    ws.at(S.SEA, l + 1) += a3_ad - 3. * a4_ad;
  };
  int k0 = 1;
  double p2k = 0., p2k1 = 0.;      // pending pe2_ad(k), pe2_ad(k+1): targets are visited in order, one update per level
  for (int k = 1; k <= km; ++k) {
    if (k > 1) ws.at(S.SP2, k - 1) += p2k;             // level k-1 is complete
    p2k = p2k1; p2k1 = 0.;
    const double g = q2_ad(k);
    const double p2t = pe2(k), p2b = pe2(k + 1);
    int l; bool found = false;
    for (l = k0; l <= km; ++l)
      if (p2t >= pe1(l) && p2t <= pe1(l + 1)) { found = true; break; }
    if (!found) continue;   // (stale-qsum path of the reference: never taken for ordered pressures)
    const double p1l = pe1(l), p1r = pe1(l + 1), dpl = p1r - p1l;
    const double a1 = q1(l), a2 = ws.at(S.SE, l), a3 = ws.at(S.SE, l + 1), a4 = 3. * (2. * a1 - (a2 + a3)), Sm = a4 + a3 - a2;
    const double pl = (p2t - p1l) / dpl;
    double pl_ad = 0.;
    if (p2b <= p1r) {
      const double pr = (p2b - p1l) / dpl, s = pr + pl, w = pr * s + pl * pl;
      addq(l, 0., (1. - 0.5 * s) * g, 0.5 * s * g, (0.5 * s - R3 * w) * g);
      const double pr_ad = (0.5 * Sm - a4 * R3 * (2. * pr + pl)) * g;
      pl_ad = (0.5 * Sm - a4 * R3 * (pr + 2. * pl)) * g;
      p2k1 += pr_ad / dpl; ws.at(S.SP1, l) -= pr_ad / dpl; ws.at(S.SDP, l) -= pr * pr_ad / dpl;
      k0 = l;
    } else {
      const double D = p2b - p2t;
      const double E = a2 + 0.5 * Sm * (1. + pl) - a4 * (R3 * (1. + pl * (1. + pl)));
      double qsum = (p1r - p2t) * E;
      int m; bool bottom = false;
      for (m = l + 1; m <= km; ++m) {
        if (p2b > pe1(m + 1)) qsum += (pe1(m + 1) - pe1(m)) * q1(m);
        else { bottom = true; break; }
      }
      const int mend = m;   // first layer not fully inside
      double dp = 0., esl = 0., Fm = 0., b1 = 0., b2 = 0., b3 = 0., b4 = 0., dpm = 1.;
      if (bottom) {
        const double p1m = pe1(m); dpm = pe1(m + 1) - p1m;
        b1 = q1(m); b2 = ws.at(S.SE, m); b3 = ws.at(S.SE, m + 1); b4 = 3. * (2. * b1 - (b2 + b3));
        dp = p2b - p1m; esl = dp / dpm;
        Fm = b2 + 0.5 * esl * (b3 - b2 + b4 * (1. - R23 * esl));
        qsum += dp * Fm;
      }
      const double q2v = qsum / D, qs_ad = g / D, D_ad = -q2v * g / D;
      p2k1 += D_ad; p2k -= D_ad;
      const double W_ad = E * qs_ad, E_ad = (p1r - p2t) * qs_ad;
      ws.at(S.SP1, l + 1) += W_ad; p2k -= W_ad;
      addq(l, 0., (1. - 0.5 * (1. + pl)) * E_ad, 0.5 * (1. + pl) * E_ad, (0.5 * (1. + pl) - R3 * (1. + pl * (1. + pl))) * E_ad);
      pl_ad = (0.5 * Sm - a4 * R3 * (1. + 2. * pl)) * E_ad;
      for (int mm = l + 1; mm < mend; ++mm) { ws.at(S.SDP, mm) += q1(mm) * qs_ad; ws.at(S.SQ1, mm) += (pe1(mm + 1) - pe1(mm)) * qs_ad; }
      if (bottom) {
        double dp_ad = Fm * qs_ad;
        const double F_ad = dp * qs_ad;
        addq(m, 0., (1. - 0.5 * esl) * F_ad, 0.5 * esl * F_ad, 0.5 * esl * (1. - R23 * esl) * F_ad);
        const double esl_ad = (0.5 * (b3 - b2 + b4 * (1. - R23 * esl)) - 0.5 * esl * b4 * R23) * F_ad;
        dp_ad += esl_ad / dpm; ws.at(S.SDP, m) -= esl * esl_ad / dpm;
        p2k1 += dp_ad; ws.at(S.SP1, m) -= dp_ad;
        k0 = m;
      }
    }
    p2k += pl_ad / dpl; ws.at(S.SP1, l) -= pl_ad / dpl; ws.at(S.SDP, l) -= pl * pl_ad / dpl;
  }
  ws.at(S.SP2, km) += p2k; ws.at(S.SP2, km + 1) += p2k1;
  {
    // reverse of  q(k) = qf(k) - gam(k+1) q(k+1), k = km-1..1 ;  q(km) = qf(km) ;  q(km+1) = qs
    double carry = 0.;
    for (int k = 1; k <= km - 1; ++k) {
      const double e = ws.at(S.SEA, k) + carry;
      ws.at(S.SFA, k) = e;
      ws.at(S.SGA, k + 1) = -ws.at(S.SE, k + 1) * e;
      carry = -ws.at(S.SG, k + 1) * e;
    }
    double a_qs = ws.at(S.SEA, km + 1), sfa, sga;
    {   // q(km) = (3 (a(km-1) + a(km)) - gr qs - qf(km-1)) / den,  den = 2 + 2 gr - gam(km)
      const double gr = ws.at(S.SD, km), den = 2. + gr + gr - ws.at(S.SG, km), t = (ws.at(S.SEA, km) + carry) / den;
      ws.at(S.SQ1, km - 1) += 3. * t; ws.at(S.SQ1, km) += 3. * t;
      a_qs -= gr * t;
      const double a_den = -ws.at(S.SF, km) * t, a_gr = -qs * t + 2. * a_den, dpk = pe1(km + 1) - pe1(km);
      sfa = ws.at(S.SFA, km - 1) - t;
      sga = ws.at(S.SGA, km) - a_den;
      ws.at(S.SDP, km - 1) += a_gr / dpk; ws.at(S.SDP, km) -= gr * a_gr / dpk;
    }
    for (int k = km - 1; k >= 2; --k) {   // qf(k) = (3 (a(k-1) + a(k)) - qf(k-1)) / bet, bet = 2 + 2 gr - gam(k), gam(k+1) = gr / bet
      const double gr = ws.at(S.SD, k), bet = 2. + gr + gr - ws.at(S.SG, k), t = sfa / bet;
      ws.at(S.SQ1, k - 1) += 3. * t; ws.at(S.SQ1, k) += 3. * t;
      const double a_bet = -ws.at(S.SF, k) * t - ws.at(S.SG, k + 1) * sga / bet, a_gr = sga / bet + 2. * a_bet, dpk = pe1(k + 1) - pe1(k);
      sfa = ws.at(S.SFA, k - 1) - t;
      sga = (k > 2 ? ws.at(S.SGA, k) : 0.) - a_bet;      // gam(2) = 0.5 is a constant
      ws.at(S.SDP, k - 1) += a_gr / dpk; ws.at(S.SDP, k) -= gr * a_gr / dpk;
    }
    ws.at(S.SQ1, 1) += 1.5 * sfa;
    if (qs_ad) *qs_ad += a_qs;
  }
  for (int k = 1; k <= km; ++k) { const double a = ws.at(S.SDP, k); ws.at(S.SP1, k + 1) += a; ws.at(S.SP1, k) -= a; }
}
// ---------------------------------------------------------------- adjoint column map, streaming form (iv != -2)
// Four sweeps, every work vector written once and read once per sweep that needs it, nothing read-modify-written:
//   A  top-down    forward elimination                      -> SF (qf), SG (gam)
//   B  bottom-up   back substitution                        -> SE (edge values)
//   C  top-down    reverse of the mapping loop FUSED with the reverse of the back substitution.  Targets are visited in
//                  order and their source layers l .. m never move up, so the adjoint accumulators of the one source level
//                  still open (q1_ad, dp1_ad, pe1_ad and the two edge-value adjoints) sit in registers; a level is closed
//                  when the search moves past it: its edge-value adjoint then enters the back-substitution recurrence
//                  (-> SFA, SGA) and its partial q1_ad / pe1_ad are stored (-> SQ, SP).  pe2_ad leaves through p2out.
//   D  bottom-up   reverse of the forward elimination (d4 = dp(k-1) / dp(k) recomputed from pe1), final q1_ad, pe1_ad
//                  through q1out / p1out, once per level.
// SQ / SP may be the very vectors q1out / p1out store to.
struct MapAdWs { int SG, SF, SE, SFA, SGA, SQ, SP; };
template <class FP1, class FQ1, class FP2, class FAd, class OQ1, class OP1, class OP2>
HD void map_col_ad_s(int km, const FP1& pe1, const FQ1& q1, const FP2& pe2, const FAd& q2_ad, const ColWs& ws, const MapAdWs& S,
                     const OQ1& q1out, const OP1& p1out, const OP2& p2out) {
  const double dp1_1 = pe1(2) - pe1(1), dp1_2 = pe1(3) - pe1(2);
  const double grat = dp1_2 / dp1_1, bet1 = grat * (grat + 0.5);
  double qbot;
  {   // ---- A
    double qf = ((grat + grat) * (grat + 1.) * q1(1) + q1(2)) / bet1, gam = (1. + grat * (grat + 1.5)) / bet1;
    ws.at(S.SF, 1) = qf; ws.at(S.SG, 1) = gam;
    double d4 = grat, a_prev = q1(1), dp_prev = dp1_1;
    for (int k = 2; k <= km; ++k) {
      const double dpk = pe1(k + 1) - pe1(k), ak_ = q1(k);
      d4 = dp_prev / dpk;
      const double bet = 2. + d4 + d4 - gam;
      qf = (3. * (a_prev + d4 * ak_) - qf) / bet;
      gam = d4 / bet;
      ws.at(S.SF, k) = qf; ws.at(S.SG, k) = gam;
      a_prev = ak_; dp_prev = dpk;
    }
    const double a_bot = 1. + d4 * (d4 + 1.5), den = d4 * (d4 + 0.5) - a_bot * gam;
    qbot = (2. * d4 * (d4 + 1.) * q1(km) + q1(km - 1) - a_bot * qf) / den;
  }
  {   // ---- B
    double qe = qbot;
    ws.at(S.SE, km + 1) = qe;
    for (int k = km; k >= 1; --k) { qe = ws.at(S.SF, k) - ws.at(S.SG, k) * qe; ws.at(S.SE, k) = qe; }
  }
  double fa_bot;
  {   // ---- C
    int b = 1;                                          // the open source level
    double aQ = 0., aD = 0., aP = 0., aPn = 0., aE = 0., aEn = 0.;   // q1_ad(b), dp1_ad(b), pe1_ad(b), pe1_ad(b+1), qe_ad(b), qe_ad(b+1)
    double ecarry = 0., dprev = 0.;                     // back-substitution carry into qe_ad(b); dp1_ad(b-1)
    // close levels b .. nb-1; the levels strictly between lo and hi lie wholly inside the current target (weight qsa)
    auto close_to = [&](int nb, int lo, int hi, double qsa) {
      while (b < nb) {
        const double e = aE + ecarry;
        ws.at(S.SFA, b) = e; ws.at(S.SGA, b) = -ws.at(S.SE, b + 1) * e; ecarry = -ws.at(S.SG, b) * e;
        ws.at(S.SQ, b) = aQ; ws.at(S.SP, b) = aP - aD + dprev; dprev = aD;
        ++b;
        const bool in = b > lo && b < hi;
        aQ = in ? (pe1(b + 1) - pe1(b)) * qsa : 0.; aD = in ? q1(b) * qsa : 0.;
        aP = aPn; aE = aEn; aPn = 0.; aEn = 0.;
      }
    };
    auto addq = [&](double a2_ad, double a3_ad, double a4_ad) {   // a4 = 3 (2 a1 - a2 - a3) of the open level
      aQ += 6. * a4_ad; aE += a2_ad - 3. * a4_ad; aEn += a3_ad - 3. * a4_ad;
    };
    double p2k = 0., p2k1 = 0.;      // pending pe2_ad(k), pe2_ad(k+1)
    for (int k = 1; k <= km; ++k) {
      if (k > 1) p2out(k - 1, p2k);
      p2k = p2k1; p2k1 = 0.;
      const double g = q2_ad(k);
      const double p2t = pe2(k), p2b = pe2(k + 1);
      int l; bool found = false;
      for (l = b; l <= km; ++l)
        if (p2t >= pe1(l) && p2t <= pe1(l + 1)) { found = true; break; }
      if (!found) continue;   // (stale-qsum path of the reference: never taken for ordered pressures)
      close_to(l, 0, 0, 0.);
      const double p1l = pe1(l), p1r = pe1(l + 1), dpl = p1r - p1l;
      const double a1 = q1(l), a2 = ws.at(S.SE, l), a3 = ws.at(S.SE, l + 1), a4 = 3. * (2. * a1 - (a2 + a3)), Sm = a4 + a3 - a2;
      const double pl = (p2t - p1l) / dpl;
      if (p2b <= p1r) {
        const double pr = (p2b - p1l) / dpl, s = pr + pl, w = pr * s + pl * pl;
        addq((1. - 0.5 * s) * g, 0.5 * s * g, (0.5 * s - R3 * w) * g);
        const double pr_ad = (0.5 * Sm - a4 * R3 * (2. * pr + pl)) * g, pl_ad = (0.5 * Sm - a4 * R3 * (pr + 2. * pl)) * g;
        p2k1 += pr_ad / dpl; aP -= pr_ad / dpl; aD -= pr * pr_ad / dpl;
        p2k += pl_ad / dpl; aP -= pl_ad / dpl; aD -= pl * pl_ad / dpl;
        continue;
      }
      const double D = p2b - p2t;
      const double E = a2 + 0.5 * Sm * (1. + pl) - a4 * (R3 * (1. + pl * (1. + pl)));
      double qsum = (p1r - p2t) * E;
      int m; bool bottom = false;
      for (m = l + 1; m <= km; ++m) {
        if (p2b > pe1(m + 1)) qsum += (pe1(m + 1) - pe1(m)) * q1(m);
        else { bottom = true; break; }
      }
      double dp = 0., esl = 0., Fm = 0., b2 = 0., b3 = 0., b4 = 0., dpm = 1.;
      if (bottom) {
        const double p1m = pe1(m); dpm = pe1(m + 1) - p1m;
        const double b1 = q1(m); b2 = ws.at(S.SE, m); b3 = ws.at(S.SE, m + 1); b4 = 3. * (2. * b1 - (b2 + b3));
        dp = p2b - p1m; esl = dp / dpm;
        Fm = b2 + 0.5 * esl * (b3 - b2 + b4 * (1. - R23 * esl));
        qsum += dp * Fm;
      }
      const double q2v = qsum / D, qs_ad = g / D, D_ad = -q2v * g / D;
      p2k1 += D_ad; p2k -= D_ad;
      const double W_ad = E * qs_ad, E_ad = (p1r - p2t) * qs_ad;
      aPn += W_ad; p2k -= W_ad;
      addq((1. - 0.5 * (1. + pl)) * E_ad, 0.5 * (1. + pl) * E_ad, (0.5 * (1. + pl) - R3 * (1. + pl * (1. + pl))) * E_ad);
      const double pl_ad = (0.5 * Sm - a4 * R3 * (1. + 2. * pl)) * E_ad;
      p2k += pl_ad / dpl; aP -= pl_ad / dpl; aD -= pl * pl_ad / dpl;
      if (bottom) {
        close_to(m, l, m, qs_ad);
        double dp_ad = Fm * qs_ad;
        const double F_ad = dp * qs_ad;
        addq((1. - 0.5 * esl) * F_ad, 0.5 * esl * F_ad, 0.5 * esl * (1. - R23 * esl) * F_ad);
        const double esl_ad = (0.5 * (b3 - b2 + b4 * (1. - R23 * esl)) - 0.5 * esl * b4 * R23) * F_ad;
        dp_ad += esl_ad / dpm; aD -= esl * esl_ad / dpm;
        p2k1 += dp_ad; aP -= dp_ad;
      } else {
        close_to(km + 1, l, km + 1, qs_ad);     // the target reaches below the last source layer: nothing further is found
      }
    }
    p2out(km, p2k); p2out(km + 1, p2k1);
    close_to(km + 1, 0, 0, 0.);
    fa_bot = aE + ecarry;                   // adjoint of the bottom edge value
    ws.at(S.SP, km + 1) = aP + dprev;
  }
  {   // ---- D
    double dpk = pe1(km + 1) - pe1(km), dpm1 = pe1(km) - pe1(km - 1);
    double d4 = dpm1 / dpk, gam_k = ws.at(S.SG, km);
    double sfa, sga, sda, qpk, qpkm1, dpend = 0., dnext = 0., q2hold = 0., d2loop = 0.;
    {
      const double d = d4, qfk = ws.at(S.SF, km);
      const double a_bot = 1. + d * (d + 1.5), den = d * (d + 0.5) - a_bot * gam_k;
      const double numb_ad = fa_bot / den, den_ad = -qbot * fa_bot / den;
      double d_ad = 2. * (2. * d + 1.) * q1(km) * numb_ad + (2. * d + 0.5) * den_ad;
      qpk = 2. * d * (d + 1.) * numb_ad; qpkm1 = numb_ad;
      const double abot_ad = -qfk * numb_ad - gam_k * den_ad;
      sfa = ws.at(S.SFA, km) - a_bot * numb_ad;
      sga = ws.at(S.SGA, km) - a_bot * den_ad;
      d_ad += (2. * d + 1.5) * abot_ad;
      sda = d_ad;
    }
    for (int k = km; k >= 2; --k) {
      const double gam = gam_k, gam_km1 = ws.at(S.SG, k - 1), qf = ws.at(S.SF, k);
      const double bet = 2. + d4 + d4 - gam_km1;
      gam_k = gam_km1;
      double d4_ad = sda + sga / bet;
      double bet_ad = -gam * sga / bet;
      const double t = sfa / bet;
      const double qk = ws.at(S.SQ, k) + qpk + 3. * d4 * t;
      if (k > 2) q1out(k, qk); else q2hold = qk;
      qpk = qpkm1 + 3. * t; qpkm1 = 0.;
      d4_ad += 3. * q1(k) * t;
      bet_ad -= qf * t;
      d4_ad += 2. * bet_ad;
      sfa = ws.at(S.SFA, k - 1) - t;
      sga = ws.at(S.SGA, k - 1) - bet_ad;
      sda = 0.;
      const double dk = dpend - d4 * d4_ad / dpk;       // dp1_ad(k) of this sweep
      dpend = d4_ad / dpk;
      if (k > 2) { p1out(k + 1, ws.at(S.SP, k + 1) + dk - dnext); dnext = dk; } else d2loop = dk;
      dpk = dpm1;
      if (k > 2) { dpm1 = pe1(k - 1) - pe1(k - 2); d4 = dpm1 / dpk; }
    }
    {
      const double gam1 = gam_k, qf1 = ws.at(S.SF, 1);
      double grat_ad = (2. * grat + 1.5) / bet1 * sga;
      double bet_ad = -gam1 * sga / bet1;
      const double t = sfa / bet1;
      bet_ad -= qf1 * t;
      grat_ad += 2. * (2. * grat + 1.) * q1(1) * t;
      q1out(2, q2hold + t);
      q1out(1, ws.at(S.SQ, 1) + qpk + 2. * grat * (grat + 1.) * t);
      grat_ad += (2. * grat + 0.5) * bet_ad;
      const double d2 = d2loop + grat_ad / dp1_1, d1 = dpend - grat * grat_ad / dp1_1;
      p1out(3, ws.at(S.SP, 3) + d2 - dnext);
      p1out(2, ws.at(S.SP, 2) + d1 - d2);
      p1out(1, ws.at(S.SP, 1) - d1);
    }
  }
}
// accumulating interface on the slots of MapAdSlots (non-hydrostatic remap, nh.h): SP1, SQ1, SP2 receive +=
template <class FP1, class FQ1, class FP2, class FAd>
HD void map_col_ad(int km, const FP1& pe1, const FQ1& q1, const FP2& pe2, const FAd& q2_ad, const ColWs& ws, const MapAdSlots& S) {
  map_col_ad_s(km, pe1, q1, pe2, q2_ad, ws, MapAdWs{S.SG, S.SF, S.SE, S.SFA, S.SGA, S.SEA, S.SDP},
               [&](int k, double x) { ws.at(S.SQ1, k) += x; }, [&](int k, double x) { ws.at(S.SP1, k) += x; }, [&](int k, double x) { ws.at(S.SP2, k) += x; });
}

// ---------------------------------------------------------------- kernels
struct RemapArgs {
  Geom g;
  Fld pe, peln, pk, pkz, pt, delp, u, v;   // state (in place)
  Fld pe2;                                 // new interface pressures (work, npz+1)
  Fld q[8]; int nq;
  const double *ak, *bk;                   // device [npz+1]
  double akap, zvir, ptop; int last_step;
  int kord_tm = -17, kord_mt = 17, kord_tr = 17;      // trajectory profiles; |kord| <= 16: the limited profile gives the values (split_kord)
  double* ws; size_t ws_stride;            // column workspace
  Fld pu_ad, pv_ad;                        // adjoint hand-over of the u/v maps to pe (npz+1 levels + ps slot)
};
HD size_t fidx(const Geom& g, const Fld& f, int tile, int i, int j, int k) { return ((size_t)(tile * f.nk + k - 1)) * g.plane + g.idx(i, j); }

// scalars: T (log p), tracers, delp, pk, peln, pkz, final pt (fv_mapz_tlm.F90:1586-1835, :2203-2250).
// Two launches: (1) one thread per (column, field) — field 0 maps T_v in log p, field n tracer n in p — each with its own
// workspace slots; (2) one thread per column for the new pressures and the final temperature conversion.
// LIM: the build of the kernels for a trajectory with limited profiles (split_kord); the default kernels carry none of that code
template <class T, bool LIM = false>
HD void remap_field_col(const RemapArgs& a, int field, int i, int j, int tile, size_t col) {
  const Geom& g = a.g; const int km = g.npz;
  const ColWs ws{a.ws + col, a.ws_stride, km + 2};
  typedef FIO<T> IO; typedef WsIO<T> W;
  const int SG = 3 * field, SE = 3 * field + 1, SO = 3 * field + 2;      // in units of T (x2 slots for Dual)
  auto pe1 = [&](int k) { return IO::ld(a.pe, fidx(g, a.pe, tile, i, j, k)); };
  auto pn1 = [&](int k) { return IO::ld(a.peln, fidx(g, a.peln, tile, i, j, k)); };
  auto pk1 = [&](int k) { return IO::ld(a.pk, fidx(g, a.pk, tile, i, j, k)); };
  const T ps = pe1(km + 1);
  auto pe2 = [&](int k) -> T { return k == 1 ? T(a.ptop) : (k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps); };
  auto outw = [&](int k, const T& x) { W::set(ws, SO, k, x); };
  if (field == 0) {
    auto pn2 = [&](int k) -> T { return (k == 1 || k == km + 1) ? pn1(k) : dlog(pe2(k)); };
    auto tv = [&](int k) -> T { return IO::ld(a.pt, fidx(g, a.pt, tile, i, j, k)) * (pk1(k + 1) - pk1(k)) / (a.akap * (pn1(k + 1) - pn1(k))); };
    if (!(LIM && kord_limited(a.kord_tm) && W::W == 1)) map_col<T>(km, pn1, tv, pn2, outw, ws, SG, SE);
    if constexpr (LIM) if (kord_limited(a.kord_tm)) {      // the values by the trajectory's limited profile (map_scalar, iv = 1, t_min = 184: fv_mapz_tlm.F90:494-509)
      auto pn1d = [&](int k) { return a.peln.t[fidx(g, a.peln, tile, i, j, k)]; };
      auto pk1d = [&](int k) { return a.pk.t[fidx(g, a.pk, tile, i, j, k)]; };
      const double psd = a.pe.t[fidx(g, a.pe, tile, i, j, km + 1)];
      auto pn2d = [&](int k) -> double { return (k == 1 || k == km + 1) ? pn1d(k) : log(k == 1 ? a.ptop : a.ak[k - 1] + a.bk[k - 1] * psd); };
      auto tvd = [&](int k) -> double { return a.pt.t[fidx(g, a.pt, tile, i, j, k)] * (pk1d(k + 1) - pk1d(k)) / (a.akap * (pn1d(k + 1) - pn1d(k))); };
      map_col_lim(LimProfile{a.kord_tm, 1, true, 184.}, km, pn1d, tvd, pn2d, [&](int k, double x) { ws.at(W::W * SO, k) = x; }, ws, W::W * SG, W::W * SE);
    }
    for (int k = 1; k <= km; ++k) IO::st(a.pt, fidx(g, a.pt, tile, i, j, k), W::get(ws, SO, k));   // T_v on the new levels
  } else {
    const Fld& qf = a.q[field - 1];
    auto q1 = [&](int k) { return IO::ld(qf, fidx(g, qf, tile, i, j, k)); };
    if (!(LIM && kord_limited(a.kord_tr) && W::W == 1)) map_col<T>(km, pe1, q1, pe2, outw, ws, SG, SE);
    if constexpr (LIM) if (kord_limited(a.kord_tr)) {      // map1_q2: scalar_profile, iv = 0, q_min = 0 (fv_mapz_tlm.F90:596-612)
      auto pe1d = [&](int k) { return a.pe.t[fidx(g, a.pe, tile, i, j, k)]; };
      const double psd = pe1d(km + 1);
      auto pe2d = [&](int k) -> double { return k == 1 ? a.ptop : (k == km + 1 ? psd : a.ak[k - 1] + a.bk[k - 1] * psd); };
      auto q1d = [&](int k) { return qf.t[fidx(g, qf, tile, i, j, k)]; };
      map_col_lim(LimProfile{a.kord_tr, 0, true, 0.}, km, pe1d, q1d, pe2d, [&](int k, double x) { ws.at(W::W * SO, k) = x; }, ws, W::W * SG, W::W * SE);
    }
    for (int k = 1; k <= km; ++k) IO::st(qf, fidx(g, qf, tile, i, j, k), W::get(ws, SO, k));
  }
}
template <class T>
HD void remap_press_col(const RemapArgs& a, int i, int j, int tile) {
  const Geom& g = a.g; const int km = g.npz;
  typedef FIO<T> IO;
  auto pe1 = [&](int k) { return IO::ld(a.pe, fidx(g, a.pe, tile, i, j, k)); };
  auto pn1 = [&](int k) { return IO::ld(a.peln, fidx(g, a.peln, tile, i, j, k)); };
  auto pk1 = [&](int k) { return IO::ld(a.pk, fidx(g, a.pk, tile, i, j, k)); };
  const T ps = pe1(km + 1);
  auto pe2 = [&](int k) -> T { return k == 1 ? T(a.ptop) : (k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps); };
  // new pressures; pk/peln end levels keep their values (:1650-1661)
  T pk_hi = pk1(1), pn_hi = pn1(1), pe_hi = pe2(1);
  const T pn_bot = pn1(km + 1), pk_bot = pk1(km + 1);
  IO::st(a.pe2, fidx(g, a.pe2, tile, i, j, 1), pe_hi);
  for (int k = 1; k <= km; ++k) {
    const T pe_lo = pe2(k + 1);
    T pn_lo, pk_lo;
    if (k == km) { pn_lo = pn_bot; pk_lo = pk_bot; } else { pn_lo = dlog(pe_lo); pk_lo = dexp(a.akap * pn_lo); }
    const T pkz = (pk_lo - pk_hi) / (a.akap * (pn_lo - pn_hi));
    const size_t n0 = fidx(g, a.pt, tile, i, j, k);
    IO::st(a.delp, n0, pe_lo - pe_hi);
    IO::st(a.pkz, n0, pkz);
    const T t2 = IO::ld(a.pt, n0);
    if (a.last_step) IO::st(a.pt, n0, t2 / (1. + a.zvir * (a.nq > 0 ? IO::ld(a.q[0], n0) : T(0.))));
    else IO::st(a.pt, n0, t2 / pkz);
    if (k > 1) { IO::st(a.peln, fidx(g, a.peln, tile, i, j, k), pn_hi); IO::st(a.pk, fidx(g, a.pk, tile, i, j, k), pk_hi); }
    IO::st(a.pe2, fidx(g, a.pe2, tile, i, j, k + 1), pe_lo);
    pe_hi = pe_lo; pn_hi = pn_lo; pk_hi = pk_lo;
  }
}

// u (dir=0, points i=1..nx, j=1..ny+1) and v (dir=1, i=1..nx+1, j=1..ny) on pressures averaged across the
// edge (fv_mapz_tlm.F90:1884-1934)
template <class T, bool LIM = false>
HD void remap_wind_col(const RemapArgs& a, int dir, int i, int j, int tile, size_t col) {
  const Geom& g = a.g; const int km = g.npz;
  const ColWs ws{a.ws + col, a.ws_stride, km + 2};
  typedef FIO<T> IO; typedef WsIO<T> W;
  const int SG = 0, SE = 1, SO = 2;
  const int im = dir == 0 ? i : i - 1, jm = dir == 0 ? j - 1 : j;
  const Fld& wf = dir == 0 ? a.u : a.v;
  auto pe0 = [&](int k) -> T {
    if (k == 1) return IO::ld(a.pe, fidx(g, a.pe, tile, i, j, 1));
    return 0.5 * (IO::ld(a.pe, fidx(g, a.pe, tile, im, jm, k)) + IO::ld(a.pe, fidx(g, a.pe, tile, i, j, k)));
  };
  const T pss = IO::ld(a.pe, fidx(g, a.pe, tile, im, jm, km + 1)) + IO::ld(a.pe, fidx(g, a.pe, tile, i, j, km + 1));
  auto pe3 = [&](int k) -> T { return (dir == 1 && k == 1) ? T(a.ak[0]) : a.ak[k - 1] + (0.5 * a.bk[k - 1]) * pss; };
  auto q1 = [&](int k) { return IO::ld(wf, fidx(g, wf, tile, i, j, k)); };
  auto outw = [&](int k, const T& x) { W::set(ws, SO, k, x); };
  if (!(LIM && kord_limited(a.kord_mt) && W::W == 1)) map_col<T>(km, pe0, q1, pe3, outw, ws, SG, SE);
  if constexpr (LIM) if (kord_limited(a.kord_mt)) {        // map1_ppm: cs_profile, iv = -1 (fv_mapz_tlm.F90:780-795, :817-830)
    auto pe0d = [&](int k) -> double {
      if (k == 1) return a.pe.t[fidx(g, a.pe, tile, i, j, 1)];
      return 0.5 * (a.pe.t[fidx(g, a.pe, tile, im, jm, k)] + a.pe.t[fidx(g, a.pe, tile, i, j, k)]);
    };
    const double pssd = a.pe.t[fidx(g, a.pe, tile, im, jm, km + 1)] + a.pe.t[fidx(g, a.pe, tile, i, j, km + 1)];
    auto pe3d = [&](int k) -> double { return (dir == 1 && k == 1) ? a.ak[0] : a.ak[k - 1] + (0.5 * a.bk[k - 1]) * pssd; };
    auto q1d = [&](int k) { return wf.t[fidx(g, wf, tile, i, j, k)]; };
    map_col_lim(LimProfile{a.kord_mt, -1, false, 0.}, km, pe0d, q1d, pe3d, [&](int k, double x) { ws.at(W::W * SO, k) = x; }, ws, W::W * SG, W::W * SE);
  }
  for (int k = 1; k <= km; ++k) IO::st(wf, fidx(g, wf, tile, i, j, k), W::get(ws, SO, k));
}

// adjoint of remap_wind_col: consumes wf.p (adjoint of the mapped wind), writes the adjoint of the
// input wind back into wf.p and hands the pressure adjoints to pu_ad / pv_ad (npz+1 levels of
// d/d pe0(k); level slot km+2 -> here stored at k = km+2 as the adjoint of the surface-pressure sum).
HD void remap_wind_col_ad(const RemapArgs& a, int dir, int i, int j, int tile, size_t col) {
  const Geom& g = a.g; const int km = g.npz;
  const ColWs ws{a.ws + col, a.ws_stride, km + 2};
  const int im = dir == 0 ? i : i - 1, jm = dir == 0 ? j - 1 : j;
  const Fld& wf = dir == 0 ? a.u : a.v;
  const Fld& ho = dir == 0 ? a.pu_ad : a.pv_ad;
  auto pe0 = [&](int k) -> double {
    if (k == 1) return a.pe.t[fidx(g, a.pe, tile, i, j, 1)];
    return 0.5 * (a.pe.t[fidx(g, a.pe, tile, im, jm, k)] + a.pe.t[fidx(g, a.pe, tile, i, j, k)]);
  };
  const double pss = a.pe.t[fidx(g, a.pe, tile, im, jm, km + 1)] + a.pe.t[fidx(g, a.pe, tile, i, j, km + 1)];
  auto pe3 = [&](int k) -> double { return (dir == 1 && k == 1) ? a.ak[0] : a.ak[k - 1] + (0.5 * a.bk[k - 1]) * pss; };
  auto q1 = [&](int k) { return wf.t[fidx(g, wf, tile, i, j, k)]; };
  auto q2ad = [&](int k) { return wf.p[fidx(g, wf, tile, i, j, k)]; };
  double pss_ad = 0.;
  // the adjoint of the input wind replaces wf.p; hand-over: level k (1..km+1) = d/d pe0(k); the ps-sum adjoint is folded into
  // level km+1 of both neighbours
  map_col_ad_s(km, pe0, q1, pe3, q2ad, ws, MapAdWs{0, 1, 2, 3, 4, 5, 6},
               [&](int k, double x) { wf.p[fidx(g, wf, tile, i, j, k)] = x; },
               [&](int k, double x) { ho.p[fidx(g, ho, tile, i, j, k)] = x; },
               [&](int k, double x) { if (!(dir == 1 && k == 1)) pss_ad += 0.5 * a.bk[k - 1] * x; });
  ho.t[fidx(g, ho, tile, i, j, 1)] = pss_ad;   // stored in the (otherwise unused) trajectory side, level 1
}

// gather of the wind-map pressure adjoints into pe.p on is-1..ie+1, js-1..je+1
HD void remap_pe_gather_ad(const RemapArgs& a, int i, int j, int tile) {
  const Geom& g = a.g; const int km = g.npz;
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je();
  const bool u0 = (i >= is && i <= ie && j >= js && j <= je + 1), u1 = (i >= is && i <= ie && j + 1 >= js && j + 1 <= je + 1);
  const bool v0 = (i >= is && i <= ie + 1 && j >= js && j <= je), v1 = (i + 1 >= is && i + 1 <= ie + 1 && j >= js && j <= je);
  for (int k = 1; k <= km + 1; ++k) {
    double acc = 0.;
    // u point (i,j) uses pe(i,j-1) and pe(i,j); u point (i,j+1) uses pe(i,j) and pe(i,j+1)
    if (k == 1) {
      if (u0) acc += a.pu_ad.p[fidx(g, a.pu_ad, tile, i, j, 1)];
      if (v0) acc += a.pv_ad.p[fidx(g, a.pv_ad, tile, i, j, 1)];
    } else {
      if (u0) acc += 0.5 * a.pu_ad.p[fidx(g, a.pu_ad, tile, i, j, k)];
      if (u1) acc += 0.5 * a.pu_ad.p[fidx(g, a.pu_ad, tile, i, j + 1, k)];
      if (v0) acc += 0.5 * a.pv_ad.p[fidx(g, a.pv_ad, tile, i, j, k)];
      if (v1) acc += 0.5 * a.pv_ad.p[fidx(g, a.pv_ad, tile, i + 1, j, k)];
    }
    if (k == km + 1) {
      if (u0) acc += a.pu_ad.t[fidx(g, a.pu_ad, tile, i, j, 1)];
      if (u1) acc += a.pu_ad.t[fidx(g, a.pu_ad, tile, i, j + 1, 1)];
      if (v0) acc += a.pv_ad.t[fidx(g, a.pv_ad, tile, i, j, 1)];
      if (v1) acc += a.pv_ad.t[fidx(g, a.pv_ad, tile, i + 1, j, 1)];
    }
    a.pe.p[fidx(g, a.pe, tile, i, j, k)] += acc;
  }
}

// adjoint of remap_scalars_col.  Trajectory inputs (pre-remap pe, peln, pk, pt, q) must be in place.
// Incoming adjoints: pt.p, q[n].p, delp.p, pk.p, peln.p, pkz.p (of the remapped fields); outgoing:
// pt.p, q[n].p (of the inputs), pk.p, peln.p replaced, pe.p accumulated, delp.p = pkz.p = 0.
// Adjoint of the scalar remap in three launches (the column maps of the different fields are independent):
//   k1, per column:          trajectory of the remapped T_v and q_v, final conversion, pkz / pk / peln / delp adjoints
//   k2, per (column, field): adjoint of one column map (field 0: T_v in log p; field n: tracer n in p), own workspace slots
//   k3, per column:          sums the pressure adjoints of all maps, T_v -> pt, pk2 / pn2 chain, write-back
// Workspace slots: 0..21 shared accumulators (as named below), 22 + 8 f .. for the map of field f: its pe1_ad, q1_ad (T map only),
// pe2_ad outputs and the five work vectors of map_col_ad_s (whose SP / SQ streams share the output vectors).
struct RemapAdSlots { static constexpr int SPE1 = 13, SPN1 = 14, SPK1 = 15, SPE2 = 16, SPN2 = 17, SPK2 = 18, ST2 = 19, SQ2 = 20, SV = 21, FBASE = 22, FN = 8; };
struct FieldAdSlots { int SP1, SQ1, SP2; MapAdWs w; };
HD FieldAdSlots remap_field_slots(int f) { const int b = RemapAdSlots::FBASE + RemapAdSlots::FN * f; return FieldAdSlots{b, b + 1, b + 2, MapAdWs{b + 3, b + 4, b + 5, b + 6, b + 7, b + 1, b}}; }

template <bool LIM = false>
HD void remap_ad_k1(const RemapArgs& a, int i, int j, int tile, size_t col) {
  const Geom& g = a.g; const int km = g.npz;
  const ColWs ws{a.ws + col, a.ws_stride, km + 2};
  typedef RemapAdSlots R_;
  auto pe1 = [&](int k) { return a.pe.t[fidx(g, a.pe, tile, i, j, k)]; };
  auto pn1 = [&](int k) { return a.peln.t[fidx(g, a.peln, tile, i, j, k)]; };
  auto pk1 = [&](int k) { return a.pk.t[fidx(g, a.pk, tile, i, j, k)]; };
  const double ps = pe1(km + 1);
  auto pe2 = [&](int k) -> double { return k == 1 ? a.ptop : (k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps); };
  auto pn2 = [&](int k) -> double { return (k == 1 || k == km + 1) ? pn1(k) : log(pe2(k)); };
  auto pk2 = [&](int k) -> double { return (k == 1 || k == km + 1) ? pk1(k) : exp(a.akap * pn2(k)); };
  auto ptin = [&](int k) { return a.pt.t[fidx(g, a.pt, tile, i, j, k)]; };
  auto tv = [&](int k) -> double { return ptin(k) * (pk1(k + 1) - pk1(k)) / (a.akap * (pn1(k + 1) - pn1(k))); };
  for (int k = 0; k <= km + 1; ++k)
    for (int s_ : {R_::SPE1, R_::SPN1, R_::SPK1, R_::SPE2, R_::SPN2, R_::SPK2}) ws.at(s_, k) = 0.;
  // trajectory of the remapped T_v and q_v (needed by the final conversion)
  {
    auto outT = [&](int k, double x) { ws.at(R_::ST2, k) = x; };
    bool done = false;
    if constexpr (LIM) if (kord_limited(a.kord_tm)) { map_col_lim(LimProfile{a.kord_tm, 1, true, 184.}, km, pn1, tv, pn2, outT, ws, 0, 1); done = true; }
    if (!done) map_col<double>(km, pn1, tv, pn2, outT, ws, 0, 1);
    if (a.nq > 0 && a.last_step) {
      auto q1 = [&](int k) { return a.q[0].t[fidx(g, a.q[0], tile, i, j, k)]; };
      auto outQ = [&](int k, double x) { ws.at(R_::SQ2, k) = x; };
      bool doneq = false;
      if constexpr (LIM) if (kord_limited(a.kord_tr)) { map_col_lim(LimProfile{a.kord_tr, 0, true, 0.}, km, pe1, q1, pe2, outQ, ws, 0, 1); doneq = true; }
      if (!doneq) map_col<double>(km, pe1, q1, pe2, outQ, ws, 0, 1);
    }
  }
  // final conversion, pkz, pk/peln outputs, delp
  for (int k = 1; k <= km; ++k) {
    const size_t n0 = fidx(g, a.pt, tile, i, j, k);
    const double pk_hi = pk2(k), pk_lo = pk2(k + 1), pn_hi = pn2(k), pn_lo = pn2(k + 1);
    const double den = a.akap * (pn_lo - pn_hi), pkz = (pk_lo - pk_hi) / den;
    const double pt_ad = a.pt.p[n0], t2 = ws.at(R_::ST2, k);
    double pkz_ad = a.pkz.p[n0], t2_ad;
    if (a.last_step) {
      const double qv = a.nq > 0 ? ws.at(R_::SQ2, k) : 0., f = 1. + a.zvir * qv;
      t2_ad = pt_ad / f;
      if (a.nq > 0) a.q[0].p[fidx(g, a.q[0], tile, i, j, k)] += -(t2 / f) * a.zvir * pt_ad / f;
    } else {
      t2_ad = pt_ad / pkz;
      pkz_ad += -(t2 / pkz) * pt_ad / pkz;
    }
    ws.at(R_::SV, k) = t2_ad;
    const double za = pkz_ad / den;
    ws.at(R_::SPK2, k + 1) += za; ws.at(R_::SPK2, k) -= za;
    const double zb = -pkz * a.akap * za;
    ws.at(R_::SPN2, k + 1) += zb; ws.at(R_::SPN2, k) -= zb;
    const double dpa = a.delp.p[n0];
    ws.at(R_::SPE2, k + 1) += dpa; ws.at(R_::SPE2, k) -= dpa;
    a.delp.p[n0] = 0.; a.pkz.p[n0] = 0.;
  }
  for (int k = 2; k <= km; ++k) {   // pk_out(k) = pk2(k), peln_out(k) = pn2(k) for interior interfaces
    ws.at(R_::SPK2, k) += a.pk.p[fidx(g, a.pk, tile, i, j, k)];
    ws.at(R_::SPN2, k) += a.peln.p[fidx(g, a.peln, tile, i, j, k)];
  }
}
HD void remap_ad_k2(const RemapArgs& a, int field, int i, int j, int tile, size_t col) {
  const Geom& g = a.g; const int km = g.npz;
  const ColWs ws{a.ws + col, a.ws_stride, km + 2};
  const FieldAdSlots S = remap_field_slots(field);
  auto pe1 = [&](int k) { return a.pe.t[fidx(g, a.pe, tile, i, j, k)]; };
  const double ps = pe1(km + 1);
  auto pe2 = [&](int k) -> double { return k == 1 ? a.ptop : (k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps); };
  auto p1out = [&](int k, double x) { ws.at(S.SP1, k) = x; };
  auto p2out = [&](int k, double x) { ws.at(S.SP2, k) = x; };
  if (field == 0) {   // T map (coordinates pn1 -> pn2)
    auto pn1 = [&](int k) { return a.peln.t[fidx(g, a.peln, tile, i, j, k)]; };
    auto pk1 = [&](int k) { return a.pk.t[fidx(g, a.pk, tile, i, j, k)]; };
    auto pn2 = [&](int k) -> double { return (k == 1 || k == km + 1) ? pn1(k) : log(pe2(k)); };
    auto tv = [&](int k) -> double { return a.pt.t[fidx(g, a.pt, tile, i, j, k)] * (pk1(k + 1) - pk1(k)) / (a.akap * (pn1(k + 1) - pn1(k))); };
    auto q2ad = [&](int k) { return ws.at(RemapAdSlots::SV, k); };
    map_col_ad_s(km, pn1, tv, pn2, q2ad, ws, S.w, [&](int k, double x) { ws.at(S.SQ1, k) = x; }, p1out, p2out);
  } else {            // tracer map (coordinates pe1 -> pe2)
    const Fld& qf = a.q[field - 1];
    auto q1 = [&](int k) { return qf.t[fidx(g, qf, tile, i, j, k)]; };
    auto q2ad = [&](int k) { return qf.p[fidx(g, qf, tile, i, j, k)]; };
    map_col_ad_s(km, pe1, q1, pe2, q2ad, ws, S.w, [&](int k, double x) { qf.p[fidx(g, qf, tile, i, j, k)] = x; }, p1out, p2out);
  }
}
HD void remap_ad_k3(const RemapArgs& a, int i, int j, int tile, size_t col) {
  const Geom& g = a.g; const int km = g.npz;
  const ColWs ws{a.ws + col, a.ws_stride, km + 2};
  typedef RemapAdSlots R_;
  auto pe1 = [&](int k) { return a.pe.t[fidx(g, a.pe, tile, i, j, k)]; };
  auto pn1 = [&](int k) { return a.peln.t[fidx(g, a.peln, tile, i, j, k)]; };
  auto pk1 = [&](int k) { return a.pk.t[fidx(g, a.pk, tile, i, j, k)]; };
  const double ps = pe1(km + 1);
  auto pe2 = [&](int k) -> double { return k == 1 ? a.ptop : (k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps); };
  auto pn2 = [&](int k) -> double { return (k == 1 || k == km + 1) ? pn1(k) : log(pe2(k)); };
  auto pk2 = [&](int k) -> double { return (k == 1 || k == km + 1) ? pk1(k) : exp(a.akap * pn2(k)); };
  {   // T map: pressure adjoints into the log-p accumulators, T_v(k) = pt * dpk / (akap * dln)
    const FieldAdSlots S = remap_field_slots(0);
    for (int k = 1; k <= km + 1; ++k) { ws.at(R_::SPN1, k) += ws.at(S.SP1, k); ws.at(R_::SPN2, k) += ws.at(S.SP2, k); }
    for (int k = 1; k <= km; ++k) {
      const double tv_ad = ws.at(S.SQ1, k), dpk = pk1(k + 1) - pk1(k), den = a.akap * (pn1(k + 1) - pn1(k)), pt0 = a.pt.t[fidx(g, a.pt, tile, i, j, k)];
      a.pt.p[fidx(g, a.pt, tile, i, j, k)] = tv_ad * dpk / den;
      const double za = pt0 * tv_ad / den;
      ws.at(R_::SPK1, k + 1) += za; ws.at(R_::SPK1, k) -= za;
      const double zb = -(pt0 * dpk / den) * a.akap * tv_ad / den;
      ws.at(R_::SPN1, k + 1) += zb; ws.at(R_::SPN1, k) -= zb;
    }
  }
  for (int n = 0; n < a.nq; ++n) {   // tracer maps: pressure adjoints, in tracer order
    const FieldAdSlots S = remap_field_slots(1 + n);
    for (int k = 1; k <= km + 1; ++k) { ws.at(R_::SPE1, k) += ws.at(S.SP1, k); ws.at(R_::SPE2, k) += ws.at(S.SP2, k); }
  }
  // pk2 = exp(akap pn2), pn2 = log pe2 on interior interfaces; end levels pass through to pk1/pn1
  for (int k = 2; k <= km; ++k) {
    const double pn_ad = ws.at(R_::SPN2, k) + a.akap * pk2(k) * ws.at(R_::SPK2, k);
    ws.at(R_::SPE2, k) += pn_ad / pe2(k);
  }
  for (int k : {1, km + 1}) {   // end interfaces pass through: pk_out = pk1, peln_out = pn1
    ws.at(R_::SPN1, k) += ws.at(R_::SPN2, k) + a.peln.p[fidx(g, a.peln, tile, i, j, k)];
    ws.at(R_::SPK1, k) += ws.at(R_::SPK2, k) + a.pk.p[fidx(g, a.pk, tile, i, j, k)];
  }
  double ps_ad = ws.at(R_::SPE2, km + 1);
  for (int k = 2; k <= km; ++k) ps_ad += a.bk[k - 1] * ws.at(R_::SPE2, k);
  ws.at(R_::SPE1, km + 1) += ps_ad;
  // write back: pk.p / peln.p replaced by the input adjoints, pe.p accumulated
  for (int k = 1; k <= km + 1; ++k) {
    a.pk.p[fidx(g, a.pk, tile, i, j, k)] = ws.at(R_::SPK1, k);
    a.peln.p[fidx(g, a.peln, tile, i, j, k)] = ws.at(R_::SPN1, k);
    a.pe.p[fidx(g, a.pe, tile, i, j, k)] += ws.at(R_::SPE1, k);
  }
}

// one kernel per mode (separate register allocations)
template <int MODE, bool LIM = false>
struct RemapFieldFn {      // z = tile * nf + field
  RemapArgs a; int nf;
  HD void operator()(int i, int j, int z) const {
    const int tile = z / nf, field = z % nf;
    const size_t col = (size_t)tile * a.g.plane + a.g.idx(i, j);
    if (MODE == MODE_NL) remap_field_col<double, LIM>(a, field, i, j, tile, col);
    else if (MODE == MODE_TL) remap_field_col<Dual, LIM>(a, field, i, j, tile, col);
    else remap_ad_k2(a, field, i, j, tile, col);
  }
};
template <int MODE, bool LIM = false>        // MODE_AD: stage 1 of the adjoint; 3: stage 3
struct RemapPressFn {
  RemapArgs a;
  HD void operator()(int i, int j, int z) const {
    const size_t col = (size_t)z * a.g.plane + a.g.idx(i, j);
    if (MODE == MODE_NL) remap_press_col<double>(a, i, j, z);
    else if (MODE == MODE_TL) remap_press_col<Dual>(a, i, j, z);
    else if (MODE == MODE_AD) remap_ad_k1<LIM>(a, i, j, z, col);
    else remap_ad_k3(a, i, j, z, col);
  }
};
template <int MODE, bool LIM = false>
struct RemapWindFn {
  RemapArgs a; int dir;
  HD void operator()(int i, int j, int z) const {
    const size_t col = (size_t)z * a.g.plane + a.g.idx(i, j);
    if (MODE == MODE_NL) remap_wind_col<double, LIM>(a, dir, i, j, z, col);
    else if (MODE == MODE_TL) remap_wind_col<Dual, LIM>(a, dir, i, j, z, col);
    else remap_wind_col_ad(a, dir, i, j, z, col);
  }
};
inline void run_remap_winds(Exec& ex, int mode, const RemapArgs& a, double bytes = 0.) {
  const Geom& g = a.g;
  const Rect U{g.is(), g.ie(), g.js(), g.je() + 1}, V{g.is(), g.ie() + 1, g.js(), g.je()};
  if (kord_limited(a.kord_mt) && mode != MODE_AD) {      // split_kord: the kernels that also run the trajectory's limited profile
    if (mode == MODE_NL) { for_points(ex, U, g.ntile, RemapWindFn<MODE_NL, true>{a, 0}, "remap_wind_lim.nl", bytes); for_points(ex, V, g.ntile, RemapWindFn<MODE_NL, true>{a, 1}, "remap_wind_lim.nl", bytes); }
    else { for_points(ex, U, g.ntile, RemapWindFn<MODE_TL, true>{a, 0}, "remap_wind_lim.tl", bytes); for_points(ex, V, g.ntile, RemapWindFn<MODE_TL, true>{a, 1}, "remap_wind_lim.tl", bytes); }
  }
  else if (mode == MODE_NL) { for_points(ex, U, g.ntile, RemapWindFn<MODE_NL>{a, 0}, "remap_wind.nl", bytes); for_points(ex, V, g.ntile, RemapWindFn<MODE_NL>{a, 1}, "remap_wind.nl", bytes); }
  else if (mode == MODE_TL) { for_points(ex, U, g.ntile, RemapWindFn<MODE_TL>{a, 0}, "remap_wind.tl", bytes); for_points(ex, V, g.ntile, RemapWindFn<MODE_TL>{a, 1}, "remap_wind.tl", bytes); }
  else { for_points(ex, V, g.ntile, RemapWindFn<MODE_AD>{a, 1}, "remap_wind.ad", bytes); for_points(ex, U, g.ntile, RemapWindFn<MODE_AD>{a, 0}, "remap_wind.ad", bytes); }
}
struct RemapGatherFn {
  RemapArgs a;
  HD void operator()(int i, int j, int z) const { remap_pe_gather_ad(a, i, j, z); }
};
// pe(k) <- pe2(k), k = 2..km, on the compute domain (fv_mapz_tlm.F90:1944-1950)
struct RemapPeFn {
  RemapArgs a; int mode;
  HD void operator()(int i, int j, int z) const {
    for (int k = 2; k <= a.g.npz; ++k) {
      const size_t n = fidx(a.g, a.pe, z, i, j, k);
      a.pe.t[n] = a.pe2.t[n];
      if (mode == MODE_TL) a.pe.p[n] = a.pe2.p[n];
    }
  }
};

// Workspace: 22 shared slots + 8 per scalar field (T and the tracers) of (npz+2) doubles per column, columns = ntile*plane.
inline int remap_ws_slots(int nq) { return RemapAdSlots::FBASE + RemapAdSlots::FN * (1 + nq); }
inline void run_remap(Exec& ex, int mode, const RemapArgs& a) {
  const Geom& g = a.g;
  const Rect A{g.is(), g.ie(), g.js(), g.je()}, H{g.is() - 1, g.ie() + 1, g.js() - 1, g.je() + 1};
  if (mode != MODE_AD) {
    const double cells = double(g.tx) * g.ty * g.ntile * g.npz, w = mode == MODE_TL ? 2. : 1.;
    // algorithmic bytes: scalars read pe,peln,pk,pt,q[nq]; write pt,q[nq],delp,pk,peln,pkz,pe2; winds read pe x2, u|v; write u|v
    const int nf = 1 + a.nq;
    const bool lim = kord_limited(a.kord_tm) || kord_limited(a.kord_tr);
    if (lim && mode == MODE_TL) {
      for_points(ex, A, g.ntile * nf, RemapFieldFn<MODE_TL, true>{a, nf}, "remap_fields_lim.tl", 8. * w * (5. + 2. * a.nq) * cells);
      for_points(ex, A, g.ntile, RemapPressFn<MODE_TL>{a}, "remap_press.tl", 8. * w * (5. + 7.) * cells);
    } else if (lim) {
      for_points(ex, A, g.ntile * nf, RemapFieldFn<MODE_NL, true>{a, nf}, "remap_fields_lim.nl", 8. * w * (5. + 2. * a.nq) * cells);
      for_points(ex, A, g.ntile, RemapPressFn<MODE_NL>{a}, "remap_press.nl", 8. * w * (5. + 7.) * cells);
    } else if (mode == MODE_TL) {
      for_points(ex, A, g.ntile * nf, RemapFieldFn<MODE_TL>{a, nf}, "remap_fields.tl", 8. * w * (5. + 2. * a.nq) * cells);
      for_points(ex, A, g.ntile, RemapPressFn<MODE_TL>{a}, "remap_press.tl", 8. * w * (5. + 7.) * cells);
    } else {
      for_points(ex, A, g.ntile * nf, RemapFieldFn<MODE_NL>{a, nf}, "remap_fields.nl", 8. * w * (5. + 2. * a.nq) * cells);
      for_points(ex, A, g.ntile, RemapPressFn<MODE_NL>{a}, "remap_press.nl", 8. * w * (5. + 7.) * cells);
    }
    run_remap_winds(ex, mode, a, 8. * w * 4. * cells);
    for_points(ex, A, g.ntile, RemapPeFn{a, mode}, "remap_pe");
  } else {
    const double cells = double(g.tx) * g.ty * g.ntile * g.npz;
    run_remap_winds(ex, MODE_AD, a, 8. * 7. * cells);
    for_points(ex, H, g.ntile, RemapGatherFn{a}, "remap_gather.ad", 8. * 4. * cells);
    const int nf = 1 + a.nq;
    if (kord_limited(a.kord_tm) || kord_limited(a.kord_tr)) for_points(ex, A, g.ntile, RemapPressFn<MODE_AD, true>{a}, "remap_scalars1_lim.ad", 8. * 12. * cells);
    else for_points(ex, A, g.ntile, RemapPressFn<MODE_AD>{a}, "remap_scalars1.ad", 8. * 12. * cells);
    for_points(ex, A, g.ntile * nf, RemapFieldFn<MODE_AD>{a, nf}, "remap_fields.ad", 8. * (4. + 3. * a.nq) * cells);
    for_points(ex, A, g.ntile, RemapPressFn<3>{a}, "remap_scalars3.ad", 8. * 8. * cells);
  }
}

}  // namespace fv3
