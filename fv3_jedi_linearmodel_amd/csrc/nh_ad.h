// fv3lm-hip: hand-written adjoints of the two vertically implicit solvers of the non-hydrostatic acoustic step
// (riem_c_col / riem3_col of nh.h).  Same statements as the generic versions, replayed on `double` with every
// intermediate kept in the column workspace, then reversed statement by statement — ≈ 20 stored values per level instead
// of the ≈ 90 tape entries of the taped run (coltape.h), which stays available (FV3LM_NH_TAPE=1) and is what the tests
// compare these against.
//   RIEM_SOLVER_C + SIM1_SOLVER  nh_utils_tlm.F90:846-933, :2760-2884     (adjoint: nh_utils_adm.F90:4769)
//   RIEM_SOLVER3 + SIM_SOLVER    nh_core_tlm.F90:245-389, nh_utils_tlm.F90:3129-3272   (adjoint: nh_core_adm.F90:75/302)
#pragma once
#include "nh.h"

namespace fv3 {

// raw workspace slots (km+2 doubles each; NH_WS_SLOTS of them per column, nh.h).  HS_*: forward values; HA_*: adjoints.  Some HS_
// slots are reused for adjoints once their forward value is dead (noted where it happens).
enum { HS_DZ = 0, HS_PEM, HS_PM, HS_DM, HS_PT, HS_W1, HS_P, HS_GR, HS_GAM, HS_BET, HS_PP, HS_AA, HS_GM2, HS_BET2, HS_W2F, HS_W2, HS_PE, HS_P1,
       HS_DZN, HS_WK, HS_CL, HA_DZ, HA_PEM, HA_PM, HA_DM, HA_PT, HA_PP, HA_PE, HA_W1, HA_GR, HA_PQ, HS_COUNT };
static_assert(HS_COUNT <= NH_WS_SLOTS, "column workspace too small for the hand-written adjoints");

struct NhAdCol {
  const ColWs& ws; int km;
  HD double& operator()(int slot, int k) const { return ws.at(slot, k); }
};

// ---- shared forward pieces (values stored) -------------------------------------------------------------------------------
// interface pressure perturbation pp(1..km+1) from the layer values (first Thomas sweep)
HD void nhad_pp_edges_fwd(const NhAdCol& W, double rgas, double gama) {
  const int km = W.km;
  for (int k = 1; k <= km; ++k) W(HS_P, k) = exp(gama * log(-(W(HS_DM, k) / W(HS_DZ, k) * rgas * W(HS_PT, k))));      // pe + pm
  for (int k = 1; k <= km - 1; ++k) W(HS_GR, k) = W(HS_DM, k) / W(HS_DM, k + 1);
  auto pq = [&](int k) { return W(HS_P, k) - W(HS_PM, k); };
  auto bb = [&](int k) { return k == km ? 2. : 2. * (1. + W(HS_GR, k)); };
  auto dd = [&](int k) { return k == km ? 3. * pq(km) : 3. * (pq(k) + W(HS_GR, k) * pq(k + 1)); };
  double bet = bb(1);
  W(HS_BET, 1) = bet;
  double ppk = dd(1) / bet;
  W(HS_PP, 1) = 0.; W(HS_PP, 2) = ppk;
  for (int k = 2; k <= km; ++k) {
    const double gm = W(HS_GR, k - 1) / bet;
    W(HS_GAM, k) = gm; bet = bb(k) - gm; W(HS_BET, k) = bet;
    ppk = (dd(k) - ppk) / bet;
    W(HS_PP, k + 1) = ppk;          // pre back-substitution value; overwritten below except k = km+1
  }
  // the reverse sweep recovers the pre back-substitution values from the final ones: ppf(k) = pp(k) + gam(k) pp(k+1)
  for (int k = km; k >= 2; --k) { ppk = W(HS_PP, k) - W(HS_GAM, k) * ppk; W(HS_PP, k) = ppk; }
}
// adjoint: consumes HA_PP(1..km+1); accumulates HA_PQ(k) (adjoint of pe = P - pm, handed on by the caller), HA_GR, and through
// them nothing else.  ppf (pre back-substitution) is recovered on the fly: ppf(k) = pp(k) + gam(k) pp(k+1).
HD void nhad_pp_edges_bwd(const NhAdCol& W) {
  const int km = W.km;
  auto pq = [&](int k) { return W(HS_P, k) - W(HS_PM, k); };
  // reverse of the back substitution pp(k) = ppf(k) - gam(k) pp(k+1), k = km..2  (processed k = 2..km)
  // a_ppf(k) = a_pp(k); a_gam(k) -= pp(k+1) a_pp(k); a_pp(k+1) -= gam(k) a_pp(k)
  // gam adjoints are needed in the elimination reverse below: kept in HS_P1 (free at this point of the caller's sweep)
  for (int k = 2; k <= km; ++k) {
    const double e = W(HA_PP, k);
    W(HS_P1, k) = -W(HS_PP, k + 1) * e;
    W(HA_PP, k + 1) -= W(HS_GAM, k) * e;
  }
  // reverse of the elimination, k = km..2:  ppf(k+1) = (dd(k) - ppf(k)) / bet_k ; bet_k = bb(k) - gam(k) ; gam(k) = gr(k-1) / bet_{k-1}
  double a_bet_next = 0.;   // adjoint of bet_k coming from gam(k+1) = gr(k) / bet_k
  for (int k = km; k >= 2; --k) {
    const double bet = W(HS_BET, k), gam = W(HS_GAM, k);
    const double ppf_k1 = (k == km) ? W(HS_PP, km + 1) : W(HS_PP, k + 1) + W(HS_GAM, k + 1) * W(HS_PP, k + 2);
    const double t = W(HA_PP, k + 1) / bet;                 // adjoint of the numerator dd(k) - ppf(k)
    double a_bet = -ppf_k1 * t + a_bet_next;
    // dd(k), bb(k)
    if (k == km) { W(HA_PQ, km) += 3. * t; }
    else { W(HA_PQ, k) += 3. * t; W(HA_PQ, k + 1) += 3. * W(HS_GR, k) * t; W(HA_GR, k) += 3. * pq(k + 1) * t + 2. * a_bet; }
    W(HA_PP, k) -= t;                                       // a_ppf(k)
    const double a_gam = W(HS_P1, k) - a_bet;
    const double betm = W(HS_BET, k - 1);
    W(HA_GR, k - 1) += a_gam / betm;
    a_bet_next = -gam * a_gam / betm;
  }
  {   // k = 1: ppf(2) = dd(1) / bet_1, bet_1 = bb(1)
    const double bet = W(HS_BET, 1);
    const double ppf2 = W(HS_PP, 2) + (km >= 2 ? W(HS_GAM, 2) * W(HS_PP, 3) : 0.);
    const double t = W(HA_PP, 2) / bet;
    const double a_bet = -ppf2 * t + a_bet_next;
    W(HA_PQ, 1) += 3. * t; W(HA_PQ, 2) += 3. * W(HS_GR, 1) * t; W(HA_GR, 1) += 3. * pq(2) * t + 2. * a_bet;
  }
}
// pe = P - pm with P = exp(gama log(-dm/dz R pt)): adjoints of dm, dz, pt, pm from HA_PQ; then gr = dm(k)/dm(k+1)
HD void nhad_pq_bwd(const NhAdCol& W, double gama) {
  const int km = W.km;
  for (int k = 1; k <= km; ++k) {
    const double ap = W(HA_PQ, k), gp = ap * gama * W(HS_P, k);
    W(HA_PM, k) -= ap;
    W(HA_DM, k) += gp / W(HS_DM, k); W(HA_DZ, k) -= gp / W(HS_DZ, k); W(HA_PT, k) += gp / W(HS_PT, k);
  }
  for (int k = 1; k <= km - 1; ++k) {
    const double ag = W(HA_GR, k), d1 = W(HS_DM, k + 1);
    W(HA_DM, k) += ag / d1; W(HA_DM, k + 1) -= W(HS_GR, k) * ag / d1;
  }
}
// new layer thickness from the pressure perturbation pe(1..km+1) (in HS_PE): forward, values p1 and dzn stored
HD void nhad_new_dz_fwd(const NhAdCol& W, double rgas, double capa1, double p_fac) {
  const int km = W.km;
  double p1 = (W(HS_PE, km) + 2. * W(HS_PE, km + 1)) * (1. / 3.);
  for (int k = km; k >= 1; --k) {
    if (k < km) p1 = (W(HS_PE, k) + 2. * (1. + W(HS_GR, k)) * W(HS_PE, k + 1) + W(HS_GR, k) * W(HS_PE, k + 2)) * (1. / 3.) - W(HS_GR, k) * p1;
    W(HS_P1, k) = p1;
    const double pm = W(HS_PM, k), lo = p_fac * pm, hi = p1 + pm, mx = (lo < hi) ? hi : lo;
    W(HS_DZN, k) = -(W(HS_DM, k) * rgas * W(HS_PT, k) * exp(capa1 * log(mx)));
  }
}
// adjoint: a_dzn(k) given by the functor; accumulates HA_DM, HA_PT, HA_PM, HA_PE, HA_GR
template <class FA>
HD void nhad_new_dz_bwd(const NhAdCol& W, double capa1, double p_fac, const FA& a_dzn) {
  const int km = W.km;
  double a_p1 = 0.;      // adjoint of p1_k: own use + recurrence from the level above
  for (int k = 1; k <= km; ++k) {
    const double ad = a_dzn(k), dzn = W(HS_DZN, k), pm = W(HS_PM, k), p1 = W(HS_P1, k);
    W(HA_DM, k) += ad * dzn / W(HS_DM, k); W(HA_PT, k) += ad * dzn / W(HS_PT, k);
    const double lo = p_fac * pm, hi = p1 + pm;
    if (lo < hi) { const double am = ad * dzn * capa1 / hi; a_p1 += am; W(HA_PM, k) += am; }
    else { W(HA_PM, k) += p_fac * ad * dzn * capa1 / lo; }
    if (k < km) {
      const double gr = W(HS_GR, k), t = a_p1 * (1. / 3.);
      W(HA_PE, k) += t; W(HA_PE, k + 1) += 2. * (1. + gr) * t; W(HA_PE, k + 2) += gr * t;
      W(HA_GR, k) += (2. * W(HS_PE, k + 1) + W(HS_PE, k + 2)) * t - W(HS_P1, k + 1) * a_p1;
      a_p1 = -gr * a_p1;
    } else {
      W(HA_PE, km) += a_p1 * (1. / 3.); W(HA_PE, km + 1) += 2. * a_p1 * (1. / 3.);
    }
  }
}
// thickness from the advected heights with the monotone fix, and its adjoint onto the heights (field slot 0)
HD void nhad_dz_fwd(const NhAdCol& W, const NhColArgs& a, int tile, int i, int j) {
  const int km = W.km;
  double below = a.f[0].t[fidx(a.g, a.f[0], tile, i, j, km + 1)];
  for (int k = km; k >= 1; --k) {
    const double z = a.f[0].t[fidx(a.g, a.f[0], tile, i, j, k)], lim = below + NH_DZ_MIN, zk = (z < lim) ? lim : z;
    W(HS_DZ, k) = below - zk;
    W(HS_CL, k) = (z < lim) ? 1. : 0.;      // clamped?
    below = zk;
  }
}
HD void nhad_dz_bwd(const NhAdCol& W, const NhColArgs& a, int tile, int i, int j, double a_wsfc, double inv_dt) {
  const int km = W.km;
  double a_zf = 0.;     // adjoint of the fixed height zfix(k), built from above
  for (int k = 1; k <= km; ++k) {
    a_zf -= W(HA_DZ, k);                       // dz(k) = zfix(k+1) - zfix(k)
    const double down = W(HA_DZ, k);           // contribution of dz(k) to zfix(k+1)
    if (W(HS_CL, k) != 0.) { a_zf = a_zf + down; }             // zfix(k) = zfix(k+1) + dz_min: hand everything to the level below
    else { a.f[0].p[fidx(a.g, a.f[0], tile, i, j, k)] += a_zf; a_zf = down; }
  }
  a.f[0].p[fidx(a.g, a.f[0], tile, i, j, km + 1)] += a_zf - a_wsfc * inv_dt;
}

// ---- RIEM_SOLVER_C -----------------------------------------------------------------------------------------------------------
// f: 0 gz_a  1 wc  2 ptc  3 delpc   ->   4 gz_c  5 pkc.  Consumes f[4].p, f[5].p (zeroed), accumulates into f[0..3].p.
HD void riem_c_col_ad(const NhColArgs& a, const ColWs& ws, int tile, int i, int j, double hs) {
  const Geom& g = a.g; const int km = g.npz;
  const NhAdCol W{ws, km};
  const double gama = 1. / (1. - a.akap), rgrav = 1. / a.grav, dt = a.dt, rdt = 1. / dt, t1g = gama * 2. * dt * dt, rgas = a.rdgas;
  auto F = [&](int sl, int k) -> size_t { return fidx(g, a.f[sl], tile, i, j, k); };
  // ---------------- forward replay with storage
  nhad_dz_fwd(W, a, tile, i, j);
  const double wsfc = (hs * rgrav - a.f[0].t[F(0, km + 1)]) * rdt;
  W(HS_PEM, 1) = a.ptop;
  for (int k = 1; k <= km; ++k) {
    const double dp = a.f[3].t[F(3, k)], p0 = W(HS_PEM, k), pn = p0 + dp;
    W(HS_PEM, k + 1) = pn; W(HS_PM, k) = dp / log(pn / p0); W(HS_DM, k) = dp * rgrav;
    W(HS_W1, k) = a.f[1].t[F(1, k)]; W(HS_PT, k) = a.f[2].t[F(2, k)];
  }
  nhad_pp_edges_fwd(W, rgas, gama);
  for (int k = 2; k <= km; ++k) W(HS_AA, k) = t1g / (W(HS_DZ, k - 1) + W(HS_DZ, k)) * (W(HS_PEM, k) + W(HS_PP, k));
  double bet = W(HS_DM, 1) - W(HS_AA, 2);
  W(HS_BET2, 1) = bet;
  double wv = (W(HS_DM, 1) * W(HS_W1, 1) + dt * W(HS_PP, 2)) / bet;
  W(HS_W2F, 1) = wv;
  for (int k = 2; k <= km - 1; ++k) {
    const double ak = W(HS_AA, k), gm = ak / bet;
    W(HS_GM2, k) = gm;
    bet = W(HS_DM, k) - (ak + W(HS_AA, k + 1) + ak * gm); W(HS_BET2, k) = bet;
    wv = (W(HS_DM, k) * W(HS_W1, k) + dt * (W(HS_PP, k + 1) - W(HS_PP, k)) - ak * wv) / bet;
    W(HS_W2F, k) = wv;
  }
  const double p1s = t1g / W(HS_DZ, km) * (W(HS_PEM, km + 1) + W(HS_PP, km + 1));
  {
    const double ak = W(HS_AA, km), gm = ak / bet;
    W(HS_GM2, km) = gm;
    bet = W(HS_DM, km) - (ak + p1s + ak * gm); W(HS_BET2, km) = bet;
    wv = (W(HS_DM, km) * W(HS_W1, km) + dt * (W(HS_PP, km + 1) - W(HS_PP, km)) - p1s * wsfc - ak * wv) / bet;
    W(HS_W2F, km) = wv;
  }
  W(HS_W2, km) = wv;
  for (int k = km - 1; k >= 1; --k) { wv = W(HS_W2F, k) - W(HS_GM2, k + 1) * wv; W(HS_W2, k) = wv; }
  W(HS_PE, 1) = 0.;
  for (int k = 1; k <= km; ++k) W(HS_PE, k + 1) = W(HS_PE, k) + W(HS_DM, k) * (W(HS_W2, k) - W(HS_W1, k)) * rdt;
  nhad_new_dz_fwd(W, rgas, a.akap - 1., a.p_fac);
  // ---------------- reverse
  for (int k = 0; k <= km + 1; ++k) for (int s_ : {HA_DZ, HA_PEM, HA_PM, HA_DM, HA_PT, HA_PP, HA_PE, HA_W1, HA_GR, HA_PQ}) W(s_, k) = 0.;
  // outputs: pkc(k) = pe(k) + pem(k), k >= 2 (level 1 is the constant ptop); gz_c(k) = hs - grav sum_{l >= k} dzn(l)
  for (int k = 2; k <= km + 1; ++k) { const double x = a.f[5].p[F(5, k)]; W(HA_PE, k) += x; W(HA_PEM, k) += x; a.f[5].p[F(5, k)] = 0.; }
  a.f[5].p[F(5, 1)] = 0.;
  {
    double G = 0.;     // running sum of the gz_c adjoints from the top: the adjoint of dzn(k) is -grav * G_k (levels visited in order)
    nhad_new_dz_bwd(W, a.akap - 1., a.p_fac, [&](int k) { G += a.f[4].p[F(4, k)]; a.f[4].p[F(4, k)] = 0.; return -a.grav * G; });
    a.f[4].p[F(4, km + 1)] = 0.;
  }
  // pe(k+1) = pe(k) + dm (w2 - w1) / dt  -> a_w2 into HS_DZN (free now)
  for (int k = km; k >= 1; --k) {
    const double ap = W(HA_PE, k + 1);
    W(HA_PE, k) += ap;
    W(HA_DM, k) += ap * (W(HS_W2, k) - W(HS_W1, k)) * rdt;
    W(HS_DZN, k) = ap * W(HS_DM, k) * rdt;
    W(HA_W1, k) -= ap * W(HS_DM, k) * rdt;
  }
  // back substitution w2(k) = w2f(k) - gm2(k+1) w2(k+1), k = km-1..1; reversed k = 1..km-1.  a_gm2 into HS_P1 (free now)
  for (int k = 2; k <= km; ++k) W(HS_P1, k) = 0.;
  for (int k = 1; k <= km - 1; ++k) {
    const double e = W(HS_DZN, k);
    W(HS_P1, k + 1) -= W(HS_W2, k + 1) * e;
    W(HS_DZN, k + 1) -= W(HS_GM2, k + 1) * e;
  }
  // now HS_DZN(k) = a_w2f(k).  Reverse of the elimination; a_aa accumulated in HS_PE (free now)
  for (int k = 0; k <= km + 1; ++k) W(HS_PE, k) = 0.;
  double a_wsfc = 0., a_betm = 0.;     // a_betm: adjoint of bet_{k-1} coming from gm2(k) = aa(k) / bet_{k-1}
  {
    const double ak = W(HS_AA, km), gm = W(HS_GM2, km), b = W(HS_BET2, km), w = W(HS_W2F, km);
    const double aN = W(HS_DZN, km) / b, ab = -w * aN;
    W(HA_DM, km) += W(HS_W1, km) * aN + ab; W(HA_W1, km) += W(HS_DM, km) * aN;
    W(HA_PP, km + 1) += dt * aN; W(HA_PP, km) -= dt * aN;
    const double a_p1s = -wsfc * aN - ab;
    a_wsfc -= p1s * aN;
    W(HS_PE, km) += -W(HS_W2F, km - 1) * aN - (1. + gm) * ab;
    W(HS_DZN, km - 1) -= ak * aN;
    const double a_gm = W(HS_P1, km) - ak * ab, bm = W(HS_BET2, km - 1);
    W(HS_PE, km) += a_gm / bm; a_betm = -gm * a_gm / bm;
    W(HA_DZ, km) -= p1s / W(HS_DZ, km) * a_p1s;
    const double c = t1g / W(HS_DZ, km) * a_p1s;
    W(HA_PEM, km + 1) += c; W(HA_PP, km + 1) += c;
  }
  for (int k = km - 1; k >= 2; --k) {
    const double ak = W(HS_AA, k), gm = W(HS_GM2, k), b = W(HS_BET2, k), w = W(HS_W2F, k);
    const double aN = W(HS_DZN, k) / b, ab = -w * aN + a_betm;
    W(HA_DM, k) += W(HS_W1, k) * aN + ab; W(HA_W1, k) += W(HS_DM, k) * aN;
    W(HA_PP, k + 1) += dt * aN; W(HA_PP, k) -= dt * aN;
    W(HS_PE, k) += -W(HS_W2F, k - 1) * aN - (1. + gm) * ab;
    W(HS_PE, k + 1) -= ab;
    W(HS_DZN, k - 1) -= ak * aN;
    const double a_gm = W(HS_P1, k) - ak * ab, bm = W(HS_BET2, k - 1);
    W(HS_PE, k) += a_gm / bm; a_betm = -gm * a_gm / bm;
  }
  {   // k = 1: w2f(1) = (dm w1 + dt pp(2)) / bet, bet = dm(1) - aa(2)
    const double b = W(HS_BET2, 1), w = W(HS_W2F, 1);
    const double aN = W(HS_DZN, 1) / b, ab = -w * aN + a_betm;
    W(HA_DM, 1) += W(HS_W1, 1) * aN + ab; W(HA_W1, 1) += W(HS_DM, 1) * aN;
    W(HA_PP, 2) += dt * aN;
    W(HS_PE, 2) -= ab;
  }
  // aa(k) = t1g / (dz(k-1) + dz(k)) (pem(k) + pp(k))
  for (int k = 2; k <= km; ++k) {
    const double aa_ad = W(HS_PE, k), dzs = W(HS_DZ, k - 1) + W(HS_DZ, k), c = t1g / dzs * aa_ad, d = -W(HS_AA, k) / dzs * aa_ad;
    W(HA_DZ, k - 1) += d; W(HA_DZ, k) += d; W(HA_PEM, k) += c; W(HA_PP, k) += c;
  }
  nhad_pp_edges_bwd(W);
  nhad_pq_bwd(W, gama);
  // inputs: pm = dp / log(pem(k+1)/pem(k)), dm = dp / g, pem = ptop + cumsum dp
  {
    double a_sum = 0.;      // suffix sum of the pem adjoints
    for (int k = km; k >= 1; --k) {
      const double dp = a.f[3].t[F(3, k)], lg = dp / W(HS_PM, k), apm = W(HA_PM, k), alg = -W(HS_PM, k) * apm / lg;
      W(HA_PEM, k + 1) += alg / W(HS_PEM, k + 1); W(HA_PEM, k) -= alg / W(HS_PEM, k);
      a_sum += W(HA_PEM, k + 1);
      a.f[3].p[F(3, k)] += apm / lg + W(HA_DM, k) * rgrav + a_sum;
      a.f[1].p[F(1, k)] += W(HA_W1, k);
      a.f[2].p[F(2, k)] += W(HA_PT, k);
    }
  }
  nhad_dz_bwd(W, a, tile, i, j, a_wsfc, rdt);
}

// ---- RIEM_SOLVER3 (semi-implicit, off-centring alpha = a_imp) -----------------------------------------------------------------
// f: 0 zh_a  1 w_m  2 pt  3 delp   ->   4 w_o  5 delz_o  6 zh_o  7 ppe  8 pk3  (last acoustic step: 9 pe  10 peln  11 pk  12 ws).
// Consumes the output adjoints (zeroed), accumulates into f[0..3].p.
HD void riem3_col_ad(const NhColArgs& a, const ColWs& ws, int tile, int i, int j, double hs) {
  const Geom& g = a.g; const int km = g.npz;
  const NhAdCol W{ws, km};
  const double gama = 1. / (1. - a.akap), rgrav = 1. / a.grav, zs = hs * rgrav, dt = a.dt, rdt = 1. / dt, rgas = a.rdgas;
  // a_imp > 0.999: the reference runs SIM1_SOLVER (nh_core_tlm.F90:176-181) = this sweep with alpha = 1 and no scale_m term
  const bool sim1 = a.a_imp > 0.999;
  const double alpha = sim1 ? 1. : a.a_imp, beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha, t1g = 2. * gama * (alpha * dt) * (alpha * dt), scale_m = sim1 ? 0. : a.scale_z;
  auto F = [&](int sl, int k) -> size_t { return fidx(g, a.f[sl], tile, i, j, k); };
  // ---------------- forward replay with storage
  nhad_dz_fwd(W, a, tile, i, j);
  const double wsfc = (zs - a.f[0].t[F(0, km + 1)]) * rdt;
  W(HS_PEM, 1) = a.ptop;
  for (int k = 1; k <= km; ++k) {
    const double dp = a.f[3].t[F(3, k)], p0 = W(HS_PEM, k), pn = p0 + dp;
    W(HS_PEM, k + 1) = pn; W(HS_PM, k) = dp / (log(pn) - log(p0)); W(HS_DM, k) = dp * rgrav;
    W(HS_W1, k) = a.f[1].t[F(1, k)]; W(HS_PT, k) = a.f[2].t[F(2, k)];
  }
  nhad_pp_edges_fwd(W, rgas, gama);
  for (int k = 2; k <= km; ++k) {
    const double ak = t1g / (W(HS_DZ, k - 1) + W(HS_DZ, k)) * (W(HS_PEM, k) + W(HS_PP, k));
    W(HS_WK, k) = t2 * ak * (W(HS_W1, k - 1) - W(HS_W1, k));
    W(HS_AA, k) = ak - scale_m * W(HS_DM, 1);
  }
  double bet = W(HS_DM, 1) - W(HS_AA, 2);
  W(HS_BET2, 1) = bet;
  double wv = (W(HS_DM, 1) * W(HS_W1, 1) + dt * W(HS_PP, 2) + W(HS_WK, 2)) / bet;
  W(HS_W2F, 1) = wv;
  for (int k = 2; k <= km - 1; ++k) {
    const double ak = W(HS_AA, k), gm = ak / bet;
    W(HS_GM2, k) = gm;
    bet = W(HS_DM, k) - (ak + W(HS_AA, k + 1) + ak * gm); W(HS_BET2, k) = bet;
    wv = (W(HS_DM, k) * W(HS_W1, k) + dt * (W(HS_PP, k + 1) - W(HS_PP, k)) + W(HS_WK, k + 1) - W(HS_WK, k) - ak * wv) / bet;
    W(HS_W2F, k) = wv;
  }
  const double wk1 = t1g / W(HS_DZ, km) * (W(HS_PEM, km + 1) + W(HS_PP, km + 1));
  {
    const double ak = W(HS_AA, km), gm = ak / bet;
    W(HS_GM2, km) = gm;
    bet = W(HS_DM, km) - (ak + wk1 + ak * gm); W(HS_BET2, km) = bet;
    wv = (W(HS_DM, km) * W(HS_W1, km) + dt * (W(HS_PP, km + 1) - W(HS_PP, km)) - W(HS_WK, km) + wk1 * (t2 * W(HS_W1, km) - ra * wsfc) - ak * wv) / bet;
    W(HS_W2F, km) = wv;
  }
  W(HS_W2, km) = wv;
  for (int k = km - 1; k >= 1; --k) { wv = W(HS_W2F, k) - W(HS_GM2, k + 1) * wv; W(HS_W2, k) = wv; }
  W(HS_PE, 1) = 0.;
  for (int k = 1; k <= km; ++k)
    W(HS_PE, k + 1) = W(HS_PE, k) + (W(HS_DM, k) * (W(HS_W2, k) - W(HS_W1, k)) * rdt - beta * (W(HS_PP, k + 1) - W(HS_PP, k))) * ra;
  nhad_new_dz_fwd(W, rgas, a.akap - 1., a.p_fac);
  // ---------------- reverse
  for (int k = 0; k <= km + 1; ++k) for (int s_ : {HA_DZ, HA_PEM, HA_PM, HA_DM, HA_PT, HA_PP, HA_PE, HA_W1, HA_GR, HA_PQ}) W(s_, k) = 0.;
  // outputs.  ppe(k) = pe(k) + beta (pp(k) - pe(k));  zh_o(k) = zs - sum_{l >= k} dzn(l);  delz_o = dzn;  w_o = w2
  for (int k = 1; k <= km + 1; ++k) {
    const double x = a.f[7].p[F(7, k)];
    W(HA_PE, k) += (1. - beta) * x; W(HA_PP, k) += beta * x;
    a.f[7].p[F(7, k)] = 0.;
  }
  {
    double G = 0.;
    nhad_new_dz_bwd(W, a.akap - 1., a.p_fac, [&](int k) {
      G += a.f[6].p[F(6, k)]; a.f[6].p[F(6, k)] = 0.;
      const double x = a.f[5].p[F(5, k)] - G; a.f[5].p[F(5, k)] = 0.;
      return x; });
    a.f[6].p[F(6, km + 1)] = 0.;
  }
  // pe(k+1) = pe(k) + (dm (w2 - w1) / dt - beta (pp(k+1) - pp(k))) / alpha ; a_w2 (incl. the w_o adjoint) into HS_DZN
  for (int k = km; k >= 1; --k) {
    const double ap = W(HA_PE, k + 1), c = ap * ra;
    W(HA_PE, k) += ap;
    W(HA_DM, k) += c * (W(HS_W2, k) - W(HS_W1, k)) * rdt;
    W(HS_DZN, k) = c * W(HS_DM, k) * rdt + a.f[4].p[F(4, k)];
    a.f[4].p[F(4, k)] = 0.;
    W(HA_W1, k) -= c * W(HS_DM, k) * rdt;
    W(HA_PP, k + 1) -= c * beta; W(HA_PP, k) += c * beta;
  }
  for (int k = 2; k <= km; ++k) W(HS_P1, k) = 0.;
  for (int k = 1; k <= km - 1; ++k) {
    const double e = W(HS_DZN, k);
    W(HS_P1, k + 1) -= W(HS_W2, k + 1) * e;
    W(HS_DZN, k + 1) -= W(HS_GM2, k + 1) * e;
  }
  // HS_DZN = a_w2f; a_aa into HS_PE, a_wk into HS_W2 (both free now)
  for (int k = 0; k <= km + 1; ++k) { W(HS_PE, k) = 0.; W(HS_W2, k) = 0.; }
  double a_wsfc = 0., a_betm = 0.;
  {
    const double ak = W(HS_AA, km), gm = W(HS_GM2, km), b = W(HS_BET2, km), w = W(HS_W2F, km);
    const double aN = W(HS_DZN, km) / b, ab = -w * aN;
    W(HA_DM, km) += W(HS_W1, km) * aN + ab; W(HA_W1, km) += (W(HS_DM, km) + wk1 * t2) * aN;
    W(HA_PP, km + 1) += dt * aN; W(HA_PP, km) -= dt * aN;
    W(HS_W2, km) -= aN;
    const double a_wk1 = (t2 * W(HS_W1, km) - ra * wsfc) * aN - ab;
    a_wsfc -= wk1 * ra * aN;
    W(HS_PE, km) += -W(HS_W2F, km - 1) * aN - (1. + gm) * ab;
    W(HS_DZN, km - 1) -= ak * aN;
    const double a_gm = W(HS_P1, km) - ak * ab, bm = W(HS_BET2, km - 1);
    W(HS_PE, km) += a_gm / bm; a_betm = -gm * a_gm / bm;
    W(HA_DZ, km) -= wk1 / W(HS_DZ, km) * a_wk1;
    const double c = t1g / W(HS_DZ, km) * a_wk1;
    W(HA_PEM, km + 1) += c; W(HA_PP, km + 1) += c;
  }
  for (int k = km - 1; k >= 2; --k) {
    const double ak = W(HS_AA, k), gm = W(HS_GM2, k), b = W(HS_BET2, k), w = W(HS_W2F, k);
    const double aN = W(HS_DZN, k) / b, ab = -w * aN + a_betm;
    W(HA_DM, k) += W(HS_W1, k) * aN + ab; W(HA_W1, k) += W(HS_DM, k) * aN;
    W(HA_PP, k + 1) += dt * aN; W(HA_PP, k) -= dt * aN;
    W(HS_W2, k + 1) += aN; W(HS_W2, k) -= aN;
    W(HS_PE, k) += -W(HS_W2F, k - 1) * aN - (1. + gm) * ab;
    W(HS_PE, k + 1) -= ab;
    W(HS_DZN, k - 1) -= ak * aN;
    const double a_gm = W(HS_P1, k) - ak * ab, bm = W(HS_BET2, k - 1);
    W(HS_PE, k) += a_gm / bm; a_betm = -gm * a_gm / bm;
  }
  {
    const double b = W(HS_BET2, 1), w = W(HS_W2F, 1);
    const double aN = W(HS_DZN, 1) / b, ab = -w * aN + a_betm;
    W(HA_DM, 1) += W(HS_W1, 1) * aN + ab; W(HA_W1, 1) += W(HS_DM, 1) * aN;
    W(HA_PP, 2) += dt * aN;
    W(HS_W2, 2) += aN;
    W(HS_PE, 2) -= ab;
  }
  // aa(k) = a_k - scale_m dm(1);  wk(k) = t2 a_k (w1(k-1) - w1(k));  a_k = t1g / (dz(k-1) + dz(k)) (pem(k) + pp(k))
  for (int k = 2; k <= km; ++k) {
    const double a_aa = W(HS_PE, k), a_wk = W(HS_W2, k), akv = W(HS_AA, k) + scale_m * W(HS_DM, 1), dw = W(HS_W1, k - 1) - W(HS_W1, k);
    W(HA_DM, 1) -= scale_m * a_aa;
    const double a_ak = a_aa + t2 * dw * a_wk;
    W(HA_W1, k - 1) += t2 * akv * a_wk; W(HA_W1, k) -= t2 * akv * a_wk;
    const double dzs = W(HS_DZ, k - 1) + W(HS_DZ, k), c = t1g / dzs * a_ak, d = -akv / dzs * a_ak;
    W(HA_DZ, k - 1) += d; W(HA_DZ, k) += d; W(HA_PEM, k) += c; W(HA_PP, k) += c;
  }
  nhad_pp_edges_bwd(W);
  nhad_pq_bwd(W, gama);
  // pressures handed out: pk3(k+1) = exp(akap log pem(k+1)) and, at the last acoustic step, pe, peln, pk; then the inputs
  if (a.last_call) { a_wsfc += a.f[12].p[F(12, 1)]; a.f[12].p[F(12, 1)] = 0.; }
  a.f[8].p[F(8, 1)] = 0.;
  if (a.last_call) { a.f[9].p[F(9, 1)] = 0.; a.f[10].p[F(10, 1)] = 0.; a.f[11].p[F(11, 1)] = 0.; }
  {
    double a_sum = 0.;
    for (int k = km; k >= 1; --k) {
      const double p = W(HS_PEM, k + 1), pk = exp(a.akap * log(p));
      double ap = a.akap * pk / p * a.f[8].p[F(8, k + 1)];
      a.f[8].p[F(8, k + 1)] = 0.;
      if (a.last_call) {
        ap += a.f[9].p[F(9, k + 1)] + a.f[10].p[F(10, k + 1)] / p + a.akap * pk / p * a.f[11].p[F(11, k + 1)];
        a.f[9].p[F(9, k + 1)] = 0.; a.f[10].p[F(10, k + 1)] = 0.; a.f[11].p[F(11, k + 1)] = 0.;
      }
      const double dp = a.f[3].t[F(3, k)], lg = dp / W(HS_PM, k), apm = W(HA_PM, k), alg = -W(HS_PM, k) * apm / lg;
      W(HA_PEM, k + 1) += ap + alg / p; W(HA_PEM, k) -= alg / W(HS_PEM, k);
      a_sum += W(HA_PEM, k + 1);
      a.f[3].p[F(3, k)] += apm / lg + W(HA_DM, k) * rgrav + a_sum;
      a.f[1].p[F(1, k)] += W(HA_W1, k);
      a.f[2].p[F(2, k)] += W(HA_PT, k);
    }
  }
  nhad_dz_bwd(W, a, tile, i, j, a_wsfc, rdt);
}

}  // namespace fv3
