// fv3lm-hip: column operators of the non-hydrostatic acoustic step, written once on the generic
// column interface of coltape.h (nonlinear / tangent / taped adjoint):
//   riem_c_col   UPDATE_DZ_C's monotone-thickness fix and surface velocity (nh_utils_tlm.F90:367-380) + RIEM_SOLVER_C
//                (:846-933) with SIM1_SOLVER (:2760-2884)
//   riem3_col    UPDATE_DZ_D's fix and surface velocity (:708-721) + RIEM_SOLVER3 (nh_core_tlm.F90:245-389) with SIM_SOLVER
//                (nh_utils_tlm.F90:3129-3272), the path of the default a_imp = 0.75
//   edge_col     EDGE_PROFILE (:3473-3584), layer means of two fields -> interface values
//   zh_init_col  interface heights from delz at the first acoustic step (dyn_core_tlm.F90:1779-1801)
//   ring_col     PK3_HALO / PE_HALO (dyn_core_tlm.F90:2716-2778, :2934-2961) on the halo rings
#pragma once
#include "coltape.h"

namespace fv3 {

constexpr double NH_DZ_MIN = 2.0;      // nh_utils: dz_min
constexpr int NH_NF = 13;
constexpr int NH_WS_SLOTS = 32;        // raw (double) workspace slots per column: 2 per generic-scalar slot, 31 for the hand-written adjoints (nh_ad.h)

struct NhColArgs {
  Geom g; Fld f[NH_NF];
  double* ws; size_t ws_stride; TapeMem tape;
  const double* hs;                    // surface geopotential [ntile][plane]
  const LevelParams* lev;
  double dt, akap, ptop, rdgas, grav, a_imp, p_fac, scale_z;
  const double *ak, *bk;               // hybrid coefficients, device [npz+1] (vertical remap)
  double zvir, cp_air;
  int last_call, what;
  int kord_tm, kord_tr, kord_wz;       // trajectory remap profiles (|kord| <= 16: the limited profile gives the values, kinds NHC_RM_*_LIM)
  int use_tape;                        // adjoint of the implicit solvers: 1 = taped run of the generic code (FV3LM_NH_TAPE=1), 0 = nh_ad.h
};
HD void riem_c_col_ad(const NhColArgs& a, const ColWs& ws, int tile, int i, int j, double hs);
HD void riem3_col_ad(const NhColArgs& a, const ColWs& ws, int tile, int i, int j, double hs);

// common tail of SIM1 / SIM: new layer thickness from the perturbation pressure (nh_utils_tlm.F90:2862-2883, :3241-3265)
template <class IO>
HD void nh_new_dz(const IO& io, int km, double rgas, double capa1, double p_fac, const WArr<IO>& pe, const WArr<IO>& bb, const WArr<IO>& g_rat,
                  const WArr<IO>& dm2, const WArr<IO>& pm2, const WArr<IO>& pt2, const WArr<IO>& dz2) {
  typedef typename IO::T T;
  T p1 = (pe(km) + 2. * pe(km + 1)) * (1. / 3.);
  for (int k = km; k >= 1; --k) {
    if (k < km) p1 = (pe(k) + bb(k) * pe(k + 1) + g_rat(k) * pe(k + 2)) * (1. / 3.) - g_rat(k) * p1;
    const T pm = pm2(k);
    const T lo = p_fac * pm, hi = p1 + pm;
    const T mx = (val(lo) < val(hi)) ? hi : lo;
    dz2.set(k, -(dm2(k) * rgas * pt2(k) * dexp(capa1 * dlog(mx))));
  }
}

// tridiagonal interpolation of the layer perturbation pressure to the interfaces (first Thomas sweep, :2790-2817 / :3160-3187)
template <class IO>
HD void nh_pp_edges(const IO& io, int km, double rgas, double gama, const WArr<IO>& pe, const WArr<IO>& dm2, const WArr<IO>& pm2, const WArr<IO>& dz2,
                    const WArr<IO>& pt2, const WArr<IO>& g_rat, const WArr<IO>& bb, const WArr<IO>& dd, const WArr<IO>& gam, const WArr<IO>& pp) {
  typedef typename IO::T T;
  for (int k = 1; k <= km; ++k) pe.set(k, dexp(gama * dlog(-(dm2(k) / dz2(k) * rgas * pt2(k)))) - pm2(k));
  for (int k = 1; k <= km - 1; ++k) {
    const T gr = dm2(k) / dm2(k + 1);
    g_rat.set(k, gr); bb.set(k, 2. * (1. + gr)); dd.set(k, 3. * (pe(k) + gr * pe(k + 1)));
  }
  // recurrences carried in registers: a value just stored is not read back through memory
  T bet = bb(1);
  T ppk = dd(1) / bet;
  pp.set(1, io.cst(0.)); pp.set(2, ppk);
  bb.set(km, io.cst(2.)); dd.set(km, 3. * pe(km));
  for (int k = 2; k <= km; ++k) {
    const T gm = g_rat(k - 1) / bet;
    gam.set(k, gm); bet = bb(k) - gm;
    ppk = (dd(k) - ppk) / bet;
    pp.set(k + 1, ppk);
  }
  for (int k = km; k >= 2; --k) { ppk = pp(k) - gam(k) * ppk; pp.set(k, ppk); }
}

// workspace slots (in units of T)
enum { NS_AA = 0, NS_BB, NS_DD, NS_W1, NS_WK, NS_GR, NS_GAM, NS_PP, NS_PE, NS_DM, NS_PM, NS_PEM, NS_W2, NS_DZ, NS_PT, NS_COUNT };

// SIM1_SOLVER: fully implicit
template <class IO>
HD void nh_sim1(const IO& io, const ColWs& ws, int km, double dt, double rgas, double gama, double kappa, const typename IO::T& wsfc, double p_fac) {
  typedef typename IO::T T;
  WArr<IO> aa{io, ws, NS_AA}, bb{io, ws, NS_BB}, dd{io, ws, NS_DD}, w1{io, ws, NS_W1}, g_rat{io, ws, NS_GR}, gam{io, ws, NS_GAM}, pp{io, ws, NS_PP},
      pe{io, ws, NS_PE}, dm2{io, ws, NS_DM}, pm2{io, ws, NS_PM}, pem{io, ws, NS_PEM}, w2{io, ws, NS_W2}, dz2{io, ws, NS_DZ}, pt2{io, ws, NS_PT};
  const double t1g = gama * 2. * dt * dt, rdt = 1. / dt;
  for (int k = 1; k <= km; ++k) w1.set(k, w2(k));
  nh_pp_edges(io, km, rgas, gama, pe, dm2, pm2, dz2, pt2, g_rat, bb, dd, gam, pp);
  for (int k = 2; k <= km; ++k) aa.set(k, t1g / (dz2(k - 1) + dz2(k)) * (pem(k) + pp(k)));
  T bet = dm2(1) - aa(2);
  T wk_ = (dm2(1) * w1(1) + dt * pp(2)) / bet;
  w2.set(1, wk_);
  for (int k = 2; k <= km - 1; ++k) {
    const T ak_ = aa(k), gm = ak_ / bet;
    gam.set(k, gm);
    bet = dm2(k) - (ak_ + aa(k + 1) + ak_ * gm);
    wk_ = (dm2(k) * w1(k) + dt * (pp(k + 1) - pp(k)) - ak_ * wk_) / bet;
    w2.set(k, wk_);
  }
  const T p1 = t1g / dz2(km) * (pem(km + 1) + pp(km + 1));
  {
    const T ak_ = aa(km), gm = ak_ / bet;
    gam.set(km, gm);
    bet = dm2(km) - (ak_ + p1 + ak_ * gm);
    wk_ = (dm2(km) * w1(km) + dt * (pp(km + 1) - pp(km)) - p1 * wsfc - ak_ * wk_) / bet;
    w2.set(km, wk_);
  }
  for (int k = km - 1; k >= 1; --k) { wk_ = w2(k) - gam(k + 1) * wk_; w2.set(k, wk_); }
  T pek = io.cst(0.);
  pe.set(1, pek);
  for (int k = 1; k <= km; ++k) { pek = pek + dm2(k) * (w2(k) - w1(k)) * rdt; pe.set(k + 1, pek); }
  nh_new_dz(io, km, rgas, kappa - 1., p_fac, pe, bb, g_rat, dm2, pm2, pt2, dz2);
}

// SIM_SOLVER: semi-implicit with off-centring alpha
template <class IO>
HD void nh_sim(const IO& io, const ColWs& ws, int km, double dt, double rgas, double gama, double kappa, const typename IO::T& wsfc, double alpha,
               double p_fac, double scale_m) {
  typedef typename IO::T T;
  WArr<IO> aa{io, ws, NS_AA}, bb{io, ws, NS_BB}, dd{io, ws, NS_DD}, w1{io, ws, NS_W1}, wk{io, ws, NS_WK}, g_rat{io, ws, NS_GR}, gam{io, ws, NS_GAM},
      pp{io, ws, NS_PP}, pe{io, ws, NS_PE}, dm2{io, ws, NS_DM}, pm2{io, ws, NS_PM}, pem{io, ws, NS_PEM}, w2{io, ws, NS_W2}, dz2{io, ws, NS_DZ},
      pt2{io, ws, NS_PT};
  const double beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha, t1g = 2. * gama * (alpha * dt) * (alpha * dt), rdt = 1. / dt;
  for (int k = 1; k <= km; ++k) w1.set(k, w2(k));
  nh_pp_edges(io, km, rgas, gama, pe, dm2, pm2, dz2, pt2, g_rat, bb, dd, gam, pp);
  for (int k = 1; k <= km + 1; ++k) pe.set(k, pem(k) + pp(k));
  for (int k = 2; k <= km; ++k) {
    const T a = t1g / (dz2(k - 1) + dz2(k)) * pe(k);
    wk.set(k, t2 * a * (w1(k - 1) - w1(k)));
    aa.set(k, a - scale_m * dm2(1));
  }
  T bet = dm2(1) - aa(2);
  T wv = (dm2(1) * w1(1) + dt * pp(2) + wk(2)) / bet;
  w2.set(1, wv);
  for (int k = 2; k <= km - 1; ++k) {
    const T ak_ = aa(k), gm = ak_ / bet;
    gam.set(k, gm);
    bet = dm2(k) - (ak_ + aa(k + 1) + ak_ * gm);
    wv = (dm2(k) * w1(k) + dt * (pp(k + 1) - pp(k)) + wk(k + 1) - wk(k) - ak_ * wv) / bet;
    w2.set(k, wv);
  }
  {
    const T wk1 = t1g / dz2(km) * pe(km + 1);
    const T ak_ = aa(km), gm = ak_ / bet;
    gam.set(km, gm);
    bet = dm2(km) - (ak_ + wk1 + ak_ * gm);
    wv = (dm2(km) * w1(km) + dt * (pp(km + 1) - pp(km)) - wk(km) + wk1 * (t2 * w1(km) - ra * wsfc) - ak_ * wv) / bet;
    w2.set(km, wv);
  }
  for (int k = km - 1; k >= 1; --k) { wv = w2(k) - gam(k + 1) * wv; w2.set(k, wv); }
  T pek = io.cst(0.);
  pe.set(1, pek);
  for (int k = 1; k <= km; ++k) { pek = pek + (dm2(k) * (w2(k) - w1(k)) * rdt - beta * (pp(k + 1) - pp(k))) * ra; pe.set(k + 1, pek); }
  nh_new_dz(io, km, rgas, kappa - 1., p_fac, pe, bb, g_rat, dm2, pm2, pt2, dz2);
  for (int k = 1; k <= km + 1; ++k) pe.set(k, pe(k) + beta * (pp(k) - pe(k)));
}

// C-grid half step.  f: 0 gz_a (advected interface heights, km+1)  1 wc  2 ptc  3 delpc   ->   4 gz_c (geopotential, km+1)  5 pkc (km+1)
template <class IO>
HD void riem_c_col(const IO& io, const NhColArgs& a, const ColWs& ws, double hs) {
  typedef typename IO::T T;
  const int km = a.g.npz;
  const double gama = 1. / (1. - a.akap), rgrav = 1. / a.grav;
  WArr<IO> pe{io, ws, NS_PE}, dm2{io, ws, NS_DM}, pm2{io, ws, NS_PM}, pem{io, ws, NS_PEM}, w2{io, ws, NS_W2}, dz2{io, ws, NS_DZ}, pt2{io, ws, NS_PT};
  // surface velocity of the terrain-following bottom and the monotone-thickness fix (UPDATE_DZ_C tail)
  T below = io.ld(0, km + 1);
  const T wsfc = (hs * rgrav - below) * (1. / a.dt);
  for (int k = km; k >= 1; --k) {
    const T z = io.ld(0, k), lim = below + NH_DZ_MIN;
    const T zk = (val(z) < val(lim)) ? lim : z;
    dz2.set(k, below - zk);
    below = zk;
  }
  io.st(5, 1, io.cst(a.ptop));
  T pk_ = io.cst(a.ptop);
  pem.set(1, pk_);
  for (int k = 1; k <= km; ++k) {
    const T dp = io.ld(3, k), pn = pk_ + dp;
    pem.set(k + 1, pn);
    pm2.set(k, dp / dlog(pn / pk_));
    dm2.set(k, dp * rgrav);
    w2.set(k, io.ld(1, k)); pt2.set(k, io.ld(2, k));
    pk_ = pn;
  }
  nh_sim1(io, ws, km, a.dt, a.rdgas, gama, a.akap, wsfc, a.p_fac);
  for (int k = 2; k <= km + 1; ++k) io.st(5, k, pe(k) + pem(k));
  T gz = io.cst(hs);
  io.st(4, km + 1, gz);
  for (int k = km; k >= 1; --k) { gz = gz - dz2(k) * a.grav; io.st(4, k, gz); }
}

// D-grid full step.  f: 0 zh_a (advected heights, km+1)  1 w_m  2 pt  3 delp   ->   4 w_o  5 delz_o  6 zh_o (km+1)  7 ppe (km+1)  8 pk3 (km+1)
//                    and at the last acoustic step 9 pe  10 peln  11 pk (km+1 each)  12 ws (one level: surface w, lower boundary of the w remap)
template <class IO>
HD void riem3_col(const IO& io, const NhColArgs& a, const ColWs& ws, double hs) {
  typedef typename IO::T T;
  const int km = a.g.npz;
  const double gama = 1. / (1. - a.akap), rgrav = 1. / a.grav, zs = hs * rgrav;
  WArr<IO> pe{io, ws, NS_PE}, dm2{io, ws, NS_DM}, pm2{io, ws, NS_PM}, pem{io, ws, NS_PEM}, w2{io, ws, NS_W2}, dz2{io, ws, NS_DZ}, pt2{io, ws, NS_PT};
  T below = io.ld(0, km + 1);
  const T wsfc = (zs - below) * (1. / a.dt);
  if (a.last_call) io.st(12, 1, wsfc);
  for (int k = km; k >= 1; --k) {
    const T z = io.ld(0, k), lim = below + NH_DZ_MIN;
    const T zk = (val(z) < val(lim)) ? lim : z;
    dz2.set(k, below - zk);      // zh(k+1) - zh(k)
    below = zk;
  }
  pem.set(1, io.cst(a.ptop));
  const double peln1 = log(a.ptop);
  io.st(8, 1, io.cst(exp(a.akap * peln1)));
  if (a.last_call) { io.st(9, 1, io.cst(a.ptop)); io.st(10, 1, io.cst(peln1)); io.st(11, 1, io.cst(exp(a.akap * peln1))); }
  T p_ = io.cst(a.ptop), l_ = io.cst(peln1);
  for (int k = 1; k <= km; ++k) {
    const T dp = io.ld(3, k), p = p_ + dp, l = dlog(p), pk = dexp(a.akap * l);
    pem.set(k + 1, p);
    io.st(8, k + 1, pk);
    if (a.last_call) { io.st(9, k + 1, p); io.st(10, k + 1, l); io.st(11, k + 1, pk); }
    pm2.set(k, dp / (l - l_));
    dm2.set(k, dp * rgrav);
    w2.set(k, io.ld(1, k)); pt2.set(k, io.ld(2, k));
    p_ = p; l_ = l;
  }
  // solver dispatch of RIEM_SOLVER3 (nh_core_tlm.F90:155-189): a_imp > 0.999 -> SIM1_SOLVER (fully implicit, no scale_m), else SIM_SOLVER;
  // a_imp <= 0.5 (RIM_2D, SIM3) is refused at create
  if (a.a_imp > 0.999) nh_sim1(io, ws, km, a.dt, a.rdgas, gama, a.akap, wsfc, a.p_fac);
  else nh_sim(io, ws, km, a.dt, a.rdgas, gama, a.akap, wsfc, a.a_imp, a.p_fac, a.scale_z);
  for (int k = 1; k <= km; ++k) { io.st(4, k, w2(k)); io.st(5, k, dz2(k)); }
  for (int k = 1; k <= km + 1; ++k) io.st(7, k, pe(k));
  T z = io.cst(zs);
  io.st(6, km + 1, z);
  for (int k = km; k >= 1; --k) { z = z - dz2(k); io.st(6, k, z); }
}

// EDGE_PROFILE (non-uniform levels, no limiter).  f: 0 q1  1 q2 (km)   ->   2 q1e  3 q2e (km+1)
template <class IO>
HD void edge_col(const IO& io, const NhColArgs& a, const ColWs& ws) {
  typedef typename IO::T T;
  const int km = a.g.npz;
  WArr<IO> e1{io, ws, 0}, e2{io, ws, 1};
  auto dp0 = [&](int k) { return a.lev[k - 1].dp_ref; };
  // the elimination factors depend on the reference thicknesses only; recomputed on the way back instead of stored
  const double g0 = dp0(2) / dp0(1);
  double bet = g0 * (g0 + 0.5), gam = (1. + g0 * (g0 + 1.5)) / bet, gk = g0;
  const double xt1 = 2. * g0 * (g0 + 1.);
  T c1 = (xt1 * io.ld(0, 1) + io.ld(0, 2)) / bet, c2 = (xt1 * io.ld(1, 1) + io.ld(1, 2)) / bet;
  e1.set(1, c1); e2.set(1, c2);
  // gam(k) depends on the reference thicknesses only: kept as plain doubles in raw workspace slot 8 (e1, e2 use raw 0..3)
  ColWs wg = ws;
  wg.at(8, 1) = gam;
  for (int k = 2; k <= km; ++k) {
    gk = dp0(k - 1) / dp0(k);
    bet = 2. + 2. * gk - gam;
    c1 = (3. * (io.ld(0, k - 1) + gk * io.ld(0, k)) - c1) / bet;
    c2 = (3. * (io.ld(1, k - 1) + gk * io.ld(1, k)) - c2) / bet;
    e1.set(k, c1); e2.set(k, c2);
    gam = gk / bet;
    wg.at(8, k) = gam;
  }
  const double a_bot = 1. + gk * (gk + 1.5), xb = 2. * gk * (gk + 1.), xt2 = gk * (gk + 0.5) - a_bot * gam;
  c1 = (xb * io.ld(0, km) + io.ld(0, km - 1) - a_bot * c1) / xt2;
  c2 = (xb * io.ld(1, km) + io.ld(1, km - 1) - a_bot * c2) / xt2;
  io.st(2, km + 1, c1); io.st(3, km + 1, c2);
  for (int k = km; k >= 1; --k) {
    const double gmk = wg.at(8, k);
    c1 = e1(k) - gmk * c1; c2 = e2(k) - gmk * c2;
    io.st(2, k, c1); io.st(3, k, c2);
  }
}

// EDGE_PROFILE is linear with coefficients that depend on the reference thicknesses only: its adjoint is the transposed
// solve, written out instead of taped.  Consumes f[2].p, f[3].p (zeroed), accumulates into f[0].p, f[1].p.
HD void edge_col_ad(const NhColArgs& a, const ColWs& ws, int tile, int i, int j) {
  const Geom& g = a.g; const int km = g.npz;
  auto dp0 = [&](int k) { return a.lev[k - 1].dp_ref; };
  // raw workspace slots: 0 bet(k), 1 gam(k), 2 ratio gk(k), 3/4 the two adjoint vectors
  const double g0 = dp0(2) / dp0(1), bet1 = g0 * (g0 + 0.5), xt1 = 2. * g0 * (g0 + 1.);
  double gam = (1. + g0 * (g0 + 1.5)) / bet1, gk = g0;
  ws.at(0, 1) = bet1; ws.at(1, 1) = gam;
  for (int k = 2; k <= km; ++k) {
    gk = dp0(k - 1) / dp0(k);
    const double bet = 2. + 2. * gk - gam;
    gam = gk / bet;
    ws.at(0, k) = bet; ws.at(1, k) = gam; ws.at(2, k) = gk;
  }
  const double a_bot = 1. + gk * (gk + 1.5), xb = 2. * gk * (gk + 1.), xt2 = gk * (gk + 0.5) - a_bot * gam;
  for (int m = 0; m < 2; ++m) {
    const Fld &fi = a.f[m], &fo = a.f[2 + m];
    // incoming adjoint, then the transposed back-substitution (k ascending)
    double prev = fo.p[fidx(g, fo, tile, i, j, 1)];
    fo.p[fidx(g, fo, tile, i, j, 1)] = 0.;
    ws.at(3, 1) = prev;
    for (int k = 1; k <= km; ++k) {
      const size_t n = fidx(g, fo, tile, i, j, k + 1);
      const double cur = fo.p[n] - ws.at(1, k) * prev;
      fo.p[n] = 0.;
      ws.at(3, k + 1) = cur; prev = cur;
    }
    // bottom closure, then the transposed elimination (k descending); q_ad(k) collects up to three contributions
    const double eb = prev;                       // adjoint of e(km+1)
    double ek = ws.at(3, km) - a_bot / xt2 * eb;  // adjoint of e(km) before the closure
    double qk = xb / xt2 * eb, qkm1 = eb / xt2;   // contributions to q(km), q(km-1)
    for (int k = km; k >= 2; --k) {
      const double bet = ws.at(0, k), gkk = ws.at(2, k);
      qk += 3. * gkk / bet * ek;
      fi.p[fidx(g, fi, tile, i, j, k)] += qk;
      qk = qkm1 + 3. / bet * ek; qkm1 = 0.;
      ek = ws.at(3, k - 1) - ek / bet;
    }
    // here qk holds the pending contribution to q(1); k = 1: e(1) = (xt1 q(1) + q(2)) / bet1
    fi.p[fidx(g, fi, tile, i, j, 1)] += qk + xt1 / bet1 * ek;
    fi.p[fidx(g, fi, tile, i, j, 2)] += ek / bet1;
  }
}

// interface heights from the layer thicknesses.  f: 0 delz (km)   ->   1 zh (km+1)
template <class IO>
HD void zh_init_col(const IO& io, const NhColArgs& a, double hs) {
  typename IO::T z = io.cst(hs / a.grav);
  io.st(1, a.g.npz + 1, z);
  for (int k = a.g.npz; k >= 1; --k) { z = z - io.ld(0, k); io.st(1, k, z); }
}

// what = 0: pk3 = p**kappa (PK3_HALO), what = 1: pe (PE_HALO).  f: 0 delp (km)   ->   1 out (km+1), levels 2.. (level 1 is set by the owner)
template <class IO>
HD void ring_col(const IO& io, const NhColArgs& a) {
  typename IO::T p = io.cst(a.ptop);
  if (a.what == 1) io.st(1, 1, p);
  for (int k = 1; k <= a.g.npz; ++k) {
    p = p + io.ld(0, k);
    if (a.what == 1) io.st(1, k + 1, p); else io.st(1, k + 1, dexp(a.akap * dlog(p)));
  }
}

// ---------------------------------------------------------------- vertical remap, non-hydrostatic (fv_mapz_tlm.F90)
// One column of map_scalar / map1_ppm / map1_q2 with the "perfectly linear" profile (|kord| > 16, the only one the TL/AD
// reference implements, fv_mapz_tlm.F90:8653-8666) on the generic column interface.  Inputs staged in the workspace:
// RS_P1 source interfaces (km+1), RS_Q1 layer means (km), RS_P2 target interfaces (km+1).  iv = -2: vertical velocity, the
// value qs at the lower boundary is given (:8549-8590).
enum { RS_P1 = 0, RS_Q1, RS_P2, RS_G, RS_E, RS_COUNT };
// LIM + a limited trajectory profile P: the values of the outputs come from the limited profile (remap.h map_loop_lim on the values, kept in
// the gam slot, which the solve no longer needs), the sensitivities stay those of the linear profile (split_kord).
template <bool LIM = false, class IO, class FOut>
HD void map_col_io(const IO& io, int km, const ColWs& ws, int iv, const typename IO::T& qs, const FOut& out0, const LimProfile& P = LimProfile{17, 1, false, 0.}) {
  typedef typename IO::T T;
  WArr<IO> pe1{io, ws, RS_P1}, q1{io, ws, RS_Q1}, pe2{io, ws, RS_P2}, gam{io, ws, RS_G}, qe{io, ws, RS_E};
  const bool lim = LIM && kord_limited(P.kord);
  auto out = [&](int k, T x) { if (LIM && lim) set_val(x, val(gam(k))); out0(k, x); };
  if (iv == -2) {
    gam.set(2, io.cst(0.5)); qe.set(1, 1.5 * q1(1));
    for (int k = 2; k <= km - 1; ++k) {
      const T grat = (pe1(k) - pe1(k - 1)) / (pe1(k + 1) - pe1(k));
      const T bet = 2. + grat + grat - gam(k);
      qe.set(k, (3. * (q1(k - 1) + q1(k)) - qe(k - 1)) / bet);
      gam.set(k + 1, grat / bet);
    }
    const T grat = (pe1(km) - pe1(km - 1)) / (pe1(km + 1) - pe1(km));
    qe.set(km, (3. * (q1(km - 1) + q1(km)) - grat * qs - qe(km - 1)) / (2. + grat + grat - gam(km)));
    qe.set(km + 1, qs);
    for (int k = km - 1; k >= 1; --k) qe.set(k, qe(k) - gam(k + 1) * qe(k + 1));
  } else {
    const T dpa = pe1(2) - pe1(1), grat = (pe1(3) - pe1(2)) / dpa;
    T bet = grat * (grat + 0.5);
    T qf = ((grat + grat) * (grat + 1.) * q1(1) + q1(2)) / bet, gm = (1. + grat * (grat + 1.5)) / bet;
    qe.set(1, qf); gam.set(1, gm);
    T d4 = grat, dp_prev = dpa;
    for (int k = 2; k <= km; ++k) {
      const T dpk = pe1(k + 1) - pe1(k);
      d4 = dp_prev / dpk;
      bet = 2. + d4 + d4 - gm;
      qf = (3. * (q1(k - 1) + d4 * q1(k)) - qf) / bet;
      gm = d4 / bet;
      qe.set(k, qf); gam.set(k, gm);
      dp_prev = dpk;
    }
    const T a_bot = 1. + d4 * (d4 + 1.5);
    qf = (2. * d4 * (d4 + 1.) * q1(km) + q1(km - 1) - a_bot * qf) / (d4 * (d4 + 0.5) - a_bot * gm);
    qe.set(km + 1, qf);
    for (int k = km; k >= 1; --k) { qf = qe(k) - gam(k) * qf; qe.set(k, qf); }
  }
  if constexpr (LIM) if (lim)
    map_loop_lim(P, km, [&](int k) { return val(pe1(k)); }, [&](int k) { return val(q1(k)); }, [&](int k) { return val(pe2(k)); }, [&](int k) { return val(qe(k)); },
                 [&](int k, double x) { gam.set(k, io.cst(x)); });
  int k0 = 1;
  T qsum = io.cst(0.);
  for (int k = 1; k <= km; ++k) {
    const T p2t = pe2(k), p2b = pe2(k + 1);
    int l; bool found = false;
    for (l = k0; l <= km; ++l)
      if (val(p2t) >= val(pe1(l)) && val(p2t) <= val(pe1(l + 1))) { found = true; break; }
    if (found) {
      const T p1l = pe1(l), p1r = pe1(l + 1), dpl = p1r - p1l;
      const T a1 = q1(l), a2 = qe(l), a3 = qe(l + 1), a4 = 3. * (2. * a1 - (a2 + a3));
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
        const T b1 = q1(m), b2 = qe(m), b3 = qe(m + 1), b4 = 3. * (2. * b1 - (b2 + b3));
        const T dp = p2b - p1m, esl = dp / dpm;
        qsum = qsum + dp * (b2 + 0.5 * esl * (b3 - b2 + b4 * (1. - R23 * esl)));
        k0 = m;
      }
    }
    out(k, qsum / (p2b - p2t));
  }
}
// Eulerian target interfaces from the surface pressure (fv_mapz_tlm.F90:1644-1661)
template <class IO>
HD void rm_target(const IO& io, const NhColArgs& a, const ColWs& ws, int slot_pe) {
  typedef typename IO::T T;
  const int km = a.g.npz;
  WArr<IO> p2{io, ws, RS_P2};
  const T ps = io.ld(slot_pe, km + 1);
  for (int k = 1; k <= km + 1; ++k) {
    const T p = k == 1 ? io.cst(a.ptop) : k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps;
    p2.set(k, p);
  }
}
// what: 0 temperature, 1 vertical velocity, 2 layer thickness, 3 tracer.
//   0  f: 0 pe 1 peln 2 pt 3 delp 4 delz -> 5 T_v on the new levels   (density pt -> density T :1607-1613, map_scalar in log p :1676-1688)
//   1  f: 0 pe 1 w 2 ws -> 3 w                                         (map1_ppm, iv = -2, :1772-1780)
//   2  f: 0 pe 1 delz 2 delp -> 3 delz                                 (specific volume / g :1635-1641, map1_ppm :1782-1796)
//   3  f: 0 pe 1 q -> 2 q                                              (map1_q2 :1746-1763)
template <bool LIM = false, class IO>
HD void remap_field_col_nh(const IO& io, const NhColArgs& a, const ColWs& ws) {
  typedef typename IO::T T;
  const int km = a.g.npz;
  WArr<IO> p1{io, ws, RS_P1}, q1{io, ws, RS_Q1}, p2{io, ws, RS_P2};
  rm_target(io, a, ws, 0);
  if (a.what == 0) {
    const double rrg = -a.rdgas / a.grav, k1k = a.rdgas / (a.cp_air - a.rdgas);
    for (int k = 1; k <= km + 1; ++k) p1.set(k, io.ld(1, k));
    for (int k = 2; k <= km; ++k) p2.set(k, dlog(p2(k)));
    p2.set(1, p1(1)); p2.set(km + 1, p1(km + 1));
    for (int k = 1; k <= km; ++k) { const T pt = io.ld(2, k); q1.set(k, pt * dexp(k1k * dlog(rrg * io.ld(3, k) / io.ld(4, k) * pt))); }
    map_col_io<LIM>(io, km, ws, 1, io.cst(0.), [&](int k, const T& x) { io.st(5, k, x); }, LimProfile{a.kord_tm, 1, true, 184.});      // map_scalar, t_min = 184
    return;
  }
  for (int k = 1; k <= km + 1; ++k) p1.set(k, io.ld(0, k));
  if (a.what == 1) {
    for (int k = 1; k <= km; ++k) q1.set(k, io.ld(1, k));
    map_col_io<LIM>(io, km, ws, -2, io.ld(2, 1), [&](int k, const T& x) { io.st(3, k, x); }, LimProfile{a.kord_wz, -2, false, 0.});
  } else if (a.what == 2) {
    for (int k = 1; k <= km; ++k) q1.set(k, -(io.ld(1, k) / io.ld(2, k)));
    map_col_io<LIM>(io, km, ws, 1, io.cst(0.), [&](int k, const T& x) { io.st(3, k, -(x * (p2(k + 1) - p2(k)))); }, LimProfile{a.kord_tm, 1, false, 0.});      // map1_ppm with abs(kord_tm) (fv_mapz_tlm.F90:627-637)
  } else {
    for (int k = 1; k <= km; ++k) q1.set(k, io.ld(1, k));
    map_col_io<LIM>(io, km, ws, 0, io.cst(0.), [&](int k, const T& x) { io.st(2, k, x); }, LimProfile{a.kord_tr, 0, true, 0.});      // map1_q2: iv = 0 (read by the limited profile only)
  }
}
// Adjoint of remap_field_col_nh without the tape: the column map itself is the hydrostatic path's hand-written
// map_col_ad (remap.h; map_col_ad_iv<-2> for the vertical velocity), the pre- and post-transforms are differentiated
// here.  FV3LM_NH_TAPE=1 keeps the w map on the tape (NHC_RM_W), which is what the tests compare against.
// Raw workspace slots: 0..12 map_col_ad, 13..15 staged trajectories.
template <bool LIM = false>
HD void remap_field_col_nh_ad(const NhColArgs& a, const ColWs& ws, int tile, int i, int j) {
  const Geom& g = a.g; const int km = g.npz;
  const MapAdSlots S{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
  auto F = [&](int sl, int k) -> size_t { return fidx(g, a.f[sl], tile, i, j, k); };
  auto pe1 = [&](int k) { return a.f[0].t[F(0, k)]; };
  const double ps = pe1(km + 1);
  auto pe2 = [&](int k) -> double { return k == 1 ? a.ptop : (k == km + 1 ? ps : a.ak[k - 1] + a.bk[k - 1] * ps); };
  for (int k = 0; k <= km + 1; ++k) { ws.at(S.SP1, k) = 0.; ws.at(S.SQ1, k) = 0.; ws.at(S.SP2, k) = 0.; }
  double ps_ad = 0.;
  if (a.what == 0) {
    const double rrg = -a.rdgas / a.grav, k1k = a.rdgas / (a.cp_air - a.rdgas);
    auto pn1 = [&](int k) { return a.f[1].t[F(1, k)]; };
    for (int k = 1; k <= km; ++k) {
      const double pt = a.f[2].t[F(2, k)];
      ws.at(13, k) = pt * exp(k1k * log(rrg * a.f[3].t[F(3, k)] / a.f[4].t[F(4, k)] * pt));
    }
    for (int k = 1; k <= km + 1; ++k) ws.at(14, k) = (k == 1 || k == km + 1) ? pn1(k) : log(pe2(k));
    auto tv = [&](int k) { return ws.at(13, k); };
    auto pn2 = [&](int k) { return ws.at(14, k); };
    auto q2ad = [&](int k) { return a.f[5].p[F(5, k)]; };
    map_col_ad(km, pn1, tv, pn2, q2ad, ws, S);
    for (int k = 1; k <= km; ++k) {
      const double ta = ws.at(S.SQ1, k), t = ws.at(13, k), pt = a.f[2].t[F(2, k)];
      a.f[2].p[F(2, k)] += ta * (t / pt) * (1. + k1k);
      a.f[3].p[F(3, k)] += ta * t * k1k / a.f[3].t[F(3, k)];
      a.f[4].p[F(4, k)] -= ta * t * k1k / a.f[4].t[F(4, k)];
      a.f[5].p[F(5, k)] = 0.;
    }
    for (int k = 1; k <= km + 1; ++k) a.f[1].p[F(1, k)] += ws.at(S.SP1, k);
    a.f[1].p[F(1, 1)] += ws.at(S.SP2, 1); a.f[1].p[F(1, km + 1)] += ws.at(S.SP2, km + 1);
    for (int k = 2; k <= km; ++k) ps_ad += a.bk[k - 1] * ws.at(S.SP2, k) / pe2(k);
    a.f[0].p[F(0, km + 1)] += ps_ad;
    return;
  }
  if (a.what == 2) {
    auto q1 = [&](int k) { return -(a.f[1].t[F(1, k)] / a.f[2].t[F(2, k)]); };
    auto outq = [&](int k, double x) { ws.at(15, k) = x; };
    bool done = false;
    if constexpr (LIM) if (kord_limited(a.kord_tm)) { map_col_lim(LimProfile{a.kord_tm, 1, false, 0.}, km, pe1, q1, pe2, outq, ws, 13, 14); done = true; }
    if (!done) map_col<double>(km, pe1, q1, pe2, outq, ws, 13, 14);          // mapped -delz/delp: needed by the product rule below
    for (int k = 1; k <= km; ++k) {
      const double oa = a.f[3].p[F(3, k)], dpa = -ws.at(15, k) * oa;
      ws.at(S.SP2, k + 1) += dpa; ws.at(S.SP2, k) -= dpa;
    }
    auto q2ad = [&](int k) { return -(pe2(k + 1) - pe2(k)) * a.f[3].p[F(3, k)]; };
    map_col_ad(km, pe1, q1, pe2, q2ad, ws, S);
    for (int k = 1; k <= km; ++k) {
      const double qa = ws.at(S.SQ1, k), dp = a.f[2].t[F(2, k)];
      a.f[1].p[F(1, k)] -= qa / dp;
      a.f[2].p[F(2, k)] += qa * a.f[1].t[F(1, k)] / (dp * dp);
      a.f[3].p[F(3, k)] = 0.;
    }
  } else if (a.what == 1) {       // vertical velocity: profile with the surface value ws as lower boundary (iv = -2)
    auto q1 = [&](int k) { return a.f[1].t[F(1, k)]; };
    auto q2ad = [&](int k) { return a.f[3].p[F(3, k)]; };
    double a_qs = 0.;
    map_col_ad_iv<-2>(km, pe1, q1, pe2, q2ad, ws, S, a.f[2].t[F(2, 1)], &a_qs);
    for (int k = 1; k <= km; ++k) { a.f[1].p[F(1, k)] += ws.at(S.SQ1, k); a.f[3].p[F(3, k)] = 0.; }
    a.f[2].p[F(2, 1)] += a_qs;
  } else {
    auto q1 = [&](int k) { return a.f[1].t[F(1, k)]; };
    auto q2ad = [&](int k) { return a.f[2].p[F(2, k)]; };
    map_col_ad(km, pe1, q1, pe2, q2ad, ws, S);
    for (int k = 1; k <= km; ++k) { a.f[1].p[F(1, k)] += ws.at(S.SQ1, k); a.f[2].p[F(2, k)] = 0.; }
  }
  for (int k = 1; k <= km + 1; ++k) a.f[0].p[F(0, k)] += ws.at(S.SP1, k);
  ps_ad = ws.at(S.SP2, km + 1);
  for (int k = 2; k <= km; ++k) ps_ad += a.bk[k - 1] * ws.at(S.SP2, k);
  a.f[0].p[F(0, km + 1)] += ps_ad;
}

// New pressures, pkz from the equation of state, and the temperature hand-over (:1644-1661, :1798-1812, :1852-1857, :2203-2250).
// f: 0 pe  1 peln  2 pk (end interfaces read, interior interfaces written)  3 T_v  4 delz  5 q_v (what = 1)  ->  6 delp  7 pkz  8 pt  9 pe2
template <class IO>
HD void remap_press_col_nh(const IO& io, const NhColArgs& a) {
  typedef typename IO::T T;
  const int km = a.g.npz;
  const double rrg = -a.rdgas / a.grav;
  const T ps = io.ld(0, km + 1), pn_bot = io.ld(1, km + 1), pk_bot = io.ld(2, km + 1);
  T pe_hi = io.cst(a.ptop), pn_hi = io.ld(1, 1), pk_hi = io.ld(2, 1);
  io.st(9, 1, pe_hi);
  for (int k = 1; k <= km; ++k) {
    T pe_lo, pn_lo, pk_lo;
    if (k == km) { pe_lo = ps; pn_lo = pn_bot; pk_lo = pk_bot; }
    else { pe_lo = a.ak[k] + a.bk[k] * ps; pn_lo = dlog(pe_lo); pk_lo = dexp(a.akap * pn_lo); }
    const T dp = pe_lo - pe_hi, t = io.ld(3, k);
    const T pkz = dexp(a.akap * dlog(rrg * dp / io.ld(4, k) * t));
    io.st(6, k, dp); io.st(7, k, pkz);
    if (a.last_call) { if (a.what) io.st(8, k, t / (1. + a.zvir * io.ld(5, k))); else io.st(8, k, t); }
    else io.st(8, k, t / pkz);
    if (k > 1) { io.st(1, k, pn_hi); io.st(2, k, pk_hi); }
    io.st(9, k + 1, pe_lo);
    pe_hi = pe_lo; pn_hi = pn_lo; pk_hi = pk_lo;
  }
}

// adjoint of zh_init_col: zh(k) = zs - sum_{l >= k} delz(l)
HD void zh_init_col_ad(const NhColArgs& a, int tile, int i, int j) {
  const Geom& g = a.g; const int km = g.npz;
  double G = 0.;
  for (int k = 1; k <= km; ++k) {
    const size_t n = fidx(g, a.f[1], tile, i, j, k);
    G += a.f[1].p[n]; a.f[1].p[n] = 0.;
    a.f[0].p[fidx(g, a.f[0], tile, i, j, k)] -= G;
  }
  a.f[1].p[fidx(g, a.f[1], tile, i, j, km + 1)] = 0.;
}
// adjoint of remap_press_col_nh.  f: 0 pe  1 peln  2 pk  3 T_v  4 delz  5 q_v (what = 1)  ->  6 delp  7 pkz  8 pt  9 pe2
HD void remap_press_col_nh_ad(const NhColArgs& a, int tile, int i, int j) {
  const Geom& g = a.g; const int km = g.npz;
  const double rrg = -a.rdgas / a.grav;
  auto F = [&](int sl, int k) -> size_t { return fidx(g, a.f[sl], tile, i, j, k); };
  const double ps = a.f[0].t[F(0, km + 1)];
  double ps_ad = 0., pe_hi = a.ptop;
  double a_hi = 0.;        // adjoint of the upper interface pressure of layer k, from dp(k) = pe_lo - pe_hi (pe_hi of layer 1 is the constant ptop)
  a.f[9].p[F(9, 1)] = 0.;
  for (int k = 1; k <= km; ++k) {
    const double pe_lo = (k == km) ? ps : a.ak[k] + a.bk[k] * ps;
    const double dp = pe_lo - pe_hi, t = a.f[3].t[F(3, k)], dz = a.f[4].t[F(4, k)];
    const double pkz = exp(a.akap * log(rrg * dp / dz * t));
    const double po = a.f[8].p[F(8, k)];
    double a_t, a_pkz = a.f[7].p[F(7, k)];
    if (a.last_call) {
      if (a.what) { const double q = a.f[5].t[F(5, k)], f1 = 1. + a.zvir * q; a_t = po / f1; a.f[5].p[F(5, k)] -= po * t * a.zvir / (f1 * f1); }
      else a_t = po;
    } else { a_t = po / pkz; a_pkz -= po * t / (pkz * pkz); }
    const double c = a_pkz * a.akap * pkz;
    a_t += c / t;
    a.f[3].p[F(3, k)] += a_t;
    a.f[4].p[F(4, k)] -= c / dz;
    const double a_dp = a.f[6].p[F(6, k)] + c / dp;
    a.f[6].p[F(6, k)] = 0.; a.f[7].p[F(7, k)] = 0.; a.f[8].p[F(8, k)] = 0.;
    // interface k (upper, pe_hi) is complete now: dp(k-1) gave +, dp(k) gives -, plus the peln / pk / pe2 outputs at level k
    if (k > 1) {
      const double pn = log(pe_hi), pkh = exp(a.akap * pn);
      const double a_pe = a_hi - a_dp + a.f[1].p[F(1, k)] / pe_hi + a.akap * pkh / pe_hi * a.f[2].p[F(2, k)] + a.f[9].p[F(9, k)];
      a.f[1].p[F(1, k)] = 0.; a.f[2].p[F(2, k)] = 0.; a.f[9].p[F(9, k)] = 0.;
      ps_ad += a.bk[k - 1] * a_pe;
    }
    a_hi = a_dp;
    pe_hi = pe_lo;
  }
  ps_ad += a_hi + a.f[9].p[F(9, km + 1)];
  a.f[9].p[F(9, km + 1)] = 0.;
  a.f[0].p[F(0, km + 1)] += ps_ad;
}
// adjoint of ring_col written out (a cumulative sum and, for pk3, one power per level): consumes f[1].p, accumulates into f[0].p
HD void ring_col_ad(const NhColArgs& a, int tile, int i, int j) {
  const Geom& g = a.g; const int km = g.npz;
  double p = a.ptop;
  for (int k = 1; k <= km; ++k) p += a.f[0].t[fidx(g, a.f[0], tile, i, j, k)];
  if (a.what == 1) a.f[1].p[fidx(g, a.f[1], tile, i, j, 1)] = 0.;
  double sum = 0.;
  for (int k = km; k >= 1; --k) {
    const size_t n = fidx(g, a.f[1], tile, i, j, k + 1);
    const double oa = a.f[1].p[n];
    a.f[1].p[n] = 0.;
    sum += (a.what == 1) ? oa : a.akap * exp(a.akap * log(p)) / p * oa;
    const size_t m = fidx(g, a.f[0], tile, i, j, k);
    a.f[0].p[m] += sum;
    p -= a.f[0].t[m];
  }
}

enum NhColKind { NHC_RIEM_C = 0, NHC_RIEM3, NHC_EDGE, NHC_ZH_INIT, NHC_RING, NHC_RM_FIELD, NHC_RM_PRESS, NHC_RM_W, NHC_RM_FIELD_LIM, NHC_RM_W_LIM };
// One kernel per (operator, mode): each gets its own register allocation (the taped remap needs ~250 VGPRs, the nonlinear
// solvers a fraction of that).
template <int KIND, int MODE>
struct NhColFn {
  NhColArgs a; Rect skip;                     // skip: rectangle left out (the rings of NHC_RING are a frame around it)
  int z0;                                     // first tile of the launch (the adjoint runs in chunks of tiles: as many as the tape holds)
  template <class IO>
  HD void body(const IO& io, const ColWs& ws, double hs) const {
    if (KIND == NHC_RIEM_C) riem_c_col(io, a, ws, hs);
    else if (KIND == NHC_RIEM3) riem3_col(io, a, ws, hs);
    else if (KIND == NHC_EDGE) edge_col(io, a, ws);
    else if (KIND == NHC_ZH_INIT) zh_init_col(io, a, hs);
    else if (KIND == NHC_RING) ring_col(io, a);
    else if (KIND == NHC_RM_FIELD || KIND == NHC_RM_W) remap_field_col_nh(io, a, ws);
    else if (KIND == NHC_RM_FIELD_LIM || KIND == NHC_RM_W_LIM) remap_field_col_nh<true>(io, a, ws);
    else remap_press_col_nh(io, a);
  }
  HD void operator()(int i, int j, int zz) const {
    const int z = zz + z0;
    if (KIND == NHC_RING && skip.has(i, j)) return;
    // corner-halo columns of a face hold no data (cell-centred operators; the flux points of NHC_EDGE all touch a valid cell)
    if (KIND != NHC_EDGE && a.g.face && (i < 1 || i > a.g.nx) && (j < 1 || j > a.g.ny)) return;
    const size_t col = (size_t)z * a.g.plane + a.g.idx(i, j);
    const ColWs ws{a.ws + col, a.ws_stride, a.g.npz + 2};
    const double hs = a.hs ? a.hs[col] : 0.;
    if (KIND == NHC_EDGE && MODE == MODE_AD) { edge_col_ad(a, ws, z, i, j); return; }
    if (KIND == NHC_RING && MODE == MODE_AD) { ring_col_ad(a, z, i, j); return; }
    if (KIND == NHC_ZH_INIT && MODE == MODE_AD && !a.use_tape) { zh_init_col_ad(a, z, i, j); return; }
    if (KIND == NHC_RM_PRESS && MODE == MODE_AD && !a.use_tape) { remap_press_col_nh_ad(a, z, i, j); return; }
    if ((KIND == NHC_RM_FIELD || KIND == NHC_RM_W) && MODE == MODE_AD && !(KIND == NHC_RM_W && a.use_tape)) { remap_field_col_nh_ad(a, ws, z, i, j); return; }
    if ((KIND == NHC_RM_FIELD_LIM || KIND == NHC_RM_W_LIM) && MODE == MODE_AD && !(KIND == NHC_RM_W_LIM && a.use_tape)) { remap_field_col_nh_ad<true>(a, ws, z, i, j); return; }
    if (KIND == NHC_RIEM_C && MODE == MODE_AD && !a.use_tape) { riem_c_col_ad(a, ws, z, i, j, hs); return; }
    if (KIND == NHC_RIEM3 && MODE == MODE_AD && !a.use_tape) { riem3_col_ad(a, ws, z, i, j, hs); return; }
    if (MODE == MODE_NL) { ColNL io{a.g, a.f, z, i, j, nullptr}; body(io, ws, hs); }
    else if (MODE == MODE_TL) { ColTL io{a.g, a.f, z, i, j, nullptr}; body(io, ws, hs); }
    else {
      Tape t{a.tape, (size_t)zz * a.g.plane + a.g.idx(i, j), 0};
      ColAD io{a.g, a.f, z, i, j, &t};
      body(io, ws, hs);
      io.reverse();
    }
  }
};
// Algorithmic bytes of one column-operator launch (bench.py roofline leg): the operator's input and output fields, each touched once
// per cell (8 B), x2 in the tangent mode, and inputs + outputs read plus input adjoints read-modify-written in the adjoint -- the same
// rule as exec.h stage_bytes.  Workspace and tape traffic is not algorithmic: it shows up as the gap between this and the PMC traffic.
inline double nh_col_bytes(const NhColArgs& a, int kind, int mode, const Rect& R, const Rect& skip) {
  int nin = 0, nout = 0;
  switch (kind) {
    case NHC_RIEM_C: nin = 4; nout = 2; break;
    case NHC_RIEM3: nin = 4; nout = a.last_call ? 9 : 5; break;
    case NHC_EDGE: nin = 2; nout = 2; break;
    case NHC_ZH_INIT: nin = 1; nout = 1; break;
    case NHC_RING: nin = 1; nout = 1; break;
    case NHC_RM_FIELD: case NHC_RM_W: case NHC_RM_FIELD_LIM: case NHC_RM_W_LIM: nin = a.what == 0 ? 5 : a.what == 3 ? 2 : 3; nout = 1; break;
    default: nin = 6; nout = 4; break;      // NHC_RM_PRESS
  }
  double cols = double(R.i1 - R.i0 + 1) * double(R.j1 - R.j0 + 1);
  if (skip.i0 <= skip.i1 && skip.j0 <= skip.j1) cols -= double(skip.i1 - skip.i0 + 1) * double(skip.j1 - skip.j0 + 1);
  const double cells = cols * a.g.ntile * a.g.npz;
  const double per = mode == MODE_NL ? nin + nout : mode == MODE_TL ? 2. * (nin + nout) : (nin + nout + 2. * nin);
  return 8. * per * cells;
}
template <int KIND, int MODE>
inline void run_nh_col_km(Exec& ex, const NhColArgs& a, const Rect& R, const Rect& skip, const char* tag) {
  const double bytes = nh_col_bytes(a, KIND, MODE, R, skip);
  if (MODE != MODE_AD || !a.use_tape) { for_points(ex, R, a.g.ntile, NhColFn<KIND, MODE>{a, skip, 0}, tag, bytes); return; }   // hand-written adjoints need no tape
  const int chunk = (int)(a.tape.stride / a.g.plane);       // tiles the tape holds at once (dycore.h sizes it from the free HBM)
  for (int z = 0; z < a.g.ntile; z += chunk) for_points(ex, R, std::min(chunk, a.g.ntile - z), NhColFn<KIND, MODE>{a, skip, z}, tag, bytes * std::min(chunk, a.g.ntile - z) / a.g.ntile);
}
template <int KIND>
inline void run_nh_col_k(Exec& ex, int mode, const NhColArgs& a, const Rect& R, const Rect& skip, const char* tag) {
  if (mode == MODE_NL) run_nh_col_km<KIND, MODE_NL>(ex, a, R, skip, tag);
  else if (mode == MODE_TL) run_nh_col_km<KIND, MODE_TL>(ex, a, R, skip, tag);
  else run_nh_col_km<KIND, MODE_AD>(ex, a, R, skip, tag);
}
inline void run_nh_col(Exec& ex, int mode, const NhColArgs& a0, int kind, const Rect& R, const Rect& skip, const char* tag) {
  NhColArgs a = a0;
  for (int n = 0; n < NH_NF; ++n) a.f[n] = ex.sh(a.f[n]);
  if (a.hs) a.hs += ex.cls_off;
  if (a.ws) a.ws += ex.cls_off;      // column workspace: columns of the class's first tile (ws_stride counts the columns of ALL resident tiles)
  switch (kind) {
    case NHC_RIEM_C: run_nh_col_k<NHC_RIEM_C>(ex, mode, a, R, skip, tag); break;
    case NHC_RIEM3: run_nh_col_k<NHC_RIEM3>(ex, mode, a, R, skip, tag); break;
    case NHC_EDGE: run_nh_col_k<NHC_EDGE>(ex, mode, a, R, skip, tag); break;
    case NHC_ZH_INIT: run_nh_col_k<NHC_ZH_INIT>(ex, mode, a, R, skip, tag); break;
    case NHC_RING: run_nh_col_k<NHC_RING>(ex, mode, a, R, skip, tag); break;
    case NHC_RM_FIELD: run_nh_col_k<NHC_RM_FIELD>(ex, mode, a, R, skip, tag); break;
    case NHC_RM_W: run_nh_col_k<NHC_RM_W>(ex, mode, a, R, skip, tag); break;
    case NHC_RM_FIELD_LIM: run_nh_col_k<NHC_RM_FIELD_LIM>(ex, mode, a, R, skip, tag); break;
    case NHC_RM_W_LIM: run_nh_col_k<NHC_RM_W_LIM>(ex, mode, a, R, skip, tag); break;
    default: run_nh_col_k<NHC_RM_PRESS>(ex, mode, a, R, skip, tag); break;
  }
}

}  // namespace fv3
