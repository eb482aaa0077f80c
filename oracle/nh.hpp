// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Non-hydrostatic pieces of the acoustic step, restated from the reference's nonlinear routines:
//   model_tlmadm/nh_utils_tlm.F90   UPDATE_DZ_C (:235-380), UPDATE_DZ_D (:590-722), RIEM_SOLVER_C (:846-933),
//                                   SIM1_SOLVER (:2760-2884), SIM_SOLVER (:3129-3272), EDGE_PROFILE (:3473-3584)
//   model_tlmadm/nh_core_tlm.F90    RIEM_SOLVER3 (:245-389)
//   model_tlmadm/dyn_core_tlm.F90   P_GRAD_C non-hydrostatic branch (:3279-3336), NH_P_GRAD (:3491-3588),
//                                   PK3_HALO (:2716-2778), PE_HALO (:2934-2961), DYN_CORE non-hydrostatic flow (:1701-2466)
// Path restated: a_imp in (0.5, 0.999] (default 0.75): SIM_SOLVER in RIEM_SOLVER3, SIM1_SOLVER in RIEM_SOLVER_C;
// use_logp = .false., beta = 0, d_con = 0, do_f3d = .false., not nested.  "Parity unpinned" like the rest of the path.
#pragma once
#include "cube.hpp"

namespace orc {

static const double nh_dz_min = 2.;          // nh_utils_tlm.F90: dz_min
static const double nh_r3 = 1. / 3.;

struct NhOpts { double a_imp = 0.75, p_fac = 0.05, scale_z = 0.; };

// 1-based column vectors
template <class T> using Col = std::vector<T>;

// SIM1_SOLVER, nh_utils_tlm.F90:2760-2884.  dm2, pm2, pem, pt2 in; w2, dz2 in/out; pe out (km+1).
template <class T>
void sim1_solver(double dt, int km, double rgas, double gama, double kappa, Col<T>& pe, const Col<T>& dm2, const Col<T>& pm2, const Col<T>& pem,
                 Col<T>& w2, Col<T>& dz2, const Col<T>& pt2, const T& ws, double p_fac) {
  const double t1g = gama * 2. * dt * dt, rdt = 1. / dt, capa1 = kappa - 1.;
  Col<T> aa(km + 2), bb(km + 2), dd(km + 2), w1(km + 2), g_rat(km + 2), gam(km + 2), pp(km + 3);
  for (int k = 1; k <= km; ++k) {
    w1[k] = w2[k];
    pe[k] = exp(gama * log(-(dm2[k] / dz2[k] * rgas * pt2[k]))) - pm2[k];
  }
  for (int k = 1; k <= km - 1; ++k) {
    g_rat[k] = dm2[k] / dm2[k + 1];
    bb[k] = 2. * (1. + g_rat[k]);
    dd[k] = 3. * (pe[k] + g_rat[k] * pe[k + 1]);
  }
  T bet = bb[1];
  pp[1] = T(0.); pp[2] = dd[1] / bet;
  bb[km] = T(2.); dd[km] = 3. * pe[km];
  for (int k = 2; k <= km; ++k) { gam[k] = g_rat[k - 1] / bet; bet = bb[k] - gam[k]; pp[k + 1] = (dd[k] - pp[k]) / bet; }
  for (int k = km; k >= 2; --k) pp[k] = pp[k] - gam[k] * pp[k + 1];
  for (int k = 2; k <= km; ++k) aa[k] = t1g / (dz2[k - 1] + dz2[k]) * (pem[k] + pp[k]);
  bet = dm2[1] - aa[2];
  w2[1] = (dm2[1] * w1[1] + dt * pp[2]) / bet;
  for (int k = 2; k <= km - 1; ++k) {
    gam[k] = aa[k] / bet;
    bet = dm2[k] - (aa[k] + aa[k + 1] + aa[k] * gam[k]);
    w2[k] = (dm2[k] * w1[k] + dt * (pp[k + 1] - pp[k]) - aa[k] * w2[k - 1]) / bet;
  }
  T p1 = t1g / dz2[km] * (pem[km + 1] + pp[km + 1]);
  gam[km] = aa[km] / bet;
  bet = dm2[km] - (aa[km] + p1 + aa[km] * gam[km]);
  w2[km] = (dm2[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - p1 * ws - aa[km] * w2[km - 1]) / bet;
  for (int k = km - 1; k >= 1; --k) w2[k] = w2[k] - gam[k + 1] * w2[k + 1];
  pe[1] = T(0.);
  for (int k = 1; k <= km; ++k) pe[k + 1] = pe[k] + dm2[k] * (w2[k] - w1[k]) * rdt;
  p1 = (pe[km] + 2. * pe[km + 1]) * nh_r3;
  {
    T mx = (p_fac * val(pm2[km]) < val(p1) + val(pm2[km])) ? p1 + pm2[km] : p_fac * pm2[km];
    dz2[km] = -(dm2[km] * rgas * pt2[km] * exp(capa1 * log(mx)));
  }
  for (int k = km - 1; k >= 1; --k) {
    p1 = (pe[k] + bb[k] * pe[k + 1] + g_rat[k] * pe[k + 2]) * nh_r3 - g_rat[k] * p1;
    T mx = (p_fac * val(pm2[k]) < val(p1) + val(pm2[k])) ? p1 + pm2[k] : p_fac * pm2[k];
    dz2[k] = -(dm2[k] * rgas * pt2[k] * exp(capa1 * log(mx)));
  }
}

// SIM_SOLVER, nh_utils_tlm.F90:3129-3272 (semi-implicit with off-centering alpha = a_imp).
template <class T>
void sim_solver(double dt, int km, double rgas, double gama, double kappa, Col<T>& pe2, const Col<T>& dm2, const Col<T>& pm2, const Col<T>& pem,
                Col<T>& w2, Col<T>& dz2, const Col<T>& pt2, const T& ws, double alpha, double p_fac, double scale_m) {
  const double beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha, t1g = 2. * gama * (alpha * dt) * (alpha * dt), rdt = 1. / dt, capa1 = kappa - 1.;
  Col<T> aa(km + 2), bb(km + 2), dd(km + 2), w1(km + 2), wk(km + 2), g_rat(km + 2), gam(km + 2), pp(km + 3);
  for (int k = 1; k <= km; ++k) {
    w1[k] = w2[k];
    pe2[k] = exp(gama * log(-(dm2[k] / dz2[k] * rgas * pt2[k]))) - pm2[k];
  }
  for (int k = 1; k <= km - 1; ++k) {
    g_rat[k] = dm2[k] / dm2[k + 1];
    bb[k] = 2. * (1. + g_rat[k]);
    dd[k] = 3. * (pe2[k] + g_rat[k] * pe2[k + 1]);
  }
  T bet = bb[1];
  pp[1] = T(0.); pp[2] = dd[1] / bet;
  bb[km] = T(2.); dd[km] = 3. * pe2[km];
  for (int k = 2; k <= km; ++k) { gam[k] = g_rat[k - 1] / bet; bet = bb[k] - gam[k]; pp[k + 1] = (dd[k] - pp[k]) / bet; }
  for (int k = km; k >= 2; --k) pp[k] = pp[k] - gam[k] * pp[k + 1];
  for (int k = 1; k <= km + 1; ++k) pe2[k] = pem[k] + pp[k];
  for (int k = 2; k <= km; ++k) {
    aa[k] = t1g / (dz2[k - 1] + dz2[k]) * pe2[k];
    wk[k] = t2 * aa[k] * (w1[k - 1] - w1[k]);
    aa[k] = aa[k] - scale_m * dm2[1];
  }
  bet = dm2[1] - aa[2];
  w2[1] = (dm2[1] * w1[1] + dt * pp[2] + wk[2]) / bet;
  for (int k = 2; k <= km - 1; ++k) {
    gam[k] = aa[k] / bet;
    bet = dm2[k] - (aa[k] + aa[k + 1] + aa[k] * gam[k]);
    w2[k] = (dm2[k] * w1[k] + dt * (pp[k + 1] - pp[k]) + wk[k + 1] - wk[k] - aa[k] * w2[k - 1]) / bet;
  }
  T wk1 = t1g / dz2[km] * pe2[km + 1];
  gam[km] = aa[km] / bet;
  bet = dm2[km] - (aa[km] + wk1 + aa[km] * gam[km]);
  w2[km] = (dm2[km] * w1[km] + dt * (pp[km + 1] - pp[km]) - wk[km] + wk1 * (t2 * w1[km] - ra * ws) - aa[km] * w2[km - 1]) / bet;
  for (int k = km - 1; k >= 1; --k) w2[k] = w2[k] - gam[k + 1] * w2[k + 1];
  pe2[1] = T(0.);
  for (int k = 1; k <= km; ++k) pe2[k + 1] = pe2[k] + (dm2[k] * (w2[k] - w1[k]) * rdt - beta * (pp[k + 1] - pp[k])) * ra;
  T p1 = (pe2[km] + 2. * pe2[km + 1]) * nh_r3;
  {
    T mx = (p_fac * val(pm2[km]) < val(p1) + val(pm2[km])) ? p1 + pm2[km] : p_fac * pm2[km];
    dz2[km] = -(dm2[km] * rgas * pt2[km] * exp(capa1 * log(mx)));
  }
  for (int k = km - 1; k >= 1; --k) {
    p1 = (pe2[k] + bb[k] * pe2[k + 1] + g_rat[k] * pe2[k + 2]) * nh_r3 - g_rat[k] * p1;
    T mx = (p_fac * val(pm2[k]) < val(p1) + val(pm2[k])) ? p1 + pm2[k] : p_fac * pm2[k];
    dz2[k] = -(dm2[k] * rgas * pt2[k] * exp(capa1 * log(mx)));
  }
  for (int k = 1; k <= km + 1; ++k) pe2[k] = pe2[k] + beta * (pp[k] - pe2[k]);
}

// RIEM_SOLVER_C, nh_utils_tlm.F90:846-933 (a_imp > 0.5: SIM1_SOLVER) on is-1..ie+1, js-1..je+1.
// gz (km+1) in/out [height*grav], pef out (full pressure at interfaces), w3 = C-grid w, ws = surface w.
template <class T>
void riem_solver_c(double dt, int km, double akap, double ptop, const Arr2<double>& hs, const Arr3<T>& w3, const Arr3<T>& pt, const Arr3<T>& delp,
                   Arr3<T>& gz, Arr3<T>& pef, const Arr2<T>& ws, const Consts& c, const NhOpts& nh, const Bounds& bd) {
  const double gama = 1. / (1. - akap), rgrav = 1. / c.grav;
  Col<T> dm(km + 2), dz2(km + 2), pm2(km + 2), w2(km + 2), pt2(km + 2), pem(km + 3), pe2(km + 3);
  for (int j = bd.js - 1; j <= bd.je + 1; ++j)
    for (int i = bd.is - 1; i <= bd.ie + 1; ++i) {
      for (int k = 1; k <= km; ++k) dm[k] = delp(i, j, k);
      pef(i, j, 1) = T(ptop); pem[1] = T(ptop);
      for (int k = 2; k <= km + 1; ++k) pem[k] = pem[k - 1] + dm[k - 1];
      for (int k = 1; k <= km; ++k) {
        dz2[k] = gz(i, j, k + 1) - gz(i, j, k);
        pm2[k] = dm[k] / log(pem[k + 1] / pem[k]);
        dm[k] = dm[k] * rgrav;
        w2[k] = w3(i, j, k);
        pt2[k] = pt(i, j, k);
      }
      sim1_solver(dt, km, c.rdgas, gama, akap, pe2, dm, pm2, pem, w2, dz2, pt2, ws(i, j), nh.p_fac);
      for (int k = 2; k <= km + 1; ++k) pef(i, j, k) = pe2[k] + pem[k];
      gz(i, j, km + 1) = T(hs(i, j));
      for (int k = km; k >= 1; --k) gz(i, j, k) = gz(i, j, k + 1) - dz2[k] * c.grav;
    }
}

// RIEM_SOLVER3, nh_core_tlm.F90:245-389 (a_imp in (0.5, 0.999]: SIM_SOLVER, a_imp > 0.999: SIM1_SOLVER; use_logp = fp_out = .false.) on is..ie, js..je.
template <class T>
void riem_solver3(double dt, int km, double akap, double ptop, const Arr2<double>& zs, Arr3<T>& w, Arr3<T>& delz, const Arr3<T>& pt,
                  const Arr3<T>& delp, Arr3<T>& zh, Arr3<T>& pe, Arr3<T>& ppe, Arr3<T>& pk3, Arr3<T>& pk, Arr3<T>& peln, const Arr2<T>& ws,
                  bool last_call, const Consts& c, const NhOpts& nh, const Bounds& bd) {
  const double gama = 1. / (1. - akap), rgrav = 1. / c.grav, peln1 = std::log(ptop), ptk = std::exp(akap * peln1);
  Col<T> dm(km + 2), dz2(km + 2), pm2(km + 2), w2(km + 2), pt2(km + 2), pem(km + 3), pe2(km + 3), peln2(km + 3);
  for (int j = bd.js; j <= bd.je; ++j)
    for (int i = bd.is; i <= bd.ie; ++i) {
      for (int k = 1; k <= km; ++k) dm[k] = delp(i, j, k);
      pem[1] = T(ptop); peln2[1] = T(peln1); pk3(i, j, 1) = T(ptk);
      for (int k = 2; k <= km + 1; ++k) { pem[k] = pem[k - 1] + dm[k - 1]; peln2[k] = log(pem[k]); pk3(i, j, k) = exp(akap * peln2[k]); }
      for (int k = 1; k <= km; ++k) {
        pm2[k] = dm[k] / (peln2[k + 1] - peln2[k]);
        dm[k] = dm[k] * rgrav;
        dz2[k] = zh(i, j, k + 1) - zh(i, j, k);
        w2[k] = w(i, j, k);
        pt2[k] = pt(i, j, k);
      }
      // nh_core_tlm.F90:155-189: a_imp > 0.999 -> SIM1_SOLVER, (0.5, 0.999] -> SIM_SOLVER (RIM_2D / SIM3 not restated)
      if (nh.a_imp > 0.999) sim1_solver(dt, km, c.rdgas, gama, akap, pe2, dm, pm2, pem, w2, dz2, pt2, ws(i, j), nh.p_fac);
      else sim_solver(dt, km, c.rdgas, gama, akap, pe2, dm, pm2, pem, w2, dz2, pt2, ws(i, j), nh.a_imp, nh.p_fac, nh.scale_z);
      for (int k = 1; k <= km; ++k) { w(i, j, k) = w2[k]; delz(i, j, k) = dz2[k]; }
      if (last_call) for (int k = 1; k <= km + 1; ++k) { peln(i, j, k) = peln2[k]; pk(i, j, k) = pk3(i, j, k); pe(i, j, k) = pem[k]; }
      for (int k = 1; k <= km + 1; ++k) ppe(i, j, k) = pe2[k];
      zh(i, j, km + 1) = T(zs(i, j));
      for (int k = km; k >= 1; --k) zh(i, j, k) = zh(i, j, k + 1) - dz2[k];
    }
}

// EDGE_PROFILE (non-uniform grid, limiter = 0), nh_utils_tlm.F90:3473-3584: layer means -> interface values,
// for one column of two fields.  q1, q2: km values; out km+1 values.
template <class T>
void edge_profile(const Col<T>& q1, const Col<T>& q2, Col<T>& qe1, Col<T>& qe2, int km, const std::vector<double>& dp0) {
  std::vector<double> gam(km + 2);
  const double g0 = dp0[2] / dp0[1];
  double xt1 = 2. * g0 * (g0 + 1.), bet = g0 * (g0 + 0.5);
  qe1[1] = (xt1 * q1[1] + q1[2]) / bet; qe2[1] = (xt1 * q2[1] + q2[2]) / bet;
  gam[1] = (1. + g0 * (g0 + 1.5)) / bet;
  double gk = g0;
  for (int k = 2; k <= km; ++k) {
    gk = dp0[k - 1] / dp0[k];
    bet = 2. + 2. * gk - gam[k - 1];
    qe1[k] = (3. * (q1[k - 1] + gk * q1[k]) - qe1[k - 1]) / bet;
    qe2[k] = (3. * (q2[k - 1] + gk * q2[k]) - qe2[k - 1]) / bet;
    gam[k] = gk / bet;
  }
  const double a_bot = 1. + gk * (gk + 1.5);
  xt1 = 2. * gk * (gk + 1.);
  const double xt2 = gk * (gk + 0.5) - a_bot * gam[km];
  qe1[km + 1] = (xt1 * q1[km] + q1[km - 1] - a_bot * qe1[km]) / xt2;
  qe2[km + 1] = (xt1 * q2[km] + q2[km - 1] - a_bot * qe2[km]) / xt2;
  for (int k = km; k >= 1; --k) { qe1[k] = qe1[k] - gam[k] * qe1[k + 1]; qe2[k] = qe2[k] - gam[k] * qe2[k + 1]; }
}

// UPDATE_DZ_C, nh_utils_tlm.F90:235-380: advect the interface heights (gz, km+1 levels, in/out on is-1..ie+1) with the
// C-grid area fluxes ut, vt interpolated to the interfaces; ws = surface vertical velocity of the terrain-following bottom.
template <class T>
void update_dz_c(int km, double dt, const std::vector<double>& dp0, const Arr2<double>& zs, const Arr3<T>& ut, const Arr3<T>& vt, Arr3<T>& gz,
                 Arr2<T>& ws, const Grid& g, const Bounds& bd) {
  const int is1 = bd.is - 1, js1 = bd.js - 1, ie1 = bd.ie + 1, je1 = bd.je + 1, ie2 = bd.ie + 2, je2 = bd.je + 2;
  const double rdt = 1. / dt, top_ratio = dp0[1] / (dp0[1] + dp0[2]), bot_ratio = dp0[km] / (dp0[km - 1] + dp0[km]);
  Arr2<T> xfx(bd), yfx(bd), fx(bd), fy(bd);
  for (int k = 1; k <= km + 1; ++k) {
    for (int j = js1; j <= je2; ++j)
      for (int i = is1; i <= ie2; ++i) {
        if (k == 1) {
          if (j <= je1) xfx(i, j) = ut(i, j, 1) + (ut(i, j, 1) - ut(i, j, 2)) * top_ratio;
          if (i <= ie1) yfx(i, j) = vt(i, j, 1) + (vt(i, j, 1) - vt(i, j, 2)) * top_ratio;
        } else if (k == km + 1) {
          if (j <= je1) xfx(i, j) = ut(i, j, km) + (ut(i, j, km) - ut(i, j, km - 1)) * bot_ratio;
          if (i <= ie1) yfx(i, j) = vt(i, j, km) + (vt(i, j, km) - vt(i, j, km - 1)) * bot_ratio;
        } else {
          const double int_ratio = 1. / (dp0[k - 1] + dp0[k]);
          if (j <= je1) xfx(i, j) = (dp0[k] * ut(i, j, k - 1) + dp0[k - 1] * ut(i, j, k)) * int_ratio;
          if (i <= ie1) yfx(i, j) = (dp0[k] * vt(i, j, k - 1) + dp0[k - 1] * vt(i, j, k)) * int_ratio;
        }
      }
    Arr2<T> gz2 = gz.plane(k);
    if (bd.any_edge()) fill_4corners(gz2, 1, bd);
    for (int j = js1; j <= je1; ++j)
      for (int i = is1; i <= ie2; ++i) fx(i, j) = xfx(i, j) * ((val(xfx(i, j)) > 0.) ? gz2(i - 1, j) : gz2(i, j));
    if (bd.any_edge()) fill_4corners(gz2, 2, bd);
    for (int j = js1; j <= je2; ++j)
      for (int i = is1; i <= ie1; ++i) fy(i, j) = yfx(i, j) * ((val(yfx(i, j)) > 0.) ? gz2(i, j - 1) : gz2(i, j));
    for (int j = js1; j <= je1; ++j)
      for (int i = is1; i <= ie1; ++i)
        gz(i, j, k) = (gz2(i, j) * g.area(i, j) + (fx(i, j) - fx(i + 1, j)) + (fy(i, j) - fy(i, j + 1))) /
                      (g.area(i, j) + (xfx(i, j) - xfx(i + 1, j)) + (yfx(i, j) - yfx(i, j + 1)));
  }
  for (int j = js1; j <= je1; ++j)
    for (int i = is1; i <= ie1; ++i) {
      ws(i, j) = (zs(i, j) - gz(i, j, km + 1)) * rdt;
      for (int k = km; k >= 1; --k)
        if (val(gz(i, j, k)) < val(gz(i, j, k + 1)) + nh_dz_min) gz(i, j, k) = gz(i, j, k + 1) + nh_dz_min;
    }
}

// UPDATE_DZ_D, nh_utils_tlm.F90:590-722: interface heights zh (km+1 levels) advected with fv_tp_2d using the D-grid
// Courant numbers / area fluxes interpolated to the interfaces (EDGE_PROFILE), plus del-2/4 damping with the vorticity
// damping coefficients of the layer above (ndif(km+1) = ndif(km)).
template <class T>
void update_dz_d(const std::vector<int>& ndif, const std::vector<double>& damp, Hord hord, int km, const std::vector<double>& dp0,
                 const Arr2<double>& zs, Arr3<T>& zh, const Arr3<T>& crx, const Arr3<T>& cry, const Arr3<T>& xfx, const Arr3<T>& yfx, Arr2<T>& ws,
                 double rdt, const Grid& g, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  Arr3<T> crx_adv(bd, km + 1), cry_adv(bd, km + 1), xfx_adv(bd, km + 1), yfx_adv(bd, km + 1);
  Col<T> a(km + 2), b(km + 2), ae(km + 3), be(km + 3);
  for (int j = jsd; j <= jed; ++j) {
    for (int i = is; i <= ie + 1; ++i) {
      for (int k = 1; k <= km; ++k) { a[k] = crx(i, j, k); b[k] = xfx(i, j, k); }
      edge_profile(a, b, ae, be, km, dp0);
      for (int k = 1; k <= km + 1; ++k) { crx_adv(i, j, k) = ae[k]; xfx_adv(i, j, k) = be[k]; }
    }
    if (j <= je + 1 && j >= js)
      for (int i = isd; i <= ied; ++i) {
        for (int k = 1; k <= km; ++k) { a[k] = cry(i, j, k); b[k] = yfx(i, j, k); }
        edge_profile(a, b, ae, be, km, dp0);
        for (int k = 1; k <= km + 1; ++k) { cry_adv(i, j, k) = ae[k]; yfx_adv(i, j, k) = be[k]; }
      }
  }
  Arr2<T> ra_x(bd), ra_y(bd), fx(bd), fy(bd), fx2(bd), fy2(bd), wk2(bd);
  for (int k = 1; k <= km + 1; ++k) {
    const int kk = (k <= km) ? k : km;
    for (int j = jsd; j <= jed; ++j)
      for (int i = is; i <= ie; ++i) ra_x(i, j) = g.area(i, j) + (xfx_adv(i, j, k) - xfx_adv(i + 1, j, k));
    for (int j = js; j <= je; ++j)
      for (int i = isd; i <= ied; ++i) ra_y(i, j) = g.area(i, j) + (yfx_adv(i, j, k) - yfx_adv(i, j + 1, k));
    Arr2<T> z2 = zh.plane(k);
    fv_tp_2d_split<T>(z2, crx_adv.plane(k), cry_adv.plane(k), hord.traj, hord.pert, fx, fy, xfx_adv.plane(k), yfx_adv.plane(k), g, bd, ra_x, ra_y, nullptr, nullptr,
                      nullptr, -1, 0.0, -1, 0.0);      // nh_utils_tlm.F90:496-560
    if (damp[kk] > 1.e-5) {
      del6_vt_flux(ndif[kk], damp[kk], z2, wk2, fx2, fy2, g, bd);
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i)
          zh(i, j, k) = (z2(i, j) * g.area(i, j) + (fx(i, j) - fx(i + 1, j)) + (fy(i, j) - fy(i, j + 1))) / (ra_x(i, j) + ra_y(i, j) - g.area(i, j)) +
                        (fx2(i, j) - fx2(i + 1, j) + (fy2(i, j) - fy2(i, j + 1))) * g.rarea(i, j);
    } else {
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i)
          zh(i, j, k) = (z2(i, j) * g.area(i, j) + (fx(i, j) - fx(i + 1, j)) + (fy(i, j) - fy(i, j + 1))) / (ra_x(i, j) + ra_y(i, j) - g.area(i, j));
    }
  }
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie; ++i) {
      ws(i, j) = (zs(i, j) - zh(i, j, km + 1)) * rdt;
      for (int k = km; k >= 1; --k)
        if (val(zh(i, j, k)) < val(zh(i, j, k + 1)) + nh_dz_min) zh(i, j, k) = zh(i, j, k + 1) + nh_dz_min;
    }
}

// PK3_HALO, dyn_core_tlm.F90:2716-2778: pk3 = p**kappa on the two halo rings from delp.
template <class T>
void pk3_halo(int npz, double ptop, double akap, Arr3<T>& pk3, const Arr3<T>& delp, const Bounds& bd) {
  auto col = [&](int i, int j) {
    T pei = T(ptop);
    for (int k = 1; k <= npz; ++k) { pei = pei + delp(i, j, k); pk3(i, j, k + 1) = exp(akap * log(pei)); }
  };
  for (int j = bd.js; j <= bd.je; ++j) for (int i : {bd.is - 2, bd.is - 1, bd.ie + 1, bd.ie + 2}) col(i, j);
  for (int i = bd.is - 2; i <= bd.ie + 2; ++i) for (int j : {bd.js - 2, bd.js - 1, bd.je + 1, bd.je + 2}) col(i, j);
}
// PE_HALO, dyn_core_tlm.F90:2934-2961
template <class T>
void pe_halo(int npz, double ptop, Arr3<T>& pe, const Arr3<T>& delp, const Bounds& bd) {
  auto col = [&](int i, int j) {
    pe(i, j, 1) = T(ptop);
    for (int k = 1; k <= npz; ++k) pe(i, j, k + 1) = pe(i, j, k) + delp(i, j, k);
  };
  for (int j = bd.js; j <= bd.je; ++j) { col(bd.is - 1, j); col(bd.ie + 1, j); }
  for (int i = bd.is - 1; i <= bd.ie + 1; ++i) { col(i, bd.js - 1); col(i, bd.je + 1); }
}

// P_GRAD_C, non-hydrostatic branch (wk = delpc), dyn_core_tlm.F90:3279-3336
template <class T>
void p_grad_c_nh(double dt2, int npz, const Arr3<T>& delpc, const Arr3<T>& pkc, const Arr3<T>& gz, Arr3<T>& uc, Arr3<T>& vc, const Grid& g,
                 const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  for (int k = 1; k <= npz; ++k) {
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i)
        uc(i, j, k) = uc(i, j, k) + dt2 * g.rdxc(i, j) / (delpc(i - 1, j, k) + delpc(i, j, k)) *
                      ((gz(i - 1, j, k + 1) - gz(i, j, k)) * (pkc(i, j, k + 1) - pkc(i - 1, j, k)) +
                       (gz(i - 1, j, k) - gz(i, j, k + 1)) * (pkc(i - 1, j, k + 1) - pkc(i, j, k)));
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i)
        vc(i, j, k) = vc(i, j, k) + dt2 * g.rdyc(i, j) / (delpc(i, j - 1, k) + delpc(i, j, k)) *
                      ((gz(i, j - 1, k + 1) - gz(i, j, k)) * (pkc(i, j, k + 1) - pkc(i, j - 1, k)) +
                       (gz(i, j - 1, k) - gz(i, j, k + 1)) * (pkc(i, j - 1, k + 1) - pkc(i, j, k)));
  }
}

// NH_P_GRAD (use_logp = .false.), dyn_core_tlm.F90:3491-3588: pp, pk, gz are replaced by their corner values.
template <class T>
void nh_p_grad(Arr3<T>& u, Arr3<T>& v, Arr3<T>& pp, Arr3<T>& gz, const Arr3<T>& delp, Arr3<T>& pk, double dt, double ptop, double akap, int npz,
               const Grid& g, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  const double ptk = std::pow(ptop, akap);
  Arr2<T> wk1(bd), wk(bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) { pp(i, j, 1) = T(0.); pk(i, j, 1) = T(ptk); }
  for (int k = 1; k <= npz + 1; ++k) {
    if (k != 1) { a2b_ord4(pp.plane(k), wk1, g, bd, true); a2b_ord4(pk.plane(k), wk1, g, bd, true); }
    a2b_ord4(gz.plane(k), wk1, g, bd, true);
  }
  for (int k = 1; k <= npz; ++k) {
    Arr2<T> dp = delp.plane(k);
    a2b_ord4(dp, wk1, g, bd, false);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) wk(i, j) = pk(i, j, k + 1) - pk(i, j, k);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i) {
        T du1 = dt / (wk(i, j) + wk(i + 1, j)) * ((gz(i, j, k + 1) - gz(i + 1, j, k)) * (pk(i + 1, j, k + 1) - pk(i, j, k)) +
                                                  (gz(i, j, k) - gz(i + 1, j, k + 1)) * (pk(i, j, k + 1) - pk(i + 1, j, k)));
        u(i, j, k) = (u(i, j, k) + du1 + dt / (wk1(i, j) + wk1(i + 1, j)) * ((gz(i, j, k + 1) - gz(i + 1, j, k)) * (pp(i + 1, j, k + 1) - pp(i, j, k)) +
                                                                             (gz(i, j, k) - gz(i + 1, j, k + 1)) * (pp(i, j, k + 1) - pp(i + 1, j, k)))) * g.rdx(i, j);
      }
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        T dv1 = dt / (wk(i, j) + wk(i, j + 1)) * ((gz(i, j, k + 1) - gz(i, j + 1, k)) * (pk(i, j + 1, k + 1) - pk(i, j, k)) +
                                                  (gz(i, j, k) - gz(i, j + 1, k + 1)) * (pk(i, j, k + 1) - pk(i, j + 1, k)));
        v(i, j, k) = (v(i, j, k) + dv1 + dt / (wk1(i, j) + wk1(i, j + 1)) * ((gz(i, j, k + 1) - gz(i, j + 1, k)) * (pp(i, j + 1, k + 1) - pp(i, j, k)) +
                                                                             (gz(i, j, k) - gz(i, j + 1, k + 1)) * (pp(i, j, k + 1) - pp(i, j + 1, k)))) * g.rdy(i, j);
      }
  }
}

// State of the non-hydrostatic acoustic loop = DynState + w, delz (+ zh carried between steps).
template <class T>
struct NhState { Arr3<T> w, delz, zh; };

// DYN_CORE, non-hydrostatic flow (dyn_core_tlm.F90:1701-2466), single tile with the periodic wrap standing in for
// mpp_update_domains.  ak, bk: hybrid coefficients (dp_ref, :1704-1706).
template <class T>
void dyn_core_nh(DynState<T>& s, NhState<T>& n, const Arr2<double>& phis, int npz, double bdt, int n_split, const DampOpts& o, const Consts& c,
                 double ptop, const std::vector<double>& ak, const std::vector<double>& bk, const NhOpts& nh, const Grid& g, const Bounds& bd,
                 Arr2<T>* ws_out = nullptr) {
  const double dt = bdt / double(n_split), dt2 = 0.5 * dt, rdt = 1. / dt, rgrav = 1. / c.grav;
  std::vector<double> dp_ref(npz + 2);
  for (int k = 1; k <= npz; ++k) dp_ref[k] = ak[k] - ak[k - 1] + (bk[k] - bk[k - 1]) * 1.e5;
  Arr2<double> zs(bd);
  for (int j = bd.jsd; j <= bd.jed; ++j) for (int i = bd.isd; i <= bd.ied; ++i) zs(i, j) = phis(i, j) * rgrav;
  Arr3<T> gz(bd, npz + 1), pkc(bd, npz + 1), pk3(bd, npz + 1), ptc(bd, npz), delpc(bd, npz), uc(bd, npz), vc(bd, npz), ua(bd, npz), va(bd, npz),
      ut(bd, npz), vt(bd, npz), divgd(bd, npz), crx(bd, npz), cry(bd, npz), xfx(bd, npz), yfx(bd, npz), wc(bd, npz);
  Arr2<T> ws3(bd), ws(bd);
  for (int k = 1; k <= npz; ++k) { s.mfx.plane(k).fill(T(0.)); s.mfy.plane(k).fill(T(0.)); s.cx.plane(k).fill(T(0.)); s.cy.plane(k).fill(T(0.)); }
  std::vector<int> ndif(npz + 2); std::vector<double> dampv(npz + 2);
  for (int k = 1; k <= npz; ++k) { LevelParams lp; level_params(o, k, npz, lp); ndif[k] = lp.nord_v; dampv[k] = lp.damp_vt; }
  for (int it = 1; it <= n_split; ++it) {
    const bool remap_step = (it == n_split);
    halo_periodic(n.w, bd);
    if (it == 1) {
      for (int j = bd.js; j <= bd.je; ++j)
        for (int i = bd.is; i <= bd.ie; ++i) {
          gz(i, j, npz + 1) = T(zs(i, j));
          for (int k = npz; k >= 1; --k) gz(i, j, k) = gz(i, j, k + 1) - n.delz(i, j, k);
        }
      halo_periodic(gz, bd);
    }
    for (int k = 1; k <= npz; ++k)
      c_sw(delpc.plane(k), s.delp.plane(k), ptc.plane(k), s.pt.plane(k), s.u.plane(k), s.v.plane(k), uc.plane(k), vc.plane(k), ua.plane(k),
           va.plane(k), ut.plane(k), vt.plane(k), divgd.plane(k), o.nord, dt2, g, bd, &n.w.plane(k), &wc.plane(k));
    if (o.nord > 0) halo_periodic(divgd, bd);
    if (it == 1) n.zh = gz; else gz = n.zh;
    update_dz_c(npz, dt2, dp_ref, zs, ut, vt, gz, ws3, g, bd);
    riem_solver_c(dt2, npz, c.akap, ptop, phis, wc, ptc, delpc, gz, pkc, ws3, c, nh, bd);
    p_grad_c_nh(dt2, npz, delpc, pkc, gz, uc, vc, g, bd);
    halo_periodic(uc, bd); halo_periodic(vc, bd);
    for (int k = 1; k <= npz; ++k) {
      LevelParams lp;
      if (!level_params(o, k, npz, lp)) { std::fprintf(stderr, "oracle: split hord at level %d not restated\n", k); std::abort(); }
      d_sw(s.delp.plane(k), s.pt.plane(k), s.u.plane(k), s.v.plane(k), uc.plane(k), vc.plane(k), ua.plane(k), va.plane(k), divgd.plane(k),
           s.mfx.plane(k), s.mfy.plane(k), s.cx.plane(k), s.cy.plane(k), crx.plane(k), cry.plane(k), xfx.plane(k), yfx.plane(k), dt, lp, o.dddmp,
           o.d4_bg, g, bd, &n.w.plane(k));
    }
    halo_periodic(s.delp, bd); halo_periodic(s.pt, bd);
    update_dz_d(ndif, dampv, Hord(o.hord_tm, o.hord_tm_pert), npz, dp_ref, zs, n.zh, crx, cry, xfx, yfx, ws, rdt, g, bd);
    riem_solver3(dt, npz, c.akap, ptop, zs, n.w, n.delz, s.pt, s.delp, n.zh, s.pe, pkc, pk3, s.pk, s.peln, ws, remap_step, c, nh, bd);
    halo_periodic(n.zh, bd); halo_periodic(pkc, bd);
    if (remap_step) pe_halo(npz, ptop, s.pe, s.delp, bd);
    pk3_halo(npz, ptop, c.akap, pk3, s.delp, bd);
    for (int k = 1; k <= npz + 1; ++k)
      for (int j = bd.js - 2; j <= bd.je + 2; ++j)
        for (int i = bd.is - 2; i <= bd.ie + 2; ++i) gz(i, j, k) = n.zh(i, j, k) * c.grav;
    nh_p_grad(s.u, s.v, pkc, gz, s.delp, pk3, dt, ptop, c.akap, npz, g, bd);
    halo_periodic(s.u, bd); halo_periodic(s.v, bd);
  }
  if (ws_out) *ws_out = ws;      // surface vertical velocity of the last acoustic step: lower boundary of the w remap
}

// Lagrangian_to_Eulerian, non-hydrostatic, remap_t (fv_mapz_tlm.F90:1607-1613 density pt -> density T, :1635-1641 delz ->
// specific volume / g, :1772-1796 w with kord_wz and the surface value ws, delz, :1852-1857 pkz from the equation of state).
template <class T>
void lagrangian_to_eulerian_nh(bool last_step, DynState<T>& s, NhState<T>& n, const Arr2<T>& ws, int km, const Consts& c, double ptop,
                               const std::vector<double>& ak, const std::vector<double>& bk, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  const int nq = (int)s.q.size();
  const double akap = c.akap, rrg = -c.rdgas / c.grav, k1k = c.rdgas / (c.cp_air - c.rdgas);
  Arr3<T> pe2s(bd, km + 1);
  std::vector<T> pe1(km + 2), pe2(km + 2), pn1(km + 2), pn2(km + 2), pk2(km + 2), q1(km + 1), q2(km + 1);
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie; ++i) {
      for (int k = 1; k <= km + 1; ++k) { pe1[k] = s.pe(i, j, k); pn1[k] = s.peln(i, j, k); }
      pe2[1] = T(ptop); pe2[km + 1] = s.pe(i, j, km + 1);
      for (int k = 1; k <= km; ++k) {
        s.pt(i, j, k) = s.pt(i, j, k) * exp(k1k * log(rrg * s.delp(i, j, k) / n.delz(i, j, k) * s.pt(i, j, k)));
        n.delz(i, j, k) = -(n.delz(i, j, k) / s.delp(i, j, k));
      }
      for (int k = 2; k <= km; ++k) pe2[k] = ak[k - 1] + bk[k - 1] * s.pe(i, j, km + 1);
      for (int k = 1; k <= km; ++k) s.delp(i, j, k) = pe2[k + 1] - pe2[k];
      pn2[1] = pn1[1]; pn2[km + 1] = pn1[km + 1]; pk2[1] = s.pk(i, j, 1); pk2[km + 1] = s.pk(i, j, km + 1);
      for (int k = 2; k <= km; ++k) { pn2[k] = log(pe2[k]); pk2[k] = exp(akap * pn2[k]); }
      for (int k = 1; k <= km; ++k) q1[k] = s.pt(i, j, k);
      map_col_split(km, pn1, q1, km, pn2, q2, 1, T(0.), Kord(remap_opts().kord_tm, remap_opts().kord_tm_pert, true, 184.));
      for (int k = 1; k <= km; ++k) s.pt(i, j, k) = q2[k];
      for (int iq = 0; iq < nq; ++iq) {
        for (int k = 1; k <= km; ++k) q1[k] = s.q[iq](i, j, k);
        map_col_split(km, pe1, q1, km, pe2, q2, 0, T(0.), Kord(remap_opts().kord_tr, remap_opts().kord_tr_pert, true, 0.));
        for (int k = 1; k <= km; ++k) s.q[iq](i, j, k) = q2[k];
      }
      for (int k = 1; k <= km; ++k) q1[k] = n.w(i, j, k);
      map_col_split(km, pe1, q1, km, pe2, q2, -2, ws(i, j), Kord(remap_opts().kord_wz, remap_opts().kord_wz_pert));
      for (int k = 1; k <= km; ++k) n.w(i, j, k) = q2[k];
      for (int k = 1; k <= km; ++k) q1[k] = n.delz(i, j, k);
      map_col_split(km, pe1, q1, km, pe2, q2, 1, T(0.), Kord(remap_opts().kord_tm, remap_opts().kord_tm_pert));      // delz: map1_ppm, iv = 1, abs(kord_tm) (fv_mapz_tlm.F90:627-637)
      for (int k = 1; k <= km; ++k) n.delz(i, j, k) = -(q2[k] * s.delp(i, j, k));
      for (int k = 1; k <= km + 1; ++k) { s.pk(i, j, k) = pk2[k]; s.peln(i, j, k) = pn2[k]; pe2s(i, j, k) = pe2[k]; }
      for (int k = 1; k <= km; ++k) s.pkz(i, j, k) = exp(akap * log(rrg * s.delp(i, j, k) / n.delz(i, j, k) * s.pt(i, j, k)));
    }
  l2e_winds(s, km, ak, bk, bd);
  for (int k = 2; k <= km; ++k)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) s.pe(i, j, k) = pe2s(i, j, k);
  for (int k = 1; k <= km; ++k)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) {
        if (last_step) s.pt(i, j, k) = s.pt(i, j, k) / (1. + c.zvir * (nq > 0 ? s.q[0](i, j, k) : T(0.)));
        else           s.pt(i, j, k) = s.pt(i, j, k) / s.pkz(i, j, k);
      }
}

// fv_dynamics, non-hydrostatic, moist_phys = .true. (fv_dynamics_tlm.F90:443-466 pkz from the equation of state, :582-590 pt ->
// virtual potential temperature, :646-860 k_split loop).  On entry pt is temperature; w, delz given.
template <class T>
void fv_dynamics_nh(DynState<T>& s, NhState<T>& n, const Arr2<double>& phis, int npz, double bdt, int n_split, int k_split, const DampOpts& o,
                    const Consts& c, double ptop, const std::vector<double>& ak, const std::vector<double>& bk, const NhOpts& nh, const Grid& g,
                    const Bounds& bd) {
  const int nq = (int)s.q.size();
  const double rdg = -c.rdgas / c.grav;
  Arr3<T> dp1(bd, npz);
  Arr2<T> ws(bd);
  for (int k = 1; k <= npz; ++k)
    for (int j = bd.js; j <= bd.je; ++j)
      for (int i = bd.is; i <= bd.ie; ++i) {
        T d = (nq > 0) ? c.zvir * s.q[0](i, j, k) : T(0.);
        s.pkz(i, j, k) = exp(c.akap * log(rdg * s.delp(i, j, k) * s.pt(i, j, k) * (1. + d) / n.delz(i, j, k)));
        s.pt(i, j, k) = s.pt(i, j, k) * (1. + d) / s.pkz(i, j, k);
      }
  const double mdt = bdt / double(k_split);
  for (int n_map = 1; n_map <= k_split; ++n_map) {
    halo_periodic(s.delp, bd); halo_periodic(s.pt, bd); halo_periodic(s.u, bd); halo_periodic(s.v, bd);
    for (int k = 1; k <= npz; ++k) dp1.plane(k) = s.delp.plane(k);
    const bool last_step = (n_map == k_split);
    dyn_core_nh(s, n, phis, npz, mdt, n_split, o, c, ptop, ak, bk, nh, g, bd, &ws);
    if (nq > 0) {
      for (auto& qq : s.q) halo_periodic(qq, bd);
      tracer_2d(s.q, dp1, s.mfx, s.mfy, s.cx, s.cy, npz, Hord(o.hord_tr, o.hord_tr_pert), g, bd);
    }
    if (npz > 4) lagrangian_to_eulerian_nh(last_step, s, n, ws, npz, c, ptop, ak, bk, bd);
  }
}

// ---- six faces (cube.hpp): the same per-face routines between the halo exchanges of the non-hydrostatic flow
// (dyn_core_tlm.F90:1771-1773 w, :1800-1802 gz, :2210-2216 zh and pkc, :2432-2434 u, v).
template <class T>
void dyn_core_nh_cube(std::vector<DynState<T>>& S, std::vector<NhState<T>>& N, const std::vector<Arr2<double>>& phis, int npz, double bdt,
                      int n_split, const DampOpts& o, const Consts& c, double ptop, const std::vector<double>& ak, const std::vector<double>& bk,
                      const NhOpts& nh, const std::vector<Grid>& G, const Bounds& bd, const CubeTables& X, std::vector<Arr2<T>>* ws_out = nullptr) {
  const int nt = (int)S.size();
  const double dt = bdt / double(n_split), dt2 = 0.5 * dt, rdt = 1. / dt, rgrav = 1. / c.grav;
  const std::vector<Arr3<T>*> none;
  std::vector<double> dp_ref(npz + 2);
  for (int k = 1; k <= npz; ++k) dp_ref[k] = ak[k] - ak[k - 1] + (bk[k] - bk[k - 1]) * 1.e5;
  std::vector<Arr2<double>> zs(nt, Arr2<double>(bd));
  for (int t = 0; t < nt; ++t)
    for (int j = bd.jsd; j <= bd.jed; ++j) for (int i = bd.isd; i <= bd.ied; ++i) zs[t](i, j) = phis[t](i, j) * rgrav;
  auto mk = [&](int nk) { return std::vector<Arr3<T>>(nt, Arr3<T>(bd, nk)); };
  auto gz = mk(npz + 1), pkc = mk(npz + 1), pk3 = mk(npz + 1), ptc = mk(npz), delpc = mk(npz), uc = mk(npz), vc = mk(npz), ua = mk(npz), va = mk(npz),
       ut = mk(npz), vt = mk(npz), divgd = mk(npz), crx = mk(npz), cry = mk(npz), xfx = mk(npz), yfx = mk(npz), wc = mk(npz);
  std::vector<Arr2<T>> ws3(nt, Arr2<T>(bd)), ws(nt, Arr2<T>(bd));
  auto fld = [&](Arr3<T> NhState<T>::*m) { std::vector<Arr3<T>*> v; for (auto& n : N) v.push_back(&(n.*m)); return v; };
  for (auto& s : S)
    for (int k = 1; k <= npz; ++k) { s.mfx.plane(k).fill(T(0.)); s.mfy.plane(k).fill(T(0.)); s.cx.plane(k).fill(T(0.)); s.cy.plane(k).fill(T(0.)); }
  std::vector<int> ndif(npz + 2); std::vector<double> dampv(npz + 2);
  for (int k = 1; k <= npz; ++k) { LevelParams lp; level_params(o, k, npz, lp); ndif[k] = lp.nord_v; dampv[k] = lp.damp_vt; }
  for (int it = 1; it <= n_split; ++it) {
    const bool remap_step = (it == n_split);
    exchange(X.rows[X_CELL], fld(&NhState<T>::w), none);
    if (it == 1) {
      for (int t = 0; t < nt; ++t)
        for (int j = bd.js; j <= bd.je; ++j)
          for (int i = bd.is; i <= bd.ie; ++i) {
            gz[t](i, j, npz + 1) = T(zs[t](i, j));
            for (int k = npz; k >= 1; --k) gz[t](i, j, k) = gz[t](i, j, k + 1) - N[t].delz(i, j, k);
          }
      exchange(X.rows[X_CELL], ptrs(gz), none);
    }
    for (int t = 0; t < nt; ++t)
      for (int k = 1; k <= npz; ++k)
        c_sw(delpc[t].plane(k), S[t].delp.plane(k), ptc[t].plane(k), S[t].pt.plane(k), S[t].u.plane(k), S[t].v.plane(k), uc[t].plane(k),
             vc[t].plane(k), ua[t].plane(k), va[t].plane(k), ut[t].plane(k), vt[t].plane(k), divgd[t].plane(k), o.nord, dt2, G[t], bd,
             &N[t].w.plane(k), &wc[t].plane(k));
    if (o.nord > 0) exchange(X.rows[X_CORNER], ptrs(divgd), none);
    for (int t = 0; t < nt; ++t) {
      if (it == 1) N[t].zh = gz[t]; else gz[t] = N[t].zh;
      update_dz_c(npz, dt2, dp_ref, zs[t], ut[t], vt[t], gz[t], ws3[t], G[t], bd);
      riem_solver_c(dt2, npz, c.akap, ptop, phis[t], wc[t], ptc[t], delpc[t], gz[t], pkc[t], ws3[t], c, nh, bd);
      p_grad_c_nh(dt2, npz, delpc[t], pkc[t], gz[t], uc[t], vc[t], G[t], bd);
    }
    exchange(X.rows[X_CVEC], ptrs(uc), ptrs(vc));
    for (int t = 0; t < nt; ++t)
      for (int k = 1; k <= npz; ++k) {
        LevelParams lp;
        if (!level_params(o, k, npz, lp)) { std::fprintf(stderr, "oracle: split hord at level %d not restated\n", k); std::abort(); }
        d_sw(S[t].delp.plane(k), S[t].pt.plane(k), S[t].u.plane(k), S[t].v.plane(k), uc[t].plane(k), vc[t].plane(k), ua[t].plane(k),
             va[t].plane(k), divgd[t].plane(k), S[t].mfx.plane(k), S[t].mfy.plane(k), S[t].cx.plane(k), S[t].cy.plane(k), crx[t].plane(k),
             cry[t].plane(k), xfx[t].plane(k), yfx[t].plane(k), dt, lp, o.dddmp, o.d4_bg, G[t], bd, &N[t].w.plane(k));
      }
    exchange(X.rows[X_CELL], per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.delp; }), none);
    exchange(X.rows[X_CELL], per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.pt; }), none);
    for (int t = 0; t < nt; ++t) {
      update_dz_d(ndif, dampv, Hord(o.hord_tm, o.hord_tm_pert), npz, dp_ref, zs[t], N[t].zh, crx[t], cry[t], xfx[t], yfx[t], ws[t], rdt, G[t], bd);
      riem_solver3(dt, npz, c.akap, ptop, zs[t], N[t].w, N[t].delz, S[t].pt, S[t].delp, N[t].zh, S[t].pe, pkc[t], pk3[t], S[t].pk, S[t].peln, ws[t],
                   remap_step, c, nh, bd);
    }
    exchange(X.rows[X_CELL], fld(&NhState<T>::zh), none);
    exchange(X.rows[X_CELL], ptrs(pkc), none);
    for (int t = 0; t < nt; ++t) {
      if (remap_step) pe_halo(npz, ptop, S[t].pe, S[t].delp, bd);
      pk3_halo(npz, ptop, c.akap, pk3[t], S[t].delp, bd);
      for (int k = 1; k <= npz + 1; ++k)
        for (int j = bd.js - 2; j <= bd.je + 2; ++j)
          for (int i = bd.is - 2; i <= bd.ie + 2; ++i) gz[t](i, j, k) = N[t].zh(i, j, k) * c.grav;
      nh_p_grad(S[t].u, S[t].v, pkc[t], gz[t], S[t].delp, pk3[t], dt, ptop, c.akap, npz, G[t], bd);
    }
    auto us = per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.u; }), vs = per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.v; });
    if (it == n_split) exchange(X.rows[X_DEDGE], us, vs); else exchange(X.rows[X_DVEC], us, vs);
  }
  if (ws_out) *ws_out = ws;
}

template <class T>
void fv_dynamics_nh_cube(std::vector<DynState<T>>& S, std::vector<NhState<T>>& N, const std::vector<Arr2<double>>& phis, int npz, double bdt,
                         int n_split, int k_split, const DampOpts& o, const Consts& c, double ptop, const std::vector<double>& ak,
                         const std::vector<double>& bk, const NhOpts& nh, const std::vector<Grid>& G, const Bounds& bd, const CubeTables& X) {
  const int nt = (int)S.size(), nq = (int)S[0].q.size();
  const std::vector<Arr3<T>*> none;
  const double rdg = -c.rdgas / c.grav;
  std::vector<Arr3<T>> dp1(nt, Arr3<T>(bd, npz));
  std::vector<Arr2<T>> ws;
  for (int t = 0; t < nt; ++t)
    for (int k = 1; k <= npz; ++k)
      for (int j = bd.js; j <= bd.je; ++j)
        for (int i = bd.is; i <= bd.ie; ++i) {
          DynState<T>& s = S[t];
          T d = (nq > 0) ? c.zvir * s.q[0](i, j, k) : T(0.);
          s.pkz(i, j, k) = exp(c.akap * log(rdg * s.delp(i, j, k) * s.pt(i, j, k) * (1. + d) / N[t].delz(i, j, k)));
          s.pt(i, j, k) = s.pt(i, j, k) * (1. + d) / s.pkz(i, j, k);
        }
  const double mdt = bdt / double(k_split);
  for (int n_map = 1; n_map <= k_split; ++n_map) {
    exchange(X.rows[X_DVEC], per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.u; }), per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.v; }));
    exchange(X.rows[X_CELL], per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.delp; }), none);
    exchange(X.rows[X_CELL], per_tile(S, [](DynState<T>& s) -> Arr3<T>& { return s.pt; }), none);
    for (int t = 0; t < nt; ++t) for (int k = 1; k <= npz; ++k) dp1[t].plane(k) = S[t].delp.plane(k);
    dyn_core_nh_cube(S, N, phis, npz, mdt, n_split, o, c, ptop, ak, bk, nh, G, bd, X, &ws);
    if (nq > 0) {
      for (int n = 0; n < nq; ++n) {
        std::vector<Arr3<T>*> qs; for (auto& s : S) qs.push_back(&s.q[n]);
        exchange(X.rows[X_CELL], qs, none);
      }
      tracer_2d_cube(S, dp1, npz, Hord(o.hord_tr, o.hord_tr_pert), G, bd, X);
    }
    if (npz > 4) for (int t = 0; t < nt; ++t) lagrangian_to_eulerian_nh(n_map == k_split, S[t], N[t], ws[t], npz, c, ptop, ak, bk, bd);
  }
}

}  // namespace orc
