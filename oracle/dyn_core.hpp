// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Restatement of model_tlmadm/dyn_core_tlm.F90: the acoustic loop DYN_CORE (:1478-2610 /
// _TLM :93-1474), geopk (:4763-4905 / _TLM :4578-4762), p_grad_c (:3279-3336 / _TLM :3194),
// one_grad_p (:4035-4159 / _TLM :3867-4034), and the per-level damping/scheme selection
// (:741-921).  Hydrostatic, beta=0, d_ext=0, d_con=0, inline_q=.false., fill_dp=.false..
// Halo exchange is FMS-internal in the reference (mpp_update_domains); for the single-tile case
// the oracle wraps the tile doubly-periodically — that data motion is
// "parity unpinned" by construction.
#pragma once
#include "sw_core.hpp"

namespace orc {

// Namelist-level options feeding the per-level selection (fv_flags_type + fv_flags_pert_type).
struct DampOpts {
  // trajectory (after run_setup_pert has copied the _pert values in when split_* = .false.)
  int hord_mt = 2, hord_vt = 2, hord_tm = 2, hord_dp = 2, hord_tr = 2;
  int nord = 1; double dddmp = 0.2, d2_bg = 0.015, d4_bg = 0.15; bool do_vort_damp = true; double vtdm4 = 0.0005;
  double d2_bg_k1 = 4., d2_bg_k2 = 2., d_con = 0., ke_bg = 0.; int n_sponge = 1;
  // perturbation
  int hord_mt_pert = 2, hord_vt_pert = 2, hord_tm_pert = 2, hord_dp_pert = 2, hord_tr_pert = 2;
  int nord_pert = 1; double dddmp_pert = 0.2, d2_bg_pert = 0.015, d4_bg_pert = 0.15; bool do_vort_damp_pert = true;
  double vtdm4_pert = 0.0005, d2_bg_k1_pert = 4., d2_bg_k2_pert = 2., d2_bg_ks_pert = 2.; int n_sponge_pert = 9;
  bool hord_ks_traj = true, hord_ks_pert = true;
  bool split_damp = false;       // fv_flags_pert_type%split_damp (the reference's default is .true.)
  int hord_mt_ks_traj = 1, hord_vt_ks_traj = 1, hord_tm_ks_traj = 1, hord_dp_ks_traj = 1, hord_tr_ks_traj = 1;
  int hord_mt_ks_pert = 1, hord_vt_ks_pert = 1, hord_tm_ks_pert = 1, hord_dp_ks_pert = 1, hord_tr_ks_pert = 1;
};

// dyn_core_tlm.F90:741-921.  Returns false if at level k the trajectory and perturbation advection
// orders differ (the "split" recompute path, not restated).
inline bool level_params(const DampOpts& o, int k, int npz, LevelParams& lp) {
  int hord_m = o.hord_mt, hord_t = o.hord_tm, hord_v = o.hord_vt, hord_p = o.hord_dp;
  int nord_k = o.nord;
  int nord_v = (2 > o.nord) ? o.nord : 2;
  double d2_divg = (0.20 > o.d2_bg) ? o.d2_bg : 0.20;
  double damp_vt = o.do_vort_damp ? o.vtdm4 : 0.;
  int nord_w = nord_v, nord_t = nord_v;
  double damp_w = damp_vt, damp_t = damp_vt;
  double d_con_k = o.d_con;
  if (npz == 1 || o.n_sponge < 0) {
    d2_divg = o.d2_bg;
  } else if (k == 1) {
    nord_k = 0;
    if (0.01 < o.d2_bg) d2_divg = (o.d2_bg < o.d2_bg_k1) ? o.d2_bg_k1 : o.d2_bg;
    else if (0.01 < o.d2_bg_k1) d2_divg = o.d2_bg_k1;
    else d2_divg = 0.01;
    nord_w = 0; damp_w = d2_divg;
    if (o.do_vort_damp) { nord_v = 0; damp_vt = 0.5 * d2_divg; }
    d_con_k = 0.;
  } else {
    int max1 = (2 < o.n_sponge - 1) ? o.n_sponge - 1 : 2;
    if (k == max1 && o.d2_bg_k2 > 0.01) {
      nord_k = 0;
      d2_divg = (o.d2_bg < o.d2_bg_k2) ? o.d2_bg_k2 : o.d2_bg;
      nord_w = 0; damp_w = d2_divg;
      if (o.do_vort_damp) { nord_v = 0; damp_vt = 0.5 * d2_divg; }
      d_con_k = 0.;
    } else {
      int max2 = (3 < o.n_sponge) ? o.n_sponge : 3;
      if (k == max2 && o.d2_bg_k2 > 0.05) {
        nord_k = 0;
        d2_divg = (o.d2_bg < 0.2 * o.d2_bg_k2) ? 0.2 * o.d2_bg_k2 : o.d2_bg;
        nord_w = 0; damp_w = d2_divg;
        d_con_k = 0.;
      }
    }
  }
  int hord_m_pert = o.hord_mt_pert, hord_t_pert = o.hord_tm_pert, hord_v_pert = o.hord_vt_pert, hord_p_pert = o.hord_dp_pert;
  int nord_k_pert = o.nord_pert;
  int nord_v_pert = (2 > o.nord_pert) ? o.nord_pert : 2;
  double d2_divg_pert = (0.20 > o.d2_bg_pert) ? o.d2_bg_pert : 0.20;
  double damp_vt_pert = o.do_vort_damp_pert ? o.vtdm4_pert : 0.;
  const int nord_t_pert = nord_v_pert; const double damp_t_pert = damp_vt_pert;     // :856-859, before the sponge rules below
  if (k <= o.n_sponge_pert) {
    nord_k_pert = 0;
    if (k <= o.n_sponge_pert - 1) {
      if (o.hord_ks_traj) { hord_m = o.hord_mt_ks_traj; hord_t = o.hord_tm_ks_traj; hord_v = o.hord_vt_ks_traj; hord_p = o.hord_dp_ks_traj; }
      if (o.hord_ks_pert) { hord_m_pert = o.hord_mt_ks_pert; hord_t_pert = o.hord_tm_ks_pert; hord_v_pert = o.hord_vt_ks_pert; hord_p_pert = o.hord_dp_ks_pert; }
    }
    double kfac = (k == 1) ? o.d2_bg_k1_pert : (k == 2) ? o.d2_bg_k2_pert : o.d2_bg_ks_pert;
    if (0.01 < o.d2_bg_pert) d2_divg_pert = (o.d2_bg_pert < kfac) ? kfac : o.d2_bg_pert;
    else if (0.01 < kfac) d2_divg_pert = kfac;
    else d2_divg_pert = 0.01;
    if (o.do_vort_damp_pert) { nord_v_pert = 0; damp_vt_pert = 0.5 * d2_divg_pert; }
  }
  lp.hord_mt = hord_m; lp.hord_vt = hord_v; lp.hord_tm = hord_t; lp.hord_dp = hord_p; lp.hord_tr = o.hord_tr;
  lp.nord = nord_k; lp.nord_v = nord_v; lp.nord_w = nord_w; lp.nord_t = nord_t;
  lp.d2_divg = d2_divg; lp.damp_vt = damp_vt; lp.damp_w = damp_w; lp.damp_t = damp_t; lp.d_con = d_con_k;
  lp.nord_v_pert = nord_v_pert; lp.damp_vt_pert = damp_vt_pert;
  lp.hord_mt_pert = hord_m_pert; lp.hord_vt_pert = hord_v_pert; lp.hord_tm_pert = hord_t_pert; lp.hord_dp_pert = hord_p_pert;
  lp.split_damp = o.split_damp; lp.nord_pert = nord_k_pert; lp.d2_divg_pert = d2_divg_pert; lp.nord_t_pert = nord_t_pert; lp.damp_t_pert = damp_t_pert;
  lp.dddmp_pert = o.dddmp_pert; lp.d4_bg_pert = o.d4_bg_pert;
  return true;
}

// Doubly-periodic wrap of every point of the padded plane that is outside 1..nx × 1..ny.
template <class T>
void halo_periodic(Arr2<T>& q, const Bounds& bd) {
  auto wrap = [](int i, int n) { int r = (i - 1) % n; if (r < 0) r += n; return r + 1; };
  for (int j = bd.jsd; j <= bd.jed + 1; ++j)
    for (int i = bd.isd; i <= bd.ied + 1; ++i) {
      if (i >= 1 && i <= bd.nx && j >= 1 && j <= bd.ny) continue;
      q(i, j) = q(wrap(i, bd.nx), wrap(j, bd.ny));
    }
}
template <class T>
void halo_periodic(Arr3<T>& q, const Bounds& bd) { for (auto& p : q.p) halo_periodic(p, bd); }

// geopk, dyn_core_tlm.F90:4763-4905.  pe/peln use the reference's (i,k,j) meaning but are stored
// here as Arr3 (i,j,k).  cg: C-grid half step (no pkz).  Range is-1..ie+1 (C) or is-2..ie+2 (D, a2b_ord=4).
template <class T>
void geopk(double ptop, Arr3<T>& pe, Arr3<T>& peln, const Arr3<T>& delp, Arr3<T>& pk, Arr3<T>& gz,
           const Arr2<double>& hs, const Arr3<T>& pt, Arr3<T>& pkz, int km, double akap, double cp_air, bool cg,
           int a2b_ord, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  int ifirst, ilast, jfirst, jlast;
  if (!cg && a2b_ord == 4) { ifirst = is - 2; ilast = ie + 2; jfirst = js - 2; jlast = je + 2; }
  else { ifirst = is - 1; ilast = ie + 1; jfirst = js - 1; jlast = je + 1; }
  const double peln1 = std::log(ptop), ptk = std::pow(ptop, akap);   // dyn_core_tlm.F90:1672-1673
  for (int j = jfirst; j <= jlast; ++j) {
    for (int i = ifirst; i <= ilast; ++i) {
      T p1d = T(ptop);
      pk(i, j, 1) = T(ptk);
      gz(i, j, km + 1) = T(hs(i, j));
      bool in_pe = (j > js - 2 && j < je + 2) && (i >= is - 1 && i <= ie + 1);
      bool in_ln = (j >= js && j <= je) && (i >= is && i <= ie);
      if (in_ln) peln(i, j, 1) = T(peln1);
      if (in_pe) pe(i, j, 1) = T(ptop);
      for (int k = 2; k <= km + 1; ++k) {
        p1d = p1d + delp(i, j, k - 1);
        T logp = log(p1d);
        pk(i, j, k) = exp(akap * logp);
        if (in_pe) pe(i, j, k) = p1d;
        if (in_ln) peln(i, j, k) = logp;
      }
      T g1d = T(hs(i, j));
      for (int k = km; k >= 1; --k) {
        g1d = g1d + cp_air * pt(i, j, k) * (pk(i, j, k + 1) - pk(i, j, k));
        gz(i, j, k) = g1d;
      }
      if (!cg && in_ln)
        for (int k = 1; k <= km; ++k)
          pkz(i, j, k) = (pk(i, j, k + 1) - pk(i, j, k)) / (akap * (peln(i, j, k + 1) - peln(i, j, k)));
    }
  }
}

// p_grad_c (hydrostatic), dyn_core_tlm.F90:3279-3336.
template <class T>
void p_grad_c(double dt2, int npz, const Arr3<T>& pkc, const Arr3<T>& gz, Arr3<T>& uc, Arr3<T>& vc, const Grid& g,
              const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  Arr2<T> wk(bd);
  for (int k = 1; k <= npz; ++k) {
    for (int j = js - 1; j <= je + 1; ++j)
      for (int i = is - 1; i <= ie + 1; ++i) wk(i, j) = pkc(i, j, k + 1) - pkc(i, j, k);
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i)
        uc(i, j, k) = uc(i, j, k) + dt2 * g.rdxc(i, j) / (wk(i - 1, j) + wk(i, j)) *
                      ((gz(i - 1, j, k + 1) - gz(i, j, k)) * (pkc(i, j, k + 1) - pkc(i - 1, j, k)) +
                       (gz(i - 1, j, k) - gz(i, j, k + 1)) * (pkc(i - 1, j, k + 1) - pkc(i, j, k)));
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i)
        vc(i, j, k) = vc(i, j, k) + dt2 * g.rdyc(i, j) / (wk(i, j - 1) + wk(i, j)) *
                      ((gz(i, j - 1, k + 1) - gz(i, j, k)) * (pkc(i, j, k + 1) - pkc(i, j - 1, k)) +
                       (gz(i, j - 1, k) - gz(i, j, k + 1)) * (pkc(i, j - 1, k + 1) - pkc(i, j, k)));
  }
}

// one_grad_p (hydrostatic, a2b_ord=4, d_ext=0), dyn_core_tlm.F90:4035-4159.  pk and gz are replaced
// by their corner (B-grid) values on is..ie+1, js..je+1.
template <class T>
void one_grad_p(Arr3<T>& u, Arr3<T>& v, Arr3<T>& pk, Arr3<T>& gz, double dt, double ptop, double akap, int npz,
                const Grid& g, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  const double ptk = std::pow(ptop, akap);
  Arr2<T> wk(bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) pk(i, j, 1) = T(ptk);
  for (int k = 2; k <= npz + 1; ++k) a2b_ord4(pk.plane(k), wk, g, bd, true);
  for (int k = 1; k <= npz + 1; ++k) a2b_ord4(gz.plane(k), wk, g, bd, true);
  for (int k = 1; k <= npz; ++k) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) wk(i, j) = pk(i, j, k + 1) - pk(i, j, k);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie; ++i)
        u(i, j, k) = g.rdx(i, j) * (u(i, j, k) + dt / (wk(i, j) + wk(i + 1, j)) *
                     ((gz(i, j, k + 1) - gz(i + 1, j, k)) * (pk(i + 1, j, k + 1) - pk(i, j, k)) +
                      (gz(i, j, k) - gz(i + 1, j, k + 1)) * (pk(i, j, k + 1) - pk(i + 1, j, k))));
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie + 1; ++i)
        v(i, j, k) = g.rdy(i, j) * (v(i, j, k) + dt / (wk(i, j) + wk(i, j + 1)) *
                     ((gz(i, j, k + 1) - gz(i, j + 1, k)) * (pk(i, j + 1, k + 1) - pk(i, j, k)) +
                      (gz(i, j, k) - gz(i, j + 1, k + 1)) * (pk(i, j, k + 1) - pk(i, j + 1, k))));
  }
}

// State carried through dyn_core.
template <class T>
struct DynState {
  Arr3<T> u, v, pt, delp;                 // prognostic (halo'd)
  Arr3<T> pe, peln, pk, pkz;              // pressure diagnostics (outputs of the D-grid geopk)
  Arr3<T> mfx, mfy, cx, cy;               // flux capacitors for tracer_2d
  std::vector<Arr3<T>> q;                 // tracers
  void init(const Bounds& bd, int npz, int nq) {
    u.init(bd, npz); v.init(bd, npz); pt.init(bd, npz); delp.init(bd, npz);
    pe.init(bd, npz + 1); peln.init(bd, npz + 1); pk.init(bd, npz + 1); pkz.init(bd, npz);
    mfx.init(bd, npz); mfy.init(bd, npz); cx.init(bd, npz); cy.init(bd, npz);
    q.assign(nq, Arr3<T>(bd, npz));
  }
};

// DYN_CORE acoustic loop, hydrostatic; dyn_core_tlm.F90:1736-2466.  Halos of delp, pt, u, v must be
// valid on entry (fv_dynamics does it, fv_dynamics_tlm.F90:646-651).
template <class T>
void dyn_core(DynState<T>& s, const Arr2<double>& phis, int npz, double bdt, int n_split, const DampOpts& o,
              const Consts& c, double ptop, const Grid& g, const Bounds& bd) {
  const double dt = bdt / double(n_split), dt2 = 0.5 * dt;
  Arr3<T> gz(bd, npz + 1), pkc(bd, npz + 1), ptc(bd, npz), delpc(bd, npz), uc(bd, npz), vc(bd, npz), ua(bd, npz),
      va(bd, npz), ut(bd, npz), vt(bd, npz), divgd(bd, npz), crx(bd, npz), cry(bd, npz), xfx(bd, npz), yfx(bd, npz);
  for (int k = 1; k <= npz; ++k) { s.mfx.plane(k).fill(T(0.)); s.mfy.plane(k).fill(T(0.)); s.cx.plane(k).fill(T(0.)); s.cy.plane(k).fill(T(0.)); }
  for (int it = 1; it <= n_split; ++it) {
    for (int k = 1; k <= npz; ++k)
      c_sw(delpc.plane(k), s.delp.plane(k), ptc.plane(k), s.pt.plane(k), s.u.plane(k), s.v.plane(k), uc.plane(k),
           vc.plane(k), ua.plane(k), va.plane(k), ut.plane(k), vt.plane(k), divgd.plane(k), o.nord, dt2, g, bd);
    if (o.nord > 0) halo_periodic(divgd, bd);
    geopk(ptop, s.pe, s.peln, delpc, pkc, gz, phis, ptc, s.pkz, npz, c.akap, c.cp_air, true, 4, bd);
    p_grad_c(dt2, npz, pkc, gz, uc, vc, g, bd);
    halo_periodic(uc, bd); halo_periodic(vc, bd);
    for (int k = 1; k <= npz; ++k) {
      LevelParams lp;
      bool ok = level_params(o, k, npz, lp);
      if (!ok) { std::fprintf(stderr, "oracle: split hord at level %d not restated\n", k); std::abort(); }
      d_sw(s.delp.plane(k), s.pt.plane(k), s.u.plane(k), s.v.plane(k), uc.plane(k), vc.plane(k), ua.plane(k),
           va.plane(k), divgd.plane(k), s.mfx.plane(k), s.mfy.plane(k), s.cx.plane(k), s.cy.plane(k), crx.plane(k),
           cry.plane(k), xfx.plane(k), yfx.plane(k), dt, lp, o.dddmp, o.d4_bg, g, bd);
    }
    halo_periodic(s.delp, bd); halo_periodic(s.pt, bd);
    geopk(ptop, s.pe, s.peln, s.delp, pkc, gz, phis, s.pt, s.pkz, npz, c.akap, c.cp_air, false, 4, bd);
    if (it == n_split)                        // remap_step .and. hydrostatic (:2364-2372)
      for (int k = 1; k <= npz + 1; ++k)
        for (int j = bd.js; j <= bd.je; ++j)
          for (int i = bd.is; i <= bd.ie; ++i) s.pk(i, j, k) = pkc(i, j, k);
    one_grad_p(s.u, s.v, pkc, gz, dt, ptop, c.akap, npz, g, bd);
    // it==n_split: mpp_get_boundary (:2418-2431); otherwise halo update of u,v (:2432-2434).  The
    // periodic wrap refreshes both the shared edge row and the halo in either case.
    halo_periodic(s.u, bd); halo_periodic(s.v, bd);
  }
}

}  // namespace orc
