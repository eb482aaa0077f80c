// fv3lm-hip: fv_dynamics level — FV_DYNAMICS_TLM (fv_dynamics_tlm.F90:87-995) and
// FV_DYNAMICS_FWD/BWD (fv_dynamics_adm.F90:110/874): pt conversion, the k_split loop
// {halo, dyn_core, tracer_2d, Lagrangian-to-Eulerian remap}, and the step_tl / step_ad entry points
// with the pressure diagnostics of fv_pressure.F90 on the device.
#pragma once
#include "dycore.h"
#include "remap.h"

namespace fv3 {

// compute_fv3_pressures{,_tlm,_bwd} (model_tlmadm/fv_pressure.F90:23-203) on is..ie, js..je
struct PressArgs { Geom g; Fld delp, pe, peln, pk, pkz; double kappa, ptop; };
template <class T>
HD void press_col(const PressArgs& a, int i, int j, int tile) {
  const Geom& g = a.g; const int km = g.npz; typedef FIO<T> IO;
  T pe = T(a.ptop), ln0 = dlog(pe), pk0 = dexp(a.kappa * ln0);
  IO::st(a.pe, fidx(g, a.pe, tile, i, j, 1), pe); IO::st(a.peln, fidx(g, a.peln, tile, i, j, 1), ln0); IO::st(a.pk, fidx(g, a.pk, tile, i, j, 1), pk0);
  for (int k = 1; k <= km; ++k) {
    pe = pe + IO::ld(a.delp, fidx(g, a.delp, tile, i, j, k));
    T ln1 = dlog(pe), pk1 = dexp(a.kappa * ln1);
    const size_t m = fidx(g, a.pe, tile, i, j, k + 1);
    IO::st(a.pe, m, pe); IO::st(a.peln, m, ln1); IO::st(a.pk, m, pk1);
    IO::st(a.pkz, fidx(g, a.pkz, tile, i, j, k), (pk1 - pk0) / (a.kappa * (ln1 - ln0)));
    ln0 = ln1; pk0 = pk1;
  }
}
HD void press_col_ad(const PressArgs& a, int i, int j, int tile) {   // fv_pressure.F90:137-203
  const Geom& g = a.g; const int km = g.npz;
  for (int k = km; k >= 1; --k) {
    const size_t n0 = fidx(g, a.pkz, tile, i, j, k), m0 = fidx(g, a.pe, tile, i, j, k), m1 = fidx(g, a.pe, tile, i, j, k + 1);
    const double t1 = a.kappa * (a.peln.t[m1] - a.peln.t[m0]);
    const double ad1 = a.pkz.p[n0] / t1, ad2 = -((a.pk.t[m1] - a.pk.t[m0]) * a.kappa * ad1 / t1);
    a.pk.p[m1] += ad1; a.pk.p[m0] -= ad1; a.peln.p[m1] += ad2; a.peln.p[m0] -= ad2; a.pkz.p[n0] = 0.;
  }
  double carry = 0.;
  for (int k = km + 1; k >= 1; --k) {
    const size_t m = fidx(g, a.pe, tile, i, j, k);
    const double ln_ad = a.peln.p[m] + a.pk.t[m] * a.kappa * a.pk.p[m];
    const double pe_ad = a.pe.p[m] + ln_ad / a.pe.t[m] + carry;
    a.pk.p[m] = 0.; a.peln.p[m] = 0.; a.pe.p[m] = 0.;
    if (k >= 2) { a.delp.p[fidx(g, a.delp, tile, i, j, k - 1)] += pe_ad; carry = pe_ad; }
  }
}
struct PressFn {
  PressArgs a; int mode;
  HD void operator()(int i, int j, int z) const {
    if (mode == MODE_NL) press_col<double>(a, i, j, z);
    else if (mode == MODE_TL) press_col<Dual>(a, i, j, z);
    else press_col_ad(a, i, j, z);
  }
};

// max Courant number per level (fv_tracer2d_tlm.F90:1248-1305); trajectory only.
struct CmaxFn {
  Geom g; Fld cx, cy; const double* sin5; double* out;   // out[ntile*npz], zeroed before the launch
  HD void operator()(int i, int, int z) const {            // one thread per (i, tile, level): column over j, coalesced in i
    const int tile = z / g.npz, k = 1 + z % g.npz;
    double cm = 0.;
    for (int j = g.js(); j <= g.je(); ++j) {
      const size_t n = (size_t)z * g.plane + g.idx(i, j);
      double ax = fabs(cx.t[n]), ay = fabs(cy.t[n]);
      double c = ax > ay ? ax : ay;
      if (!(k < g.npz / 6)) c += 1. - sin5[(size_t)tile * g.plane + g.idx(i, j)];
      if (cm < c) cm = c;
    }
#ifdef FV3LM_HOST_EMUL
    if (out[z] < cm) out[z] = cm;
#else
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(&out[z]), (unsigned long long)__double_as_longlong(cm));
#endif
  }
};

// dp1 <- dp2 between sub-steps for the levels still sub-cycling (fv_tracer2d_tlm.F90:1431-1437)
struct TrDp1Fn {
  Geom g; Fld dp1, dp2; const LevelParams* lev; int it, mode;
  HD void operator()(int i, int j, int z) const {
    const int k = 1 + z % g.npz;
    if (it > lev[k - 1].tr_ksplt) return;
    const size_t n = (size_t)z * g.plane + g.idx(i, j);
    if (mode == MODE_AD) { dp2.p[n] += dp1.p[n]; dp1.p[n] = 0.; return; }
    dp1.t[n] = dp2.t[n];
    if (mode == MODE_TL) dp1.p[n] = dp2.p[n];
  }
};
// max over the resident tiles: out[k] = max_t in[t*npz + k]  (one thread per level)
struct CmaxTilesFn {
  int ntile, npz; double* buf;
  HD void operator()(int k, int, int) const {
    double m = buf[k];
    for (int t = 1; t < ntile; ++t) { const double x = buf[(size_t)t * npz + k]; if (m < x) m = x; }
    buf[k] = m;
  }
};
// Boundary copies on the device (traj_to_fv3 / pert_to_fv3 / fv3_to_pert, DYN/fv3jedi_lm_dynamics_mod.F90:717-933): the host's arrays are
// compact -- (isc:iec, jsc:jec, nk) per tile, no halo.  Unpack: padded plane <- compact, halo zeroed (the reference zeroes the whole
// FV_Atm array first); pack: compact <- interior of the padded plane.
struct CompactFn {
  Geom g; double* pad; double* cmp; int nk, dir;      // dir 0: unpack (pad <- cmp, zeros outside), 1: pack (cmp <- pad)
  HD void operator()(int i, int j, int z) const {
    const size_t n = (size_t)z * g.plane + g.idx(i, j);
    const bool in = i >= g.is() && i <= g.ie() && j >= g.js() && j <= g.je();
    const size_t m = ((size_t)z * g.ty + (j - g.j0)) * g.tx + (i - g.i0);
    if (dir == 0) pad[n] = in ? cmp[m] : 0.0;
    else if (in) cmp[m] = pad[n];
  }
};
typedef void (*fv3lm_allreduce_fn)(void* user, double* buf, int n);   // in-place max over ranks (host buffer)
struct AllReduce { fv3lm_allreduce_fn cb = nullptr; void* user = nullptr; };
inline AllReduce& allreduce_max_hook() { static AllReduce a; return a; }

struct Dynamics : Dycore {
  Arena tshared, twork;
  Progs tracer_scale, tracer_pre, tracer_q, pt_in;
  Fld tr_dp2;
  int cur_km = 0;                                   // k_split iteration being run (tracer checkpoints are per iteration)
  std::vector<std::vector<int>> tr_ksplt_km;        // per k_split iteration: sub-steps per level, from the forward sweep
  std::vector<int> tr_nsplt_km;
  std::map<long, double*> sub_ck;                   // sub-step checkpoints (nsplt > 1 only): key (km, it, field)
  std::vector<Fld> q;
  Fld dp1, qc, qc_o, pe2, pu_ad, pv_ad;
  double *ak_dev = nullptr, *bk_dev = nullptr, *remap_ws = nullptr, *cmax_dev = nullptr;
  bool remap_ws_own = false;
  // non-hydrostatic vertical remap (nh.h): column operators into the staging fields, handed back to the state afterwards
  Progs remap_nh;
  Fld t_m, w_m, dz_m; std::vector<Fld> q_m;
  double* ck_nh = nullptr;     // per k_split step: delp, w, delz before the remap + ws
  double* cknh(int km, int n) { return ck_nh + (size_t)km * (3 * n3 + (size_t)ntile_all * g.plane) + (size_t)n * n3; }
  void build_remap_nh();
  void remap_nh_run(int mode, int km);
  double* ck_k = nullptr;      // per-k_split checkpoints
  double* ck_0 = nullptr;      // initial T and pkz
  size_t ck_k_stride = 0;
  int nsplt_max = 1;
  bool tracer_subcycle_error = false;
  std::vector<std::pair<double*, size_t>> tracer_zero;     // plan_adjoint(tracer_q, twork)
  std::vector<double*> snap;   // device snapshot of the prognostic state (fv3lm_state_save)
  double* stage_dev = nullptr;   // compact staging buffer of the boundary copies (one field)
  // one field between the host's compact array and the device state.  which: 0 trajectory, 1 perturbation / adjoint
  void compact_in(const Fld& f, int which, const double* host) {
    const size_t n = (size_t)ntile_all * f.nk * g.tx * g.ty;
    if (!stage_dev) stage_dev = (double*)dev_alloc((size_t)ntile_all * (g.npz + 1) * g.tx * g.ty * 8);
    h2d(ex, stage_dev, host, n * 8);
    each_class([&]() {
      const Fld fc = ex.sh(f);
      double* cmp = stage_dev + (size_t)classes[(size_t)cur_cls].t0 * f.nk * g.tx * g.ty;
      for_points(ex, Rect{g.isd(), g.ied() + 1, g.jsd(), g.jed() + 1}, g.ntile * f.nk, CompactFn{g, which ? fc.p : fc.t, cmp, f.nk, 0}, "boundary_unpack");
    });
  }
  void compact_out(const Fld& f, int which, double* host) {
    const size_t n = (size_t)ntile_all * f.nk * g.tx * g.ty;
    if (!stage_dev) stage_dev = (double*)dev_alloc((size_t)ntile_all * (g.npz + 1) * g.tx * g.ty * 8);
    each_class([&]() {
      const Fld fc = ex.sh(f);
      double* cmp = stage_dev + (size_t)classes[(size_t)cur_cls].t0 * f.nk * g.tx * g.ty;
      for_points(ex, Rect{g.is(), g.ie(), g.js(), g.je()}, g.ntile * f.nk, CompactFn{g, which ? fc.p : fc.t, cmp, f.nk, 1}, "boundary_pack");
    });
    d2h(ex, host, stage_dev, n * 8);
  }
  bool traj_to_fv3(const double* u, const double* v, const double* t, const double* delp, const double* const* qs, const double* w,
                   const double* delz, const double* phis);
  bool pert_to_fv3(const double* u, const double* v, const double* t, const double* delp, const double* const* qs, const double* w,
                   const double* delz);
  bool fv3_to_pert(double* u, double* v, double* t, double* delp, double* const* qs, double* w, double* delz);

  bool init2(const double* ak, const double* bk);
  void destroy2();
  void build_tracer();
  RemapArgs remap_args(bool last_step);
  void pressures(int mode);
  void tracer_fwd(int mode);
  void set_tracer_levels(const std::vector<int>& ksplt);
  double* subck(int km, int it, int n) {
    const long key = ((long)km * 64 + it) * 64 + n;
    auto f_ = sub_ck.find(key);
    if (f_ != sub_ck.end()) return f_->second;
    double* p = (double*)dev_alloc(n3 * 8); sub_ck[key] = p; return p;
  }
  void tracer_ad();
  void fv_dynamics(int mode);
  // non-hydrostatic: pe, peln, pk come out of the last acoustic step and pkz from the equation of state; nothing to prepare
  // "Edge of pert always needs to be filled" (DYN/fv3jedi_lm_dynamics_mod.F90:386-399): the perturbation's D-grid edge rows up(:, jec+1),
  // vp(iec+1, :) come from the neighbour faces (mpp_get_boundary) before compute_fv3_pressures_tlm / FV_DYNAMICS_TLM; the adjoint of that
  // fill (mpp_get_boundary_ad, :651-665) follows FV_DYNAMICS_BWD and compute_fv3_pressures_bwd.  Both belong to the operator whatever way
  // the host moved its arrays in (pert_to_fv3 zeroes those rows, :848-849); the forward fill is idempotent, the adjoint one clears the rows
  // it has moved, so a host that repeats either changes nothing.
  void step_tl() { halo(MODE_TL, H_DEDGE, f("u"), f("v")); if (!nh) pressures(MODE_TL); fv_dynamics(MODE_TL); }
  void step_nl() { if (!nh) pressures(MODE_NL); fv_dynamics(MODE_NL); }
  // after step_nl() has stored the checkpoints.  The backward sweep leaves the initial delp trajectory
  // in place (first acoustic checkpoint), from which the initial pressures are recomputed.
  void step_ad() { fv_dynamics(MODE_AD); if (!nh) { pressures(MODE_NL); pressures(MODE_AD); } halo(MODE_AD, H_DEDGE, f("u"), f("v")); }

  double* ckq(int km, int n) { return ck_k + (size_t)km * ck_k_stride + (size_t)n * n3; }                       // q before tracer
  double* ckm(int km, int n) { return ck_k + (size_t)km * ck_k_stride + (size_t)(nq + n) * n3; }                // mfx mfy cx cy
  double* ckr3(int km, int n) { return ck_k + (size_t)km * ck_k_stride + (size_t)(nq + 4 + n) * n3; }           // pt u v q'[nq]
  double* ckrp(int km, int n) { return ck_k + (size_t)km * ck_k_stride + (size_t)(2 * nq + 7) * n3 + (size_t)n * n3p; }  // pe peln pk
};

inline bool Dynamics::init2(const double* ak, const double* bk) {
  const int npz = g.npz;
  if (nq > 8) { err = "at most 8 tracers"; return false; }
  ak_dev = (double*)dev_alloc((npz + 1) * 8); bk_dev = (double*)dev_alloc((npz + 1) * 8);
  if (ak) h2d(ex, ak_dev, ak, (npz + 1) * 8);
  if (bk) h2d(ex, bk_dev, bk, (npz + 1) * 8);
  if (nh) {
    if (!ak || !bk) { err = "non-hydrostatic: ak, bk needed (reference layer thicknesses)"; return false; }
    for (int k = 1; k <= npz; ++k) lev_host[k - 1].dp_ref = ak[k] - ak[k - 1] + (bk[k] - bk[k - 1]) * 1.e5;
    lev_host[npz].dp_ref = lev_host[npz - 1].dp_ref;
    h2d(ex, lev_dev, lev_host.data(), sizeof(LevelParams) * (npz + 1));
  }
  for (int n = 0; n < nq; ++n) { char nm[16]; std::snprintf(nm, sizeof nm, "q%d", n + 1); q.push_back(S(nm, npz)); }
  dp1 = S("dp1", npz); qc = S("qc", npz); qc_o = S("qc_o", npz);
  pe2 = S("pe2", npz + 1); pu_ad = S("pu_ad", npz + 1); pv_ad = S("pv_ad", npz + 1);
  {   // the column workspace of the remap lives in the perturbation side of the acoustic work arena when it fits: every work
      // array is dead between acoustic steps (each step's first touch of a work tangent / adjoint is a store or a planned clear)
    const size_t need = (size_t)remap_ws_slots(nq) * (npz + 2) * ntile_all * g.plane;
    remap_ws_own = need > work.cap;
    remap_ws = remap_ws_own ? (double*)dev_alloc(need * 8) : work.p;
  }
  cmax_dev = (double*)dev_alloc((size_t)ntile_all * npz * 8);
  tshared.init(n3 * 9);       // build_tracer's nine shared fields; its per-tracer work arena is sized by a dry run
  tr_ksplt_km.assign(k_split, std::vector<int>(npz, 1)); tr_nsplt_km.assign(k_split, 1);
  for (Progs* P_ : {&tracer_scale, &tracer_pre, &tracer_q, &pt_in, &remap_nh}) { P_->c.assign(classes.size(), Program{}); P_->cur = &cur_cls; }
  std::map<double*, size_t> tzero;
  for (int c = 0; c < (int)classes.size(); ++c) {      // one set of programs per tile class, the fields shared
    set_class(c); reuse_fields = c > 0;
    build_tracer();
    for (auto& zr : tracer_zero) { auto it = tzero.find(zr.first); if (it == tzero.end() || it->second < zr.second) tzero[zr.first] = zr.second; }
    const Rect A = R(g.is(), g.ie(), g.js(), g.je());
    if (nh) {
      DynPtInNh s; s.in[0] = f("pt"); s.in[1] = nq > 0 ? q[0] : Fld{}; if (!s.in[1].t) s.in[1].nk = npz; s.in[2] = f("delp"); s.in[3] = f("delz");
      s.out[0] = f("pt_o"); s.out[1] = f("pkz"); s.orect[0] = s.orect[1] = A; s.k1 = npz; s.zvir = opt.zvir; s.akap = opt.akap;
      s.rdg = -opt.rdgas / opt.grav; s.has_q = nq > 0; add(pt_in, "pt_in", s);
      build_remap_nh();
    } else {
      DynPtIn s; s.in[0] = f("pt"); s.in[1] = nq > 0 ? q[0] : Fld{}; if (!s.in[1].t) s.in[1].nk = npz; s.in[2] = f("pkz"); s.out[0] = f("pt_o");
      s.orect[0] = A; s.k1 = npz; s.zvir = opt.zvir; s.has_q = nq > 0; add(pt_in, "pt_in", s);
    }
  }
  reuse_fields = false; set_class(-1);
  tracer_zero.assign(tzero.begin(), tzero.end());
  if (nh) ck_nh = (double*)dev_alloc((3 * n3 + (size_t)ntile_all * g.plane) * k_split * 8);
  ck_k_stride = (size_t)(2 * nq + 7) * n3 + 3 * n3p;
  ck_k = (double*)dev_alloc(ck_k_stride * k_split * 8);
  ck_0 = (double*)dev_alloc(2 * n3 * 8);
  init_traj_slots();
  return true;
}
inline void Dynamics::destroy2() {
  dev_free(stage_dev); stage_dev = nullptr;
  dev_free(ak_dev); dev_free(bk_dev); if (remap_ws_own) dev_free(remap_ws); dev_free(cmax_dev); dev_free(ck_k); dev_free(ck_0); dev_free(ck_nh);
  tshared.destroy(); twork.destroy();
  for (double* p : snap) dev_free(p);
  for (auto& kv : sub_ck) dev_free(kv.second);
}

// traj_to_fv3 (DYN/fv3jedi_lm_dynamics_mod.F90:717-807): halos zeroed, interiors from the host's compact arrays, the D-grid edge rows
// u(:, jec+1), v(iec+1, :) from the neighbours (mpp_get_boundary :781-793), halo of phis (:798), pe / peln / pk / pkz (:803-805)
inline bool Dynamics::traj_to_fv3(const double* u, const double* v, const double* t, const double* delp, const double* const* qs, const double* w,
                                  const double* delz, const double* phis) {
  if (!u || !v || !t || !delp || (nq > 0 && !qs) || (nh && (!w || !delz))) { err = "traj_to_fv3: null array"; return false; }
  for (int n = 0; n < nq; ++n) if (!qs[n]) { err = "traj_to_fv3: null tracer array"; return false; }      // before anything is touched
  compact_in(f("u"), 0, u); compact_in(f("v"), 0, v); compact_in(f("pt"), 0, t); compact_in(f("delp"), 0, delp);
  for (int n = 0; n < nq; ++n) compact_in(q[n], 0, qs[n]);
  if (nh) { compact_in(f("w"), 0, w); compact_in(f("delz"), 0, delz); }
  halo(MODE_NL, H_DEDGE, f("u"), f("v"));
  if (phis) { Fld hs; hs.t = hs_dev; hs.p = nullptr; hs.nk = 1; compact_in(hs, 0, phis); halo(MODE_NL, H_CELL, hs); }
  pressures(MODE_NL);
  return true;
}
// pert_to_fv3 (:846-889): perturbation / adjoint arrays, halos zeroed
inline bool Dynamics::pert_to_fv3(const double* u, const double* v, const double* t, const double* delp, const double* const* qs, const double* w,
                                  const double* delz) {
  if (!u || !v || !t || !delp || (nq > 0 && !qs) || (nh && (!w || !delz))) { err = "pert_to_fv3: null array"; return false; }
  for (int n = 0; n < nq; ++n) if (!qs[n]) { err = "pert_to_fv3: null tracer array"; return false; }
  compact_in(f("u"), 1, u); compact_in(f("v"), 1, v); compact_in(f("pt"), 1, t); compact_in(f("delp"), 1, delp);
  for (int n = 0; n < nq; ++n) compact_in(q[n], 1, qs[n]);
  if (nh) { compact_in(f("w"), 1, w); compact_in(f("delz"), 1, delz); }
  return true;
}
// fv3_to_pert (:893-933): compute-domain values back to the host; the device perturbation is cleared as the reference clears FV_AtmP
inline bool Dynamics::fv3_to_pert(double* u, double* v, double* t, double* delp, double* const* qs, double* w, double* delz) {
  if (!u || !v || !t || !delp || (nq > 0 && !qs) || (nh && (!w || !delz))) { err = "fv3_to_pert: null array"; return false; }
  for (int n = 0; n < nq; ++n) if (!qs[n]) { err = "fv3_to_pert: null tracer array"; return false; }
  compact_out(f("u"), 1, u); compact_out(f("v"), 1, v); compact_out(f("pt"), 1, t); compact_out(f("delp"), 1, delp);
  for (int n = 0; n < nq; ++n) compact_out(q[n], 1, qs[n]);
  if (nh) { compact_out(f("w"), 1, w); compact_out(f("delz"), 1, delz); }
  std::vector<Fld> fs{f("u"), f("v"), f("pt"), f("delp")};
  for (auto& x : q) fs.push_back(x);
  if (nh) { fs.push_back(f("w")); fs.push_back(f("delz")); }
  for (const Fld& x : fs) dev_zero(ex, x.p, n3 * 8);
  return true;
}

inline void Dynamics::build_remap_nh() {
  const int npz = g.npz;
  const Rect A = R(g.is(), g.ie(), g.js(), g.je());
  t_m = S("remap_t", npz); w_m = S("remap_w", npz); dz_m = S("remap_dz", npz);
  q_m.clear();
  for (int n = 0; n < nq; ++n) { char nm[24]; std::snprintf(nm, sizeof nm, "remap_q%d", n + 1); q_m.push_back(S(nm, npz)); }
  // split_kord: the column kernels that also run the trajectory's limited profile (instantiations of their own)
  const bool anylim = kord_limited(opt.kord_tm) || kord_limited(opt.kord_tr) || kord_limited(opt.kord_wz);
  const int kfield = anylim ? NHC_RM_FIELD_LIM : NHC_RM_FIELD, kw = anylim ? NHC_RM_W_LIM : NHC_RM_W;
  auto args = [&](int what) { NhColArgs a = nh_args(0.); a.ak = ak_dev; a.bk = bk_dev; a.what = what; return a; };
  { NhColArgs a = args(0); a.f[0] = f("pe"); a.f[1] = f("peln"); a.f[2] = f("pt"); a.f[3] = f("delp"); a.f[4] = f("delz"); a.f[5] = t_m;
    add_col(remap_nh, "remap", kfield, a, A, Rect{1, 0, 1, 0}, 3); }
  { NhColArgs a = args(1); a.f[0] = f("pe"); a.f[1] = f("w"); a.f[2] = f("ws"); a.f[3] = w_m; add_col(remap_nh, "remap", kw, a, A, Rect{1, 0, 1, 0}, 3); }
  { NhColArgs a = args(2); a.f[0] = f("pe"); a.f[1] = f("delz"); a.f[2] = f("delp"); a.f[3] = dz_m; add_col(remap_nh, "remap", kfield, a, A, Rect{1, 0, 1, 0}, 3); }
  for (int n = 0; n < nq; ++n) { NhColArgs a = args(3); a.f[0] = f("pe"); a.f[1] = q[n]; a.f[2] = q_m[n]; add_col(remap_nh, "remap", kfield, a, A, Rect{1, 0, 1, 0}, 3); }
  { NhColArgs a = args(nq > 0 ? 1 : 0); a.f[0] = f("pe"); a.f[1] = f("peln"); a.f[2] = f("pk"); a.f[3] = t_m; a.f[4] = dz_m; a.f[5] = nq > 0 ? q_m[0] : Fld{};
    a.f[6] = f("delp"); a.f[7] = f("pkz"); a.f[8] = f("pt"); a.f[9] = pe2; add_col(remap_nh, "remap", NHC_RM_PRESS, a, A, Rect{1, 0, 1, 0}, 3);
    remap_nh.now().back().accum = true; }     // overwrites delp, peln, pk: left out of the adjoint's trajectory recompute
}
// Vertical remap, non-hydrostatic: scalars by the column operators above, winds by the kernels of remap.h.
// Adjoint: the pre-remap trajectory (pt u v q pe peln pk delp w delz ws) must be in place.
inline void Dynamics::remap_nh_run(int mode, int km) {
  const size_t b3 = n3 * 8;
  const bool last = km == k_split - 1;
  remap_last = last;
  std::vector<std::pair<Fld, Fld>> back{{f("w"), w_m}, {f("delz"), dz_m}};
  for (int n = 0; n < nq; ++n) back.push_back({q[n], q_m[n]});
  if (mode != MODE_AD) {
    run_group(remap_nh, nullptr, mode);
    for (auto& pr : back) { dev_copy(ex, pr.first.t, pr.second.t, b3); if (mode == MODE_TL) dev_copy(ex, pr.first.p, pr.second.p, b3); }
    each_class([&]() {
      RemapArgs ra = remap_args(last);
      run_remap_winds(ex, mode, ra);
      for_points(ex, Rect{g.is(), g.ie(), g.js(), g.je()}, g.ntile, RemapPeFn{ra, mode}, "remap_pe");
    });
    return;
  }
  each_class([&]() {
    RemapArgs ra = remap_args(last);
    run_remap_winds(ex, MODE_AD, ra);
    for_points(ex, Rect{g.is() - 1, g.ie() + 1, g.js() - 1, g.je() + 1}, g.ntile, RemapGatherFn{ra}, "remap_gather.ad");
  });
  run_group(remap_nh, nullptr, MODE_NL, true);      // trajectory of the staging fields
  dev_zero(ex, t_m.p, b3); dev_zero(ex, pe2.p, n3p * 8);     // pe after the remap is not read again (pe2 is only copied into it)
  for (auto& pr : back) { dev_copy(ex, pr.second.p, pr.first.p, b3); dev_zero(ex, pr.first.p, b3); }
  run_group(remap_nh, nullptr, MODE_AD);
}

inline void Dynamics::build_tracer() {
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je(), isd = g.isd(), ied = g.ied(), jsd = g.jsd(), jed = g.jed(), npz = g.npz;
  auto TS = [&](const char* n) { if (reuse_fields) { auto it = F.find(n); if (it != F.end()) return it->second; } Fld x = tshared.take((size_t)ntile_all * npz * g.plane, npz); F[n] = x; return x; };
  Fld xfx = TS("tr_xfx"), yfx = TS("tr_yfx"), dp2 = TS("tr_dp2"), rax = TS("tr_rax"), ray = TS("tr_ray");
  Fld cxs = TS("tr_cxs"), cys = TS("tr_cys"), mfxs = TS("tr_mfxs"), mfys = TS("tr_mfys");
  tr_dp2 = dp2;
  Fld cx = f("cx"), cy = f("cy"), mfx = f("mfx"), mfy = f("mfy");
  // once per call: scaled Courant numbers / mass fluxes and the area fluxes
  { TrScale s; s.in[0] = cx; s.in[1] = cy; s.in[2] = mfx; s.in[3] = mfy; s.out[0] = cxs; s.out[1] = cys; s.out[2] = mfxs; s.out[3] = mfys;
    s.orect[0] = R(is, ie + 1, jsd, jed); s.orect[1] = R(isd, ied, js, je + 1); s.orect[2] = R(is, ie + 1, js, je); s.orect[3] = R(is, ie, js, je + 1);
    s.k1 = npz; add(tracer_scale, "tracer", s); }
  { TrFlux s; s.in[0] = cxs; s.in[1] = cys; s.out[0] = xfx; s.out[1] = yfx; s.orect[0] = R(is, ie + 1, jsd, jed); s.orect[1] = R(isd, ied, js, je + 1);
    s.k1 = npz; add(tracer_scale, "tracer", s); }
  // once per sub-step
  { TrDp2Ra s; s.in[0] = dp1; s.in[1] = mfxs; s.in[2] = mfys; s.in[3] = xfx; s.in[4] = yfx; s.out[0] = dp2; s.out[1] = rax; s.out[2] = ray;
    s.orect[0] = R(is, ie, js, je); s.orect[1] = R(is, ie, jsd, jed); s.orect[2] = R(isd, ied, js, je); s.k1 = npz; add(tracer_pre, "tracer", s); }
  // per tracer and sub-step, on the staging field qc -> qc_o, work arrays in twork
  auto build_q = [&]() {
    Arena save = work; work = twork;
    Fld fx = W("tr_fx", npz), fy = W("tr_fy", npz);
    build_tp(tracer_q, "tracer", "tpq", qc, cxs, cys, xfx, yfx, rax, ray, mfxs, mfys, Fld{}, HORD_TR, DAMP_NONE, false, fx, fy);
    { TrUpdate s; s.in[0] = qc; s.in[1] = dp1; s.in[2] = dp2; s.in[3] = fx; s.in[4] = fy; s.out[0] = qc_o; s.orect[0] = R(is, ie, js, je); s.k1 = npz;
      add(tracer_q, "tracer", s); }
    twork = work; work = save;
  };
  if (!reuse_fields) {       // first class: size the per-tracer work arena by a dry run
    const std::map<std::string, Fld> F_mark = F;
    twork.measure(); build_q();
    const size_t need = twork.used; tracer_q.clear(); F = F_mark; twork.init(need);
  }
  build_q();
  tracer_zero = plan_adjoint(tracer_q, twork);
}

inline void Dynamics::set_tracer_levels(const std::vector<int>& ksplt) {
  for (int k = 0; k < g.npz; ++k) { lev_host[k].tr_ksplt = ksplt[k]; lev_host[k].tr_frac = 1. / double(ksplt[k]); }
  h2d(ex, lev_dev, lev_host.data(), sizeof(LevelParams) * g.npz);
}

inline RemapArgs Dynamics::remap_args(bool last_step) {      // for the class being run: its geometry, its tiles' fields and workspace columns
  RemapArgs a; a.g = g; a.pe = ex.sh(f("pe")); a.peln = ex.sh(f("peln")); a.pk = ex.sh(f("pk")); a.pkz = ex.sh(f("pkz")); a.pt = ex.sh(f("pt")); a.delp = ex.sh(f("delp"));
  a.u = ex.sh(f("u")); a.v = ex.sh(f("v")); a.pe2 = ex.sh(pe2); a.nq = nq;
  for (int n = 0; n < nq; ++n) a.q[n] = ex.sh(q[n]);
  a.ak = ak_dev; a.bk = bk_dev; a.akap = opt.akap; a.zvir = opt.zvir; a.ptop = opt.ptop; a.last_step = last_step;
  a.kord_tm = opt.kord_tm; a.kord_mt = opt.kord_mt; a.kord_tr = opt.kord_tr;
  a.ws = remap_ws + ex.cls_off; a.ws_stride = (size_t)ntile_all * g.plane; a.pu_ad = ex.sh(pu_ad); a.pv_ad = ex.sh(pv_ad);
  return a;
}

inline void Dynamics::pressures(int mode) {
  each_class([&]() {
    PressArgs a{g, ex.sh(f("delp")), ex.sh(f("pe")), ex.sh(f("peln")), ex.sh(f("pk")), ex.sh(f("pkz")), opt.akap, opt.ptop};
    for_points(ex, Rect{g.is(), g.ie(), g.js(), g.je()}, g.ntile, PressFn{a, mode}, "pressures");
  });
}

// tracer_2d forward (nonlinear or tangent); q halos must be valid.  The sub-step count comes from the trajectory's maximum
// Courant number per level over all faces and ranks (fv_tracer2d_tlm.F90:1248-1317).
inline void Dynamics::tracer_fwd(int mode) {
  const size_t b3 = n3 * 8;
  const int npz = g.npz, km = cur_km;
  {
    dev_zero(ex, cmax_dev, (size_t)ntile_all * npz * 8);
    each_class([&]() { for_points(ex, Rect{g.is(), g.ie(), g.js(), g.js()}, g.ntile * npz, CmaxFn{g, ex.sh(f("cx")), ex.sh(f("cy")), ctx.m.sin_sg[5], cmax_dev + (size_t)classes[(size_t)cur_cls].t0 * npz}, "tracer_cmax"); });
    // mp_reduce_max (fv_tracer2d_tlm.F90:1306): over the resident tiles on the device, over the ranks by ncclAllReduce(max) on the
    // library stream when an RCCL communicator is up (the one collective of the path: npz doubles per k_split step); the host hook
    // serves transports without RCCL (gloo rehearsals, host emulation)
    for_points(ex, Rect{0, npz - 1, 0, 0}, 1, CmaxTilesFn{ntile_all, npz, cmax_dev}, "tracer_cmax_tiles");
#ifndef FV3LM_HOST_EMUL
    { Transport& T = transport();
      if (T.comm && !allreduce_max_hook().cb) {
        const ncclResult_t r = T.pAllReduce(cmax_dev, cmax_dev, (size_t)npz, ncclDouble, ncclMax, T.comm, ex.stream);
        if (r != ncclSuccess) set_sticky("ncclAllReduce(max) of the tracer Courant numbers failed (code " + std::to_string((int)r) + ")");
      } }
#endif
    std::vector<double> cl(npz, 0.);
    d2h(ex, cl.data(), cmax_dev, (size_t)npz * 8);
    if (allreduce_max_hook().cb) allreduce_max_hook().cb(allreduce_max_hook().user, cl.data(), npz);
    double cg = 0.; for (double c : cl) if (!(c < cg)) cg = c;
    const int nsplt = int(1. + cg);
    std::vector<int> ks(npz, 1);
    if (nsplt != 1) for (int k = 0; k < npz; ++k) ks[k] = int(1. + cl[k]);
    if (nsplt > 60) { tracer_subcycle_error = true; return; }
    tr_ksplt_km[km] = ks; tr_nsplt_km[km] = nsplt;
    if (nsplt > nsplt_max) nsplt_max = nsplt;
  }
  const int nsplt = tr_nsplt_km[km];
  set_tracer_levels(tr_ksplt_km[km]);
  run_group(tracer_scale, nullptr, mode);
  for (int it = 1; it <= nsplt; ++it) {
    ctx.tr_it = it;
    if (mode == MODE_NL && nsplt > 1) {     // trajectory of the later sub-steps for the backward sweep
      dev_copy(ex, subck(km, it, nq), dp1.t, b3);
      for (int n = 0; n < nq; ++n) dev_copy(ex, subck(km, it, n), q[n].t, b3);
    }
    run_group(tracer_pre, nullptr, mode);
    for (int n = 0; n < nq; ++n) {
      ex.nrt = ex.nrp = 0;             // the transport program reads tracer n where it is (exec.h Redir); its output cannot go there (neighbours' halos)
      ex.redirect_t(qc.t, q[n].t); ex.redirect_p(qc.p, q[n].p);
      run_group(tracer_q, nullptr, mode);
      ex.nrt = ex.nrp = 0;
      dev_copy(ex, q[n].t, qc_o.t, b3);
      if (mode == MODE_TL) dev_copy(ex, q[n].p, qc_o.p, b3);
    }
    if (it != nsplt) {
      each_class([&]() { for_points(ex, Rect{g.is(), g.ie(), g.js(), g.je()}, g.ntile * npz, TrDp1Fn{g, ex.sh(dp1), ex.sh(tr_dp2), lev_dev, it, mode}, "tracer_dp1"); });
      for (int n = 0; n < nq; ++n) halo(mode, H_CELL, q[n]);
    }
  }
  ctx.tr_it = 1;
}
// adjoint: trajectory of dp1, mfx..cy and the pre-transport q[n] must be in place; q[n].p holds the
// adjoint of the transported tracers on entry, of the inputs on exit; mfx..cy.p, dp1.p accumulate.
inline void Dynamics::tracer_ad() {
  const size_t b3 = n3 * 8;
  const int npz = g.npz, km = cur_km, nsplt = tr_nsplt_km[km];
  set_tracer_levels(tr_ksplt_km[km]);
  run_group(tracer_scale, nullptr, MODE_NL);
  dev_zero(ex, tshared.p, tshared.used * 8);
  for (int it = nsplt; it >= 1; --it) {
    ctx.tr_it = it;
    if (it != nsplt) {
      for (int n = nq - 1; n >= 0; --n) halo(MODE_AD, H_CELL, q[n]);
      each_class([&]() { for_points(ex, Rect{g.is(), g.ie(), g.js(), g.je()}, g.ntile * npz, TrDp1Fn{g, ex.sh(dp1), ex.sh(tr_dp2), lev_dev, it, MODE_AD}, "tracer_dp1"); });
    }
    if (nsplt > 1) dev_copy(ex, dp1.t, subck(km, it, nq), b3);
    run_group(tracer_pre, nullptr, MODE_NL);
    for (int n = nq - 1; n >= 0; --n) {
      ex.nrt = ex.nrp = 0;             // trajectory of tracer n read where it is, the incoming adjoint likewise; the result is built in qc.p
      ex.redirect_t(qc.t, nsplt > 1 ? subck(km, it, n) : q[n].t); ex.redirect_p(qc_o.p, q[n].p);
      run_group(tracer_q, nullptr, MODE_NL);
      for (auto& zr : tracer_zero) dev_zero(ex, zr.first, zr.second * 8);       // plan_adjoint: the rest is stored by its first stage launch
      dev_zero(ex, qc.p, b3);
      run_group(tracer_q, nullptr, MODE_AD);
      ex.nrt = ex.nrp = 0;
      dev_copy(ex, q[n].p, qc.p, b3);
    }
    run_group(tracer_pre, nullptr, MODE_AD);
    if (it != 1) for (const char* nm : {"tr_dp2", "tr_rax", "tr_ray"}) dev_zero(ex, f(nm).p, b3);   // per-sub-step adjoints
  }
  run_group(tracer_scale, nullptr, MODE_AD);
  ctx.tr_it = 1;
}

inline void Dynamics::fv_dynamics(int mode) {
  const size_t b3 = n3 * 8, b3p = n3p * 8;
  if (mode != MODE_AD) {
    if (mode == MODE_NL) { dev_copy(ex, ck_0, f("pt").t, b3); dev_copy(ex, ck_0 + n3, f("pkz").t, b3); }
    run_group(pt_in, nullptr, mode);
    dev_copy(ex, f("pt").t, f("pt_o").t, b3);
    if (mode == MODE_TL) dev_copy(ex, f("pt").p, f("pt_o").p, b3);
    for (int km = 0; km < k_split; ++km) {
      halo(mode, H_DVEC, f("u"), f("v")); halo(mode, H_CELL, f("delp")); halo(mode, H_CELL, f("pt"));
      dev_copy(ex, dp1.t, f("delp").t, b3);
      if (mode == MODE_TL) dev_copy(ex, dp1.p, f("delp").p, b3);
      ck_base = km * n_split; cur_km = km;
      dyn_core(mode);
      if (nq > 0) {
        for (int n = 0; n < nq; ++n) halo(mode, H_CELL, q[n]);
        if (mode == MODE_NL) {
          for (int n = 0; n < nq; ++n) dev_copy(ex, ckq(km, n), q[n].t, b3);
          const char* mf[4] = {"mfx", "mfy", "cx", "cy"};
          for (int n = 0; n < 4; ++n) dev_copy(ex, ckm(km, n), f(mf[n]).t, b3);
        }
        tracer_fwd(mode);
      }
      if (g.npz > 4) {
        if (mode == MODE_NL) {
          const char* r3[3] = {"pt", "u", "v"}; const char* rp[3] = {"pe", "peln", "pk"};
          for (int n = 0; n < 3; ++n) dev_copy(ex, ckr3(km, n), f(r3[n]).t, b3);
          for (int n = 0; n < nq; ++n) dev_copy(ex, ckr3(km, 3 + n), q[n].t, b3);
          for (int n = 0; n < 3; ++n) dev_copy(ex, ckrp(km, n), f(rp[n]).t, b3p);
          if (nh) {
            const char* r4[3] = {"delp", "w", "delz"};
            for (int n = 0; n < 3; ++n) dev_copy(ex, cknh(km, n), f(r4[n]).t, b3);
            dev_copy(ex, cknh(km, 3), f("ws").t, (size_t)ntile_all * g.plane * 8);
          }
        }
        if (nh) remap_nh_run(mode, km); else
        each_class([&]() { run_remap(ex, mode, remap_args(km == k_split - 1)); });
      }
    }
    return;
  }
  // ------------------------------------------------------------------ adjoint sweep
  // entry: u,v,pt,delp,q[n] .p = adjoint of the step outputs.  pe..pkz after the last remap are dead.
  for (const char* n : {"pe", "peln", "pk"}) dev_zero(ex, f(n).p, b3p);
  dev_zero(ex, f("pkz").p, b3);
  if (nh) dev_zero(ex, f("ws").p, (size_t)ntile_all * g.plane * 8);
  for (int km = k_split - 1; km >= 0; --km) {
    if (g.npz > 4) {
      const char* r3[3] = {"pt", "u", "v"}; const char* rp[3] = {"pe", "peln", "pk"};
      for (int n = 0; n < 3; ++n) dev_copy(ex, f(r3[n]).t, ckr3(km, n), b3);
      for (int n = 0; n < nq; ++n) dev_copy(ex, q[n].t, ckr3(km, 3 + n), b3);
      for (int n = 0; n < 3; ++n) dev_copy(ex, f(rp[n]).t, ckrp(km, n), b3p);
      dev_zero(ex, f("pe").p, b3p);
      if (nh) {
        const char* r4[3] = {"delp", "w", "delz"};
        for (int n = 0; n < 3; ++n) dev_copy(ex, f(r4[n]).t, cknh(km, n), b3);
        dev_copy(ex, f("ws").t, cknh(km, 3), (size_t)ntile_all * g.plane * 8);
        remap_nh_run(MODE_AD, km);
      } else
      each_class([&]() { run_remap(ex, MODE_AD, remap_args(km == k_split - 1)); });
    }
    const char* mf[4] = {"mfx", "mfy", "cx", "cy"};
    for (int n = 0; n < 4; ++n) dev_zero(ex, f(mf[n]).p, b3);
    dev_zero(ex, dp1.p, b3);
    if (nq > 0) {
      for (int n = 0; n < nq; ++n) dev_copy(ex, q[n].t, ckq(km, n), b3);
      for (int n = 0; n < 4; ++n) dev_copy(ex, f(mf[n]).t, ckm(km, n), b3);
      dev_copy(ex, dp1.t, ckpt + (size_t)(km * n_split) * ck_stride + 2 * n3, b3);   // delp at the start of this k_split step
      cur_km = km;
      tracer_ad();
      for (int n = 0; n < nq; ++n) halo(MODE_AD, H_CELL, q[n]);
    }
    ck_base = km * n_split;
    dyn_core(MODE_AD);
    each_class([&]() { for_points(ex, Rect{g.isd(), g.ied() + 1, g.jsd(), g.jed() + 1}, g.ntile * g.npz, AccumFn{g, ex.sh(dp1), ex.sh(f("delp")), MODE_AD}, "accum"); });   // delp.p += dp1.p
    halo(MODE_AD, H_CELL, f("pt")); halo(MODE_AD, H_CELL, f("delp")); halo(MODE_AD, H_DVEC, f("u"), f("v"));
  }
  // pt_in: pt(theta_v) = T (1 + zvir qv) / pkz
  dev_copy(ex, f("pt").t, ck_0, b3); dev_copy(ex, f("pkz").t, ck_0 + n3, b3);
  if (nq > 0) dev_copy(ex, q[0].t, ckq(0, 0), b3);
  dev_copy(ex, f("pt_o").p, f("pt").p, b3); dev_zero(ex, f("pt").p, b3);
  run_group(pt_in, nullptr, MODE_AD);
}

}  // namespace fv3
