// TEST INFRASTRUCTURE — C entry points of the CPU oracle (liboracle.so), loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.  See scalar.hpp.
//
// Array layout (shared with the product's C-ABI): every field is [nk][pj][pi] doubles,
// pi = nx+2*ng+1, pj = ny+2*ng+1, element (i,j,k) at ((k-1)*pj + (j-jsd))*pi + (i-isd), ng=3.
// mode: 0 = nonlinear, 1 = tangent linear (pert = tangent), 2 = adjoint (on entry out[].pert
// holds the adjoint of the outputs; on exit in[].pert holds the adjoint of the inputs).
#include "dyn_core.hpp"
#include "fv_dynamics.hpp"
#include "fv_pressure.hpp"
#include "cube.hpp"
#include "nh.hpp"
#include <memory>
#include <string>

using namespace orc;

struct OrcHandle {
  Bounds bd; Grid g; int npz = 0, nq = 0;
  DampOpts o; Consts c; double ptop = 1.0;
  Arr2<double> phis;
  std::vector<double> ak, bk;
  RemapOpts ro;
};

struct IO { double* traj; double* pert; int nk; };

static void load(Arr3<double>& a, const IO& io, const Bounds& bd, Tape*, std::vector<int32_t>*) {
  size_t n = (size_t)bd.pi() * bd.pj();
  for (int k = 0; k < io.nk; ++k) std::memcpy(a.p[k].d.data(), io.traj + k * n, n * sizeof(double));
}
static void load(Arr3<Dual>& a, const IO& io, const Bounds& bd, Tape*, std::vector<int32_t>*) {
  size_t n = (size_t)bd.pi() * bd.pj();
  for (int k = 0; k < io.nk; ++k)
    for (size_t m = 0; m < n; ++m) a.p[k].d[m] = Dual(io.traj[k * n + m], io.pert ? io.pert[k * n + m] : 0.0);
}
static void load(Arr3<Rev>& a, const IO& io, const Bounds& bd, Tape* t, std::vector<int32_t>* ids) {
  size_t n = (size_t)bd.pi() * bd.pj();
  for (int k = 0; k < io.nk; ++k)
    for (size_t m = 0; m < n; ++m) {
      int32_t id = io.pert ? t->fresh() : -1;
      a.p[k].d[m] = Rev(io.traj[k * n + m], id);
      ids->push_back(id);
    }
}
static void store(const Arr3<double>& a, const IO& io, const Bounds& bd) {
  size_t n = (size_t)bd.pi() * bd.pj();
  for (int k = 0; k < io.nk; ++k) std::memcpy(io.traj + k * n, a.p[k].d.data(), n * sizeof(double));
}
static void store(const Arr3<Dual>& a, const IO& io, const Bounds& bd) {
  size_t n = (size_t)bd.pi() * bd.pj();
  for (int k = 0; k < io.nk; ++k)
    for (size_t m = 0; m < n; ++m) { io.traj[k * n + m] = a.p[k].d[m].v; if (io.pert) io.pert[k * n + m] = a.p[k].d[m].d; }
}

// Generic NL/TL/AD driver.  f(tag, in, out) runs the templated routine on Arr3<T> lists.
template <class F>
static void drive(int mode, const Bounds& bd, std::vector<IO>& in, std::vector<IO>& out, F f) {
  if (mode == 0) {
    std::vector<Arr3<double>> xi(in.size()), xo(out.size());
    for (size_t n = 0; n < in.size(); ++n) { xi[n].init(bd, in[n].nk); load(xi[n], in[n], bd, nullptr, nullptr); }
    for (size_t n = 0; n < out.size(); ++n) xo[n].init(bd, out[n].nk);
    f(xi, xo);
    for (size_t n = 0; n < out.size(); ++n) store(xo[n], out[n], bd);
  } else if (mode == 1) {
    std::vector<Arr3<Dual>> xi(in.size()), xo(out.size());
    for (size_t n = 0; n < in.size(); ++n) { xi[n].init(bd, in[n].nk); load(xi[n], in[n], bd, nullptr, nullptr); }
    for (size_t n = 0; n < out.size(); ++n) xo[n].init(bd, out[n].nk);
    f(xi, xo);
    for (size_t n = 0; n < out.size(); ++n) store(xo[n], out[n], bd);
  } else {
    Tape tape; active_tape() = &tape;
    std::vector<Arr3<Rev>> xi(in.size()), xo(out.size());
    std::vector<std::vector<int32_t>> ids(in.size());
    for (size_t n = 0; n < in.size(); ++n) { xi[n].init(bd, in[n].nk); load(xi[n], in[n], bd, &tape, &ids[n]); }
    for (size_t n = 0; n < out.size(); ++n) xo[n].init(bd, out[n].nk);
    f(xi, xo);
    std::vector<double> adj(tape.size(), 0.0);
    size_t np = (size_t)bd.pi() * bd.pj();
    for (size_t n = 0; n < out.size(); ++n)
      for (int k = 0; k < out[n].nk; ++k)
        for (size_t m = 0; m < np; ++m) {
          const Rev& r = xo[n].p[k].d[m];
          out[n].traj[k * np + m] = r.v;
          if (out[n].pert && r.id >= 0) adj[r.id] += out[n].pert[k * np + m];
        }
    tape.reverse(adj);
    for (size_t n = 0; n < in.size(); ++n)
      if (in[n].pert)
        for (size_t m = 0; m < ids[n].size(); ++m) in[n].pert[m] = adj[ids[n][m]];
    active_tape() = nullptr;
  }
}

static std::vector<IO> mkio(int n, double** traj, double** pert, const int* nk) {
  std::vector<IO> v(n);
  for (int i = 0; i < n; ++i) { v[i].traj = traj[i]; v[i].pert = pert ? pert[i] : nullptr; v[i].nk = nk[i]; }
  return v;
}

extern "C" {

// Metric order for orc_create (each a [pj][pi] plane):
static const char* METRIC_NAMES =
    "area,rarea,rarea_c,dx,dy,dxa,dya,dxc,dyc,rdx,rdy,rdxa,rdya,rdxc,rdyc,cosa,sina,rsina,cosa_u,cosa_v,cosa_s,"
    "sina_u,sina_v,rsin_u,rsin_v,rsin2,f0,fC,del6_u,del6_v,divg_u,divg_v,"
    "sin_sg1,sin_sg2,sin_sg3,sin_sg4,sin_sg5,sin_sg6,sin_sg7,sin_sg8,sin_sg9,"
    "cos_sg1,cos_sg2,cos_sg3,cos_sg4,cos_sg5,cos_sg6,cos_sg7,cos_sg8,cos_sg9";
const char* orc_metric_names() { return METRIC_NAMES; }

// iopt: hord_mt,hord_vt,hord_tm,hord_dp,hord_tr, nord, do_vort_damp, n_sponge,
//       hord_*_pert(5), nord_pert, do_vort_damp_pert, n_sponge_pert, hord_ks_traj, hord_ks_pert,
//       hord_*_ks_traj(5), hord_*_ks_pert(5), kord_tm, kord_mt, kord_wz, kord_tr, kord_*_pert(4), split_damp        (37 ints)
// ropt: dddmp,d2_bg,d4_bg,vtdm4,d2_bg_k1,d2_bg_k2,d_con,ke_bg, dddmp_pert,d2_bg_pert,d4_bg_pert,vtdm4_pert,
//       d2_bg_k1_pert,d2_bg_k2_pert,d2_bg_ks_pert, akap,cp,zvir,grav_jedi, cp_air,rdgas,rvgas,grav,radius,omega,hlv,
//       ptop, da_min, da_min_c                                                            (29 doubles)
void* orc_create(int nx, int ny, int npz, int nq, const double* const* metrics, const int* iopt, const double* ropt,
                 const double* phis, const double* ak, const double* bk) {
  OrcHandle* h = new OrcHandle;
  h->bd.set(nx, ny); h->npz = npz; h->nq = nq;
  h->g.init(h->bd);
  Arr2<double>* all[] = {&h->g.area, &h->g.rarea, &h->g.rarea_c, &h->g.dx, &h->g.dy, &h->g.dxa, &h->g.dya, &h->g.dxc,
                         &h->g.dyc, &h->g.rdx, &h->g.rdy, &h->g.rdxa, &h->g.rdya, &h->g.rdxc, &h->g.rdyc, &h->g.cosa,
                         &h->g.sina, &h->g.rsina, &h->g.cosa_u, &h->g.cosa_v, &h->g.cosa_s, &h->g.sina_u, &h->g.sina_v,
                         &h->g.rsin_u, &h->g.rsin_v, &h->g.rsin2, &h->g.f0, &h->g.fC, &h->g.del6_u, &h->g.del6_v,
                         &h->g.divg_u, &h->g.divg_v};
  size_t np = (size_t)h->bd.pi() * h->bd.pj();
  int m = 0;
  for (auto* a : all) std::memcpy(a->d.data(), metrics[m++], np * sizeof(double));
  for (int n = 1; n <= 9; ++n) std::memcpy(h->g.sin_sg[n].d.data(), metrics[m++], np * sizeof(double));
  for (int n = 1; n <= 9; ++n) std::memcpy(h->g.cos_sg[n].d.data(), metrics[m++], np * sizeof(double));
  DampOpts& o = h->o; const int* p = iopt;
  o.hord_mt = *p++; o.hord_vt = *p++; o.hord_tm = *p++; o.hord_dp = *p++; o.hord_tr = *p++;
  o.nord = *p++; o.do_vort_damp = *p++; o.n_sponge = *p++;
  o.hord_mt_pert = *p++; o.hord_vt_pert = *p++; o.hord_tm_pert = *p++; o.hord_dp_pert = *p++; o.hord_tr_pert = *p++;
  o.nord_pert = *p++; o.do_vort_damp_pert = *p++; o.n_sponge_pert = *p++; o.hord_ks_traj = *p++; o.hord_ks_pert = *p++;
  o.hord_mt_ks_traj = *p++; o.hord_vt_ks_traj = *p++; o.hord_tm_ks_traj = *p++; o.hord_dp_ks_traj = *p++; o.hord_tr_ks_traj = *p++;
  o.hord_mt_ks_pert = *p++; o.hord_vt_ks_pert = *p++; o.hord_tm_ks_pert = *p++; o.hord_dp_ks_pert = *p++; o.hord_tr_ks_pert = *p++;
  h->ro.kord_tm = *p++; h->ro.kord_mt = *p++; h->ro.kord_wz = *p++; h->ro.kord_tr = *p++;
  h->ro.kord_tm_pert = *p++; h->ro.kord_mt_pert = *p++; h->ro.kord_wz_pert = *p++; h->ro.kord_tr_pert = *p++;
  o.split_damp = *p++ != 0;
  const double* r = ropt;
  o.dddmp = *r++; o.d2_bg = *r++; o.d4_bg = *r++; o.vtdm4 = *r++; o.d2_bg_k1 = *r++; o.d2_bg_k2 = *r++; o.d_con = *r++; o.ke_bg = *r++;
  o.dddmp_pert = *r++; o.d2_bg_pert = *r++; o.d4_bg_pert = *r++; o.vtdm4_pert = *r++; o.d2_bg_k1_pert = *r++;
  o.d2_bg_k2_pert = *r++; o.d2_bg_ks_pert = *r++;
  Consts& c = h->c;
  c.akap = *r++; c.cp = *r++; c.zvir = *r++; c.grav_jedi = *r++;
  c.cp_air = *r++; c.rdgas = *r++; c.rvgas = *r++; c.grav = *r++; c.radius = *r++; c.omega = *r++; c.hlv = *r++;
  h->ptop = *r++; h->g.da_min = *r++; h->g.da_min_c = *r++;
  h->phis.init(h->bd);
  if (phis) std::memcpy(h->phis.d.data(), phis, np * sizeof(double));
  h->ak.assign(ak, ak + npz + 1); h->bk.assign(bk, bk + npz + 1);
  return h;
}
void orc_destroy(void* h) { delete (OrcHandle*)h; }

// Switch the tile to "whole cube face" mode (is=1, ie=npx-1, all edges and corners) and load the a2b_ord4
// edge weights (edge[4][pj]: w, e, s, n, indexed by padded-plane position) and extrap_corner coefficients.
void orc_set_face(void* hv, const double* edge, const double* ecorner) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro;
  const int n = h->bd.nx, pj = h->bd.pj();
  h->bd.set_face(n);
  std::vector<double>* dst[4] = {&h->g.edge_w, &h->g.edge_e, &h->g.edge_s, &h->g.edge_n};
  for (int e = 0; e < 4; ++e) {
    dst[e]->assign(n + 3, 0.0);
    for (int i = 1; i <= n + 1; ++i) (*dst[e])[i] = edge[e * pj + (i + h->bd.ng - 1)];
  }
  for (int c = 0; c < 4; ++c) for (int k = 0; k < 3; ++k) h->g.ecorner[c][k] = ecorner[c * 3 + k];
}

// Per-level parameters as the reference would hand them to d_sw; returns 0 if traj/pert hord split.
int orc_level_params(void* hv, int k, int* ip, double* rp) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; LevelParams lp;
  bool ok = level_params(h->o, k, h->npz, lp);
  ip[0] = lp.hord_mt; ip[1] = lp.hord_vt; ip[2] = lp.hord_tm; ip[3] = lp.hord_dp; ip[4] = lp.hord_tr;
  ip[5] = lp.nord; ip[6] = lp.nord_v; ip[7] = lp.nord_w; ip[8] = lp.nord_t; ip[9] = lp.nord_v_pert;
  rp[0] = lp.d2_divg; rp[1] = lp.damp_vt; rp[2] = lp.damp_w; rp[3] = lp.damp_t; rp[4] = lp.d_con; rp[5] = lp.damp_vt_pert;
  return ok ? 1 : 0;
}

// in: q, crx, cry, xfx, yfx, ra_x, ra_y, [mfx, mfy, mass]   out: fx, fy   (single planes)
void orc_fv_tp_2d(void* hv, int mode, int hord, int nord, double damp_c, int use_mf, int use_mass, double** in_t,
                  double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro;
  int nin = 7 + (use_mf ? 2 : 0) + (use_mass ? 1 : 0);
  std::vector<int> nk(nin, 1), nko(2, 1);
  auto in = mkio(nin, in_t, in_p, nk.data()); auto out = mkio(2, out_t, out_p, nko.data());
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    const Arr2<T>* mfx = use_mf ? &x[7].p[0] : nullptr; const Arr2<T>* mfy = use_mf ? &x[8].p[0] : nullptr;
    const Arr2<T>* mass = use_mass ? &x[use_mf ? 9 : 7].p[0] : nullptr;
    fv_tp_2d<T>(x[0].p[0], x[1].p[0], x[2].p[0], hord, y[0].p[0], y[1].p[0], x[3].p[0], x[4].p[0], h->g, h->bd,
                x[5].p[0], x[6].p[0], mfx, mfy, mass, nord, damp_c);
  });
}

// c_sw on all levels.  in: delp, pt, u, v   out: delpc, ptc, uc, vc, ua, va, ut, vt, divgd
void orc_c_sw(void* hv, int mode, double dt2, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(4, npz), nko(9, npz);
  auto in = mkio(4, in_t, in_p, nk.data()); auto out = mkio(9, out_t, out_p, nko.data());
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    for (int k = 1; k <= npz; ++k)
      c_sw(y[0].plane(k), x[0].plane(k), y[1].plane(k), x[1].plane(k), x[2].plane(k), x[3].plane(k), y[2].plane(k),
           y[3].plane(k), y[4].plane(k), y[5].plane(k), y[6].plane(k), y[7].plane(k), y[8].plane(k), h->o.nord, dt2,
           h->g, h->bd);
  });
}

// d_sw on all levels.  in: delp, pt, u, v, uc, vc, ua, va, divgd, mfx, mfy, cx, cy
//                      out: delp, pt, u, v, mfx, mfy, cx, cy, crx, cry, xfx, yfx
void orc_d_sw(void* hv, int mode, double dt, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(13, npz), nko(12, npz);
  auto in = mkio(13, in_t, in_p, nk.data()); auto out = mkio(12, out_t, out_p, nko.data());
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    for (int n : {0, 1, 2, 3}) y[n] = x[n];
    for (int n : {0, 1, 2, 3}) y[4 + n] = x[9 + n];
    for (int k = 1; k <= npz; ++k) {
      LevelParams lp; level_params(h->o, k, npz, lp);
      d_sw(y[0].plane(k), y[1].plane(k), y[2].plane(k), y[3].plane(k), x[4].plane(k), x[5].plane(k), x[6].plane(k),
           x[7].plane(k), x[8].plane(k), y[4].plane(k), y[5].plane(k), y[6].plane(k), y[7].plane(k), y[8].plane(k),
           y[9].plane(k), y[10].plane(k), y[11].plane(k), dt, lp, h->o.dddmp, h->o.d4_bg, h->g, h->bd);
    }
  });
}

// geopk.  in: delp, pt (npz)   out: pe, peln, pk, gz (npz+1), pkz (npz)
void orc_geopk(void* hv, int mode, int cg, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  int nk[2] = {npz, npz}, nko[5] = {npz + 1, npz + 1, npz + 1, npz + 1, npz};
  auto in = mkio(2, in_t, in_p, nk); auto out = mkio(5, out_t, out_p, nko);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    geopk(h->ptop, y[0], y[1], x[0], y[2], y[3], h->phis, x[1], y[4], npz, h->c.akap, h->c.cp_air, cg != 0, 4, h->bd);
  });
}

// p_grad_c.  in: pkc, gz (npz+1), uc, vc (npz)   out: uc, vc
void orc_p_grad_c(void* hv, int mode, double dt2, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  int nk[4] = {npz + 1, npz + 1, npz, npz}, nko[2] = {npz, npz};
  auto in = mkio(4, in_t, in_p, nk); auto out = mkio(2, out_t, out_p, nko);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    y[0] = x[2]; y[1] = x[3];
    p_grad_c(dt2, npz, x[0], x[1], y[0], y[1], h->g, h->bd);
  });
}

// one_grad_p.  in: u, v (npz), pk, gz (npz+1)   out: u, v
void orc_one_grad_p(void* hv, int mode, double dt, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  int nk[4] = {npz, npz, npz + 1, npz + 1}, nko[2] = {npz, npz};
  auto in = mkio(4, in_t, in_p, nk); auto out = mkio(2, out_t, out_p, nko);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    y[0] = x[0]; y[1] = x[1];
    one_grad_p(y[0], y[1], x[2], x[3], dt, h->ptop, h->c.akap, npz, h->g, h->bd);
  });
}

// compute_fv3_pressures (fv_pressure.F90:23-72 / _tlm :74-135 / _bwd :137-203).
// in: delp (npz)   out: pe (npz+1), pk (npz+1), pkz (npz), peln (npz+1)
void orc_fv_pressures(void* hv, int mode, double kappa, double ptop, double** in_t, double** in_p, double** out_t,
                      double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  int nk[1] = {npz}, nko[4] = {npz + 1, npz + 1, npz, npz + 1};
  auto in = mkio(1, in_t, in_p, nk); auto out = mkio(4, out_t, out_p, nko);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    compute_fv3_pressures(h->bd, npz, kappa, ptop, x[0], y[0], y[1], y[2], y[3]);
  });
}

// dyn_core (n_split acoustic steps).  in: u, v, pt, delp   out: u, v, pt, delp, mfx, mfy, cx, cy, pe, peln, pk, pkz
void orc_dyn_core(void* hv, int mode, double bdt, int n_split, double** in_t, double** in_p, double** out_t,
                  double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(4, npz); int nko[12] = {npz, npz, npz, npz, npz, npz, npz, npz, npz + 1, npz + 1, npz + 1, npz};
  auto in = mkio(4, in_t, in_p, nk.data()); auto out = mkio(12, out_t, out_p, nko);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    DynState<T> s; s.init(h->bd, npz, 0);
    s.u = x[0]; s.v = x[1]; s.pt = x[2]; s.delp = x[3];
    dyn_core(s, h->phis, npz, bdt, n_split, h->o, h->c, h->ptop, h->g, h->bd);
    y[0] = s.u; y[1] = s.v; y[2] = s.pt; y[3] = s.delp; y[4] = s.mfx; y[5] = s.mfy; y[6] = s.cx; y[7] = s.cy;
    y[8] = s.pe; y[9] = s.peln; y[10] = s.pk; y[11] = s.pkz;
  });
}

// non-hydrostatic dyn_core (n_split acoustic steps).  in: u, v, pt, delp, w, delz   out: u, v, pt, delp, w, delz (npz), pe, peln, pk, zh (npz+1)
void orc_dyn_core_nh(void* hv, int mode, double bdt, int n_split, double a_imp, double p_fac, double scale_z, double** in_t, double** in_p,
                     double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(6, npz); int nko[10] = {npz, npz, npz, npz, npz, npz, npz + 1, npz + 1, npz + 1, npz + 1};
  auto in = mkio(6, in_t, in_p, nk.data()); auto out = mkio(10, out_t, out_p, nko);
  NhOpts nh; nh.a_imp = a_imp; nh.p_fac = p_fac; nh.scale_z = scale_z;
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    DynState<T> s; s.init(h->bd, npz, 0);
    NhState<T> n; n.zh.init(h->bd, npz + 1);
    s.u = x[0]; s.v = x[1]; s.pt = x[2]; s.delp = x[3]; n.w = x[4]; n.delz = x[5];
    dyn_core_nh(s, n, h->phis, npz, bdt, n_split, h->o, h->c, h->ptop, h->ak, h->bk, nh, h->g, h->bd);
    y[0] = s.u; y[1] = s.v; y[2] = s.pt; y[3] = s.delp; y[4] = n.w; y[5] = n.delz; y[6] = s.pe; y[7] = s.peln; y[8] = s.pk; y[9] = n.zh;
  });
}

// fv_dynamics, non-hydrostatic.  in: u, v, pt (temperature), delp, w, delz, q[nq]   out: the same
void orc_fv_dynamics_nh(void* hv, int mode, int nq, double bdt, int n_split, int k_split, double a_imp, double p_fac, double scale_z, double** in_t,
                        double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(6 + nq, npz);
  auto in = mkio(6 + nq, in_t, in_p, nk.data()); auto out = mkio(6 + nq, out_t, out_p, nk.data());
  NhOpts nh; nh.a_imp = a_imp; nh.p_fac = p_fac; nh.scale_z = scale_z;
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    DynState<T> s; s.init(h->bd, npz, nq);
    NhState<T> n; n.zh.init(h->bd, npz + 1);
    s.u = x[0]; s.v = x[1]; s.pt = x[2]; s.delp = x[3]; n.w = x[4]; n.delz = x[5];
    for (int m = 0; m < nq; ++m) s.q[m] = x[6 + m];
    fv_dynamics_nh(s, n, h->phis, npz, bdt, n_split, k_split, h->o, h->c, h->ptop, h->ak, h->bk, nh, h->g, h->bd);
    y[0] = s.u; y[1] = s.v; y[2] = s.pt; y[3] = s.delp; y[4] = n.w; y[5] = n.delz;
    for (int m = 0; m < nq; ++m) y[6 + m] = s.q[m];
  });
}

// tracer_2d.  in: dp1, mfx, mfy, cx, cy, q[nq]   out: q[nq]
void orc_tracer_2d(void* hv, int mode, int nq, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(5 + nq, npz), nko(nq, npz);
  auto in = mkio(5 + nq, in_t, in_p, nk.data()); auto out = mkio(nq, out_t, out_p, nko.data());
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    std::vector<Arr3<T>> q(nq);
    for (int n = 0; n < nq; ++n) q[n] = x[5 + n];
    tracer_2d(q, x[0], x[1], x[2], x[3], x[4], npz, Hord(h->o.hord_tr, h->o.hord_tr_pert), h->g, h->bd);
    for (int n = 0; n < nq; ++n) y[n] = q[n];
  });
}

// Lagrangian_to_Eulerian.  in: pe, peln, pk (npz+1), pt, delp, u, v, q[nq]
//                          out: pe, peln, pk (npz+1), pkz, pt, delp, u, v, q[nq]
void orc_remap(void* hv, int mode, int nq, int last_step, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(7 + nq, npz), nko(8 + nq, npz);
  nk[0] = nk[1] = nk[2] = npz + 1; nko[0] = nko[1] = nko[2] = npz + 1;
  auto in = mkio(7 + nq, in_t, in_p, nk.data()); auto out = mkio(8 + nq, out_t, out_p, nko.data());
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    DynState<T> s; s.init(h->bd, npz, nq);
    s.pe = x[0]; s.peln = x[1]; s.pk = x[2]; s.pt = x[3]; s.delp = x[4]; s.u = x[5]; s.v = x[6];
    for (int n = 0; n < nq; ++n) s.q[n] = x[7 + n];
    lagrangian_to_eulerian(last_step != 0, s, npz, h->c.akap, h->c.zvir, h->ptop, h->ak, h->bk, h->bd);
    y[0] = s.pe; y[1] = s.peln; y[2] = s.pk; y[3] = s.pkz; y[4] = s.pt; y[5] = s.delp; y[6] = s.u; y[7] = s.v;
    for (int n = 0; n < nq; ++n) y[8 + n] = s.q[n];
  });
}

// fv_dynamics (k_split x (dyn_core, tracer_2d, remap)).  in: u, v, pt(=T), delp, pe, peln, pk (npz+1), pkz, q[nq]
//                                                       out: u, v, pt(=T), delp, q[nq]
void orc_fv_dynamics(void* hv, int mode, int nq, double bdt, int n_split, int k_split, double** in_t, double** in_p,
                     double** out_t, double** out_p) {
  OrcHandle* h = (OrcHandle*)hv; remap_opts() = h->ro; int npz = h->npz;
  std::vector<int> nk(8 + nq, npz), nko(4 + nq, npz);
  nk[4] = nk[5] = nk[6] = npz + 1;
  auto in = mkio(8 + nq, in_t, in_p, nk.data()); auto out = mkio(4 + nq, out_t, out_p, nko.data());
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    DynState<T> s; s.init(h->bd, npz, nq);
    s.u = x[0]; s.v = x[1]; s.pt = x[2]; s.delp = x[3]; s.pe = x[4]; s.peln = x[5]; s.pk = x[6]; s.pkz = x[7];
    for (int n = 0; n < nq; ++n) s.q[n] = x[8 + n];
    fv_dynamics(s, h->phis, npz, bdt, n_split, k_split, h->o, h->c, h->ptop, h->ak, h->bk, h->g, h->bd);
    y[0] = s.u; y[1] = s.v; y[2] = s.pt; y[3] = s.delp;
    for (int n = 0; n < nq; ++n) y[4 + n] = s.q[n];
  });
}

// ---- six-face cube: handles of the six faces (each after orc_set_face) + the exchange tables of cube.py
struct OrcCube { OrcHandle* f[6]; CubeTables X; std::vector<Grid> G; std::vector<Arr2<double>> phis; };
void* orc_cube_create(void** faces, const int* const* tables, const int* nrows) {
  OrcCube* c = new OrcCube;
  for (int t = 0; t < 6; ++t) { c->f[t] = (OrcHandle*)faces[t]; c->G.push_back(c->f[t]->g); c->phis.push_back(c->f[t]->phis); }
  for (int k = 0; k < X_NKIND; ++k) c->X.rows[k].assign(tables[k], tables[k] + 7 * (size_t)nrows[k]);
  return c;
}
void orc_cube_destroy(void* cv) { delete (OrcCube*)cv; }
// every field is [6][nk][pj][pi]; one IO per (field, tile), field-major
static std::vector<IO> mkio6(int n, double** traj, double** pert, const int* nk, size_t np) {
  std::vector<IO> v((size_t)n * 6);
  for (int i = 0; i < n; ++i)
    for (int t = 0; t < 6; ++t) {
      IO& io = v[(size_t)i * 6 + t];
      io.traj = traj[i] + (size_t)t * nk[i] * np; io.pert = pert ? pert[i] + (size_t)t * nk[i] * np : nullptr; io.nk = nk[i];
    }
  return v;
}
// dyn_core on the cube.  in: u, v, pt, delp   out: u, v, pt, delp, mfx, mfy, cx, cy, pe, peln, pk, pkz
void orc_cube_dyn_core(void* cv, int mode, double bdt, int n_split, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcCube* c = (OrcCube*)cv; OrcHandle* h = c->f[0]; remap_opts() = h->ro; const int npz = h->npz;
  const size_t np = (size_t)h->bd.pi() * h->bd.pj();
  std::vector<int> nk(4, npz); int nko[12] = {npz, npz, npz, npz, npz, npz, npz, npz, npz + 1, npz + 1, npz + 1, npz};
  auto in = mkio6(4, in_t, in_p, nk.data(), np); auto out = mkio6(12, out_t, out_p, nko, np);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    std::vector<DynState<T>> S(6);
    for (int t = 0; t < 6; ++t) { S[t].init(h->bd, npz, 0); S[t].u = x[0 * 6 + t]; S[t].v = x[1 * 6 + t]; S[t].pt = x[2 * 6 + t]; S[t].delp = x[3 * 6 + t]; }
    dyn_core_cube(S, c->phis, npz, bdt, n_split, h->o, h->c, h->ptop, c->G, h->bd, c->X);
    for (int t = 0; t < 6; ++t) {
      Arr3<T>* o[12] = {&S[t].u, &S[t].v, &S[t].pt, &S[t].delp, &S[t].mfx, &S[t].mfy, &S[t].cx, &S[t].cy, &S[t].pe, &S[t].peln, &S[t].pk, &S[t].pkz};
      for (int n = 0; n < 12; ++n) y[n * 6 + t] = *o[n];
    }
  });
}
// tracer_2d on the cube (global sub-cycling count, halo exchange between sub-steps).  in: dp1, mfx, mfy, cx, cy, q[nq]   out: q[nq]
void orc_cube_tracer_2d(void* cv, int mode, int nq, double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcCube* c = (OrcCube*)cv; OrcHandle* h = c->f[0]; remap_opts() = h->ro; const int npz = h->npz;
  const size_t np = (size_t)h->bd.pi() * h->bd.pj();
  std::vector<int> nk(5 + nq, npz), nko(nq, npz);
  auto in = mkio6(5 + nq, in_t, in_p, nk.data(), np); auto out = mkio6(nq, out_t, out_p, nko.data(), np);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    std::vector<DynState<T>> S(6); std::vector<Arr3<T>> dp1(6);
    for (int t = 0; t < 6; ++t) {
      S[t].init(h->bd, npz, nq);
      dp1[t] = x[0 * 6 + t]; S[t].mfx = x[1 * 6 + t]; S[t].mfy = x[2 * 6 + t]; S[t].cx = x[3 * 6 + t]; S[t].cy = x[4 * 6 + t];
      for (int n = 0; n < nq; ++n) S[t].q[n] = x[(5 + n) * 6 + t];
    }
    tracer_2d_cube(S, dp1, npz, Hord(h->o.hord_tr, h->o.hord_tr_pert), c->G, h->bd, c->X);
    for (int t = 0; t < 6; ++t) for (int n = 0; n < nq; ++n) y[n * 6 + t] = S[t].q[n];
  });
}

// fv_dynamics on the cube.  in: u, v, pt(=T), delp, pe, peln, pk (npz+1), pkz, q[nq]   out: u, v, pt(=T), delp, q[nq]
void orc_cube_fv_dynamics(void* cv, int mode, int nq, double bdt, int n_split, int k_split, double** in_t, double** in_p,
                          double** out_t, double** out_p) {
  OrcCube* c = (OrcCube*)cv; OrcHandle* h = c->f[0]; remap_opts() = h->ro; const int npz = h->npz;
  const size_t np = (size_t)h->bd.pi() * h->bd.pj();
  std::vector<int> nk(8 + nq, npz), nko(4 + nq, npz);
  nk[4] = nk[5] = nk[6] = npz + 1;
  auto in = mkio6(8 + nq, in_t, in_p, nk.data(), np); auto out = mkio6(4 + nq, out_t, out_p, nko.data(), np);
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    std::vector<DynState<T>> S(6);
    for (int t = 0; t < 6; ++t) {
      DynState<T>& s = S[t]; s.init(h->bd, npz, nq);
      s.u = x[0 * 6 + t]; s.v = x[1 * 6 + t]; s.pt = x[2 * 6 + t]; s.delp = x[3 * 6 + t]; s.pe = x[4 * 6 + t]; s.peln = x[5 * 6 + t];
      s.pk = x[6 * 6 + t]; s.pkz = x[7 * 6 + t];
      for (int n = 0; n < nq; ++n) s.q[n] = x[(8 + n) * 6 + t];
    }
    fv_dynamics_cube(S, c->phis, npz, bdt, n_split, k_split, h->o, h->c, h->ptop, h->ak, h->bk, c->G, h->bd, c->X);
    for (int t = 0; t < 6; ++t) {
      y[0 * 6 + t] = S[t].u; y[1 * 6 + t] = S[t].v; y[2 * 6 + t] = S[t].pt; y[3 * 6 + t] = S[t].delp;
      for (int n = 0; n < nq; ++n) y[(4 + n) * 6 + t] = S[t].q[n];
    }
  });
}

// six faces, non-hydrostatic.  in / out: u, v, pt (temperature), delp, w, delz, q[nq]
void orc_cube_fv_dynamics_nh(void* cv, int mode, int nq, double bdt, int n_split, int k_split, double a_imp, double p_fac, double scale_z,
                             double** in_t, double** in_p, double** out_t, double** out_p) {
  OrcCube* c = (OrcCube*)cv; OrcHandle* h = c->f[0]; remap_opts() = h->ro; const int npz = h->npz;
  const size_t np = (size_t)h->bd.pi() * h->bd.pj();
  std::vector<int> nk(6 + nq, npz);
  auto in = mkio6(6 + nq, in_t, in_p, nk.data(), np); auto out = mkio6(6 + nq, out_t, out_p, nk.data(), np);
  NhOpts nh; nh.a_imp = a_imp; nh.p_fac = p_fac; nh.scale_z = scale_z;
  drive(mode, h->bd, in, out, [&](auto& x, auto& y) {
    using T = typename std::decay<decltype(x[0].p[0].d[0])>::type;
    std::vector<DynState<T>> S(6); std::vector<NhState<T>> N(6);
    for (int t = 0; t < 6; ++t) {
      DynState<T>& s = S[t]; s.init(h->bd, npz, nq); N[t].zh.init(h->bd, npz + 1);
      s.u = x[0 * 6 + t]; s.v = x[1 * 6 + t]; s.pt = x[2 * 6 + t]; s.delp = x[3 * 6 + t]; N[t].w = x[4 * 6 + t]; N[t].delz = x[5 * 6 + t];
      for (int n = 0; n < nq; ++n) s.q[n] = x[(6 + n) * 6 + t];
    }
    fv_dynamics_nh_cube(S, N, c->phis, npz, bdt, n_split, k_split, h->o, h->c, h->ptop, h->ak, h->bk, nh, c->G, h->bd, c->X);
    for (int t = 0; t < 6; ++t) {
      y[0 * 6 + t] = S[t].u; y[1 * 6 + t] = S[t].v; y[2 * 6 + t] = S[t].pt; y[3 * 6 + t] = S[t].delp; y[4 * 6 + t] = N[t].w; y[5 * 6 + t] = N[t].delz;
      for (int n = 0; n < nq; ++n) y[(6 + n) * 6 + t] = S[t].q[n];
    }
  });
}

}  // extern "C"
