// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).  Fortran-style arrays and the grid/flag
// containers the reference routines take (fv_grid_bounds_type NLM/fv_arrays_nlm.F90:601-609,
// fv_grid_type :115-234, fv_flags_type :236-506, fv_flags_pert_type TLM/fv_arrays_tlmadm.F90:37-92).
#pragma once
#include "scalar.hpp"
#include <cassert>
#include <cstring>

namespace orc {

// Tile bounds.  is..ie / js..je is the compute domain (local numbering is=js=1), isd..jed the
// halo'd data domain (ng=3, TOOLS/fv_mp_nlm_mod.F90:67).  Every 2-D array of the oracle is
// allocated on the padded plane (isd:ied+1, jsd:jed+1) — the superset of the A/C/D/corner
// staggerings — so all fields share one index map; this is also the layout the C-ABI uses.
struct Bounds {
  int nx = 0, ny = 0, ng = 3;
  int is, ie, js, je, isd, ied, jsd, jed;
  int npx, npy;          // global face size + 1 (cells 1..npx-1)
  int ioff = 0, joff = 0;  // global index = local index + offset
  // Which cube-face edges this tile touches.  All false = interior MPI rank of the reference
  // (is>5, ie<npx-5): every `is .EQ. 1` / `i .EQ. npx` branch is skipped.
  bool edge_w = false, edge_e = false, edge_s = false, edge_n = false;
  bool sw_corner = false, se_corner = false, nw_corner = false, ne_corner = false;
  int pi() const { return nx + 2 * ng + 1; }
  int pj() const { return ny + 2 * ng + 1; }
  void set(int nx_, int ny_) {
    nx = nx_; ny = ny_; is = 1; ie = nx; js = 1; je = ny;
    isd = is - ng; ied = ie + ng; jsd = js - ng; jed = je + ng;
    // interior-rank placement: far from every face edge
    ioff = 1000; joff = 1000; npx = 1 << 28; npy = 1 << 28;
  }
  bool any_edge() const { return edge_w || edge_e || edge_s || edge_n; }
  // whole cube face: is=1, ie=npx-1, js=1, je=npy-1, all four edges and corners present
  void set_face(int n) {
    set(n, n); ioff = joff = 0; npx = n + 1; npy = n + 1;
    edge_w = edge_e = edge_s = edge_n = true; sw_corner = se_corner = nw_corner = ne_corner = true;
  }
};

template <class T>
struct Arr2 {
  int isd = 0, jsd = 0, pi = 0, pj = 0;
  std::vector<T> d;
  Arr2() {}
  explicit Arr2(const Bounds& b) { init(b); }
  void init(const Bounds& b) { isd = b.isd; jsd = b.jsd; pi = b.pi(); pj = b.pj(); d.assign((size_t)pi * pj, T(0.0)); }
  inline T& operator()(int i, int j) {
#ifdef ORC_BOUNDS_CHECK
    if (i < isd || i >= isd + pi || j < jsd || j >= jsd + pj) { std::fprintf(stderr, "Arr2 OOB %d %d\n", i, j); std::abort(); }
#endif
    return d[(size_t)(j - jsd) * pi + (i - isd)];
  }
  inline const T& operator()(int i, int j) const { return d[(size_t)(j - jsd) * pi + (i - isd)]; }
  void fill(T v) { for (auto& x : d) x = v; }
};

template <class T>
struct Arr3 {
  int nk = 0;
  std::vector<Arr2<T>> p;  // p[k-1]
  Arr3() {}
  Arr3(const Bounds& b, int nk_) { init(b, nk_); }
  void init(const Bounds& b, int nk_) { nk = nk_; p.assign(nk, Arr2<T>(b)); }
  inline T& operator()(int i, int j, int k) { return p[k - 1](i, j); }
  inline const T& operator()(int i, int j, int k) const { return p[k - 1](i, j); }
  Arr2<T>& plane(int k) { return p[k - 1]; }
  const Arr2<T>& plane(int k) const { return p[k - 1]; }
};

// Metric terms (always plain doubles: the grid is not differentiated).
struct Grid {
  Arr2<double> area, rarea, rarea_c, dx, dy, dxa, dya, dxc, dyc, rdx, rdy, rdxa, rdya, rdxc, rdyc;
  Arr2<double> cosa, sina, rsina, cosa_u, cosa_v, cosa_s, sina_u, sina_v, rsin_u, rsin_v, rsin2;
  Arr2<double> f0, fC, del6_u, del6_v, divg_u, divg_v;
  Arr2<double> sin_sg[10], cos_sg[10];  // index 1..9 used (sin_sg(i,j,1:9))
  double da_min = 0, da_min_c = 0;
  // a2b_ord4 cube-edge data: edge_w(j), edge_e(j), edge_s(i), edge_n(i) (fv_grid_type edge_w..edge_n) and the
  // extrap_corner coefficients x1/(x2-x1) for the 3 extrapolations of each corner (order sw, se, ne, nw),
  // precomputed from grid/agrid with great_circle_dist (a2b_edge_tlm.F90:1478-1487).
  std::vector<double> edge_w, edge_e, edge_s, edge_n;
  double ecorner[4][3] = {};
  static constexpr int NMETRIC = 33 + 18;
  void init(const Bounds& b) {
    Arr2<double>* all[] = {&area, &rarea, &rarea_c, &dx, &dy, &dxa, &dya, &dxc, &dyc, &rdx, &rdy, &rdxa, &rdya, &rdxc,
                           &rdyc, &cosa, &sina, &rsina, &cosa_u, &cosa_v, &cosa_s, &sina_u, &sina_v, &rsin_u, &rsin_v,
                           &rsin2, &f0, &fC, &del6_u, &del6_v, &divg_u, &divg_v};
    for (auto* a : all) a->init(b);
    for (int n = 0; n < 10; ++n) { sin_sg[n].init(b); cos_sg[n].init(b); }
  }
};

// Physical constants: two sets are mixed on the hot path: the JEDI set passed as
// arguments (kappa, cp, zvir; DYN/fv3jedi_lm_dynamics_mod.F90:423-424) and FMS constants_mod
// (cp_air, rdgas, grav, radius ...; TLM/dyn_core_tlm.F90:22).  Both are runtime inputs here.
struct Consts {
  double akap, cp, zvir, grav_jedi;        // JEDI-side
  double cp_air, rdgas, rvgas, grav, radius, omega, hlv;  // FMS-side
};

// Per-level resolved scheme/damping choices (the block TLM/dyn_core_tlm.F90:741-921 computes
// for each k before calling D_SW_TLM).  With split_hord/split_damp = .false. the trajectory and
// perturbation values coincide (TLM/fv_control_tlmadm.F90:219-252).
struct LevelParams {
  int hord_mt, hord_vt, hord_tm, hord_dp, hord_tr;
  int nord, nord_v, nord_w, nord_t;
  double d2_divg, damp_vt, damp_w, damp_t, d_con;
  // Vorticity damping (del6_vt_flux) is always evaluated twice by D_SW_TLM: trajectory with
  // (nord_v, damp_vt), perturbation with the *_pert pair (sw_core_tlm.F90:2436-2452, :2502-2530).
  int nord_v_pert; double damp_vt_pert;
  // perturbation advection schemes where they differ from the trajectory's (split_hord): fv_tp_2d runs twice then (tp_core.hpp fv_tp_2d_split)
  int hord_mt_pert, hord_vt_pert, hord_tm_pert, hord_dp_pert;
  // split_damp (fv_arrays_tlmadm.F90:76, sw_core_tlm.F90:1664, :1787, :2341-2369): the perturbation's own divergence damping
  // (nord_k_pert, d2_divg_pert of dyn_core_tlm.F90:839-918 with dddmp_pert, d4_bg_pert) and its own damping pair in the heat
  // transport -- nord_t_pert / damp_t_pert are taken BEFORE the perturbation sponge overrides nord_v_pert / damp_vt_pert
  // (dyn_core_tlm.F90:856-859 vs :913-917)
  bool split_damp = false;
  int nord_pert = 0, nord_t_pert = 0;
  double d2_divg_pert = 0., damp_t_pert = 0., dddmp_pert = 0., d4_bg_pert = 0.;
};

struct Flags {
  bool hydrostatic = true;
  int n_split = 1, k_split = 1, nq = 0;
  double dddmp = 0.2, d4_bg = 0.15, d_ext = 0.0, beta = 0.0, ke_bg = 0.0;
  int a2b_ord = 4, nf_omega = 1;
  bool use_old_omega = false;   // fv_flags_type default .true. is reset? (checked in dyn_core)
  int kord_tm = -17, kord_mt = 17, kord_wz = 17, kord_tr = 17;
  int remap_option = 0;
  double ptop = 1.0;
};

}  // namespace orc
