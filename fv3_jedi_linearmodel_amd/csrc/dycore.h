// fv3lm-hip: the dynamics driver — device state, the acoustic-step program (DYN_CORE_TLM /
// DYN_CORE_FWD+BWD, dyn_core_tlm.F90:467-1346, dyn_core_adm.F90:115/1686) and its execution in
// nonlinear, tangent-linear and adjoint order.
//
// Adjoint strategy (replaces the Tapenade tape, utils/tapenade/adStack.c): the forward nonlinear
// sweep stores only the prognostic state at the start of every acoustic step (4 fields); the
// backward sweep, for each step in reverse, reloads that checkpoint, recomputes the step's
// nonlinear intermediates (every work array is written once per step), zeroes the work adjoints
// with one memset and runs the stage list backwards in gather form.
#pragma once
#include "../../include/fv3lm.h"
#include "stages.h"
#include "column.h"
#include "exchange.h"
#include "nh.h"
#include "nh_ad.h"
#include "tpfused.h"
#include "tpad.h"
#include "tp2.h"
#include "dampt.h"
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

namespace fv3 {

// ------------------------------------------------------------------ device memory helpers
#ifdef FV3LM_HOST_EMUL
inline void* dev_alloc(size_t n) { void* p = std::calloc(n ? n : 1, 1); return p; }
inline void dev_free(void* p) { std::free(p); }
inline void dev_zero(Exec&, void* p, size_t n) { std::memset(p, 0, n); }
inline void dev_copy(Exec&, void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
inline void h2d(Exec&, void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
inline void d2h(Exec&, void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
inline void dev_sync(Exec&) {}
#else
// no abort(): the failure is recorded (exec.h sticky_error) and surfaces as the status of the C-ABI call
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { char b_[320]; std::snprintf(b_, sizeof b_, "HIP error '%s' at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); set_sticky(b_); } } while (0)
inline void* dev_alloc(size_t n) {
  void* p = nullptr;
  if (!sticky_error().empty()) return nullptr;                // after a failure (e.g. out of memory at C384) nothing more is allocated
  const hipError_t e = hipMalloc(&p, n ? n : 8);
  if (e != hipSuccess) { char b_[200]; std::snprintf(b_, sizeof b_, "hipMalloc of %zu bytes failed: %s", n, hipGetErrorString(e)); set_sticky(b_); (void)hipGetLastError(); return nullptr; }
  HIPCHK(hipMemset(p, 0, n ? n : 8));
  return p;
}
inline void dev_free(void* p) { if (p) (void)hipFree(p); }
// clearing a field: 16-byte stores from a grid that covers the chip (measured: the runtime's fill kernel reaches 2.3 TB/s on these sizes)
__global__ void __launch_bounds__(256) k_zero16(double2* p, size_t n2) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) p[i] = make_double2(0., 0.);
}
inline void dev_zero(Exec& ex, void* p, size_t n) {
  if (n >= ((size_t)1 << 20) && ((uintptr_t)p & 7) == 0) {
    char* q = (char*)p;
    if ((uintptr_t)q & 15) { HIPCHK(hipMemsetAsync(q, 0, 8, ex.stream)); q += 8; n -= 8; }      // fields are 8-byte aligned; the kernel wants 16
    const size_t n2 = n / 16, want = (n2 + 255) / 256;
    hipLaunchKernelGGL(k_zero16, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, ex.stream, (double2*)q, n2);
    if (n & 15) HIPCHK(hipMemsetAsync(q + n2 * 16, 0, n & 15, ex.stream));
    return;
  }
  HIPCHK(hipMemsetAsync(p, 0, n, ex.stream));
}
inline void dev_copy(Exec& ex, void* d, const void* s, size_t n) { HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, ex.stream)); }
inline void h2d(Exec& ex, void* d, const void* s, size_t n) { HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, ex.stream)); HIPCHK(hipStreamSynchronize(ex.stream)); }
inline void d2h(Exec& ex, void* d, const void* s, size_t n) { HIPCHK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, ex.stream)); HIPCHK(hipStreamSynchronize(ex.stream)); }
inline void dev_sync(Exec& ex) { HIPCHK(hipStreamSynchronize(ex.stream)); }
#endif

// One arena = one traj buffer + one pert buffer carved into fields; the pert side of the work arena
// is cleared with a single memset per backward acoustic step.
struct Arena {
  double* t = nullptr; double* p = nullptr; size_t cap = 0, used = 0;
  // sizing pass: hands out distinct non-null addresses that are never dereferenced (the program built on them is thrown away)
  void measure() { cap = (size_t)1 << 56; t = p = (double*)(uintptr_t)4096; used = 0; }
  void init(size_t ndoubles) { cap = ndoubles; t = (double*)dev_alloc(cap * 8); p = (double*)dev_alloc(cap * 8); used = 0; }
  void destroy() { dev_free(t); dev_free(p); t = p = nullptr; }
  Fld take(size_t n, int nk) {
    if (used + n > cap) { set_sticky("internal sizing error: field arena overflow"); Fld f; f.t = t; f.p = p; f.nk = nk; return f; }   // create() fails; nothing runs on it
    Fld f; f.t = t + used; f.p = p + used; f.nk = nk; used += n; return f;
  }
};

typedef fv3lm_options Options;   // include/fv3lm.h

// Per-level selection, dyn_core_tlm.F90:741-921: both the schemes the tangent / adjoint is taken of and the trajectory's.
inline bool resolve_level(const Options& o, int k, int npz, LevelParams& lp) {
  int hord_m = o.hord_mt, hord_t = o.hord_tm, hord_v = o.hord_vt, hord_p = o.hord_dp;
  int nord_k = o.nord, nord_v = (2 > o.nord) ? o.nord : 2;
  double d2_divg = (0.20 > o.d2_bg) ? o.d2_bg : 0.20;
  double damp_vt = o.do_vort_damp ? o.vtdm4 : 0.;
  int nord_w = nord_v, nord_t = nord_v;
  double damp_w = damp_vt, damp_t = damp_vt, d_con_k = o.d_con;
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
    const int max1 = (2 < o.n_sponge - 1) ? o.n_sponge - 1 : 2;
    if (k == max1 && o.d2_bg_k2 > 0.01) {
      nord_k = 0; d2_divg = (o.d2_bg < o.d2_bg_k2) ? o.d2_bg_k2 : o.d2_bg;
      nord_w = 0; damp_w = d2_divg;
      if (o.do_vort_damp) { nord_v = 0; damp_vt = 0.5 * d2_divg; }
      d_con_k = 0.;
    } else {
      const int max2 = (3 < o.n_sponge) ? o.n_sponge : 3;
      if (k == max2 && o.d2_bg_k2 > 0.05) {
        nord_k = 0; d2_divg = (o.d2_bg < 0.2 * o.d2_bg_k2) ? 0.2 * o.d2_bg_k2 : o.d2_bg;
        nord_w = 0; damp_w = d2_divg; d_con_k = 0.;
      }
    }
  }
  int hmp = o.hord_mt_pert, htp = o.hord_tm_pert, hvp = o.hord_vt_pert, hpp = o.hord_dp_pert;
  int nord_k_pert = o.nord_pert;
  int nord_v_pert = (2 > o.nord_pert) ? o.nord_pert : 2;
  double damp_vt_pert = o.do_vort_damp_pert ? o.vtdm4_pert : 0.;
  double d2_divg_pert = (0.20 > o.d2_bg_pert) ? o.d2_bg_pert : 0.20;
  const int nord_t_pert = nord_v_pert; const double damp_t_pert = damp_vt_pert;      // :856-859: before the sponge rules below touch nord_v_pert / damp_vt_pert
  if (k <= o.n_sponge_pert) {
    nord_k_pert = 0;
    if (k <= o.n_sponge_pert - 1) {
      if (o.hord_ks_traj) { hord_m = o.hord_mt_ks_traj; hord_t = o.hord_tm_ks_traj; hord_v = o.hord_vt_ks_traj; hord_p = o.hord_dp_ks_traj; }
      if (o.hord_ks_pert) { hmp = o.hord_mt_ks_pert; htp = o.hord_tm_ks_pert; hvp = o.hord_vt_ks_pert; hpp = o.hord_dp_ks_pert; }
    }
    const double kfac = (k == 1) ? o.d2_bg_k1_pert : (k == 2) ? o.d2_bg_k2_pert : o.d2_bg_ks_pert;
    double d2p;
    if (0.01 < o.d2_bg_pert) d2p = (o.d2_bg_pert < kfac) ? kfac : o.d2_bg_pert;
    else if (0.01 < kfac) d2p = kfac;
    else d2p = 0.01;
    d2_divg_pert = d2p;
    if (o.do_vort_damp_pert) { nord_v_pert = 0; damp_vt_pert = 0.5 * d2p; }
  }
  lp.hord_mt = hmp; lp.hord_vt = hvp; lp.hord_tm = htp; lp.hord_dp = hpp; lp.hord_tr = o.hord_tr_pert;
  lp.hord_tm_g = o.hord_tm_pert;
  lp.hord_mt_t = hord_m; lp.hord_vt_t = hord_v; lp.hord_tm_t = hord_t; lp.hord_dp_t = hord_p; lp.hord_tr_t = o.hord_tr; lp.hord_tm_g_t = o.hord_tm;
  lp.nord = nord_k; lp.nord_v = nord_v; lp.nord_w = nord_w; lp.nord_t = nord_t; lp.nord_v_pert = nord_v_pert;
  lp.d2_divg = d2_divg; lp.damp_vt = damp_vt; lp.damp_w = damp_w; lp.damp_t = damp_t; lp.d_con = d_con_k;
  lp.damp_vt_pert = damp_vt_pert;
  lp.split_damp = o.split_damp ? 1 : 0; lp.nord_t_pert = nord_t_pert; lp.damp_t_pert = damp_t_pert;
  lp.dddmp = o.dddmp; lp.d4_bg = o.d4_bg;
  if (o.split_damp) { lp.nord_p = nord_k_pert; lp.d2_divg_p = d2_divg_pert; lp.dddmp_p = o.dddmp_pert; lp.d4_bg_p = o.d4_bg_pert; }
  else { lp.nord_p = nord_k; lp.d2_divg_p = d2_divg; lp.dddmp_p = o.dddmp; lp.d4_bg_p = o.d4_bg; }
  return true;
}

struct Op {
  std::string group;
  std::function<void(Exec&, int)> fn;   // mode-aware
  bool accum = false;                   // flux-capacitor update: skipped in the adjoint's trajectory recompute
  unsigned modes = 7u;                  // bit m: the op takes part in mode m (fused forward kernels: nonlinear + tangent; their staged form: adjoint only)
  // adjoint bookkeeping for Dycore::plan_adjoint: the adjoint buffers (.p) the op's adjoint accumulates into (its forward
  // inputs) and the ones it reads (its forward outputs); stage ops can be switched to store-instead-of-accumulate per input
  std::vector<double*> ad_in, ad_out;
  std::vector<int> ad_in_slot;                 // stage input index of each ad_in entry
  std::function<void(unsigned, Rect)> set_wmask;     // stage ops only: write-mode inputs and the rectangle the stores must cover
  std::vector<Rect> ad_out_rect;               // where the op writes each ad_out in the forward direction (empty: unknown -> whole plane)
  std::vector<double*> ad_store;               // adjoints the op's adjoint STORES over everything read later (hand-written fused adjoints)
  std::string name;                            // stage name (diagnostics)
  bool global = false;                         // an exchange: runs once over all resident tiles, between the classes' segments
  Op() = default;
  Op(std::string g_, std::function<void(Exec&, int)> f_, bool acc_ = false) : group(std::move(g_)), fn(std::move(f_)), accum(acc_) {}
};
typedef std::vector<Op> Program;
// One program per tile class (Dycore::classes): the builders run once per class, in that class's global face indices, and leave the same
// sequence of exchanges (Op::global) in each.  Converts to the program of the class being built / run.
struct Progs {
  std::vector<Program> c; const int* cur = nullptr;
  Program& now() { return c[(size_t)(*cur < 0 ? 0 : *cur)]; }
  operator Program&() { return now(); }
  void clear() { now().clear(); }
};

struct Dycore {
  Geom g{}; Options opt; Exec ex; Ctx ctx{};
  double bdt = 0; int n_split = 1, k_split = 1, nq = 0;
  std::vector<double*> metric_dev;
  LevelParams* lev_dev = nullptr;
  std::vector<LevelParams> lev_host;
  double* hs_dev = nullptr;
  Arena state, work;
  std::map<std::string, Fld> F;       // field registry (debug/test access + driver)
  // Tile classes: runs of resident tiles with the same window of their face (one class when every tile is a whole face).
  struct TileClass { int t0, n, i0, j0; };
  std::vector<TileClass> classes;
  int ntile_all = 0, cur_cls = -1; Metrics m_all{};
  bool subtile = false;               // some resident tile is a window of its face (fv_control_nlm.F90:556 layout > 1 x 1)
  void set_class(int c);              // c = -1: all tiles, no shift (exchanges, whole-field copies)
  template <class F_> void each_class(const F_& fn) { for (int c = 0; c < (int)classes.size(); ++c) { set_class(c); fn(); } set_class(-1); }
  Progs acoustic;                     // one acoustic step
  double* ckpt = nullptr;             // [n_split*k_split][step state: u v delp pt (+ w delz zh)]
  std::vector<const char*> st_in, st_out;   // per-step prognostic fields and the work buffers the step writes them to
  size_t ck_stride = 0;
  bool nh = false;                    // non-hydrostatic
  double* nh_ws = nullptr; TapeMem nh_tape;   // column workspace and reverse-mode tape of the column solvers (nh.h)
  NhColArgs nh_args(double dt_) const;
  void add_col(Program& P, const char* group, int kind, const NhColArgs& a, Rect r, Rect skip = Rect{1, 0, 1, 0}, int when = 0);
  bool nh_overflow();
  // Trajectory slots: as many acoustic steps as free HBM allows keep the trajectory of ALL their intermediates (one copy of
  // the work arena's trajectory side + pe, peln, pk, pkz each), written by the forward sweep of step_nl; the backward sweep
  // then skips the nonlinear recompute of those steps.  Steps without a slot are recomputed from their 4-field checkpoint.
  std::vector<double*> traj_slot, traj_slot_p;   // work.t copy, (pe, peln, pk, pkz).t copy
  void init_traj_slots();
  bool has_slot(int a) const { return a < (int)traj_slot.size(); }
  int ck_base = 0;                    // first acoustic-step slot of the current k_split iteration
  size_t n3 = 0, n3p = 0;             // doubles per npz / npz+1 field
  std::string err;
  ExTable xt[H_NKIND];                // cube-face exchange tables (face mode), set by set_exchange
  ExRemote xr[H_NKIND];               // ... and the part of each exchange that crosses ranks (set_exchange_remote)
  double *edge_dev = nullptr, *ecorner_dev = nullptr;
  bool last_acoustic = false;         // the acoustic step being run is the last of its dyn_core call
  bool remap_last = false;            // the vertical remap being run is the one of the last k_split step
  bool halo_missing = false;          // an exchange was needed before its table was set

  Fld& f(const char* n) {
    auto it = F.find(n);
    if (it == F.end()) { set_sticky(std::string("internal error: unknown field ") + n); static Fld none; none = Fld{}; return none; }
    return it->second;
  }
  bool reuse_fields = false;          // building the program of a class after the first: the fields exist
  Fld S(const char* n, int nk) {
    if (reuse_fields) { auto it = F.find(n); if (it != F.end()) return it->second; }
    (nk == g.npz ? nS3 : nS3p)++; Fld x = state.take((size_t)ntile_all * nk * g.plane, nk); F[n] = x; return x; }
  int nW3 = 0, nW3p = 0, nS3 = 0, nS3p = 0;
  Fld W(const char* n, int nk) {
    if (reuse_fields) { auto it = F.find(n); if (it != F.end()) return it->second; }
    (nk == g.npz ? nW3 : nW3p)++; Fld x = work.take((size_t)ntile_all * nk * g.plane, nk); F[n] = x; return x; }

  Rect R(int i0, int i1, int j0, int j1) const { return Rect{i0, i1, j0, j1}; }
  static Rect empty_in(const Rect& r) { return Rect{r.i0 + 1, r.i0, r.j0 + 1, r.j0}; }   // contains nothing, leaves the union with r alone
  static Rect isect(const Rect& a, const Rect& b) { return Rect{std::max(a.i0, b.i0), std::min(a.i1, b.i1), std::max(a.j0, b.j0), std::min(a.j1, b.j1)}; }
  static bool is_empty(const Rect& r) { return r.i0 > r.i1 || r.j0 > r.j1; }
  // A stage whose body has face-edge branches (stages.h Edged): on a face, one bulk launch over the outputs more than W points
  // away from every edge with the branches compiled out, and up to four strip launches (south, north, west, east) with them.
  // Inputs only the edge formulas read are not handed to the bulk instance.
  template <class D>
  void add_face(Program& P, const char* grp, const D& s, int W) {
    if (!g.face) { add(P, grp, Edged<D, false>(s)); return; }
    const int lo = 1 + W, hi = g.nx - W, BIG = 1 << 20;       // global face indices: a tile away from every cube edge gets the bulk launch only
    const Rect regs[5] = {{lo, hi, lo, hi}, {-BIG, BIG, -BIG, lo - 1}, {-BIG, BIG, hi + 1, BIG}, {-BIG, lo - 1, lo, hi}, {hi + 1, BIG, lo, hi}};
    // program order: strips first, bulk last -- the backward sweep then runs the bulk launch first, which is the one that can
    // store the input adjoints instead of accumulating them (plan_adjoint); the five launches are independent of each other
    std::vector<Edged<D, true>> strips; bool have_bulk = false; Edged<D, false> bulk;
    for (int r = 0; r < 5; ++r) {
      D t = s; Rect anchor{1, 0, 1, 0}; bool any = false;
      for (int n = 0; n < D::NOUT; ++n) {
        t.orect[n] = isect(s.orect[n], regs[r]);
        if (!is_empty(t.orect[n]) && !any) { anchor = t.orect[n]; any = true; }
      }
      if (!any) continue;
      for (int n = 0; n < D::NOUT; ++n) if (is_empty(t.orect[n])) t.orect[n] = empty_in(anchor);
      if (r == 0) {
        for (int m = 0; m < D::NIN; ++m) if ((edge_only_inputs((const D*)nullptr) >> m) & 1u) { const int nk = t.in[m].nk; t.in[m] = Fld{}; t.in[m].nk = nk; }
        bulk = Edged<D, false>(t); have_bulk = true;
      } else strips.push_back(Edged<D, true>(t));
    }
    const size_t n0 = P.size();
    add_multi(P, grp, strips);
    const size_t n1 = P.size();
    if (have_bulk) add(P, grp, bulk);
#ifndef FV3LM_HOST_EMUL
    // Forward modes: the strips go to the strip stream and run beside the bulk launch (they read the same inputs and write disjoint
    // regions of the outputs); the main stream waits for them after the bulk launch.  A strip launch is 2 % of the points but
    // latency-bound (40-140 us): in line it cost 45 ms per step.  The adjoint keeps the order bulk, then strips (the strips accumulate
    // into what the bulk launch stores).
    if (have_bulk && n1 == n0 + 1 && P.size() == n1 + 1) {
      auto fs = P[n0].fn; auto fb = P[n1].fn;
      P[n0].fn = [fs](Exec& e, int mode) {
        if (mode == MODE_AD || !e.sstream) { fs(e, mode); return; }
        HIPCHK(hipEventRecord(e.ev_s0, e.stream)); HIPCHK(hipStreamWaitEvent(e.sstream, e.ev_s0, 0));
        hipStream_t main = e.stream; e.stream = e.sstream;
        fs(e, mode);
        e.stream = main;
        HIPCHK(hipEventRecord(e.ev_s1, e.sstream)); e.side_pending = true;
      };
      P[n1].fn = [fb](Exec& e, int mode) {
        fb(e, mode);
        if (e.side_pending) { HIPCHK(hipStreamWaitEvent(e.stream, e.ev_s1, 0)); e.side_pending = false; }
      };
    }
#endif
  }

  template <class St>
  static void declare(Op& op, const St& s) {
    for (int m = 0; m < St::NIN; ++m) if (St::wants(m) && s.in[m].p) { op.ad_in.push_back(s.in[m].p); op.ad_in_slot.push_back(m); }
    for (int n = 0; n < St::NOUT; ++n) { op.ad_out.push_back(s.out[n].p); op.ad_out_rect.push_back(s.orect[n]); }
  }
  template <class St>
  void add(Program& P, const char* group, const St& s) {
    Ctx* cp = &ctx;
    auto sp = std::make_shared<St>(s);
    Op op{group, [sp, cp](Exec& e, int mode) { run(e, mode, *sp, *cp); }};
    declare(op, s);
    op.set_wmask = [sp](unsigned w, Rect r) { sp->wmask = w; sp->wrect = r; };
    op.name = St::name();
    P.push_back(op);
  }
  // several instances of one stage type (the edge strips of a face) as one launch (exec.h run_multi)
  template <class St>
  void add_multi(Program& P, const char* group, const std::vector<St>& v) {
    if (v.empty()) return;
    if (v.size() == 1) { add(P, group, v[0]); return; }
    Ctx* cp = &ctx;
    Op op{group, [v, cp](Exec& e, int mode) { run_multi(e, mode, v.data(), (int)v.size(), *cp); }};
    for (const St& s : v) declare(op, s);
    op.ad_in_slot.clear();            // strips only ever accumulate
    op.name = St::name();
    P.push_back(op);
  }
  // Halo update of one field or of a staggered vector pair.  Single tile: doubly-periodic wrap; cube faces:
  // table-driven exchange (exchange.h).
  void halo(int mode, int kind, const Fld& f0, const Fld& f1 = Fld{}) {
    if (!g.face) { run_halo(ex, mode, g, f0); if (f1.t) run_halo(ex, mode, g, f1); return; }
    if (xt[kind].n == 0 && !xr[kind].active) { halo_missing = true; return; }
    // forward: local rows and remote rows write disjoint halo elements; adjoint: both add into the sources, one after the other
    run_exchange(ex, mode, g, xt[kind], ex.sh(f0), ex.sh(f1));
    if (!run_exchange_remote(ex, mode, g, xr[kind], ex.sh(f0), ex.sh(f1), err)) halo_missing = true;
  }
  // when: 0 every acoustic step, 1 all but the last, 2 the last only (face mode; the periodic wrap does
  // both jobs at once and ignores it)
  void add_halo(Program& P, const char* group, int kind, Fld f0, Fld f1 = Fld{}, int when = 0) {
    Dycore* self = this;
    Op op{group, [self, kind, f0, f1, when](Exec&, int mode) {
      if (self->g.face && ((when == 1 && self->last_acoustic) || (when == 2 && !self->last_acoustic))) return;
      if (!self->g.face && when == 2) return;
      self->halo(mode, kind, f0, f1);
    }};
    op.ad_in = {f0.p, f1.p};          // the adjoint exchange moves halo adjoints onto their source elements: both must hold defined values
    op.name = "halo"; op.global = true;
    P.push_back(op);
  }
  // Exchange beside compute.  A halo exchange whose field no launch touches between its producer and its first consumer is split
  // into a start (right after the last launch that touches the field) and a join (where the exchange used to stand): pack ->
  // ncclSend/ncclRecv -> unpack (or the local table kernel) run on the exchange stream while the main stream goes on; the adjoint
  // sweep meets the two in the opposite order and does the same.  Between start and join the field is untouched on the main
  // stream, so both directions are race-free by construction.  FV3LM_NO_ASYNC_HALO=1 keeps every exchange in line.
  void halo_window(int mode, bool start, int win, int kind, const Fld& f0, const Fld& f1) {
#ifndef FV3LM_HOST_EMUL
    if (ex.xstream) {
      if (start) {
        HIPCHK(hipEventRecord(ex.ev_a[win], ex.stream)); HIPCHK(hipStreamWaitEvent(ex.xstream, ex.ev_a[win], 0));
        hipStream_t main = ex.stream; ex.stream = ex.xstream;
        halo(mode, kind, f0, f1);
        ex.stream = main;
        HIPCHK(hipEventRecord(ex.ev_b[win], ex.xstream));
      } else HIPCHK(hipStreamWaitEvent(ex.stream, ex.ev_b[win], 0));
      return;
    }
#endif
    (void)win;
    if (!start) halo(mode, kind, f0, f1);       // no second stream (host emulation): the exchange stays where it was
  }
  // replaces add_halo(P, group, kind, f0, f1) at the point where the exchanged halo is first needed
  void add_halo_async(Program& P, const char* group, int kind, Fld f0, Fld f1 = Fld{}) {
    const char* env = std::getenv("FV3LM_NO_ASYNC_HALO");
    if ((env && env[0] == '1') || n_win >= 4) { add_halo(P, group, kind, f0, f1); return; }
    size_t at = P.size();        // start goes right after the last op that reads or writes the field
    while (at > 0) {
      const Op& o = P[at - 1]; bool touch = false;
      for (double* q_ : o.ad_in) if (q_ && (q_ == f0.p || q_ == f1.p)) touch = true;
      for (double* q_ : o.ad_out) if (q_ && (q_ == f0.p || q_ == f1.p)) touch = true;
      if (touch || o.ad_in.empty()) break;       // ops that declare nothing (none today) are taken to touch everything
      --at;
    }
    const int win = n_win++;
    Dycore* self = this;
    Op a{group, [self, kind, f0, f1, win](Exec&, int mode) { self->halo_window(mode, mode != MODE_AD, win, kind, f0, f1); }};
    Op b{group, [self, kind, f0, f1, win](Exec&, int mode) { self->halo_window(mode, mode == MODE_AD, win, kind, f0, f1); }};
    a.ad_in = b.ad_in = {f0.p, f1.p}; a.name = "halo_start"; b.name = "halo_join"; a.global = b.global = true;
    P.insert(P.begin() + at, a);
    P.push_back(b);
  }
  int n_win = 0;
  bool set_face_data(const double* edge, const double* ecorner);
  bool set_exchange(int kind, const int* rows, int n);
  bool set_exchange_remote(int kind, int npeers, const int* peers, const int* nsend, const int* send_rows, const int* nrecv, const int* recv_rows);
  void add_accum(Program& P, const char* group, Fld acc, Fld x, Rect r) {
    Geom gg = g;
    Op op{group, [acc, x, r, gg](Exec& e, int mode) { run_accum(e, mode, gg, acc, x, r); }, true};
    op.ad_in = {x.p}; op.ad_out = {acc.p}; op.ad_out_rect = {r};
    P.push_back(op);
  }
  // Backward sweep without clearing the work adjoints.  Walking the program in reverse (= the order of the backward sweep), the
  // first op that touches the adjoint of a work array decides: a stage launch stores it (bit in its wmask; exec.h writes the whole
  // padded plane, zeros outside the stage's reach), anything else (column operators, exchanges, strips) needs it cleared first.  Work
  // arrays whose adjoint is read by their producer but written by nobody (outputs without a consumer) are cleared as well.
  // Returns the ranges of A.p to clear before every backward pass over P.
  // preset: work arrays whose adjoint the driver sets before the backward pass (the step outputs u_o .. receive the incoming adjoint).
  std::vector<std::pair<double*, size_t>> plan_adjoint(Program& P, const Arena& A, const std::vector<double*>& preset = {}) {
    std::map<double*, size_t> size;
    for (auto& kv : F) if (kv.second.p >= A.p && kv.second.p < A.p + A.cap) size[kv.second.p] = (size_t)ntile_all * kv.second.nk * g.plane;
    auto inA = [&](double* q_) { return q_ && q_ >= A.p && q_ < A.p + A.cap; };
    std::set<double*> written(preset.begin(), preset.end()), zero;
    // A stored adjoint must be defined wherever somebody reads it later in the sweep: where its producer wrote the field (the
    // producer's adjoint reads the whole of that), and on the whole padded plane if an exchange moves its halo adjoints around.
    const Rect whole{g.isd(), g.ied() + 1, g.jsd(), g.jed() + 1};
    std::map<double*, Rect> prod; std::set<double*> everywhere;
    for (const Op& o : P) {
      if (!((o.modes >> MODE_AD) & 1u)) continue;
      if (o.name == "halo" || o.name == "halo_start" || o.name == "halo_join") for (double* q_ : o.ad_in) if (q_) everywhere.insert(q_);
      for (size_t n = 0; n < o.ad_out.size(); ++n) {
        double* q_ = o.ad_out[n];
        if (!q_) continue;
        if (n >= o.ad_out_rect.size()) { everywhere.insert(q_); continue; }
        const Rect& r = o.ad_out_rect[n];
        if (r.i0 > r.i1 || r.j0 > r.j1) continue;
        auto f_ = prod.find(q_);
        if (f_ == prod.end()) prod[q_] = r;
        else { Rect& u = f_->second; u.i0 = std::min(u.i0, r.i0); u.i1 = std::max(u.i1, r.i1); u.j0 = std::min(u.j0, r.j0); u.j1 = std::max(u.j1, r.j1); }
      }
    }
    const bool off = std::getenv("FV3LM_NO_AD_WRITE_MODE") != nullptr;     // debugging aid: every work adjoint cleared, every launch accumulates
    for (auto it = P.rbegin(); it != P.rend(); ++it) {
      if (!((it->modes >> MODE_AD) & 1u)) continue;
      unsigned w = 0; Rect wr{1, 0, 1, 0};
      auto grow = [&](const Rect& r) { if (wr.i0 > wr.i1) wr = r; else { wr.i0 = std::min(wr.i0, r.i0); wr.i1 = std::max(wr.i1, r.i1); wr.j0 = std::min(wr.j0, r.j0); wr.j1 = std::max(wr.j1, r.j1); } };
      for (double* q_ : it->ad_out) if (inA(q_) && !written.count(q_)) zero.insert(q_);
      for (double* q_ : it->ad_store) if (inA(q_)) written.insert(q_);
      for (size_t n = 0; n < it->ad_in.size(); ++n) {
        double* q_ = it->ad_in[n];
        if (!inA(q_) || written.count(q_)) continue;
        const char* skip = std::getenv("FV3LM_AD_WRITE_SKIP");        // debugging aid: stage names (comma separated) kept in accumulate mode
        const bool skipped = skip && ("," + std::string(skip) + ",").find("," + it->name + ",") != std::string::npos;
        if (std::getenv("FV3LM_AD_WRITE_LIST") && it->set_wmask && !skipped && !zero.count(q_)) std::fprintf(stderr, "write-mode %s input %d\n", it->name.c_str(), it->ad_in_slot[n]);
        if (!off && !skipped && it->set_wmask && n < it->ad_in_slot.size() && !zero.count(q_)) {
          w |= 1u << it->ad_in_slot[n];
          grow((everywhere.count(q_) || !prod.count(q_)) ? whole : prod[q_]);
        }
        else zero.insert(q_);
        written.insert(q_);
      }
      if (it->set_wmask) it->set_wmask(w, wr);
    }
    std::vector<std::pair<double*, size_t>> out;
    for (double* q_ : zero) {      // std::set iterates in address order: merge neighbours
      const size_t n = size.count(q_) ? size[q_] : 0;
      if (!n) { set_sticky("internal error: adjoint plan met an unregistered work array"); continue; }
      if (!out.empty() && out.back().first + out.back().second == q_) out.back().second += n; else out.push_back({q_, n});
    }
    if (std::getenv("FV3LM_AD_WRITE_LIST")) { size_t tot = 0; for (auto& z_ : out) tot += z_.second; std::fprintf(stderr, "plan_adjoint: %zu work adjoints cleared per pass (%zu ranges, %.1f field-equivalents of %zu in the arena)\n", zero.size(), out.size(), double(tot) / double(n3 ? n3 : 1), size.size()); }
    return out;
  }
  std::vector<std::pair<double*, size_t>> acoustic_zero;     // plan_adjoint(acoustic, work)

  // fv_tp_2d as a stage sequence (tp_core_tlm.F90:83-236): q -> fx, fy.  mx/my = xfx/yfx or mass fluxes.
  void build_tp(Program& P, const char* grp, const std::string& pre, Fld q, Fld crx, Fld cry, Fld xfx, Fld yfx, Fld rax,
                Fld ray, Fld mx, Fld my, Fld mass, int hsel, int dsel, bool use_mass, Fld fx, Fld fy, int nk = 0, const Fld* acc4 = nullptr) {
    const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je(), isd = g.isd(), ied = g.ied(), jsd = g.jsd(), jed = g.jed(), npz = nk ? nk : g.npz;
    // nonlinear and tangent modes: the whole routine as one LDS-tiled launch (tpfused.h); the staged launches below then serve the
    // adjoint only (FV3LM_TP_FUSED=0 runs them in every mode -- the two forms agree bit for bit)
    const char* fenv = std::getenv("FV3LM_TP_FUSED");
    const bool fused = !(fenv && fenv[0] == '0');
    // The outer fluxes fxo, fyo are trajectory arrays of their own only while something reads them: the staged flux assembly.  With the
    // hand-written outer adjoint (which re-evaluates them from q_i, q_j in passing) they are neither stored by the nonlinear launch nor
    // part of a trajectory slot; the names then alias fx2 / fy2 for the staged ops that never run.
    const char* aenv = std::getenv("FV3LM_TP_AD_FUSED");
    // 2: the whole adjoint as one launch (tpad.h; faces, sub-face tiles and the periodic tile); 1: outer half fused, inner half staged; 0: staged
    const int adf = !fused ? 0 : aenv ? (aenv[0] == '0' ? 0 : aenv[0] == '1' ? 1 : 2) : 2;
    const bool own_fo = adf == 0, own_mid = adf < 2;
    // the inner fluxes and the intermediate fields are trajectory arrays of their own only while a staged adjoint launch reads them; the
    // fused adjoint recomputes them in LDS, so neither the nonlinear launch stores them nor a trajectory slot holds them -- the names then
    // alias fx / fy for the staged ops that never run
    Fld fy2 = own_mid ? W((pre + "_fy2").c_str(), npz) : fy, q_i = own_mid ? W((pre + "_qi").c_str(), npz) : fx;
    Fld fx2 = own_mid ? W((pre + "_fx2").c_str(), npz) : fx, q_j = own_mid ? W((pre + "_qj").c_str(), npz) : fy;
    Fld fxo = own_fo ? W((pre + "_fxo").c_str(), npz) : fx2, fyo = own_fo ? W((pre + "_fyo").c_str(), npz) : fy2;
    const size_t first_op = P.size();
    // the four 1-D PPM sweeps; on a face each is a bulk launch (4-point edge values everywhere) plus two strips three flux
    // points wide next to the face edges (one-sided edge values, corner views of the inner sweeps)
    auto ppm_y = [&](Fld qq, Fld out, Rect r, int cdir) {
      TpPpmY_<false> a; a.in[0] = qq; a.in[1] = cry; a.out[0] = out; a.k1 = npz; a.hsel = hsel;
      if (!g.face) { a.orect[0] = r; add(P, grp, a); return; }
      const int npy = g.ny + 1;          // global: the three flux points next to a cube edge take the one-sided edge values
      a.orect[0] = isect(r, R(r.i0, r.i1, 4, npy - 3));
      TpPpmY_<true> e; e.in[0] = qq; e.in[1] = cry; e.out[0] = out; e.k1 = npz; e.hsel = hsel; e.cdir = cdir;
      std::vector<TpPpmY_<true>> ev;
      for (const Rect& er : {isect(r, R(r.i0, r.i1, r.j0, 3)), isect(r, R(r.i0, r.i1, npy - 2, r.j1))}) if (!is_empty(er)) { e.orect[0] = er; ev.push_back(e); }
      add_multi(P, grp, ev);
      if (!is_empty(a.orect[0])) add(P, grp, a);          // bulk last: first in the backward sweep (plan_adjoint)
    };
    auto ppm_x = [&](Fld qq, Fld out, Rect r, int cdir) {
      TpPpmX_<false> a; a.in[0] = qq; a.in[1] = crx; a.out[0] = out; a.k1 = npz; a.hsel = hsel;
      if (!g.face) { a.orect[0] = r; add(P, grp, a); return; }
      const int npx = g.nx + 1;
      a.orect[0] = isect(r, R(4, npx - 3, r.j0, r.j1));
      TpPpmX_<true> e; e.in[0] = qq; e.in[1] = crx; e.out[0] = out; e.k1 = npz; e.hsel = hsel; e.cdir = cdir;
      std::vector<TpPpmX_<true>> ev;
      for (const Rect& er : {isect(r, R(r.i0, 3, r.j0, r.j1)), isect(r, R(npx - 2, r.i1, r.j0, r.j1))}) if (!is_empty(er)) { e.orect[0] = er; ev.push_back(e); }
      add_multi(P, grp, ev);
      if (!is_empty(a.orect[0])) add(P, grp, a);
    };
    ppm_y(q, fy2, R(isd, ied, js, je + 1), 2);
    TpQi b; b.in[0] = q; b.in[1] = fy2; b.in[2] = yfx; b.in[3] = ray; b.out[0] = q_i; b.orect[0] = R(isd, ied, js, je); b.k1 = npz;
    add(P, grp, b);
    const size_t o1a = P.size();
    ppm_x(q_i, fxo, R(is, ie + 1, js, je), 0);
    const size_t o1b = P.size();
    ppm_x(q, fx2, R(is, ie + 1, jsd, jed), 1);
    TpQj e; e.in[0] = q; e.in[1] = fx2; e.in[2] = xfx; e.in[3] = rax; e.out[0] = q_j; e.orect[0] = R(is, ie, jsd, jed); e.k1 = npz;
    add(P, grp, e);
    const size_t o2a = P.size();
    ppm_y(q_j, fyo, R(is, ie, js, je + 1), 0);
    const size_t o2b = P.size();
    Fld d2b{};
    if (dsel != DAMP_NONE) {
      d2b = W((pre + "_d2b").c_str(), npz);
      TpD2D h; h.in[0] = q; h.out[0] = d2b; h.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); h.k1 = npz; h.dsel = dsel; h.use_mass = use_mass; h.hsel = hsel;
      add_face(P, grp, h, 1);
    }
    // split_hord: some level runs this transport with a trajectory scheme other than the scheme the tangent / adjoint is taken of
    bool any_split = false;
    for (int k = 0; k < npz && k < (int)lev_host.size(); ++k) any_split = any_split || level_split(lev_host[k], hsel, dsel);
    TpFlux t; t.in[0] = fxo; t.in[1] = fx2; t.in[2] = mx; t.in[3] = fyo; t.in[4] = fy2; t.in[5] = my;
    t.in[6] = (dsel != DAMP_NONE) ? q : Fld{}; t.in[7] = d2b; t.in[8] = use_mass ? mass : Fld{};
    for (int n = 6; n < 9; ++n) if (!t.in[n].t) t.in[n].nk = npz;
    t.out[0] = fx; t.out[1] = fy; t.orect[0] = R(is, ie + 1, js, je); t.orect[1] = R(is, ie, js, je + 1); t.k1 = npz;
    t.dsel = dsel; t.use_mass = use_mass; t.hsel = hsel;
    const size_t o3 = P.size();
    add(P, grp, t);
    if (any_split && !fused) set_sticky("split_hord / split_damp need the fused fv_tp_2d (unset FV3LM_TP_FUSED)");
    if (fused) {
      size_t d2_op = P.size();      // the damping Laplacian (TpD2) stays a stage of its own in every mode
      for (size_t n = first_op; n < P.size(); ++n) if (P[n].name == "TpD2" || P[n].name == "TpD2e") d2_op = n;
      for (size_t n = first_op; n < P.size(); ++n) if (!(P[n].name == "TpD2" || P[n].name == "TpD2e")) P[n].modes = 1u << MODE_AD;
      (void)d2_op;
      TpFusedArgs a; a.q = q; a.crx = crx; a.cry = cry; a.xfx = xfx; a.yfx = yfx; a.rax = rax; a.ray = ray; a.mx = mx; a.my = my;
      a.mass = use_mass ? mass : Fld{}; a.d2b = d2b; a.fx = fx; a.fy = fy;
      a.fy2 = fy2; a.q_i = q_i; a.fxo = fxo; a.fx2 = fx2; a.q_j = q_j; a.fyo = fyo;
      a.hsel = hsel; a.dsel = dsel; a.use_mass = use_mass ? 1 : 0; a.nk = npz;
      a.do_acc = 0; a.store_fo = own_fo ? 1 : 0; a.store_mid = own_mid ? 1 : 0;
      if (acc4) { a.acx = acc4[0]; a.acy = acc4[1]; a.amfx = acc4[2]; a.amfy = acc4[3]; }
      Ctx* cp = &ctx;
      Op op{grp, [a, cp](Exec& e, int mode) { run_tp_fused(e, mode, a, *cp); }};
      op.modes = (1u << MODE_NL) | (1u << MODE_TL);
      op.name = "TpFused";
      // what the launch reads / writes (add_halo_async looks for the last op that touches a field; plan_adjoint skips this op: not an adjoint op)
      for (const Fld* f_ : {&a.q, &a.crx, &a.cry, &a.xfx, &a.yfx, &a.rax, &a.ray, &a.mx, &a.my, &a.mass, &a.d2b}) if (f_->p) op.ad_in.push_back(f_->p);
      for (const Fld* f_ : {&a.fx, &a.fy, &a.fy2, &a.q_i, &a.fxo, &a.fx2, &a.q_j, &a.fyo, &a.acx, &a.acy, &a.amfx, &a.amfy}) if (f_->p) op.ad_out.push_back(f_->p);
      P.push_back(op);
      if (any_split) {       // the trajectory pass of the split levels: its own damping Laplacian (run as a nonlinear launch in the tangent mode too), then the chain
        if (dsel != DAMP_NONE) {
          Fld d2t = W((pre + "_d2t").c_str(), npz);
          // a trajectory nord of 2 (split_damp, nord >= 2): two Laplacians, the first into a scratch array on the range widened by one
          bool nord2 = false;
          for (int k = 0; k < npz && k < (int)lev_host.size(); ++k) { int n_; double d_; damp_of(lev_host[k], dsel, n_, d_, false); nord2 = nord2 || (n_ == 2 && d_ > 1.e-4 && level_split(lev_host[k], hsel, dsel)); }
          const size_t n0 = P.size();
          TpD2D h; h.in[0] = q; h.out[0] = d2t; h.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); h.k1 = npz; h.dsel = dsel; h.use_mass = use_mass; h.hsel = hsel; h.traj = 1;
          if (nord2) {
            Fld d2a = W((pre + "_d2a").c_str(), npz);
            TpD2D h1 = h; h1.out[0] = d2a; h1.orect[0] = R(is - 2, ie + 2, js - 2, je + 2); h1.pass = 1;
            add_face(P, grp, h1, 2);
            h.in[0] = d2a;
          }
          add_face(P, grp, h, 1);
          for (size_t n = n0; n < P.size(); ++n) {
            auto f = P[n].fn;
            P[n].fn = [f](Exec& e, int) { f(e, MODE_NL); };
            P[n].modes = (1u << MODE_NL) | (1u << MODE_TL); P[n].ad_in.clear(); P[n].ad_in_slot.clear(); P[n].ad_out.clear(); P[n].ad_out_rect.clear(); P[n].set_wmask = nullptr;
          }
          a.d2b_t = d2t;
        }
        Op tr{grp, [a, cp](Exec& e, int mode) { run_tp_fused(e, mode, a, *cp, true); }};
        tr.modes = (1u << MODE_NL) | (1u << MODE_TL); tr.name = "TpFusedTraj";
        P.push_back(tr);
      }
      // adjoint: the whole routine as ONE hand-written launch (tpad.h); the staged launches then never run, the damping part of the flux keeps a
      // (small) stage of its own.  FV3LM_TP_AD_FUSED=1: round 2's form (outer half fused, inner half staged); =0: all staged.
      if (adf > 0) {
        if (adf == 2) { for (size_t n = first_op; n < o3; ++n) if (!(P[n].name == "TpD2" || P[n].name == "TpD2e")) P[n].modes = 0; }
        else { for (size_t n = o1a; n < o1b; ++n) P[n].modes = 0; for (size_t n = o2a; n < o2b; ++n) P[n].modes = 0; }
        P[o3].modes = 0;
        if (dsel != DAMP_NONE) {
          TpDamp dm; dm.in[0] = q; dm.in[1] = d2b; dm.in[2] = use_mass ? mass : Fld{};
          for (int n = 1; n < 3; ++n) if (!dm.in[n].t) dm.in[n].nk = npz;
          dm.out[0] = fx; dm.out[1] = fy; dm.orect[0] = R(is, ie + 1, js, je); dm.orect[1] = R(is, ie, js, je + 1); dm.k1 = npz;
          dm.dsel = dsel; dm.use_mass = use_mass; dm.hsel = hsel;
          add(P, grp, dm); P.back().modes = 1u << MODE_AD;
        }
        Op ad{grp, [a, cp, adf](Exec& e, int) { if (adf == 2) run_tp_ad(e, a, *cp); else run_tp_outer_ad(e, a, *cp); }};
        ad.modes = 1u << MODE_AD; ad.name = adf == 2 ? "TpAd" : "TpOuter";
        ad.ad_out = {fx.p, fy.p}; ad.ad_out_rect = {R(is, ie + 1, js, je), R(is, ie, js, je + 1)};
        if (adf == 2) ad.ad_in = {q.p, crx.p, cry.p, xfx.p, yfx.p, rax.p, ray.p, mx.p, my.p};
        else { ad.ad_store = {q_i.p, q_j.p, fx2.p, fy2.p}; ad.ad_in = {crx.p, cry.p, mx.p, my.p}; }
        P.push_back(ad);
      }
    }
  }
  // a2b_ord4 (a2b_edge_tlm.F90:48-542): q (nk levels) -> qb on is..ie+1, js..je+1
  void build_a2b(Program& P, const char* grp, const std::string& pre, Fld q, Fld qb, int nk) {
    const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je();
    Fld qx = W((pre + "_qx").c_str(), nk), qy = W((pre + "_qy").c_str(), nk);
    const int npx = g.nx + 1, npy = g.ny + 1;
    A2bA_<false> a; a.in[0] = q; a.out[0] = qx; a.out[1] = qy; a.k1 = nk;
    A2bB_<false> b; b.in[0] = qx; b.in[1] = qy; b.in[2] = Fld{}; b.out[0] = qb; b.k1 = nk;
    if (!g.face) {
      a.orect[0] = R(is, ie + 1, js - 2, je + 2); a.orect[1] = R(is - 2, ie + 2, js, je + 1); add(P, grp, a);
      b.orect[0] = R(is, ie + 1, js, je + 1); add(P, grp, b);
      return;
    }
    // faces: the 4-point formulas away from the cube edges, the edge formulas on strips two points wide (column strips run with the wave
    // along j, exec.h strip_tr).  Global indices: a sub-face tile takes what falls into its window -- rows / columns beyond a cube edge
    // are not interpolated (the edge formulas stand in), the ones beyond an interior tile boundary are (like the periodic tile).
    const Rect qxr = R(is, ie + 1, std::max(1, js - 2), std::min(npy - 1, je + 2)), qyr = R(std::max(1, is - 2), std::min(npx - 1, ie + 2), js, je + 1);
    a.orect[0] = isect(qxr, R(3, npx - 2, qxr.j0, qxr.j1)); a.orect[1] = isect(qyr, R(qyr.i0, qyr.i1, 3, npy - 2));
    const bool a0 = !is_empty(a.orect[0]), a1 = !is_empty(a.orect[1]);
    if (a0 && !a1) a.orect[1] = empty_in(a.orect[0]);
    if (a1 && !a0) a.orect[0] = empty_in(a.orect[1]);
    A2bA_<true> ae; ae.in[0] = q; ae.out[0] = qx; ae.out[1] = qy; ae.k1 = nk;
    std::vector<A2bA_<true>> av;
    for (int e = 0; e < 4; ++e) {
      const Rect r = e == 0 ? isect(qxr, R(1, 2, qxr.j0, qxr.j1)) : e == 1 ? isect(qxr, R(npx - 1, npx, qxr.j0, qxr.j1)) : e == 2 ? isect(qyr, R(qyr.i0, qyr.i1, 1, 2)) : isect(qyr, R(qyr.i0, qyr.i1, npy - 1, npy));
      if (is_empty(r)) continue;
      ae.orect[e < 2 ? 0 : 1] = r; ae.orect[e < 2 ? 1 : 0] = empty_in(r); av.push_back(ae);
    }
    add_multi(P, grp, av);
    if (a0 || a1) add(P, grp, a);            // bulk last: first in the backward sweep (plan_adjoint)
    const Rect qbr = R(is, ie + 1, js, je + 1);
    b.orect[0] = isect(qbr, R(3, npx - 2, 3, npy - 2));
    A2bB_<true> be; be.in[0] = qx; be.in[1] = qy; be.in[2] = q; be.out[0] = qb; be.k1 = nk;
    std::vector<A2bB_<true>> bv;
    for (int e = 0; e < 4; ++e) {
      const Rect r = isect(qbr, e == 0 ? R(1, npx, 1, 2) : e == 1 ? R(1, npx, npy - 1, npy) : e == 2 ? R(1, 2, 3, npy - 2) : R(npx - 1, npx, 3, npy - 2));
      if (is_empty(r)) continue;
      be.orect[0] = r; bv.push_back(be);
    }
    add_multi(P, grp, bv);
    if (!is_empty(b.orect[0])) add(P, grp, b);
  }

  void build_acoustic();
  bool init(int nx, int ny, int npz, int ntile, int face, int nq_, double bdt_, int n_split_, int k_split_, const Options& o,
            const double* const* metrics_host, double da_min, double da_min_c, const double* phis_host, int nface = 0, const int* ij0 = nullptr);
  void destroy();

  void run_group(Progs& P, const char* group, int mode, bool skip_accum = false);
  void zero_work_adjoint() { dev_zero(ex, work.p, work.used * 8); }
  void dyn_core(int mode);
};

// One pass over a program.  The classes' programs hold the same exchanges in the same order; between two exchanges every class runs its
// own launches (any order: they touch disjoint tiles), then the exchange runs once over all tiles.  Adjoint: the same backwards.
inline void Dycore::run_group(Progs& PS, const char* group, int mode, bool skip_accum) {
  const bool all = (group == nullptr) || (group[0] == 0);
  ex.skip_accum = skip_accum;
  const int nc = (int)PS.c.size();
  auto take = [&](const Op& op) { return (all || op.group == group) && !(mode != MODE_AD && skip_accum && op.accum) && ((op.modes >> mode) & 1u); };
  if (nc == 1) {
    set_class(0);
    const Program& P = PS.c[0];
    if (mode != MODE_AD) { for (const Op& op : P) if (take(op)) { if (op.global) set_class(-1); op.fn(ex, mode); if (op.global) set_class(0); } }
    else for (auto it = P.rbegin(); it != P.rend(); ++it) if (take(*it)) { if (it->global) set_class(-1); it->fn(ex, mode); if (it->global) set_class(0); }
    set_class(-1);
    ex.skip_accum = false;
    return;
  }
  // segment boundaries: positions of the global ops in each class's program
  std::vector<std::vector<size_t>> gl(nc);
  for (int c = 0; c < nc; ++c) for (size_t n = 0; n < PS.c[c].size(); ++n) if (PS.c[c][n].global) gl[c].push_back(n);
  for (int c = 1; c < nc; ++c) if (gl[c].size() != gl[0].size()) { set_sticky("internal error: the tile classes' programs hold different exchange sequences"); ex.skip_accum = false; return; }
  const size_t nseg = gl[0].size() + 1;
  auto seg = [&](int c, size_t sg, size_t& a, size_t& b) { a = sg == 0 ? 0 : gl[c][sg - 1] + 1; b = sg < gl[c].size() ? gl[c][sg] : PS.c[c].size(); };
  if (mode != MODE_AD) {
    for (size_t sg = 0; sg < nseg; ++sg) {
      for (int c = 0; c < nc; ++c) { size_t a, b; seg(c, sg, a, b); set_class(c); for (size_t n = a; n < b; ++n) if (take(PS.c[c][n])) PS.c[c][n].fn(ex, mode); }
      if (sg < gl[0].size()) { const Op& op = PS.c[0][gl[0][sg]]; set_class(-1); if (take(op)) op.fn(ex, mode); }
    }
  } else {
    for (size_t sg = nseg; sg-- > 0;) {
      if (sg < gl[0].size()) { const Op& op = PS.c[0][gl[0][sg]]; set_class(-1); if (take(op)) op.fn(ex, mode); }
      for (int c = 0; c < nc; ++c) { size_t a, b; seg(c, sg, a, b); set_class(c); for (size_t n = b; n-- > a;) if (take(PS.c[c][n])) PS.c[c][n].fn(ex, mode); }
    }
  }
  set_class(-1);
  ex.skip_accum = false;
}

inline void Dycore::set_class(int c) {
  cur_cls = c;
  g.ntile = ntile_all; g.i0 = 1; g.j0 = 1; ctx.m = m_all; ex.cls_off = 0;
  if (c >= 0 && c < (int)classes.size()) {
    const TileClass& k = classes[(size_t)c];
    g.ntile = k.n; g.i0 = k.i0; g.j0 = k.j0;
    const size_t op = (size_t)k.t0 * g.plane;
    Metrics& M = ctx.m;
    const double** slots[] = {&M.area, &M.rarea, &M.rarea_c, &M.dx, &M.dy, &M.dxa, &M.dya, &M.dxc, &M.dyc, &M.rdx, &M.rdy, &M.rdxa,
                              &M.rdya, &M.rdxc, &M.rdyc, &M.cosa, &M.sina, &M.rsina, &M.cosa_u, &M.cosa_v, &M.cosa_s, &M.sina_u,
                              &M.sina_v, &M.rsin_u, &M.rsin_v, &M.rsin2, &M.f0, &M.fC, &M.del6_u, &M.del6_v, &M.divg_u, &M.divg_v};
    for (auto sl : slots) *sl += op;
    for (int n = 1; n <= 9; ++n) { M.sin_sg[n] += op; M.cos_sg[n] += op; }
    if (M.edge) M.edge += (size_t)k.t0 * 4 * g.pj;
    if (M.ecorner) M.ecorner += (size_t)k.t0 * 12;
    ex.cls_off = op;
  }
  ctx.g = g;
}

inline bool Dycore::init(int nx, int ny, int npz, int ntile, int face, int nq_, double bdt_, int n_split_, int k_split_,
                         const Options& o, const double* const* metrics_host, double da_min, double da_min_c,
                         const double* phis_host, int nface, const int* ij0) {
  // nx, ny: cells of a resident tile; nface: cells per face edge (0: the tile is the whole face); ij0: (is, js) of each resident tile in
  // the global indices of its face (null: 1, 1)
  g.tx = nx; g.ty = ny; g.nx = (face && nface > 0) ? nface : nx; g.ny = (face && nface > 0) ? nface : ny;
  g.ng = 3; g.npz = npz; g.ntile = ntile; g.face = face ? 1 : 0; g.pi = nx + 2 * g.ng + 1; g.pj = ny + 2 * g.ng + 1;
  g.plane = g.pi * g.pj; g.i0 = g.j0 = 1;
  ntile_all = ntile;
  classes.clear();
  for (int t = 0; t < ntile; ++t) {
    const int i0 = ij0 ? ij0[2 * t] : 1, j0 = ij0 ? ij0[2 * t + 1] : 1;
    if (i0 < 1 || j0 < 1 || i0 + nx - 1 > g.nx || j0 + ny - 1 > g.ny) { err = "tile window outside its face"; return false; }
    if (!classes.empty() && classes.back().i0 == i0 && classes.back().j0 == j0) classes.back().n++;
    else classes.push_back(TileClass{t, 1, i0, j0});
  }
  subtile = face && !(g.tx == g.nx && g.ty == g.ny);
  if (!subtile) for (const TileClass& k : classes) if (k.i0 != 1 || k.j0 != 1) { err = "a whole-face tile starts at (1, 1)"; return false; }
  if (subtile && g.nx != g.ny) { err = "cube faces are square"; return false; }
  opt = o; bdt = bdt_; n_split = n_split_; k_split = k_split_; nq = nq_;
  if (nx < 8 || ny < 8 || npz < 1) { err = "tile too small (need nx,ny >= 8)"; return false; }
  if (!face && ntile != 1) { err = "ntile > 1 needs face = 1 (whole cube faces); face = 0 is the single doubly-periodic tile"; return false; }
  if (face && nx != ny) { err = "cube faces (and their tiles) are square: nx must equal ny"; return false; }
  if (ntile < 1) { err = "ntile < 1"; return false; }
  nh = !o.hydrostatic;
  if (o.nord_pert > 1 || o.nord_pert < 0) { err = "nord_pert in {0,1} only (the divergence damping whose tangent / adjoint is taken; the trajectory's nord may be 2 or 3 with split_damp)"; return false; }
  if (o.nord > 3 || o.nord < 0) { err = "nord in 0..3 only"; return false; }
  if (!o.split_damp) {
    // run_setup_pert (fv_control_tlmadm.F90:220-229) has overwritten the trajectory's damping options by the perturbation's: two different
    // sets with split_damp = 0 describe no state the reference can be in
    if (o.nord != o.nord_pert || o.dddmp != o.dddmp_pert || o.d2_bg != o.d2_bg_pert || o.d4_bg != o.d4_bg_pert || (o.do_vort_damp != 0) != (o.do_vort_damp_pert != 0) ||
        o.vtdm4 != o.vtdm4_pert || o.d2_bg_k1 != o.d2_bg_k1_pert || o.d2_bg_k2 != o.d2_bg_k2_pert) {
      err = "split_damp = 0 but the trajectory's damping options (nord, dddmp, d2_bg, d4_bg, do_vort_damp, vtdm4, d2_bg_k1, d2_bg_k2) differ from the *_pert ones: set split_damp = 1 "
            "(fv_arrays_tlmadm.F90:76) or hand over equal values (run_setup_pert, fv_control_tlmadm.F90:220-229)";
      return false;
    }
  } else {
    if (o.nord == 0 && o.nord_pert > 0) { err = "split_damp: nord = 0 with nord_pert > 0 -- c_sw computes divg_d only for the trajectory's nord > 0 (sw_core_tlm.F90: IF (nord .GT. 0) divergence_corner), the perturbation's damping would read an undefined array"; return false; }
    if (nh && o.nord > 1) { err = "non-hydrostatic with a trajectory nord > 1: the w / height damping would need the tangent of del6_vt_flux with nord 2 (sw_core_tlm.F90:1713-1726), not built"; return false; }
  }
  if (nh && npz < 3) { err = "non-hydrostatic solver needs npz >= 3"; return false; }
  if (nh && !(o.a_imp > 0.5)) { err = "non-hydrostatic: a_imp must be > 0.5 (semi-implicit solver; the reference's a_imp <= 0.5 Riemann-invariant solver is not built)"; return false; }
  // options whose other values are not built are refused, never silently replaced by what is built:
  //   remap profiles: only the linear one, |kord| > 16 (fv_mapz_tlm.F90:8653-8666; the limited profiles are the split_kord work of DESIGN.md §8)
  //   a trajectory kord in {9, 10, 11} with a linear perturbation kord (split_kord): the limited profile gives the values (remap.h)
  for (int kd : {o.kord_tm_pert, o.kord_mt_pert, o.kord_wz_pert, o.kord_tr_pert})
    if ((kd < 0 ? -kd : kd) <= 16) { err = "kord_*_pert: only |kord| > 16 (the linear profile, the one the TL/AD reference differentiates) is built"; return false; }
  for (int kd : {o.kord_tm, o.kord_mt, o.kord_wz, o.kord_tr}) {
    const int ak = kd < 0 ? -kd : kd;
    if (ak <= 16 && !(ak >= 8 && ak <= 15)) { err = "kord_tm/kord_mt/kord_wz/kord_tr: the linear profile (|kord| > 16) or, for the trajectory, the limited profiles 8 .. 15 of cs_profile / scalar_profile (not 16, not ppm_profile's 7 and below)"; return false; }
    if (ak <= 16 && npz < 6) { err = "kord: limited trajectory profiles need npz >= 6"; return false; }
  }
  //   tracer advection: hord_tr / hord_tr_pert split like the others (fv_tracer2d_tlm.F90:1047-1110); hord_tr_ks_* are read by the
  //   reference's namelist but used nowhere on the path (fv_control_tlmadm.F90:166-172), so they are accepted and ignored here too
  //   advection schemes: the tangent / adjoint exists for 1, 2, 333 (tp_core_tlm.F90:2393-2487); a trajectory scheme that differs
  //   (split_hord) gives the values -- a second pass of fv_tp_2d, a second evaluation inside the xtp_u / ytp_v stage -- and may
  //   also be one of the limited 3 .. 7 or the monotone 8 .. 13
  lev_host.resize(npz + 1);
  for (int k = 1; k <= npz; ++k) {
    resolve_level(o, k, npz, lev_host[k - 1]);
    const LevelParams& l = lev_host[k - 1];
    for (int h : {l.hord_mt, l.hord_vt, l.hord_tm, l.hord_dp, l.hord_tr, l.hord_tm_g})
      if (h != 1 && h != 2 && h != 333) { err = "perturbation hord must be 1, 2 or 333 (the schemes the TL/AD reference implements)"; return false; }
    for (int h : {l.hord_mt_t, l.hord_vt_t, l.hord_tm_t, l.hord_dp_t, l.hord_tr_t, l.hord_tm_g_t})
      if (h != 333 && !(h >= 1 && h <= 13)) { err = "trajectory hord must be 1, 2, 333 (differentiated) or, with a different perturbation scheme, 3 .. 13"; return false; }
    const int pairs[6][2] = {{l.hord_mt_t, l.hord_mt}, {l.hord_vt_t, l.hord_vt}, {l.hord_tm_t, l.hord_tm}, {l.hord_dp_t, l.hord_dp}, {l.hord_tr_t, l.hord_tr}, {l.hord_tm_g_t, l.hord_tm_g}};
    for (auto& pr : pairs) if (pr[0] == pr[1] && pr[0] != 1 && pr[0] != 2 && pr[0] != 333) { err = "hord must be 1, 2 or 333 where trajectory and perturbation share the scheme"; return false; }
  }
#ifndef FV3LM_HOST_EMUL
  HIPCHK(hipStreamCreate(&ex.stream));
  { const char* env = std::getenv("FV3LM_NO_ASYNC_HALO");
    if (face && !(env && env[0] == '1')) {
      HIPCHK(hipStreamCreate(&ex.xstream));
      for (int w = 0; w < Exec::NWIN; ++w) { HIPCHK(hipEventCreateWithFlags(&ex.ev_a[w], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ex.ev_b[w], hipEventDisableTiming)); }
    }
    const char* es = std::getenv("FV3LM_NO_SIDE_STRIPS");
    if (face && !(es && es[0] == '1')) {
      HIPCHK(hipStreamCreate(&ex.sstream));
      HIPCHK(hipEventCreateWithFlags(&ex.ev_s0, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ex.ev_s1, hipEventDisableTiming));
    } }
#endif
  lev_host[npz] = lev_host[npz - 1];          // interface npz+1 of the height transport uses the last layer's schemes
  lev_dev = (LevelParams*)dev_alloc(sizeof(LevelParams) * (npz + 1));
  h2d(ex, lev_dev, lev_host.data(), sizeof(LevelParams) * (npz + 1));
  const size_t np = (size_t)ntile * g.plane;
  metric_dev.resize(NMETRIC);
  for (int m = 0; m < NMETRIC; ++m) { metric_dev[m] = (double*)dev_alloc(np * 8); h2d(ex, metric_dev[m], metrics_host[m], np * 8); }
  Metrics& M = ctx.m; int m = 0;
  const double** slots[] = {&M.area, &M.rarea, &M.rarea_c, &M.dx, &M.dy, &M.dxa, &M.dya, &M.dxc, &M.dyc, &M.rdx, &M.rdy, &M.rdxa,
                            &M.rdya, &M.rdxc, &M.rdyc, &M.cosa, &M.sina, &M.rsina, &M.cosa_u, &M.cosa_v, &M.cosa_s, &M.sina_u,
                            &M.sina_v, &M.rsin_u, &M.rsin_v, &M.rsin2, &M.f0, &M.fC, &M.del6_u, &M.del6_v, &M.divg_u, &M.divg_v};
  for (auto s : slots) *s = metric_dev[m++];
  M.sin_sg[0] = M.cos_sg[0] = nullptr;
  for (int n = 1; n <= 9; ++n) M.sin_sg[n] = metric_dev[m++];
  for (int n = 1; n <= 9; ++n) M.cos_sg[n] = metric_dev[m++];
  M.da_min = da_min; M.da_min_c = da_min_c;
  edge_dev = (double*)dev_alloc((size_t)ntile * 4 * g.pj * 8); ecorner_dev = (double*)dev_alloc((size_t)ntile * 12 * 8);
  M.edge = edge_dev; M.ecorner = ecorner_dev;
  m_all = M;
  hs_dev = (double*)dev_alloc(np * 8);
  if (phis_host) h2d(ex, hs_dev, phis_host, np * 8);
  ctx.g = g; ctx.lev = lev_dev; ctx.nlev = npz;
  n3 = np * npz; n3p = np * (npz + 1);
  // the state arena holds exactly the fields build_acoustic and Dynamics::init2 take (an overflow fails create)
  state.init(n3 * (12 + (size_t)nq + (nh ? 5 + (size_t)nq : 0)) + n3p * (6 + (nh ? 2 : 0)) + np);
  if (nh) {
    nh_ws = (double*)dev_alloc((size_t)NH_WS_SLOTS * (npz + 2) * np * 8);
    const char* tape_env = std::getenv("FV3LM_NH_TAPE");
    const bool tape_on = tape_env && tape_env[0] == '1';      // the taped adjoints are the reference path of the tests (nh_ad.h is the default)
    nh_tape.cap = tape_on ? 104 * (npz + 2) : 1;
    int tape_tiles = tape_on ? ntile : 1;
    if (tape_on) {                   // as many tiles per adjoint launch as a third of the free HBM holds (32 B per entry)
#ifndef FV3LM_HOST_EMUL
    { size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
      const size_t per_tile = (size_t)nh_tape.cap * g.plane * 32;
      tape_tiles = (int)std::max<size_t>(1, std::min<size_t>((size_t)ntile, fr / 3 / per_tile)); }
#endif
    }
    nh_tape.stride = (size_t)tape_tiles * g.plane;
    nh_tape.part = (TapePart*)dev_alloc((size_t)nh_tape.cap * nh_tape.stride * sizeof(TapePart));
    nh_tape.idx = (TapeIdx*)dev_alloc((size_t)nh_tape.cap * nh_tape.stride * sizeof(TapeIdx));
    nh_tape.adj = (double*)dev_alloc((size_t)nh_tape.cap * nh_tape.stride * 8);
    nh_tape.overflow = (int*)dev_alloc(8);
  }
  acoustic.c.assign(classes.size(), Program{}); acoustic.cur = &cur_cls;
  {   // the work arena is sized by a dry run of the builder: which fields exist depends on the options
    const size_t state_mark = state.used;
    const std::map<std::string, Fld> F_mark = F;
    work.measure();
    set_class(0);
    build_acoustic();
    const size_t need = work.used;
    acoustic.clear(); n_win = 0; state.used = state_mark; nW3 = nW3p = nS3 = nS3p = 0; F = F_mark;
    work.init(need);
  }
  std::map<double*, size_t> zero_ranges;
  for (int c = 0; c < (int)classes.size(); ++c) {       // one program per tile class, the fields shared
    set_class(c); reuse_fields = c > 0; n_win = 0;
    build_acoustic();
    std::vector<double*> pre; for (const char* n_ : st_out) pre.push_back(f(n_).p);
    for (auto& zr : plan_adjoint(acoustic, work, pre)) { auto it = zero_ranges.find(zr.first); if (it == zero_ranges.end() || it->second < zr.second) zero_ranges[zr.first] = zr.second; }
  }
  reuse_fields = false; set_class(-1);
  acoustic_zero.assign(zero_ranges.begin(), zero_ranges.end());
  ck_stride = 0;
  for (const char* n_ : st_in) ck_stride += (size_t)f(n_).nk * np;
  ckpt = (double*)dev_alloc((size_t)n_split * k_split * ck_stride * 8);
  ex.wlo = work.t; ex.whi = work.t + work.cap;
  return true;
}

inline bool Dycore::set_face_data(const double* edge, const double* ecorner) {
  if (!g.face) { err = "set_face_data: handle was not created with face = 1"; return false; }
  h2d(ex, edge_dev, edge, (size_t)g.ntile * 4 * g.pj * 8);
  h2d(ex, ecorner_dev, ecorner, (size_t)g.ntile * 12 * 8);
  return true;
}
inline bool Dycore::set_exchange(int kind, const int* rows, int n) {
  if (!g.face) { err = "set_exchange: handle was not created with face = 1"; return false; }
  if (kind < 0 || kind >= H_NKIND || n < 0) { err = "set_exchange: bad kind"; return false; }
  std::vector<int> src, ptr, dst; const char* why = "";
  if (!build_extable_host(rows, n, g.ntile, g.plane, src, ptr, dst, &why)) { err = why; return false; }
  ExTable& t = xt[kind];
  dev_free(t.rows); dev_free(t.src); dev_free(t.ptr); dev_free(t.dst);
  t.n = n; t.ns = (int)ptr.size() - 1;
  t.rows = (int*)dev_alloc((size_t)n * 7 * 4 + 4); t.src = (int*)dev_alloc(src.size() * 4 + 4); t.ptr = (int*)dev_alloc(ptr.size() * 4); t.dst = (int*)dev_alloc(dst.size() * 4 + 4);
  if (n) { h2d(ex, t.rows, rows, (size_t)n * 7 * 4); h2d(ex, t.src, src.data(), src.size() * 4); h2d(ex, t.dst, dst.data(), dst.size() * 4); }
  h2d(ex, t.ptr, ptr.data(), ptr.size() * 4);
  return true;
}

inline bool Dycore::set_exchange_remote(int kind, int npeers, const int* peers, const int* nsend, const int* send_rows, const int* nrecv,
                                        const int* recv_rows) {
  if (!g.face) { err = "set_exchange_remote: handle was not created with face = 1"; return false; }
  if (kind < 0 || kind >= H_NKIND || npeers < 0) { err = "set_exchange_remote: bad kind"; return false; }
  ExRemote& x = xr[kind];
  dev_free(x.send_rows); dev_free(x.recv_rows); dev_free(x.asrc); dev_free(x.aptr); dev_free(x.apos); dev_free(x.sendbuf); dev_free(x.recvbuf);
  x = ExRemote{};
  for (int p = 0; p < npeers; ++p) {
    x.peer.push_back(peers[p]); x.nsend.push_back(nsend[p]); x.nrecv.push_back(nrecv[p]);
    x.soff.push_back(x.nsend_tot); x.roff.push_back(x.nrecv_tot);
    x.nsend_tot += nsend[p]; x.nrecv_tot += nrecv[p];
  }
  for (int r = 0; r < x.nsend_tot; ++r) { const int* w = send_rows + 3 * (size_t)r; if ((w[0] | 1) != 1 || w[1] < 0 || w[1] >= g.ntile || w[2] < 0 || w[2] >= g.plane) { err = "send row out of range"; return false; } }
  for (int r = 0; r < x.nrecv_tot; ++r) { const int* w = recv_rows + 4 * (size_t)r; if ((w[0] | 1) != 1 || w[1] < 0 || w[1] >= g.ntile || w[2] < 0 || w[2] >= g.plane || (w[3] != 1 && w[3] != -1)) { err = "recv row out of range"; return false; } }
  // adjoint grouping: send rows by source element
  std::vector<int> order(x.nsend_tot);
  for (int r = 0; r < x.nsend_tot; ++r) order[r] = r;
  auto key = [&](int r) { const int* w = send_rows + 3 * (size_t)r; return ((long long)w[0] << 48) | ((long long)w[1] << 32) | (long long)w[2]; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
  std::vector<int> asrc, aptr, apos; long long last = -1;
  for (int m = 0; m < x.nsend_tot; ++m) {
    const int r = order[m]; const int* w = send_rows + 3 * (size_t)r;
    if (key(r) != last) { last = key(r); aptr.push_back(m); asrc.push_back(w[0]); asrc.push_back(w[1]); asrc.push_back(w[2]); }
    apos.push_back(r);
  }
  aptr.push_back(x.nsend_tot);
  x.ns = (int)aptr.size() - 1;
  auto up = [&](const int* h, size_t n) { int* d = (int*)dev_alloc(n * 4 + 4); if (n) h2d(ex, d, h, n * 4); return d; };
  x.send_rows = up(send_rows, (size_t)x.nsend_tot * 3); x.recv_rows = up(recv_rows, (size_t)x.nrecv_tot * 4);
  x.asrc = up(asrc.data(), asrc.size()); x.aptr = up(aptr.data(), aptr.size()); x.apos = up(apos.data(), apos.size());
  x.cap = (size_t)std::max(x.nsend_tot, x.nrecv_tot) * (g.npz + 1) * 2;
  x.sendbuf = (double*)dev_alloc(x.cap * 8); x.recvbuf = (double*)dev_alloc(x.cap * 8);
  x.active = npeers > 0;
  return true;
}

// called after every other allocation (Dynamics::init2): take what is left of the HBM, minus a reserve
inline void Dycore::init_traj_slots() {
  const size_t slot_bytes = work.used * 8, extra_bytes = (3 * n3p + n3) * 8;     // the fields taken, not the arena's capacity
  if (std::getenv("FV3LM_VERBOSE")) std::fprintf(stderr, "fv3lm: face %d nh %d nq %d: work arena %zu of %zu doubles (%d + %d fields), state %zu of %zu (%d + %d fields)\n", g.face, (int)nh, nq, work.used, work.cap, nW3, nW3p, state.used, state.cap, nS3, nS3p);
  int want = n_split * k_split;
  if (const char* e = std::getenv("FV3LM_TRAJ_SLOTS")) want = std::min(want, std::max(0, std::atoi(e)));
#ifndef FV3LM_HOST_EMUL
  size_t fr = 0, tot = 0;
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = 0;
  const size_t reserve = (size_t)6 << 30;
  const int fit = fr > reserve ? (int)((fr - reserve) / (slot_bytes + extra_bytes)) : 0;
  if (want > fit) want = fit;
#endif
  for (int n = 0; n < want; ++n) { traj_slot.push_back((double*)dev_alloc(slot_bytes)); traj_slot_p.push_back((double*)dev_alloc(extra_bytes)); }
}

inline void Dycore::destroy() {
  for (double* p : metric_dev) dev_free(p);
  dev_free(lev_dev); dev_free(hs_dev); dev_free(ckpt); dev_free(edge_dev); dev_free(ecorner_dev);
  dev_free(nh_ws); dev_free(nh_tape.part); dev_free(nh_tape.idx); dev_free(nh_tape.adj); dev_free(nh_tape.overflow);
  for (ExTable& t : xt) { dev_free(t.rows); dev_free(t.src); dev_free(t.ptr); dev_free(t.dst); }
  for (ExRemote& x : xr) { dev_free(x.send_rows); dev_free(x.recv_rows); dev_free(x.asrc); dev_free(x.aptr); dev_free(x.apos); dev_free(x.sendbuf); dev_free(x.recvbuf); }
  state.destroy(); work.destroy();
  for (double* q_ : traj_slot) dev_free(q_);
  for (double* q_ : traj_slot_p) dev_free(q_);
#ifndef FV3LM_HOST_EMUL
  if (ex.stream) (void)hipStreamDestroy(ex.stream);
  if (ex.sstream) { (void)hipStreamDestroy(ex.sstream); if (ex.ev_s0) (void)hipEventDestroy(ex.ev_s0); if (ex.ev_s1) (void)hipEventDestroy(ex.ev_s1); }
  if (ex.xstream) { (void)hipStreamDestroy(ex.xstream); for (int w = 0; w < Exec::NWIN; ++w) { if (ex.ev_a[w]) (void)hipEventDestroy(ex.ev_a[w]); if (ex.ev_b[w]) (void)hipEventDestroy(ex.ev_b[w]); } }
#endif
}

inline NhColArgs Dycore::nh_args(double dt_) const {
  NhColArgs a{};
  a.g = g; a.ws = nh_ws; a.ws_stride = (size_t)ntile_all * g.plane; a.tape = nh_tape; a.hs = hs_dev; a.lev = lev_dev;
  a.zvir = opt.zvir; a.cp_air = opt.cp_air;
  a.kord_tm = opt.kord_tm; a.kord_tr = opt.kord_tr; a.kord_wz = opt.kord_wz;
  { const char* e = std::getenv("FV3LM_NH_TAPE"); a.use_tape = (e && e[0] == '1') ? 1 : 0; }
  a.dt = dt_; a.akap = opt.akap; a.ptop = opt.ptop; a.rdgas = opt.rdgas; a.grav = opt.grav; a.a_imp = opt.a_imp; a.p_fac = opt.p_fac; a.scale_z = opt.scale_z;
  return a;
}
// column operator of nh.h as a program step.  when: 0 every acoustic step, 2 the last only, 3 vertical remap (last_call = last k_split step)
inline void Dycore::add_col(Program& P, const char* group, int kind, const NhColArgs& a, Rect r, Rect skip, int when) {
  Dycore* self = this;
  static const char* tags[][3] = {{"riem_c.nl", "riem_c.tl", "riem_c.ad"}, {"riem3.nl", "riem3.tl", "riem3.ad"}, {"edge_profile.nl", "edge_profile.tl", "edge_profile.ad"},
                                  {"zh_init.nl", "zh_init.tl", "zh_init.ad"}, {"p_ring.nl", "p_ring.tl", "p_ring.ad"},
                                  {"remap_field_nh.nl", "remap_field_nh.tl", "remap_field_nh.ad"}, {"remap_press_nh.nl", "remap_press_nh.tl", "remap_press_nh.ad"},
                                  {"remap_w_nh.nl", "remap_w_nh.tl", "remap_w_nh.ad"}, {"remap_field_nh_lim.nl", "remap_field_nh_lim.tl", "remap_field_nh_lim.ad"},
                                  {"remap_w_nh_lim.nl", "remap_w_nh_lim.tl", "remap_w_nh_lim.ad"}};
  Op op{group, [self, kind, a, r, skip, when](Exec& e, int mode) {
    if (when == 2 && !self->last_acoustic) return;
    NhColArgs b = a; b.last_call = (when == 3 ? self->remap_last : self->last_acoustic) ? 1 : 0;
    run_nh_col(e, mode, b, kind, r, skip, tags[kind][mode]);
  }};
  // forward inputs / outputs of each column operator (nh.h), for plan_adjoint
  const int nin = kind == NHC_RIEM_C ? 4 : kind == NHC_RIEM3 ? 4 : kind == NHC_EDGE ? 2 : kind == NHC_RING ? 1 : 0;
  const int nout = kind == NHC_RIEM_C ? 2 : kind == NHC_RIEM3 ? 9 : kind == NHC_EDGE ? 2 : kind == NHC_RING ? 1 : 0;
  for (int n = 0; n < nin; ++n) op.ad_in.push_back(a.f[n].p);
  for (int n = 0; n < nout; ++n) { op.ad_out.push_back(a.f[nin + n].p); op.ad_out_rect.push_back(r); }
  P.push_back(op);
}
inline bool Dycore::nh_overflow() {
  if (!nh) return false;
  int flag = 0; d2h(ex, &flag, nh_tape.overflow, 4);
  return flag != 0;
}

// One acoustic step (dyn_core_tlm.F90:1736-2466).  Inputs u,v,delp,pt (halos valid);
// outputs u_o,v_o,delp_o,pt_o (halos valid) + accumulated mfx,mfy,cx,cy + pe,peln,pk,pkz.
inline void Dycore::build_acoustic() {
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je(), isd = g.isd(), ied = g.ied(), jsd = g.jsd(), jed = g.jed(), npz = g.npz;
  const double dt = bdt / double(n_split) / double(k_split), dt2 = 0.5 * dt;
  Program& P = acoustic;
  // prognostic state (step input / output) and accumulators
  Fld u = S("u", npz), v = S("v", npz), delp = S("delp", npz), pt = S("pt", npz);
  Fld u_o = W("u_o", npz), v_o = W("v_o", npz), delp_o = W("delp_o", npz), pt_o = W("pt_o", npz);   // step outputs: per-step trajectory like every work array
  Fld mfx = S("mfx", npz), mfy = S("mfy", npz), cx = S("cx", npz), cy = S("cy", npz);
  Fld pe = S("pe", npz + 1), peln = S("peln", npz + 1), pk = S("pk", npz + 1), pkz = S("pkz", npz);
  st_in = {"u", "v", "delp", "pt"}; st_out = {"u_o", "v_o", "delp_o", "pt_o"};
  Fld w{}, delz{}, zh{}, w_o{}, delz_o{}, zh_o{};
  if (nh) {
    w = S("w", npz); delz = S("delz", npz); zh = S("zh", npz + 1);
    w_o = W("w_o", npz); delz_o = W("delz_o", npz); zh_o = W("zh_o", npz + 1);
    for (const char* n_ : {"w", "delz", "zh"}) st_in.push_back(n_);
    for (const char* n_ : {"w_o", "delz_o", "zh_o"}) st_out.push_back(n_);
  }
  // ---- c_sw
  Fld utmp = W("utmp", npz), vtmp = W("vtmp", npz), ua = W("ua", npz), va = W("va", npz);
  { CswInterpAD s; s.in[0] = u; s.in[1] = v; s.out[0] = utmp; s.out[1] = vtmp; s.out[2] = ua; s.out[3] = va;
    s.orect[0] = R(isd, ied, js - 1, je + 1); s.orect[1] = R(is - 1, ie + 1, jsd, jed);
    s.orect[2] = s.orect[3] = R(is - 1, ie + 1, js - 1, je + 1);
    if (g.face) {   // next to a cube edge the 2-point bands + corner views reach the whole halo; beyond an interior tile boundary: as the periodic tile
      const int il = is == 1 ? isd : is - 1, ih = ie == g.nx ? ied : ie + 1, jl = js == 1 ? jsd : js - 1, jh = je == g.ny ? jed : je + 1;
      s.orect[0] = R(isd, ied, jl, jh); s.orect[1] = R(il, ih, jsd, jed); s.orect[2] = s.orect[3] = R(il, ih, jl, jh);
    }
    s.k1 = npz; add_face(P, "c_sw", s, 3); }
  Fld uc0 = W("uc0", npz), utf = W("utf", npz), vc0 = W("vc0", npz), vtf = W("vtf", npz);
  const Fld none{};
  { CswInterpC_<false> s; s.in[0] = utmp; s.in[1] = vtmp; s.in[2] = u; s.in[3] = v; s.out[0] = uc0; s.out[1] = utf; s.out[2] = vc0; s.out[3] = vtf;
    s.dt2 = dt2; s.k1 = npz;
    if (!g.face) {
      s.orect[0] = s.orect[1] = R(is - 1, ie + 2, js - 1, je + 1); s.orect[2] = s.orect[3] = R(is - 1, ie + 1, js - 1, je + 2); add(P, "c_sw", s);
    } else {   // 4-point interpolation away from the cube edges; one-sided / edge formulas and corner views on 3-wide strips (global indices)
      const int npx = g.nx + 1, npy = g.ny + 1;
      const Rect ur = R(is - 1, ie + 2, js - 1, je + 1), vr = R(is - 1, ie + 1, js - 1, je + 2);
      s.orect[0] = s.orect[1] = isect(ur, R(3, npx - 2, ur.j0, ur.j1)); s.orect[2] = s.orect[3] = isect(vr, R(vr.i0, vr.i1, 3, npy - 2));
      const bool b0 = !is_empty(s.orect[0]), b2 = !is_empty(s.orect[2]);
      if (b0 && !b2) s.orect[2] = s.orect[3] = empty_in(s.orect[0]);
      if (b2 && !b0) s.orect[0] = s.orect[1] = empty_in(s.orect[2]);
      CswInterpC_<true> e; for (int n = 0; n < 4; ++n) { e.in[n] = s.in[n]; e.out[n] = s.out[n]; }
      e.in[4] = ua; e.in[5] = va; e.dt2 = dt2; e.k1 = npz;
      std::vector<CswInterpC_<true>> ev;
      for (int m = 0; m < 4; ++m) {
        const Rect r = m == 0 ? isect(ur, R(ur.i0, 2, ur.j0, ur.j1)) : m == 1 ? isect(ur, R(npx - 1, ur.i1, ur.j0, ur.j1)) : m == 2 ? isect(vr, R(vr.i0, vr.i1, vr.j0, 2)) : isect(vr, R(vr.i0, vr.i1, npy - 1, vr.j1));
        if (is_empty(r)) continue;
        e.orect[0] = e.orect[1] = m < 2 ? r : empty_in(r); e.orect[2] = e.orect[3] = m < 2 ? empty_in(r) : r; ev.push_back(e);
      }
      add_multi(P, "c_sw", ev);
      if (b0 || b2) add(P, "c_sw", s);       // bulk last: first in the backward sweep (plan_adjoint)
    } }
  Fld divgd = W("divgd", npz);
  if (opt.nord > 0) {
    CswDivgD s; s.in[0] = u; s.in[1] = v; s.in[2] = ua; s.in[3] = va; s.out[0] = divgd; s.orect[0] = R(is, ie + 1, js, je + 1); s.k1 = npz;
    add_face(P, "c_sw", s, 1);
  }
  Fld delpc = W("delpc", npz), ptc = W("ptc", npz);
  { CswTransportD s; s.in[0] = delp; s.in[1] = pt; s.in[2] = utf; s.in[3] = vtf; s.out[0] = delpc; s.out[1] = ptc;
    s.orect[0] = s.orect[1] = R(is - 1, ie + 1, js - 1, je + 1); s.k1 = npz; add_face(P, "c_sw", s, 1); }
  Fld wc{};
  if (nh) {
    wc = W("wc", npz);
    CswTransportWD s; s.in[0] = delp; s.in[1] = w; s.in[2] = utf; s.in[3] = vtf; s.in[4] = delpc; s.out[0] = wc;
    s.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); s.k1 = npz; add_face(P, "c_sw", s, 1);
  }
  Fld ke_c = W("ke_c", npz), vort_c = W("vort_c", npz);
  { CswKeVortD s; s.in[0] = ua; s.in[1] = va; s.in[2] = uc0; s.in[3] = vc0; s.in[4] = g.face ? u : none; s.in[5] = g.face ? v : none; s.out[0] = ke_c; s.out[1] = vort_c;
    s.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); s.orect[1] = R(is, ie + 1, js, je + 1); s.dt2 = dt2; s.k1 = npz; add_face(P, "c_sw", s, 1); }
  Fld uc1 = W("uc1", npz), vc1 = W("vc1", npz);
  { CswUpdateD s; s.in[0] = uc0; s.in[1] = vc0; s.in[2] = u; s.in[3] = v; s.in[4] = vort_c; s.in[5] = ke_c; s.out[0] = uc1; s.out[1] = vc1;
    s.orect[0] = R(is, ie + 1, js, je); s.orect[1] = R(is, ie, js, je + 1); s.dt2 = dt2; s.k1 = npz; add_face(P, "c_sw", s, 1); }
  // ---- geopk (C grid) + p_grad_c
  Fld pe_c = W("pe_c", npz + 1), peln_c = W("peln_c", npz + 1), pkc = W("pkc", npz + 1), gz = W("gz", npz + 1);
  Fld uc = W("uc", npz), vc = W("vc", npz);
  if (nh) {
    // interface heights advected with the C-grid fluxes, the vertically implicit solver, pressure gradient (dyn_core_tlm.F90:1860-1960)
    Fld gz_a = W("gz_a", npz + 1);
    Fld xfz = W("xfz", npz + 1), yfz = W("yfz", npz + 1);
    { DzFluxC s; s.in[0] = utf; s.in[1] = vtf; s.out[0] = xfz; s.out[1] = yfz; s.orect[0] = R(is - 1, ie + 2, js - 1, je + 1);
      s.orect[1] = R(is - 1, ie + 1, js - 1, je + 2); s.k1 = npz + 1; add(P, "update_dz_c", s); }
    { UpdateDzCD s; s.in[0] = zh; s.in[1] = xfz; s.in[2] = yfz; s.out[0] = gz_a; s.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); s.k1 = npz + 1;
      add_face(P, "update_dz_c", s, 1); }
    { NhColArgs a = nh_args(dt2); a.f[0] = gz_a; a.f[1] = wc; a.f[2] = ptc; a.f[3] = delpc; a.f[4] = gz; a.f[5] = pkc;
      add_col(P, "riem_c", NHC_RIEM_C, a, R(is - 1, ie + 1, js - 1, je + 1)); }
    PGradCNh s; s.in[0] = pkc; s.in[1] = gz; s.in[2] = uc1; s.in[3] = vc1; s.in[4] = delpc; s.out[0] = uc; s.out[1] = vc;
    s.orect[0] = R(is, ie + 1, js, je); s.orect[1] = R(is, ie, js, je + 1); s.dt2 = dt2; s.k1 = npz; add(P, "p_grad_c", s);
  } else {
  { GeopkArgs a; a.g = g; a.R = R(is - 1, ie + 1, js - 1, je + 1); a.delp = delpc; a.pt = ptc; a.pe = pe_c; a.peln = peln_c; a.pk = pkc;
    a.gz = gz; a.pkz = Fld{}; a.hs = hs_dev; a.ptop = opt.ptop; a.akap = opt.akap; a.cp_air = opt.cp_air; a.cg = 1;
    Op op{"geopk_c", [a](Exec& e, int mode) { run_geopk(e, mode, a); }};
    op.ad_in = {a.delp.p, a.pt.p}; op.ad_out = {a.pe.p, a.peln.p, a.pk.p, a.gz.p}; op.ad_out_rect.assign(4, a.R);
    P.push_back(op); }
  { PGradC s; s.in[0] = pkc; s.in[1] = gz; s.in[2] = uc1; s.in[3] = vc1; s.out[0] = uc; s.out[1] = vc;
    s.orect[0] = R(is, ie + 1, js, je); s.orect[1] = R(is, ie, js, je + 1); s.dt2 = dt2; s.k1 = npz; add(P, "p_grad_c", s); }
  }
  add_halo(P, "halo_uc", H_CVEC, uc, vc);
  // ---- d_sw
  Fld ut = W("ut", npz), crx = W("crx", npz), xfx = W("xfx", npz), vt = W("vt", npz), cry = W("cry", npz), yfx = W("yfx", npz);
  if (!g.face) {
    DswWinds s; s.in[0] = uc; s.in[1] = vc; s.out[0] = ut; s.out[1] = crx; s.out[2] = xfx; s.out[3] = vt; s.out[4] = cry; s.out[5] = yfx;
    s.orect[0] = R(is - 1, ie + 2, jsd, jed); s.orect[1] = s.orect[2] = R(is, ie + 1, jsd, jed);
    s.orect[3] = R(isd, ied, js - 1, je + 2); s.orect[4] = s.orect[5] = R(isd, ied, js, je + 1); s.dt = dt; s.k1 = npz; add(P, "d_sw", s);
  } else {
    Fld ut_a = W("ut_a", npz), vt_a = W("vt_a", npz);
    { DswWindsAD s; s.in[0] = uc; s.in[1] = vc; s.out[0] = ut_a; s.out[1] = vt_a; s.orect[0] = R(is - 1, ie + 2, jsd, jed); s.orect[1] = R(isd, ied, js - 1, je + 2);
      s.dt = dt; s.k1 = npz; add_face(P, "d_sw", s, 1); }
    Fld ut_e = W("ut_e", npz), vt_e = W("vt_e", npz);
    const int npx = g.nx + 1, npy = g.ny + 1;
    DswWindsE se; se.in[0] = ut_a; se.in[1] = vt_a; se.in[2] = uc; se.in[3] = vc; se.out[0] = ut_e; se.out[1] = vt_e; se.k1 = npz;
    std::vector<DswWindsE> sv;
    for (int e = 0; e < 4; ++e) {      // the rows / columns next to a cube edge that fall into this tile's range (global indices)
      const Rect r = e == 0 ? isect(R(is - 1, ie + 2, jsd, jed), R(is - 1, ie + 2, 0, 1)) : e == 1 ? isect(R(is - 1, ie + 2, jsd, jed), R(is - 1, ie + 2, npy - 1, npy))
                   : e == 2 ? isect(R(isd, ied, js - 1, je + 2), R(0, 1, js - 1, je + 2)) : isect(R(isd, ied, js - 1, je + 2), R(npx - 1, npx, js - 1, je + 2));
      if (is_empty(r)) continue;
      se.orect[e < 2 ? 0 : 1] = r; se.orect[e < 2 ? 1 : 0] = empty_in(r); sv.push_back(se);
    }
    add_multi(P, "d_sw", sv);
    DswWindsCD s; s.in[0] = ut_a; s.in[1] = vt_a; s.in[2] = ut_e; s.in[3] = vt_e; s.out[0] = ut; s.out[1] = crx; s.out[2] = xfx; s.out[3] = vt; s.out[4] = cry; s.out[5] = yfx;
    s.orect[0] = R(is - 1, ie + 2, jsd, jed); s.orect[1] = s.orect[2] = R(is, ie + 1, jsd, jed);
    s.orect[3] = R(isd, ied, js - 1, je + 2); s.orect[4] = s.orect[5] = R(isd, ied, js, je + 1); s.dt = dt; s.k1 = npz; add_face(P, "d_sw", s, 1);
  }
  Fld rax = W("ra_x", npz), ray = W("ra_y", npz);
  { DswRa s; s.in[0] = xfx; s.in[1] = yfx; s.out[0] = rax; s.out[1] = ray; s.orect[0] = R(is, ie, jsd, jed); s.orect[1] = R(isd, ied, js, je);
    s.k1 = npz; add(P, "d_sw", s); }
  Fld fx = W("fx", npz), fy = W("fy", npz), gx = W("gx", npz), gy = W("gy", npz);
  // the flux capacitors ride along in the fused fv_tp_2d launch of delp (nonlinear + tangent); their staged form serves the adjoint
  const Fld acc4[4] = {cx, cy, mfx, mfy};
  const bool tpf_on = !(std::getenv("FV3LM_TP_FUSED") && std::getenv("FV3LM_TP_FUSED")[0] == '0');
  build_tp(P, "d_sw", "tpd", delp, crx, cry, xfx, yfx, rax, ray, xfx, yfx, Fld{}, HORD_DP, DAMP_V, false, fx, fy, 0, tpf_on ? acc4 : nullptr);
  { const size_t n0 = P.size();
    add_accum(P, "d_sw", cx, crx, R(is, ie + 1, jsd, jed)); add_accum(P, "d_sw", mfx, fx, R(is, ie + 1, js, je));
    add_accum(P, "d_sw", cy, cry, R(isd, ied, js, je + 1)); add_accum(P, "d_sw", mfy, fy, R(is, ie, js, je + 1));
    if (tpf_on) for (size_t n = n0; n < P.size(); ++n) P[n].modes = 1u << MODE_AD; }
  Fld w_m{}, dw{}, gxw{}, gyw{};
  if (nh) {   // w: damping increment and transport with the mass fluxes (sw_core_tlm.F90:3020-3062)
    Fld d6w = W("del6_w", npz); dw = W("dw", npz); gxw = W("gxw", npz); gyw = W("gyw", npz); w_m = W("w_m", npz);
    { Del6AD s; s.in[0] = w; s.out[0] = d6w; s.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); s.k1 = npz; add_face(P, "d_sw", s, 1); }
    { DswDw s; s.in[0] = w; s.in[1] = d6w; s.out[0] = dw; s.orect[0] = R(is, ie, js, je); s.k1 = npz; add(P, "d_sw", s); }
    build_tp(P, "d_sw", "tpw", w, crx, cry, xfx, yfx, rax, ray, fx, fy, Fld{}, HORD_VT, DAMP_NONE, false, gxw, gyw);
  }
  build_tp(P, "d_sw", "tpt", pt, crx, cry, xfx, yfx, rax, ray, fx, fy, delp, HORD_TM, DAMP_T, true, gx, gy);
  { DswUpdateDp s; s.in[0] = delp; s.in[1] = pt; s.in[2] = fx; s.in[3] = fy; s.in[4] = gx; s.in[5] = gy; s.out[0] = delp_o; s.out[1] = pt_o;
    s.orect[0] = s.orect[1] = R(is, ie, js, je); s.k1 = npz; add(P, "d_sw", s); }
  if (nh) {
    DswUpdateW s; s.in[0] = w; s.in[1] = delp; s.in[2] = delp_o; s.in[3] = gxw; s.in[4] = gyw; s.in[5] = dw; s.out[0] = w_m;
    s.orect[0] = R(is, ie, js, je); s.k1 = npz; add(P, "d_sw", s);
  }
  Fld vb = W("vb", npz), ub = W("ub", npz), ke = W("ke", npz);
  { DswKeWindsD s; s.in[0] = uc; s.in[1] = vc; s.in[2] = g.face ? ut : none; s.in[3] = g.face ? vt : none; s.out[0] = vb; s.out[1] = ub; s.orect[0] = s.orect[1] = R(is, ie + 1, js, je + 1); s.dt = dt;
    s.k1 = npz; add_face(P, "d_sw", s, 1); }
  { bool split_mt = false;
    for (int k = 0; k < npz; ++k) split_mt = split_mt || level_split(lev_host[k], HORD_MT);
    auto ke_stage = [&](auto s) { s.in[0] = vb; s.in[1] = ub; s.in[2] = u; s.in[3] = v; s.in[4] = g.face ? ut : none; s.in[5] = g.face ? vt : none; s.dt = dt; s.out[0] = ke; s.orect[0] = R(is, ie + 1, js, je + 1); s.k1 = npz; add_face(P, "d_sw", s, 3); };
    if (split_mt) ke_stage(DswKeSD{}); else ke_stage(DswKeD{}); }
  Fld wk = W("wk", npz), vorta = W("vort_abs", npz);
  { DswVort s; s.in[0] = u; s.in[1] = v; s.out[0] = wk; s.out[1] = vorta; s.orect[0] = s.orect[1] = R(isd, ied, jsd, jed); s.k1 = npz; add(P, "d_sw", s); }
  Fld da = W("dd_a", npz), db = W("dd_b", npz), dc = W("dd_c", npz), vortb = W("vort_b", npz), ke2 = W("ke2", npz);
  // the corner-scalar exchange of divg_d is first needed here: its start is hoisted to right after the last launch that touches the field
  // (CswDivg in c_sw), so it runs beside the rest of c_sw, geopk, p_grad_c and the transports of d_sw (the join used to sit in c_sw)
  if (opt.nord > 0) add_halo_async(P, "halo_divgd", H_CORNER, divgd);
  { DdAD s; s.in[0] = divgd; s.in[1] = u; s.in[2] = v; s.in[3] = ua; s.in[4] = va; s.in[5] = g.face ? uc : none; s.in[6] = g.face ? vc : none; s.out[0] = da; s.out[1] = db;
    s.orect[0] = R(is - 1, ie + 1, js, je + 1); s.orect[1] = R(is, ie + 1, js - 1, je + 1); s.k1 = npz; add_face(P, "d_sw", s, 1); }
  { DdBD s; s.in[0] = da; s.in[1] = db; s.out[0] = dc; s.orect[0] = R(is, ie + 1, js, je + 1); s.k1 = npz; add_face(P, "d_sw", s, 1); }
  build_a2b(P, "d_sw", "a2bw", wk, vortb, npz);
  { DdC s; s.in[0] = ke; s.in[1] = dc; s.in[2] = divgd; s.in[3] = vortb; s.out[0] = ke2; s.orect[0] = R(is, ie + 1, js, je + 1);
    s.dt = dt; s.k1 = npz; add(P, "d_sw", s); }
  // split_damp (sw_core_tlm.F90:2350-2369): the stages above are COMPUTE_DIVERGENCE_DAMPING_TLM with the perturbation's coefficients, whose
  // tangent / adjoint is kept; the trajectory's own damping -- nord up to 3 -- gives the VALUES of ke2 (dampt.h), nonlinear and tangent mode alike
  { bool need = false; int max_nord = 0;
    for (int k = 0; k < npz; ++k) { const LevelParams& l = lev_host[k]; max_nord = std::max(max_nord, l.nord);
      need = need || (l.split_damp && (l.nord != l.nord_p || l.d2_divg != l.d2_divg_p || l.dddmp != l.dddmp_p || l.d4_bg != l.d4_bg_p)); }
    if (need) {
      Fld s1 = max_nord >= 1 ? W("ddt_s1", npz) : Fld{}, s2 = max_nord >= 2 ? W("ddt_s2", npz) : Fld{};
      Dycore* self = this;
      Op op{"d_sw", [self, u, v, ua, va, uc, vc, divgd, vortb, ke, ke2, s1, s2, dt, max_nord](Exec& e, int) {
        DampTArgs a; a.g = self->g; a.m = self->ctx.m; a.lev = self->lev_dev;
        a.u = e.sh(u).t; a.v = e.sh(v).t; a.ua = e.sh(ua).t; a.va = e.sh(va).t; a.uc = e.sh(uc).t; a.vc = e.sh(vc).t;
        a.divgd = e.sh(divgd).t; a.vortb = e.sh(vortb).t; a.ke = e.sh(ke).t; a.ke2 = e.sh(ke2).t; a.s1 = e.sh(s1).t; a.s2 = e.sh(s2).t; a.dt = dt;
        run_damp_t(e, a, max_nord);
      }};
      op.modes = (1u << MODE_NL) | (1u << MODE_TL); op.name = "DampT";
      P.push_back(op);
    } }
  Fld fxv = W("fxv", npz), fyv = W("fyv", npz);
  build_tp(P, "d_sw", "tpv", vorta, crx, cry, xfx, yfx, rax, ray, xfx, yfx, Fld{}, HORD_VT, DAMP_NONE, false, fxv, fyv);
  Fld d6 = W("del6_d2", npz);
  { Del6AD s; s.in[0] = wk; s.out[0] = d6; s.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); s.k1 = npz; add_face(P, "d_sw", s, 1); }
  Fld u_m = W("u_m", npz), v_m = W("v_m", npz);
  { bool n2 = false;       // the trajectory's vorticity damping with nord_v = 2 (split_damp, nord >= 2): two more Laplacians, values only
    for (int k = 0; k < npz; ++k) n2 = n2 || (lev_host[k].nord_v == 2 && lev_host[k].damp_vt > 1.e-5);
    auto uv_stage = [&](auto s) { s.in[0] = u; s.in[1] = v; s.in[2] = ke2; s.in[3] = fxv; s.in[4] = fyv; s.in[5] = wk; s.in[6] = d6; s.out[0] = u_m; s.out[1] = v_m;
      s.orect[0] = R(is, ie, js, je + 1); s.orect[1] = R(is, ie + 1, js, je); s.k1 = npz; return s; };
    if (!n2) add(P, "d_sw", uv_stage(DswUpdateUV{}));
    else {
      Fld l1 = W("d6t_a", npz), l2 = W("d6t_b", npz);
      Dycore* self = this;
      Op op{"d_sw", [self, wk, l1, l2](Exec& e, int) {
        const Geom& g_ = self->g;
        LapTArgs a{g_, self->ctx.m, self->lev_dev, e.sh(wk).t, e.sh(l1).t, 1};
        for_points(e, Rect{g_.is() - 2, g_.ie() + 2, g_.js() - 2, g_.je() + 2}, g_.ntile * g_.npz, LapTPass{a}, "LapT.1", 0.);
        a.src = e.sh(l1).t; a.dst = e.sh(l2).t; a.pass = 2;
        for_points(e, Rect{g_.is() - 1, g_.ie() + 1, g_.js() - 1, g_.je() + 1}, g_.ntile * g_.npz, LapTPass{a}, "LapT.2", 0.);
      }};
      op.modes = (1u << MODE_NL) | (1u << MODE_TL); op.name = "LapT";
      P.push_back(op);
      auto s2 = uv_stage(DswUpdateUVT<true>{}); s2.in[7] = l2; add(P, "d_sw", s2);
    } }
  // delp, pt of the step: first needed with their halo by geopk / the pressure gradient; the exchange runs beside the KE / vorticity /
  // damping launches of d_sw
  add_halo_async(P, "halo_dp", H_CELL, delp_o); add_halo_async(P, "halo_dp", H_CELL, pt_o);
  if (nh) {
    // ---- update_dz_d (nh_utils_tlm.F90:590-722): Courant numbers and area fluxes at the interfaces, transport of the heights
    Fld crx_e = W("crx_e", npz + 1), xfx_e = W("xfx_e", npz + 1), cry_e = W("cry_e", npz + 1), yfx_e = W("yfx_e", npz + 1);
    { NhColArgs a = nh_args(dt); a.f[0] = crx; a.f[1] = xfx; a.f[2] = crx_e; a.f[3] = xfx_e; add_col(P, "update_dz_d", NHC_EDGE, a, R(is, ie + 1, jsd, jed)); }
    { NhColArgs a = nh_args(dt); a.f[0] = cry; a.f[1] = yfx; a.f[2] = cry_e; a.f[3] = yfx_e; add_col(P, "update_dz_d", NHC_EDGE, a, R(isd, ied, js, je + 1)); }
    Fld rax_e = W("ra_x_e", npz + 1), ray_e = W("ra_y_e", npz + 1);
    { DswRa s; s.in[0] = xfx_e; s.in[1] = yfx_e; s.out[0] = rax_e; s.out[1] = ray_e; s.orect[0] = R(is, ie, jsd, jed); s.orect[1] = R(isd, ied, js, je);
      s.k1 = npz + 1; add(P, "update_dz_d", s); }
    Fld fxz = W("fxz", npz + 1), fyz = W("fyz", npz + 1), d6z = W("del6_z", npz + 1), zh_a = W("zh_a", npz + 1);
    build_tp(P, "update_dz_d", "tpz", zh, crx_e, cry_e, xfx_e, yfx_e, rax_e, ray_e, xfx_e, yfx_e, Fld{}, HORD_TM_G, DAMP_NONE, false, fxz, fyz, npz + 1);   // the global hord_tm for every interface (dyn_core_tlm.F90:1091, nh_utils_tlm.F90:496-514)
    { Del6AD s; s.in[0] = zh; s.out[0] = d6z; s.orect[0] = R(is - 1, ie + 1, js - 1, je + 1); s.k1 = npz + 1; add_face(P, "update_dz_d", s, 1); }
    { UpdateDzD s; s.in[0] = zh; s.in[1] = d6z; s.in[2] = fxz; s.in[3] = fyz; s.in[4] = rax_e; s.in[5] = ray_e; s.out[0] = zh_a;
      s.orect[0] = R(is, ie, js, je); s.k1 = npz + 1; add(P, "update_dz_d", s); }
    // ---- riem_solver3 + halo rings of the pressures + nh_p_grad (dyn_core_tlm.F90:2165-2400)
    Fld ppe = W("ppe", npz + 1), pk3 = W("pk3", npz + 1);
    { NhColArgs a = nh_args(dt); a.f[0] = zh_a; a.f[1] = w_m; a.f[2] = pt_o; a.f[3] = delp_o; a.f[4] = w_o; a.f[5] = delz_o; a.f[6] = zh_o; a.f[7] = ppe;
      a.f[8] = pk3; a.f[9] = pe; a.f[10] = peln; a.f[11] = pk; a.f[12] = S("ws", 1); add_col(P, "riem3", NHC_RIEM3, a, R(is, ie, js, je)); }
    add_halo(P, "halo_zh", H_CELL, zh_o); add_halo(P, "halo_zh", H_CELL, ppe);
    { NhColArgs a = nh_args(dt); a.f[0] = delp_o; a.f[1] = pe; a.what = 1;
      add_col(P, "p_ring", NHC_RING, a, R(is - 1, ie + 1, js - 1, je + 1), R(is, ie, js, je), 2); }
    { NhColArgs a = nh_args(dt); a.f[0] = delp_o; a.f[1] = pk3; a.what = 0;
      add_col(P, "p_ring", NHC_RING, a, R(is - 2, ie + 2, js - 2, je + 2), R(is, ie, js, je)); }
    Fld pp_b = W("pp_b", npz + 1), pk_b = W("pk3_b", npz + 1), zh_b = W("zh_b", npz + 1), dp_b = W("dp_b", npz);
    build_a2b(P, "nh_p_grad", "a2bpp", ppe, pp_b, npz + 1);
    build_a2b(P, "nh_p_grad", "a2bpk", pk3, pk_b, npz + 1);
    build_a2b(P, "nh_p_grad", "a2bzh", zh_o, zh_b, npz + 1);
    build_a2b(P, "nh_p_grad", "a2bdp", delp_o, dp_b, npz);
    { NhPGrad s; s.in[0] = u_m; s.in[1] = v_m; s.in[2] = dp_b; s.in[3] = pp_b; s.in[4] = pk_b; s.in[5] = zh_b; s.out[0] = u_o; s.out[1] = v_o;
      s.orect[0] = R(is, ie, js, je + 1); s.orect[1] = R(is, ie + 1, js, je); s.dt = dt; s.ptk = std::pow(opt.ptop, opt.akap); s.grav = opt.grav; s.k1 = npz;
      add(P, "nh_p_grad", s); }
    add_halo(P, "halo_uv", H_DVEC, u_o, v_o, 1); add_halo(P, "halo_uv", H_DEDGE, u_o, v_o, 2);
    add_halo(P, "halo_w", H_CELL, w_o);      // the reference fills it at the start of the next step (:1741-1748); same data, and the step's
    return;                                   // checkpointed input then carries a valid halo
  }
  // ---- geopk (D grid) + one_grad_p
  Fld pkd = pk, gzd = W("gzd", npz + 1);
  { GeopkArgs a; a.g = g; a.R = R(is - 2, ie + 2, js - 2, je + 2); a.delp = delp_o; a.pt = pt_o; a.pe = pe; a.peln = peln; a.pk = pkd;
    a.gz = gzd; a.pkz = pkz; a.hs = hs_dev; a.ptop = opt.ptop; a.akap = opt.akap; a.cp_air = opt.cp_air; a.cg = 0;
    Op op{"geopk_d", [a](Exec& e, int mode) { run_geopk(e, mode, a); }};
    op.ad_in = {a.delp.p, a.pt.p}; op.ad_out = {a.pe.p, a.peln.p, a.pk.p, a.gz.p, a.pkz.p}; op.ad_out_rect.assign(5, a.R);
    P.push_back(op); }
  Fld pkb = W("pk_b", npz + 1), gzb = W("gz_b", npz + 1);
  build_a2b(P, "one_grad_p", "a2bp", pkd, pkb, npz + 1);
  build_a2b(P, "one_grad_p", "a2bg", gzd, gzb, npz + 1);
  { OneGradP s; s.in[0] = u_m; s.in[1] = v_m; s.in[2] = pkb; s.in[3] = gzb; s.out[0] = u_o; s.out[1] = v_o;
    s.orect[0] = R(is, ie, js, je + 1); s.orect[1] = R(is, ie + 1, js, je); s.dt = dt; s.ptk = std::pow(opt.ptop, opt.akap); s.k1 = npz;
    add(P, "one_grad_p", s); }
  // it < n_split: halo update of u, v (:2432-2434); last step: shared edge rows only, mpp_get_boundary (:2418-2431)
  add_halo(P, "halo_uv", H_DVEC, u_o, v_o, 1); add_halo(P, "halo_uv", H_DEDGE, u_o, v_o, 2);
}

// n_split acoustic steps.  NL/TL: state fields u,v,delp,pt are advanced in place (via the *_o
// buffers).  AD: on entry the .p of u,v,delp,pt (and of mfx..cy, pe..pkz) hold the adjoint of the
// outputs; on exit the adjoint of the inputs.  Trajectory checkpoints must have been stored by a
// preceding MODE_NL sweep (store_ckpt=true).
inline void Dycore::dyn_core(int mode) {
  const std::vector<const char*>&names = st_in, &onames = st_out;
  const int ns = (int)names.size();
  const char* pnames[4] = {"pe", "peln", "pk", "pkz"};
  const size_t b3 = n3 * 8, np = (size_t)g.ntile * g.plane;
  auto bytes = [&](int n) { return (size_t)f(names[n]).nk * np * 8; };
  auto ck = [&](int a, int n) { double* q_ = ckpt + (size_t)a * ck_stride; for (int m = 0; m < n; ++m) q_ += (size_t)f(names[m]).nk * np; return q_; };
  // first acoustic step of the call: interface heights from the layer thicknesses, halo filled (dyn_core_tlm.F90:1779-1801)
  auto zh_init = [&](int md) {
    NhColArgs a = nh_args(0.); a.f[0] = f("delz"); a.f[1] = f("zh");
    halo(md, H_CELL, f("w"));
    auto cols = [&]() { each_class([&]() { NhColArgs b = a; b.g = g; run_nh_col(ex, md, b, NHC_ZH_INIT, R(g.is(), g.ie(), g.js(), g.je()), R(1, 0, 1, 0), "zh_init"); }); };
    if (md != MODE_AD) { cols(); halo(md, H_CELL, f("zh")); }
    else { halo(md, H_CELL, f("zh")); cols(); }
  };
  auto slot_io = [&](int a, bool save) {      // pe, peln, pk, pkz trajectory of step a <-> its slot
    double* q_ = traj_slot_p[a];
    for (int n = 0; n < 4; ++n) {
      const size_t nb = (n < 3 ? n3p : n3) * 8;
      if (save) dev_copy(ex, q_, f(pnames[n]).t, nb); else dev_copy(ex, f(pnames[n]).t, q_, nb);
      q_ += (n < 3 ? n3p : n3);
    }
  };
  // State rotation.  The prognostic fields are not copied between the acoustic steps: the step's program is pointed (exec.h Redir) at
  // where its inputs already are and at where its outputs are wanted.
  //   NL  the outputs (*_o, work arena) of a step with a trajectory slot stay in that slot; the next step reads them there, and so does
  //       the backward sweep (no checkpoint copy for that step).  pe, peln, pk, pkz are written straight into the slot (hydrostatic;
  //       the non-hydrostatic step does not write pkz).  The last step's outputs are copied to the state fields.
  //   TL  inputs and outputs swap roles every step (both sides); one copy at the end when n_split is odd.
  //   AD  the adjoint of a step's inputs is accumulated where the next (earlier) step expects the adjoint of its outputs.
  const bool p_in_slot = !nh;
  auto slot_out = [&](int a, int n) { return f(onames[n]).t + (traj_slot[a] - work.t); };      // step a's output n inside its slot
  auto point_p_at_slot = [&](int a) {
    double* q_ = traj_slot_p[a];
    for (int n = 0; n < 4; ++n) { ex.redirect_t(f(pnames[n]).t, q_); q_ += (n < 3 ? n3p : n3); }
  };
  if (mode != MODE_AD) {
    for (const char* a : {"mfx", "mfy", "cx", "cy"}) { dev_zero(ex, f(a).t, b3); if (mode == MODE_TL) dev_zero(ex, f(a).p, b3); }
    if (nh) zh_init(mode);
    for (int it = 0; it < n_split; ++it) {
      const int a = ck_base + it;
      const bool last = it == n_split - 1;
      ex.nrt = ex.nrp = 0;
      if (mode == MODE_NL) {
        if (it > 0 && has_slot(a - 1)) { for (int n = 0; n < ns; ++n) ex.redirect_t(f(names[n]).t, slot_out(a - 1, n)); }
        else for (int n = 0; n < ns; ++n) dev_copy(ex, ck(a, n), f(names[n]).t, bytes(n));
        ex.tshift = has_slot(a) ? traj_slot[a] - work.t : 0;
        if (has_slot(a) && p_in_slot) point_p_at_slot(a);
      } else if (it & 1) {
        for (int n = 0; n < ns; ++n) {
          ex.redirect_t(f(names[n]).t, f(onames[n]).t); ex.redirect_t(f(onames[n]).t, f(names[n]).t);
          ex.redirect_p(f(names[n]).p, f(onames[n]).p); ex.redirect_p(f(onames[n]).p, f(names[n]).p);
        }
      }
      last_acoustic = last;
      run_group(acoustic, nullptr, mode);
      if (mode == MODE_NL) {
        if (last || !has_slot(a)) for (int n = 0; n < ns; ++n) dev_copy(ex, f(names[n]).t, f(onames[n]).t + ex.tshift, bytes(n));
        if (has_slot(a)) {
          if (!p_in_slot) slot_io(a, true);
          else if (last) { ex.nrt = 0; slot_io(a, false); }         // what follows dyn_core reads the pressures in their own fields
        }
      } else if (last && !(it & 1)) {
        for (int n = 0; n < ns; ++n) { dev_copy(ex, f(names[n]).t, f(onames[n]).t, bytes(n)); dev_copy(ex, f(names[n]).p, f(onames[n]).p, bytes(n)); }
      }
      ex.tshift = 0; ex.nrt = ex.nrp = 0;
    }
  } else {
    // the incoming adjoint lives in the input-named buffers, which stand for the *_o side of the last step
    if (nh) dev_zero(ex, f("zh").p, n3p * 8);      // the heights leave dyn_core unused: no incoming adjoint
    for (int it = n_split - 1; it >= 0; --it) {
      const int a = ck_base + it, r = n_split - 1 - it;
      ex.nrt = ex.nrp = 0;
      if (it > 0 && has_slot(a - 1)) { for (int n = 0; n < ns; ++n) ex.redirect_t(f(names[n]).t, slot_out(a - 1, n)); }
      else for (int n = 0; n < ns; ++n) dev_copy(ex, f(names[n]).t, ck(a, n), bytes(n));
      last_acoustic = (it == n_split - 1);
      if (has_slot(a)) {       // this step's intermediates were kept by the forward sweep
        ex.tshift = traj_slot[a] - work.t;
        if (p_in_slot) point_p_at_slot(a); else slot_io(a, false);
      } else {                 // recompute this step's nonlinear intermediates (flux capacitors left alone)
        ex.tshift = 0;
        run_group(acoustic, nullptr, MODE_NL, true);
      }
      for (auto& zr : acoustic_zero) dev_zero(ex, zr.first, zr.second * 8);      // the few work adjoints no stage launch stores first (plan_adjoint)
      if (!(r & 1)) {          // adjoint of the outputs in the input-named buffers: swap the two sides
        for (int n = 0; n < ns; ++n) { ex.redirect_p(f(names[n]).p, f(onames[n]).p); ex.redirect_p(f(onames[n]).p, f(names[n]).p); dev_zero(ex, f(onames[n]).p, bytes(n)); }
      } else for (int n = 0; n < ns; ++n) dev_zero(ex, f(names[n]).p, bytes(n));
      run_group(acoustic, nullptr, MODE_AD);
      // pe, peln, pk, pkz of earlier steps are overwritten by later ones: their adjoint is zero there
      for (const char* a_ : {"pe", "peln", "pk"}) dev_zero(ex, f(a_).p, n3p * 8);
      dev_zero(ex, f("pkz").p, b3);
      ex.tshift = 0; ex.nrt = ex.nrp = 0;
    }
    if (n_split & 1) for (int n = 0; n < ns; ++n) dev_copy(ex, f(names[n]).p, f(onames[n]).p, bytes(n));      // an odd number of swaps
    if (nh) zh_init(MODE_AD);
  }
}

}  // namespace fv3
