// fv3lm-hip: generic stage launchers (nonlinear / tangent / adjoint) — see core.h.
//
// Thread mapping (gfx950): one thread per grid point, block = 64 x 4 (one wavefront per row
// segment, i fastest so that every global access of a wave is a contiguous 512-byte row piece),
// grid.z = tiles x levels — at C48L72 that is already >10^3 workgroups per launch, C192L127 x 6
// faces ~10^5, far above the 256 CUs.  Levels/tiles are the outermost index so that the planes of
// one level stay together in an XCD's L2 while the stencil rows are re-read.
#pragma once
#include "core.h"
#include <cstddef>
#include <type_traits>
#include <string>
#include <vector>

namespace fv3 {

// A library a Fortran host links must not abort (include/fv3lm.h: every entry point returns a status).  The first failure of
// a HIP call, allocation or internal sizing check is recorded here and the work goes on as no-ops where it can; every C-ABI
// entry point reports it through its status and fv3lm_last_error() (fv3lm_capi.cpp status()).  One process drives one GPU
// (like one MPI rank of the reference), so the record is per process.
inline std::string& sticky_error() { static std::string e; return e; }
inline void set_sticky(const std::string& m) { if (sticky_error().empty()) sticky_error() = m; }

// Per-kernel measurement: HIP events recorded on the library's own stream around every launch while
// profiling is on (bench.py's roofline leg), aggregated by kernel name together with the algorithmic
// bytes of each launch (compulsory reads+writes of the launch's fields).
struct ProfRec { std::string name; double bytes; 
#ifndef FV3LM_HOST_EMUL
  hipEvent_t e0, e1;
#endif
};
struct Exec {
#ifndef FV3LM_HOST_EMUL
  hipStream_t stream = nullptr;
  // exchange stream: a halo exchange whose field nobody touches for a while runs here beside the launches of the main
  // stream (dycore.h add_halo_async); ev_a / ev_b: "main reached the start" / "exchange done", one pair per window
  hipStream_t xstream = nullptr;
  // strip stream (dycore.h add_face): in the forward modes the edge strips of a stage run beside its bulk launch (disjoint outputs);
  // ev_s0 = "main reached the strips", ev_s1 = "strips done"
  hipStream_t sstream = nullptr; hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr; bool side_pending = false;
  static constexpr int NWIN = 4;
  hipEvent_t ev_a[NWIN] = {}, ev_b[NWIN] = {};
#endif
  bool check_boxes = false;   // host emulation only: verify declared stencil boxes
  bool skip_accum = false;    // the adjoint's trajectory recompute: flux accumulators are left alone (dycore.h run_group)
  bool no_wmask = false;      // adjoint launches accumulate only (a single kernel group run through fv3lm_run_group: the caller clears and seeds the adjoints)
  // Trajectory slots (dycore.h): while tshift != 0 every trajectory pointer into the work arena [wlo, whi) is redirected
  // to the slot of the acoustic step being run, so that the backward sweep finds that step's intermediates without
  // recomputing them.
  double* wlo = nullptr; double* whi = nullptr; std::ptrdiff_t tshift = 0;
  // Tile classes (dycore.h set_class): the resident tiles are run one class at a time -- runs of tiles that share a window is..ie x
  // js..je of their face (one class when every tile is a whole face).  Each class has its own program, built in the reference's global
  // face indices; what moves at run time is every field pointer, to the class's first tile (cls_off doubles into a one-level field).
  // Exchanges run once, over all tiles, with the shift off.
  size_t cls_off = 0;
  // State rotation (dycore.h dyn_core): instead of copying the prognostic fields between the acoustic steps, the step's programs are
  // pointed at where the values already are -- whole fields, matched by their base pointer (trajectory side rt, perturbation side rp).
  struct Redir { const double* from; double* to; };
  static constexpr int NREDIR = 24;
  Redir rt[NREDIR], rp[NREDIR]; int nrt = 0, nrp = 0;
  void redirect_t(const double* from, double* to) { if (nrt < NREDIR) rt[nrt++] = Redir{from, to}; else set_sticky("internal error: redirect table full"); }
  void redirect_p(const double* from, double* to) { if (nrp < NREDIR) rp[nrp++] = Redir{from, to}; else set_sticky("internal error: redirect table full"); }
  Fld sh(const Fld& f) const {
    Fld r = f;
    if (tshift && f.t >= wlo && f.t < whi) r.t = f.t + tshift;
    for (int k = 0; k < nrt; ++k) if (f.t == rt[k].from) { r.t = rt[k].to; break; }
    for (int k = 0; k < nrp; ++k) if (f.p == rp[k].from) { r.p = rp[k].to; break; }
    if (cls_off) { if (r.t) r.t += cls_off * (size_t)f.nk; if (r.p) r.p += cls_off * (size_t)f.nk; }
    return r;
  }
  long launches = 0;
  bool profiling = false;
  std::vector<ProfRec> recs;
  void mark_begin(const char* name, const char* suffix, double bytes) {
    if (!profiling) return;
    ProfRec r; r.name = std::string(name) + suffix; r.bytes = bytes;
#ifndef FV3LM_HOST_EMUL
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess || hipEventRecord(r.e0, stream) != hipSuccess) { set_sticky("profiling: hipEventCreate/Record failed"); profiling = false; return; }
#endif
    recs.push_back(r);
  }
  void mark_end() {
    if (!profiling) return;
#ifndef FV3LM_HOST_EMUL
    (void)hipEventRecord(recs.back().e1, stream);
#endif
  }
};

HD Rect rect_union(const Rect* r, int n) {
  Rect u = r[0];
  for (int m = 1; m < n; ++m) {
    if (r[m].i0 < u.i0) u.i0 = r[m].i0;
    if (r[m].i1 > u.i1) u.i1 = r[m].i1;
    if (r[m].j0 < u.j0) u.j0 = r[m].j0;
    if (r[m].j1 > u.j1) u.j1 = r[m].j1;
  }
  return u;
}

// ---- per-point bodies ---------------------------------------------------------------------
template <class S>
HD void body_nl(const S& s, const Ctx& c, int i, int j, int z) {
  const int nl = s.k1 - s.k0 + 1, tile = z / nl, k = s.k0 + z % nl;
  double o[S::NOUT];
  AccNL<S> a{s, c, tile, k};
  s.template eval<double>(a, c, tile, i, j, k, o);
  for (int n = 0; n < S::NOUT; ++n)
    if (s.orect[n].has(i, j)) s.out[n].t[(size_t)(tile * s.out[n].nk + k - 1) * c.g.plane + c.g.idx(i, j)] = o[n];
}
template <class S>
HD void body_tl(const S& s, const Ctx& c, int i, int j, int z) {
  const int nl = s.k1 - s.k0 + 1, tile = z / nl, k = s.k0 + z % nl;
  Dual o[S::NOUT];
  AccTL<S> a{s, c, tile, k};
  s.template eval<Dual>(a, c, tile, i, j, k, o);
  for (int n = 0; n < S::NOUT; ++n)
    if (s.orect[n].has(i, j)) {
      size_t m = (size_t)(tile * s.out[n].nk + k - 1) * c.g.plane + c.g.idx(i, j);
      s.out[n].t[m] = o[n].v;
      s.out[n].p[m] = o[n].d;
    }
}
// Adjoint of input M at point (i,j) of plane z (z counts levels 1..in[M].nk per tile).
// The stencil box is a compile-time constant and the offset loops are fully unrolled: after inlining,
// "is this load the seeded one" folds to a constant for every load of the stage, so each of the K
// evaluations keeps only the derivative chain that starts at its seeded load (forward-mode partial
// evaluation; built with -fno-signed-zeros -ffinite-math-only so that 0*x and x+0 fold).
template <class S, int M>
HD void body_ad_one(const S& s, const Ctx& c, const Rect& R, int i, int j, int z) {
  const Fld& f = s.in[M];
  if (!f.p) return;
  const int tile = z / f.nk, kk = 1 + z % f.nk;
  constexpr Box b = S::box(M);
  constexpr unsigned want = S::wants(M);
  if (!want) return;
  if (i < R.i0 + b.di0 || i > R.i1 + b.di1 || j < R.j0 + b.dj0 || j > R.j1 + b.dj1) return;
  double acc = 0.0;
#pragma unroll
  for (int dk = b.dk0; dk <= b.dk1; ++dk) {
    const int k = kk - dk;
    if (k < s.k0 || k > s.k1) continue;
    AccAD<S, M> a{s, c, tile, k, i, j, kk, want};
#pragma unroll
    for (int dj = b.dj0; dj <= b.dj1; ++dj) {
      const int oj = j - dj;
      if (oj < R.j0 || oj > R.j1) continue;
#pragma unroll
      for (int di = b.di0; di <= b.di1; ++di) {
        const int oi = i - di;
        if (oi < R.i0 || oi > R.i1) continue;
        if (!S::uses(M, di, dj, dk)) continue;
        Dual o[S::NOUT];
        s.template eval<Dual>(a, c, tile, oi, oj, k, o);
#pragma unroll
        for (int n = 0; n < S::NOUT; ++n)
          if (((want >> n) & 1u) && s.orect[n].has(oi, oj))
            acc += o[n].d * s.out[n].p[(size_t)(tile * s.out[n].nk + k - 1) * c.g.plane + c.g.idx(oi, oj)];
      }
    }
  }
  f.p[(size_t)(tile * f.nk + kk - 1) * c.g.plane + c.g.idx(i, j)] += acc;
}
// Joint form of the same gather: all inputs of a level class at once.  For every output offset in the union of their
// boxes the stage is evaluated ONCE with one tangent direction per input (DualV), instead of once per input: the loads,
// the branches and the common subexpressions of an evaluation are shared, which is what the per-input form spent most
// of its time on for stages with many inputs (flux assembly, pointwise updates).
// Level classes: inputs of one class have the same number of levels (npz vs npz+1); kclass_of() names the class.
template <class S> constexpr int kclass_of(const S*, int) { return 0; }
template <class S> constexpr bool joint_ad(const S*) { return true; }
template <class S, int KC>
HD constexpr Box class_box() {
  Box u{0, 0, 0, 0, 0, 0}; bool first = true;
  for (int m = 0; m < S::NIN; ++m) {
    if (kclass_of((const S*)nullptr, m) != KC || !S::wants(m)) continue;
    const Box b = S::box(m);
    if (first) { u = b; first = false; continue; }
    if (b.di0 < u.di0) u.di0 = b.di0; if (b.di1 > u.di1) u.di1 = b.di1;
    if (b.dj0 < u.dj0) u.dj0 = b.dj0; if (b.dj1 > u.dj1) u.dj1 = b.dj1;
    if (b.dk0 < u.dk0) u.dk0 = b.dk0; if (b.dk1 > u.dk1) u.dk1 = b.dk1;
  }
  return u;
}
template <class S, int KC>
HD constexpr bool class_any() {
  for (int m = 0; m < S::NIN; ++m) if (kclass_of((const S*)nullptr, m) == KC && S::wants(m)) return true;
  return false;
}
template <class S, int KC>
HD void body_ad_joint(const S& s, const Ctx& c, const Rect& R, int i, int j, int z) {
  if constexpr (class_any<S, KC>()) {
    constexpr int N = S::NIN;
    int nk = 0;
#pragma unroll
    for (int m = 0; m < N; ++m) if (kclass_of((const S*)nullptr, m) == KC && S::wants(m) && s.in[m].p && nk == 0) nk = s.in[m].nk;
    if (nk == 0 || z >= c.g.ntile * nk) return;
    const int tile = z / nk, kk = 1 + z % nk;
    constexpr Box ub = class_box<S, KC>();
    // outside the reach of every output: nothing to gather, but a write-mode input (s.wmask, below) still stores its zero
    const bool inreach = !(i < R.i0 + ub.di0 || i > R.i1 + ub.di1 || j < R.j0 + ub.dj0 || j > R.j1 + ub.dj1);
    if (!inreach && !s.wmask) return;
    double acc[N];
#pragma unroll
    for (int m = 0; m < N; ++m) acc[m] = 0.0;
    if (inreach) {
#pragma unroll
    for (int dk = ub.dk0; dk <= ub.dk1; ++dk) {
      const int k = kk - dk;
      if (k < s.k0 || k > s.k1) continue;
#pragma unroll
      for (int dj = ub.dj0; dj <= ub.dj1; ++dj) {
        const int oj = j - dj;
        if (oj < R.j0 || oj > R.j1) continue;
#pragma unroll
        for (int di = ub.di0; di <= ub.di1; ++di) {
          const int oi = i - di;
          if (oi < R.i0 || oi > R.i1) continue;
          unsigned mask = 0u, want = 0u;
#pragma unroll
          for (int m = 0; m < N; ++m) {
            const Box b = S::box(m);
            if (kclass_of((const S*)nullptr, m) == KC && S::wants(m) && di >= b.di0 && di <= b.di1 && dj >= b.dj0 && dj <= b.dj1 && dk >= b.dk0 &&
                dk <= b.dk1 && S::uses(m, di, dj, dk)) { mask |= 1u << m; want |= S::wants(m); }
          }
          if (!mask) continue;
          AccADV<S> a{s, c, tile, k, i, j, kk, mask, want};
          DualV<N> o[S::NOUT];
          s.template eval<DualV<N>>(a, c, tile, oi, oj, k, o);
#pragma unroll
          for (int n = 0; n < S::NOUT; ++n)
            if (((want >> n) & 1u) && s.orect[n].has(oi, oj)) {
              const double oa = s.out[n].p[(size_t)(tile * s.out[n].nk + k - 1) * c.g.plane + c.g.idx(oi, oj)];
#pragma unroll
              for (int m = 0; m < N; ++m) if (((mask >> m) & 1u) && ((S::wants(m) >> n) & 1u)) acc[m] += o[n].d[m] * oa;
            }
        }
      }
    }
    }
    // Store.  Default: accumulate onto what later stages (earlier in this backward sweep) left.  Write mode (bit m of s.wmask,
    // set by Dycore::plan_adjoint for the first stage of the backward sweep that touches the buffer): store, with zeros outside
    // the stage's reach, over the launch rectangle (widened to where the buffer is read later, s.wrect) -- which is what lets the sweep
    // run without clearing the work adjoints.
#pragma unroll
    for (int m = 0; m < N; ++m)
      if (kclass_of((const S*)nullptr, m) == KC && S::wants(m) && s.in[m].p) {
        const Box b = S::box(m);
        const bool in_m = !(i < R.i0 + b.di0 || i > R.i1 + b.di1 || j < R.j0 + b.dj0 || j > R.j1 + b.dj1);
        double* q = &s.in[m].p[(size_t)(tile * nk + kk - 1) * c.g.plane + c.g.idx(i, j)];
        if ((s.wmask >> m) & 1u) *q = in_m ? acc[m] : 0.0;
        else if (in_m) *q += acc[m];
      }
  }
}

// Corner-halo reads through an index map (edges.h): input point (i,j) is also read at its alias
// locations.  Outputs reached only that way are added here, by a second, small launch over the points
// around the four face corners (the main kernel keeps its fully unrolled fast path).  mask = outputs of
// this visit not already covered by the main loop or by an earlier alias of the same point.
template <class S, int M>
HD void body_ad_alias(const S& s, const Ctx& c, const Rect& R, int i, int j, int z) {
  const Fld& f = s.in[M];
  if (!f.p) return;
  constexpr int MA = S::alias_box(M);
  constexpr Box b = S::box(M), ba = S::box(MA);
  constexpr unsigned wm = S::wants(M), wa = S::wants(MA);
  if (!wa) return;
  const int tile = z / f.nk, kk = 1 + z % f.nk;
  double acc = 0.0;
  for (int n = 0; n < S::NALIAS; ++n) {
    int qi, qj;
    if (!s.alias(c, M, i, j, n, qi, qj)) continue;
    for (int dk = ba.dk0; dk <= ba.dk1; ++dk) {
      const int k = kk - dk;
      if (k < s.k0 || k > s.k1) continue;
      for (int dj = ba.dj0; dj <= ba.dj1; ++dj)
        for (int di = ba.di0; di <= ba.di1; ++di) {
          const int oi = qi - di, oj = qj - dj;
          if (!R.has(oi, oj) || !S::uses(MA, di, dj, dk)) continue;
          unsigned mask = wa;
          const int ei = i - oi, ej = j - oj;
          if (ei >= b.di0 && ei <= b.di1 && ej >= b.dj0 && ej <= b.dj1 && dk >= b.dk0 && dk <= b.dk1 && S::uses(M, ei, ej, dk)) mask &= ~wm;
          for (int n2 = 0; n2 < n; ++n2) {
            int pi2, pj2;
            if (!s.alias(c, M, i, j, n2, pi2, pj2)) continue;
            const int fi = pi2 - oi, fj = pj2 - oj;
            if (fi >= ba.di0 && fi <= ba.di1 && fj >= ba.dj0 && fj <= ba.dj1 && S::uses(MA, fi, fj, dk)) mask = 0u;
          }
          if (!mask) continue;
          AccAD<S, M> a{s, c, tile, k, i, j, kk, mask};
          Dual o[S::NOUT];
          s.template eval<Dual>(a, c, tile, oi, oj, k, o);
          for (int m = 0; m < S::NOUT; ++m)
            if (((mask >> m) & 1u) && s.orect[m].has(oi, oj))
              acc += o[m].d * s.out[m].p[(size_t)(tile * s.out[m].nk + k - 1) * c.g.plane + c.g.idx(oi, oj)];
        }
    }
  }
  if (acc != 0.0) f.p[(size_t)(tile * f.nk + kk - 1) * c.g.plane + c.g.idx(i, j)] += acc;
}
template <class S, int M, bool END = (M >= S::NIN)>
struct AdAliasLoop {
  HD static void run(const S& s, const Ctx& c, const Rect& R, int i, int j, int z) {
    if (z < c.g.ntile * s.in[M].nk) body_ad_alias<S, M>(s, c, R, i, j, z);
    AdAliasLoop<S, M + 1>::run(s, c, R, i, j, z);
  }
};
template <class S, int M>
struct AdAliasLoop<S, M, true> {
  HD static void run(const S&, const Ctx&, const Rect&, int, int, int) {}
};
// the (2 ng)^2 points around face corner cn (0 sw, 1 se, 2 ne, 3 nw), n = 0 .. 4 ng^2 - 1
HD void corner_block_point(const Geom& g, int cn, int n, int& i, int& j) {
  const int w = 2 * g.ng;
  const int i0 = (cn == 0 || cn == 3) ? 1 - g.ng : g.nx + 1 - g.ng, j0 = (cn < 2) ? 1 - g.ng : g.ny + 1 - g.ng;
  i = i0 + n % w; j = j0 + n / w;
}

template <class S, int M, bool END = (M >= S::NIN)>
struct AdLoop {
  HD static void run(const S& s, const Ctx& c, const Rect& R, int i, int j, int z, int nkmax) {
    if (z < c.g.ntile * s.in[M].nk) body_ad_one<S, M>(s, c, R, i, j, z);
    AdLoop<S, M + 1>::run(s, c, R, i, j, z, nkmax);
  }
};
template <class S, int M>
struct AdLoop<S, M, true> {
  HD static void run(const S&, const Ctx&, const Rect&, int, int, int, int) {}
};
// A stage may bring its own gather adjoint (static constexpr bool HAND_AD; hand_ad(c, R, i, j, z) with the store rules of body_ad_joint):
// worth it where the stage is linear with constant weights and the seeded re-evaluation per offset costs several times the arithmetic.
template <class S, class = void> struct hand_ad_of { static constexpr bool value = false; };
template <class S> struct hand_ad_of<S, typename std::enable_if<S::HAND_AD>::type> { static constexpr bool value = true; };
template <class S>
HD void ad_point(const S& s, const Ctx& c, const Rect& R, int i, int j, int z, int nkmax) {
  if constexpr (hand_ad_of<S>::value) { (void)nkmax; s.hand_ad(c, R, i, j, z); }
  else if constexpr (joint_ad((const S*)nullptr)) { body_ad_joint<S, 0>(s, c, R, i, j, z); body_ad_joint<S, 1>(s, c, R, i, j, z); }
  else AdLoop<S, 0>::run(s, c, R, i, j, z, nkmax);
}

template <class S>
inline Rect ad_input_rect(const S& s, const Ctx& c, const Rect& R) {
  Rect q = R;
  for (int m = 0; m < S::NIN; ++m) {
    Box b = S::box(m);
    if (R.i0 + b.di0 < q.i0) q.i0 = R.i0 + b.di0;
    if (R.i1 + b.di1 > q.i1) q.i1 = R.i1 + b.di1;
    if (R.j0 + b.dj0 < q.j0) q.j0 = R.j0 + b.dj0;
    if (R.j1 + b.dj1 > q.j1) q.j1 = R.j1 + b.dj1;
  }
  if (s.wmask && s.wrect.i0 <= s.wrect.i1) {     // write mode: also where the stored adjoints are read later although this stage does not reach
    if (s.wrect.i0 < q.i0) q.i0 = s.wrect.i0; if (s.wrect.i1 > q.i1) q.i1 = s.wrect.i1;
    if (s.wrect.j0 < q.j0) q.j0 = s.wrect.j0; if (s.wrect.j1 > q.j1) q.j1 = s.wrect.j1;
  }
  if (q.i0 < c.g.isd()) q.i0 = c.g.isd();
  if (q.j0 < c.g.jsd()) q.j0 = c.g.jsd();
  if (q.i1 > c.g.ied() + 1) q.i1 = c.g.ied() + 1;
  if (q.j1 > c.g.jed() + 1) q.j1 = c.g.jed() + 1;
  return q;
}

enum Mode { MODE_NL = 0, MODE_TL = 1, MODE_AD = 2 };
// Algorithmic bytes of one stage launch: every active input read once, every output written once
// (trajectory; + perturbation in TL; adjoint: trajectory inputs + output adjoints read, input adjoints
// read-modify-written), 8 B each, over the launch's cells.
template <class S>
inline double stage_bytes(const S& s, const Ctx& c, const Rect& R, int mode) {
  const double cells = double(R.i1 - R.i0 + 1) * double(R.j1 - R.j0 + 1) * c.g.ntile * (s.k1 - s.k0 + 1);
  int nin = 0;
  for (int m = 0; m < S::NIN; ++m) if (s.in[m].t) nin++;
  const double per = mode == MODE_NL ? nin + S::NOUT : mode == MODE_TL ? 2. * (nin + S::NOUT) : (nin + S::NOUT + 2. * nin);
  return 8. * per * cells;
}

#ifndef FV3LM_HOST_EMUL
// ---- HIP kernels ---------------------------------------------------------------------------
#ifndef FV3LM_BY
#define FV3LM_BY 4
#endif
constexpr int BX = 64, BY = FV3LM_BY;
// experiment switch: minimum waves per SIMD asked of the compiler for the generic stage kernels (register cap); unset = the compiler's choice
#ifdef FV3LM_STAGE_WAVES
#define FV3LM_STAGE_ATTR __attribute__((amdgpu_waves_per_eu(FV3LM_STAGE_WAVES, 8)))
#else
#define FV3LM_STAGE_ATTR
#endif
// tr: narrow column strips (face-edge stages) run with the 64 lanes of a wave along j instead of i
// XCD-aware block order.  Workgroups go to the 8 XCDs round-robin by linear id, so with the plain (x, y) order the
// blocks above and below a block — which re-read 2/3 of its stencil rows — sit on other XCDs and their private L2s
// never see the reuse (rocprofv3 FETCH_SIZE showed 1.6-1.9x the compulsory reads for the y-stencil kernels).  Here
// the blocks of one residue class mod 8 (one XCD within a plane) take a contiguous band of rows instead.
HD void xcd_block(int gx, int gy, int& bx, int& by) {
  const int n = gx * gy, l = bx + gx * by, q = n / 8, r = n % 8, res = l % 8;
  const int logical = (res < r ? res * (q + 1) : r * (q + 1) + (res - r) * q) + l / 8;
  bx = logical % gx; by = logical / gx;
}
// tr = lanes per row.  0: the normal mapping, a wave = 64 consecutive points of one row.  8 or 16 (narrow column strips of
// the face-edge stages): a wave covers 64/tr rows of tr points, so its loads touch 64/tr short row pieces instead of one
// point of 64 different rows.
HD void thread_point(const Rect& R, int tr, int bx, int by, int tx, int ty, int& i, int& j) {
  if (tr) { const int rows = BX / tr; i = R.i0 + bx * tr + tx % tr; j = R.j0 + (by * BY + ty) * rows + tx / tr; }
  else { i = R.i0 + bx * BX + tx; j = R.j0 + by * BY + ty; }
}
template <class S>
__global__ void __launch_bounds__(BX* BY) FV3LM_STAGE_ATTR k_stage_nl(S s, Ctx c, Rect R, int tr) {
  int i, j, bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by); thread_point(R, tr, bx, by, threadIdx.x, threadIdx.y, i, j);
  if (i <= R.i1 && j <= R.j1) body_nl(s, c, i, j, blockIdx.z);
}
template <class S>
__global__ void __launch_bounds__(BX* BY) FV3LM_STAGE_ATTR k_stage_tl(S s, Ctx c, Rect R, int tr) {
  int i, j, bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by); thread_point(R, tr, bx, by, threadIdx.x, threadIdx.y, i, j);
  if (i <= R.i1 && j <= R.j1) body_tl(s, c, i, j, blockIdx.z);
}
// ---- LDS-staged forward launch (nonlinear / tangent) ----------------------------------------------------------------------
// A stage opts in with `static constexpr bool LDS_FW / LDS_AD = true` (bulk stencil stages whose reads stay inside their declared
// box).  Every input with a horizontal stencil is loaded ONCE per block — tile + halo, all lanes issuing their loads
// back to back — into LDS; the stencil then reads LDS.  A 64x4 block of a 6-point y-stencil goes from 7 global loads per
// point to 2.25, of an x-stencil to 1.1.  The block is 64 x S::LDS_BY (taller for y-stencils: less halo per output row).
// opt-in per stage and per mode (measured: small-box stages with many inputs lose occupancy to their tiles in the adjoint)
template <class S, class = void> struct lds_fw { static constexpr bool value = false; };
template <class S> struct lds_fw<S, typename std::enable_if<S::LDS_FW>::type> { static constexpr bool value = true; };
template <class S, class = void> struct lds_ad { static constexpr bool value = false; };
template <class S> struct lds_ad<S, typename std::enable_if<S::LDS_AD>::type> { static constexpr bool value = true; };
template <class S>
struct TileLayout {
  HD static constexpr bool staged(int M) { return (S::box(M).di0 != 0 || S::box(M).di1 != 0 || S::box(M).dj0 != 0 || S::box(M).dj1 != 0) && S::box(M).dk0 == 0 && S::box(M).dk1 == 0; }
  HD static constexpr int w(int M) { return BX + S::box(M).di1 - S::box(M).di0; }
  HD static constexpr int h(int M) { return S::LDS_BY + S::box(M).dj1 - S::box(M).dj0; }
  HD static constexpr int off(int M) { int o = 0; for (int m = 0; m < M; ++m) if (staged(m)) o += w(m) * h(m); return o; }
  static constexpr int total = off(S::NIN);
};
#ifndef FV3LM_LDS_NLEV
#define FV3LM_LDS_NLEV 1
#endif
constexpr int LDS_NLEV = FV3LM_LDS_NLEV;    // planes per block: all their tile loads are in flight together before the first use
template <class S, bool TL>
struct AccLds {
  const S& s; const Ctx& c; int tile, k; int bi0, bj0; const double* lds;     // lds: this plane's tiles
  static constexpr unsigned want = ~0u;
  typedef typename std::conditional<TL, Dual, double>::type T;
  template <int M> HD T in(int i, int j, int dk = 0) const {
    typedef TileLayout<S> L;
    if constexpr (L::staged(M)) {
      const int e = L::off(M) + (j - (bj0 + S::box(M).dj0)) * L::w(M) + (i - (bi0 + S::box(M).di0));
      if constexpr (TL) return Dual(lds[e], lds[L::total + e]); else return lds[e];
    } else {
      const Fld& f = s.in[M];
      const size_t n = (size_t)(tile * f.nk + k - 1 + dk) * c.g.plane + c.g.idx(i, j);
      if constexpr (TL) return Dual(f.t[n], f.p ? f.p[n] : 0.0); else return f.t[n];
    }
  }
};
template <class S, bool TL, int M>
struct TileLoad {
  __device__ static void run(const S& s, const Ctx& c, int tile, int k, int bi0, int bj0, double* lds) {
    typedef TileLayout<S> L;
    if constexpr (L::staged(M)) {
      const Fld& f = s.in[M];
      const size_t base = (size_t)(tile * f.nk + k - 1) * c.g.plane;
      const int tid = threadIdx.y * BX + threadIdx.x;
      constexpr int W = L::w(M), N = W * L::h(M);
      for (int e = tid; e < N; e += BX * S::LDS_BY) {
        const int gi = bi0 + S::box(M).di0 + e % W, gj = bj0 + S::box(M).dj0 + e / W;
        if (f.t && gi >= c.g.isd() && gi <= c.g.ied() + 1 && gj >= c.g.jsd() && gj <= c.g.jed() + 1) {   // f.t null: input not handed to this instance
          const size_t n = base + c.g.idx(gi, gj);
          lds[L::off(M) + e] = f.t[n];
          if constexpr (TL) lds[L::total + L::off(M) + e] = f.p ? f.p[n] : 0.0;
        }
      }
    }
    if constexpr (M + 1 < S::NIN) TileLoad<S, TL, M + 1>::run(s, c, tile, k, bi0, bj0, lds);
  }
};
template <class S, bool TL>
__global__ void __launch_bounds__(BX* S::LDS_BY) k_stage_fw_lds(S s, Ctx c, Rect R, int nz) {
  extern __shared__ double lds[];
  constexpr int PER = TileLayout<S>::total * (TL ? 2 : 1);
  int bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by);
  const int bi0 = R.i0 + bx * BX, bj0 = R.j0 + by * S::LDS_BY, i = bi0 + threadIdx.x, j = bj0 + threadIdx.y;
  const int nl = s.k1 - s.k0 + 1;
#pragma unroll
  for (int l = 0; l < LDS_NLEV; ++l) {
    const int z = blockIdx.z * LDS_NLEV + l;
    if (z < nz) TileLoad<S, TL, 0>::run(s, c, z / nl, s.k0 + z % nl, bi0, bj0, lds + l * PER);
  }
  __syncthreads();
  if (i > R.i1 || j > R.j1) return;
#pragma unroll
  for (int l = 0; l < LDS_NLEV; ++l) {
    const int z = blockIdx.z * LDS_NLEV + l;
    if (z >= nz) break;
    const int tile = z / nl, k = s.k0 + z % nl;
    typename AccLds<S, TL>::T o[S::NOUT];
    AccLds<S, TL> a{s, c, tile, k, bi0, bj0, lds + l * PER};
    s.template eval<typename AccLds<S, TL>::T>(a, c, tile, i, j, k, o);
    for (int n = 0; n < S::NOUT; ++n)
      if (s.orect[n].has(i, j)) {
        const size_t m = (size_t)(tile * s.out[n].nk + k - 1) * c.g.plane + c.g.idx(i, j);
        if constexpr (TL) { s.out[n].t[m] = o[n].v; s.out[n].p[m] = o[n].d; } else s.out[n].t[m] = o[n];
      }
  }
}
template <class S>
__global__ void __launch_bounds__(BX* BY) FV3LM_STAGE_ATTR k_stage_ad(S s, Ctx c, Rect R, Rect Q, int nkmax, int tr) {
  int i, j, bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by); thread_point(Q, tr, bx, by, threadIdx.x, threadIdx.y, i, j);
  if (i <= Q.i1 && j <= Q.j1) ad_point(s, c, R, i, j, blockIdx.z, nkmax);
}
template <class S>
__global__ void __launch_bounds__(64) k_stage_ad_alias(S s, Ctx c, Rect R) {
  if ((int)threadIdx.x >= 4 * c.g.ng * c.g.ng) return;
  int i, j;
  corner_block_point(c.g, blockIdx.x, threadIdx.x, i, j);
  if (!c.g.in_plane(i, j)) return;            // a sub-face tile holds at most one face corner
  AdAliasLoop<S, 0>::run(s, c, R, i, j, blockIdx.y);
}
inline int strip_tr(const Rect& R) {
  const int w = R.i1 - R.i0 + 1, h = R.j1 - R.j0 + 1;
  if (h < BX / 2) return 0;
  return w <= 8 ? 8 : w <= 16 ? 16 : 0;
}
inline dim3 grid_for(const Rect& R, int nz, int tr = 0) {
  if (tr) { const int rows = BY * (BX / tr); return dim3((R.i1 - R.i0 + tr) / tr, (R.j1 - R.j0 + rows) / rows, nz); }
  return dim3((R.i1 - R.i0 + BX) / BX, (R.j1 - R.j0 + BY) / BY, nz);
}
// ---- LDS-staged adjoint launch (joint gather form, single level class, no vertical stencil, no corner aliases) -----
// The thread that owns input point p evaluates the stage at the outputs o = p - u (u in the union box U of the inputs);
// those evaluations read input M at p + (b_M - u) and the output adjoints at p - u.  Slot sl < NIN is the trajectory tile
// of input sl, slot NIN + n the adjoint tile of output n; each global element is loaded once per block.
template <class S>
struct TileLayoutAd {
  static constexpr Box U = class_box<S, 0>();
  static constexpr int NS = S::NIN + S::NOUT;
  HD static constexpr int lo_i(int sl) { return (sl < S::NIN ? S::box(sl).di0 : 0) - U.di1; }
  HD static constexpr int hi_i(int sl) { return (sl < S::NIN ? S::box(sl).di1 : 0) - U.di0; }
  HD static constexpr int lo_j(int sl) { return (sl < S::NIN ? S::box(sl).dj0 : 0) - U.dj1; }
  HD static constexpr int hi_j(int sl) { return (sl < S::NIN ? S::box(sl).dj1 : 0) - U.dj0; }
  HD static constexpr int w(int sl) { return BX + hi_i(sl) - lo_i(sl); }
  HD static constexpr int h(int sl) { return S::LDS_BY + hi_j(sl) - lo_j(sl); }
  HD static constexpr int off(int sl) { int o = 0; for (int m = 0; m < sl; ++m) o += w(m) * h(m); return o; }
  static constexpr int total = off(NS);
  HD static constexpr bool ok() {
    for (int m = 0; m < S::NIN; ++m) if (kclass_of((const S*)nullptr, m) != 0 || S::box(m).dk0 != 0 || S::box(m).dk1 != 0) return false;
    return S::NALIAS == 0 && joint_ad((const S*)nullptr);
  }
};
template <class S>
struct AccADVLds {
  const S& s; const Ctx& c; int tile, k; int si, sj, sk; unsigned mask, want; int bi0, bj0; const double* lds;
  template <int M> HD DualV<S::NIN> in(int i, int j, int dk = 0) const {
    typedef TileLayoutAd<S> L;
    DualV<S::NIN> r(lds[L::off(M) + (j - (bj0 + L::lo_j(M))) * L::w(M) + (i - (bi0 + L::lo_i(M)))]);
    r.d[M] = (((mask >> M) & 1u) && i == si && j == sj && k + dk == sk) ? 1.0 : 0.0;
    return r;
  }
};
template <class S, int SL>
struct TileLoadAd {
  __device__ static void run(const S& s, const Ctx& c, int tile, int kk, int bi0, int bj0, double* lds) {
    typedef TileLayoutAd<S> L;
    const Fld& f = SL < S::NIN ? s.in[SL < S::NIN ? SL : 0] : s.out[SL < S::NIN ? 0 : SL - S::NIN];
    const double* src = SL < S::NIN ? f.t : f.p;
    const size_t base = (size_t)(tile * f.nk + kk - 1) * c.g.plane;
    const int tid = threadIdx.y * BX + threadIdx.x;
    constexpr int W = L::w(SL), N = W * L::h(SL);
    for (int e = tid; e < N; e += BX * S::LDS_BY) {
      const int gi = bi0 + L::lo_i(SL) + e % W, gj = bj0 + L::lo_j(SL) + e / W;
      double v = 0.;
      if (src && gi >= c.g.isd() && gi <= c.g.ied() + 1 && gj >= c.g.jsd() && gj <= c.g.jed() + 1) v = src[base + c.g.idx(gi, gj)];
      lds[L::off(SL) + e] = v;
    }
    if constexpr (SL + 1 < L::NS) TileLoadAd<S, SL + 1>::run(s, c, tile, kk, bi0, bj0, lds);
  }
};
template <class S>
__global__ void __launch_bounds__(BX* S::LDS_BY) k_stage_ad_lds(S s, Ctx c, Rect R, Rect Q) {
  extern __shared__ double lds[];
  typedef TileLayoutAd<S> L;
  constexpr int N = S::NIN;
  int bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by);
  const int bi0 = Q.i0 + bx * BX, bj0 = Q.j0 + by * S::LDS_BY, i = bi0 + threadIdx.x, j = bj0 + threadIdx.y;
  const int nk = s.in[0].nk, tile = blockIdx.z / nk, kk = 1 + blockIdx.z % nk, k = kk;
  const bool lev_ok = (k >= s.k0 && k <= s.k1);
  if (lev_ok) TileLoadAd<S, 0>::run(s, c, tile, kk, bi0, bj0, lds);
  __syncthreads();
  if (i > Q.i1 || j > Q.j1) return;
  if (!lev_ok && !s.wmask) return;
  constexpr Box ub = L::U;
  const bool inreach = lev_ok && !(i < R.i0 + ub.di0 || i > R.i1 + ub.di1 || j < R.j0 + ub.dj0 || j > R.j1 + ub.dj1);
  if (!inreach && !s.wmask) return;
  double acc[N];
#pragma unroll
  for (int m = 0; m < N; ++m) acc[m] = 0.0;
  if (inreach) {
#pragma unroll
  for (int dj = ub.dj0; dj <= ub.dj1; ++dj) {
    const int oj = j - dj;
    if (oj < R.j0 || oj > R.j1) continue;
#pragma unroll
    for (int di = ub.di0; di <= ub.di1; ++di) {
      const int oi = i - di;
      if (oi < R.i0 || oi > R.i1) continue;
      unsigned mask = 0u, want = 0u;
#pragma unroll
      for (int m = 0; m < N; ++m) {
        const Box b = S::box(m);
        if (S::wants(m) && di >= b.di0 && di <= b.di1 && dj >= b.dj0 && dj <= b.dj1 && S::uses(m, di, dj, 0)) { mask |= 1u << m; want |= S::wants(m); }
      }
      if (!mask) continue;
      AccADVLds<S> a{s, c, tile, k, i, j, kk, mask, want, bi0, bj0, lds};
      DualV<N> o[S::NOUT];
      s.template eval<DualV<N>>(a, c, tile, oi, oj, k, o);
#pragma unroll
      for (int n = 0; n < S::NOUT; ++n)
        if (((want >> n) & 1u) && s.orect[n].has(oi, oj)) {
          const double oa = lds[L::off(N + n) + (oj - (bj0 + L::lo_j(N + n))) * L::w(N + n) + (oi - (bi0 + L::lo_i(N + n)))];
#pragma unroll
          for (int m = 0; m < N; ++m) if (((mask >> m) & 1u) && ((S::wants(m) >> n) & 1u)) acc[m] += o[n].d[m] * oa;
        }
    }
  }
  }
#pragma unroll
  for (int m = 0; m < N; ++m)
    if (S::wants(m) && s.in[m].p) {        // store: see body_ad_joint
      const Box b = S::box(m);
      const bool in_m = lev_ok && !(i < R.i0 + b.di0 || i > R.i1 + b.di1 || j < R.j0 + b.dj0 || j > R.j1 + b.dj1);
      double* q = &s.in[m].p[(size_t)(tile * nk + kk - 1) * c.g.plane + c.g.idx(i, j)];
      if ((s.wmask >> m) & 1u) *q = in_m ? acc[m] : 0.0;
      else if (in_m) *q += acc[m];
    }
}
template <class S, bool TL>
inline void launch_fw_lds(Exec& ex, const S& s, const Ctx& c, const Rect& R) {
  const int nz = c.g.ntile * (s.k1 - s.k0 + 1);
  const dim3 g((R.i1 - R.i0 + BX) / BX, (R.j1 - R.j0 + S::LDS_BY) / S::LDS_BY, (nz + LDS_NLEV - 1) / LDS_NLEV);
  hipLaunchKernelGGL((k_stage_fw_lds<S, TL>), g, dim3(BX, S::LDS_BY), TileLayout<S>::total * (TL ? 16 : 8) * LDS_NLEV, ex.stream, s, c, R, nz);
}
template <class S>
void run_nl(Exec& ex, const S& s, const Ctx& c) {
  Rect R = rect_union(s.orect, S::NOUT);
  ex.mark_begin(S::name(), ".nl", stage_bytes(s, c, R, MODE_NL));
  hipLaunchKernelGGL(k_stage_nl<S>, grid_for(R, c.g.ntile * (s.k1 - s.k0 + 1), strip_tr(R)), dim3(BX, BY), 0, ex.stream, s, c, R, strip_tr(R));
  ex.mark_end();
  ex.launches++;
}
template <class S>
void run_tl(Exec& ex, const S& s, const Ctx& c) {
  Rect R = rect_union(s.orect, S::NOUT);
  ex.mark_begin(S::name(), ".tl", stage_bytes(s, c, R, MODE_TL));
  if constexpr (lds_fw<S>::value) if constexpr (TileLayout<S>::total > 0 && TileLayout<S>::total * 16 <= 60000) { if (strip_tr(R) == 0) {
    launch_fw_lds<S, true>(ex, s, c, R);
    ex.mark_end(); ex.launches++; return; } }
  hipLaunchKernelGGL(k_stage_tl<S>, grid_for(R, c.g.ntile * (s.k1 - s.k0 + 1), strip_tr(R)), dim3(BX, BY), 0, ex.stream, s, c, R, strip_tr(R));
  ex.mark_end();
  ex.launches++;
}
template <class S>
void run_ad(Exec& ex, const S& s, const Ctx& c) {
  Rect R = rect_union(s.orect, S::NOUT);
  Rect Q = ad_input_rect(s, c, R);
  int nkmax = 0;
  for (int m = 0; m < S::NIN; ++m) if (s.in[m].nk > nkmax) nkmax = s.in[m].nk;
  ex.mark_begin(S::name(), ".ad", stage_bytes(s, c, R, MODE_AD));
  if constexpr (lds_ad<S>::value) if constexpr (TileLayoutAd<S>::ok() && TileLayoutAd<S>::total * 8 <= 60000) { if (strip_tr(Q) == 0) {
    bool same = true;
    for (int m = 0; m < S::NIN; ++m) if (s.in[m].nk != s.in[0].nk) same = false;
    if (same) {
      const dim3 gq((Q.i1 - Q.i0 + BX) / BX, (Q.j1 - Q.j0 + S::LDS_BY) / S::LDS_BY, c.g.ntile * s.in[0].nk);
      hipLaunchKernelGGL(k_stage_ad_lds<S>, gq, dim3(BX, S::LDS_BY), TileLayoutAd<S>::total * 8, ex.stream, s, c, R, Q);
      ex.mark_end(); ex.launches++; return; } } }
  hipLaunchKernelGGL(k_stage_ad<S>, grid_for(Q, c.g.ntile * nkmax, strip_tr(Q)), dim3(BX, BY), 0, ex.stream, s, c, R, Q, nkmax, strip_tr(Q));
  ex.mark_end();
  ex.launches++;
  if constexpr (S::NALIAS > 0) if (c.g.face) {
    ex.mark_begin(S::name(), ".ad_corner", 0.);
    hipLaunchKernelGGL(k_stage_ad_alias<S>, dim3(4, c.g.ntile * nkmax), dim3(64), 0, ex.stream, s, c, R);
    ex.mark_end();
    ex.launches++;
  }
}
// ---- several instances of one stage in one launch (the edge strips of a face: same stage type, different output rectangles) ----
// A strip launch covers ~1 % of a face and is launch-latency-bound; the four strips of a stage go out together.  Adjoint: the
// input regions of the strips overlap at the face corners, so each input point is owned by the first strip whose region
// contains it and that thread runs the gather of every strip that reaches the point, in strip order (no race, fixed order).
constexpr int MAXSTRIP = 4;
template <class S>
struct Multi { S s[MAXSTRIP]; Rect R[MAXSTRIP]; int n, boff[MAXSTRIP + 1], gx[MAXSTRIP], tr[MAXSTRIP]; };
template <class S, bool TL>
__global__ void __launch_bounds__(BX* BY) k_stage_multi_fw(Multi<S> m, Ctx c) {
  int b = blockIdx.x, r = 0;
  while (r + 1 < m.n && b >= m.boff[r + 1]) ++r;
  b -= m.boff[r];
  int i, j; thread_point(m.R[r], m.tr[r], b % m.gx[r], b / m.gx[r], threadIdx.x, threadIdx.y, i, j);
  if (i > m.R[r].i1 || j > m.R[r].j1) return;
  if (TL) body_tl(m.s[r], c, i, j, blockIdx.z); else body_nl(m.s[r], c, i, j, blockIdx.z);
}
template <class S>
__global__ void __launch_bounds__(BX* BY) k_stage_multi_ad(Multi<S> m, Rect Q0, Rect Q1, Rect Q2, Rect Q3, Ctx c, int nkmax) {
  const Rect Q[MAXSTRIP] = {Q0, Q1, Q2, Q3};
  int b = blockIdx.x, r = 0;
  while (r + 1 < m.n && b >= m.boff[r + 1]) ++r;
  b -= m.boff[r];
  int i, j; thread_point(Q[r], m.tr[r], b % m.gx[r], b / m.gx[r], threadIdx.x, threadIdx.y, i, j);
  if (i > Q[r].i1 || j > Q[r].j1) return;
  for (int r2 = 0; r2 < r; ++r2) if (Q[r2].has(i, j)) return;          // owned by an earlier strip
  // static indices (unrolled): a dynamically indexed stage struct would be copied to scratch
  if (0 >= r && 0 < m.n && Q0.has(i, j)) ad_point(m.s[0], c, m.R[0], i, j, blockIdx.z, nkmax);
  if (1 >= r && 1 < m.n && Q1.has(i, j)) ad_point(m.s[1], c, m.R[1], i, j, blockIdx.z, nkmax);
  if (2 >= r && 2 < m.n && Q2.has(i, j)) ad_point(m.s[2], c, m.R[2], i, j, blockIdx.z, nkmax);
  if (3 >= r && 3 < m.n && Q3.has(i, j)) ad_point(m.s[3], c, m.R[3], i, j, blockIdx.z, nkmax);
}
// Two strips whose input regions are disjoint (south + north, west + east of a face wider than twice the stencil reach) share one adjoint
// launch: one gather per thread, as in the single-strip launch -- none of the register cost of the four-strip merge below.
template <class S>
struct Pair { S s[2]; Rect R[2], Q[2]; int boff[3], gx[2], tr[2]; };
template <class S>
__global__ void __launch_bounds__(BX* BY) k_stage_pair_ad(Pair<S> m, Ctx c, int nkmax) {
  int b = blockIdx.x;
  if (b < m.boff[1]) {
    int i, j; thread_point(m.Q[0], m.tr[0], b % m.gx[0], b / m.gx[0], threadIdx.x, threadIdx.y, i, j);
    if (i <= m.Q[0].i1 && j <= m.Q[0].j1) ad_point(m.s[0], c, m.R[0], i, j, blockIdx.z, nkmax);
  } else {
    b -= m.boff[1];
    int i, j; thread_point(m.Q[1], m.tr[1], b % m.gx[1], b / m.gx[1], threadIdx.x, threadIdx.y, i, j);
    if (i <= m.Q[1].i1 && j <= m.Q[1].j1) ad_point(m.s[1], c, m.R[1], i, j, blockIdx.z, nkmax);
  }
}
inline bool rects_disjoint(const Rect& a, const Rect& b) { return a.i1 < b.i0 || b.i1 < a.i0 || a.j1 < b.j0 || b.j1 < a.j0; }
// The merged adjoint (four inlined gathers) measured faster for some stages and much slower for others (register
// pressure): opt-in per stage with `static constexpr bool MERGE_AD_STRIPS = true`; the forward modes always merge.
template <class S, class = void> struct merge_ad_strips { static constexpr bool value = false; };
template <class S> struct merge_ad_strips<S, typename std::enable_if<S::MERGE_AD_STRIPS>::type> { static constexpr bool value = true; };
template <class S>
void run_multi(Exec& ex, int mode, const S* s0, int n, const Ctx& c) {
  if (mode == MODE_AD && !merge_ad_strips<S>::value && n <= MAXSTRIP && !std::getenv("FV3LM_NO_PAIR_STRIPS")) {
    // strips in pairs with disjoint input regions (add_face hands them over as south, north, west, east); a strip without a partner alone
    bool done[MAXSTRIP] = {false, false, false, false};
    for (int r = 0; r < n; ++r) {
      if (done[r]) continue;
      S a = s0[r];
      if (ex.no_wmask) a.wmask = 0;
      for (int k = 0; k < S::NIN; ++k) a.in[k] = ex.sh(a.in[k]);
      for (int k = 0; k < S::NOUT; ++k) a.out[k] = ex.sh(a.out[k]);
      const Rect Ra = rect_union(a.orect, S::NOUT), Qa = ad_input_rect(a, c, Ra);
      int partner = -1; S b = a; Rect Rb = Ra, Qb = Qa;
      for (int q = r + 1; q < n && partner < 0; ++q) {
        if (done[q]) continue;
        S t = s0[q];
        if (ex.no_wmask) t.wmask = 0;
        for (int k = 0; k < S::NIN; ++k) t.in[k] = ex.sh(t.in[k]);
        for (int k = 0; k < S::NOUT; ++k) t.out[k] = ex.sh(t.out[k]);
        const Rect Rt = rect_union(t.orect, S::NOUT), Qt = ad_input_rect(t, c, Rt);
        if (rects_disjoint(Qa, Qt) && a.wmask == 0 && t.wmask == 0) { partner = q; b = t; Rb = Rt; Qb = Qt; }
      }
      if (partner < 0) { run(ex, mode, s0[r], c); done[r] = true; continue; }
      done[r] = done[partner] = true;
      Pair<S> m; m.s[0] = a; m.s[1] = b; m.R[0] = Ra; m.R[1] = Rb; m.Q[0] = Qa; m.Q[1] = Qb;
      int nkmax = 0;
      for (int k = 0; k < S::NIN; ++k) if (a.in[k].nk > nkmax) nkmax = a.in[k].nk;
      m.boff[0] = 0;
      for (int q = 0; q < 2; ++q) { m.tr[q] = strip_tr(m.Q[q]); const dim3 gd = grid_for(m.Q[q], 1, m.tr[q]); m.gx[q] = gd.x; m.boff[q + 1] = m.boff[q] + gd.x * gd.y; }
      ex.mark_begin(S::name(), ".ad", stage_bytes(a, c, Ra, MODE_AD) + stage_bytes(b, c, Rb, MODE_AD));
      hipLaunchKernelGGL(k_stage_pair_ad<S>, dim3(m.boff[2], 1, c.g.ntile * nkmax), dim3(BX, BY), 0, ex.stream, m, c, nkmax);
      ex.mark_end();
      ex.launches++;
      if constexpr (S::NALIAS > 0) if (c.g.face)
        for (int q = 0; q < 2; ++q) {
          ex.mark_begin(S::name(), ".ad_corner", 0.);
          hipLaunchKernelGGL(k_stage_ad_alias<S>, dim3(4, c.g.ntile * nkmax), dim3(64), 0, ex.stream, m.s[q], c, m.R[q]);
          ex.mark_end();
          ex.launches++;
        }
    }
    return;
  }
  if (n > MAXSTRIP || (mode == MODE_AD && !merge_ad_strips<S>::value)) { for (int r = 0; r < n; ++r) run(ex, mode, s0[r], c); return; }
  Multi<S> m; m.n = n;
  Rect Q[MAXSTRIP]; int nkmax = 0; double bytes = 0.;
  for (int r = 0; r < MAXSTRIP; ++r) {
    const int q = r < n ? r : n - 1;
    S t = s0[q];
    for (int k = 0; k < S::NIN; ++k) t.in[k] = ex.sh(t.in[k]);
    for (int k = 0; k < S::NOUT; ++k) t.out[k] = ex.sh(t.out[k]);
    m.s[r] = t; m.R[r] = rect_union(t.orect, S::NOUT); Q[r] = ad_input_rect(t, c, m.R[r]);
    if (r < n) bytes += stage_bytes(t, c, m.R[r], mode);
  }
  for (int k = 0; k < S::NIN; ++k) if (m.s[0].in[k].nk > nkmax) nkmax = m.s[0].in[k].nk;
  const int nz = mode == MODE_AD ? c.g.ntile * nkmax : c.g.ntile * (m.s[0].k1 - m.s[0].k0 + 1);
  m.boff[0] = 0;
  for (int r = 0; r < n; ++r) {
    const Rect& X = mode == MODE_AD ? Q[r] : m.R[r];
    m.tr[r] = strip_tr(X);
    const dim3 gd = grid_for(X, 1, m.tr[r]);
    m.gx[r] = gd.x; m.boff[r + 1] = m.boff[r] + gd.x * gd.y;
  }
  ex.mark_begin(S::name(), mode == MODE_NL ? ".nl" : mode == MODE_TL ? ".tl" : ".ad", bytes);
  const dim3 grid(m.boff[n], 1, nz);
  if (mode == MODE_NL) hipLaunchKernelGGL((k_stage_multi_fw<S, false>), grid, dim3(BX, BY), 0, ex.stream, m, c);
  else if (mode == MODE_TL) hipLaunchKernelGGL((k_stage_multi_fw<S, true>), grid, dim3(BX, BY), 0, ex.stream, m, c);
  else if constexpr (merge_ad_strips<S>::value) hipLaunchKernelGGL(k_stage_multi_ad<S>, grid, dim3(BX, BY), 0, ex.stream, m, Q[0], Q[1], Q[2], Q[3], c, nkmax);
  ex.mark_end();
  ex.launches++;
  if constexpr (S::NALIAS > 0) if (mode == MODE_AD && c.g.face)
    for (int r = 0; r < n; ++r) {
      ex.mark_begin(S::name(), ".ad_corner", 0.);
      hipLaunchKernelGGL(k_stage_ad_alias<S>, dim3(4, c.g.ntile * nkmax), dim3(64), 0, ex.stream, m.s[r], c, m.R[r]);
      ex.mark_end();
      ex.launches++;
    }
}
// generic per-point functor launch: f(i, j, z)
template <class F>
__global__ void __launch_bounds__(BX* BY) k_points(F f, Rect R) {
  const int i = R.i0 + blockIdx.x * BX + threadIdx.x, j = R.j0 + blockIdx.y * BY + threadIdx.y;
  if (i <= R.i1 && j <= R.j1) f(i, j, (int)blockIdx.z);
}
// one row of points (exchange tables, per-level reductions): the whole block along i
template <class F>
__global__ void __launch_bounds__(BX* BY) k_points_row(F f, Rect R) {
  const int i = R.i0 + blockIdx.x * (BX * BY) + threadIdx.x;
  if (i <= R.i1) f(i, R.j0, (int)blockIdx.z);
}
template <class F>
void for_points(Exec& ex, const Rect& R, int nz, const F& f, const char* tag = "points", double bytes = 0.) {
  if (nz <= 0) return;
  ex.mark_begin(tag, "", bytes);
  if (R.j0 == R.j1) hipLaunchKernelGGL(k_points_row<F>, dim3((R.i1 - R.i0 + BX * BY) / (BX * BY), 1, nz), dim3(BX * BY), 0, ex.stream, f, R);
  else hipLaunchKernelGGL(k_points<F>, grid_for(R, nz), dim3(BX, BY), 0, ex.stream, f, R);
  ex.mark_end();
  ex.launches++;
}
#else
// ---- host emulation (tests only) -----------------------------------------------------------
template <class S>
void run_nl(Exec& ex, const S& s, const Ctx& c) {
  Rect R = rect_union(s.orect, S::NOUT);
  for (int z = 0; z < c.g.ntile * (s.k1 - s.k0 + 1); ++z)
    for (int j = R.j0; j <= R.j1; ++j)
      for (int i = R.i0; i <= R.i1; ++i) body_nl(s, c, i, j, z);
  ex.launches++;
}
template <class S>
void run_tl(Exec& ex, const S& s, const Ctx& c) {
  Rect R = rect_union(s.orect, S::NOUT);
  const int nl = s.k1 - s.k0 + 1;
  for (int z = 0; z < c.g.ntile * nl; ++z)
    for (int j = R.j0; j <= R.j1; ++j)
      for (int i = R.i0; i <= R.i1; ++i) {
        if (ex.check_boxes) {
          Dual o[S::NOUT];
          AccChk<S> a{s, c, z / nl, s.k0 + z % nl, i, j};
          s.template eval<Dual>(a, c, z / nl, i, j, s.k0 + z % nl, o);
        }
        body_tl(s, c, i, j, z);
      }
  ex.launches++;
}
template <class S>
void run_ad(Exec& ex, const S& s, const Ctx& c) {
  Rect R = rect_union(s.orect, S::NOUT);
  Rect Q = ad_input_rect(s, c, R);
  int nkmax = 0;
  for (int m = 0; m < S::NIN; ++m) if (s.in[m].nk > nkmax) nkmax = s.in[m].nk;
  for (int z = 0; z < c.g.ntile * nkmax; ++z)
    for (int j = Q.j0; j <= Q.j1; ++j)
      for (int i = Q.i0; i <= Q.i1; ++i) ad_point(s, c, R, i, j, z, nkmax);
  if constexpr (S::NALIAS > 0) if (c.g.face)
    for (int z = 0; z < c.g.ntile * nkmax; ++z)
      for (int cn = 0; cn < 4; ++cn)
        for (int n = 0; n < 4 * c.g.ng * c.g.ng; ++n) {
          int i, j;
          corner_block_point(c.g, cn, n, i, j);
          if (!c.g.in_plane(i, j)) continue;
          AdAliasLoop<S, 0>::run(s, c, R, i, j, z);
        }
  ex.launches++;
}
template <class F>
void for_points(Exec& ex, const Rect& R, int nz, const F& f, const char* tag = "points", double bytes = 0.) {
  for (int z = 0; z < nz; ++z)
    for (int j = R.j0; j <= R.j1; ++j)
      for (int i = R.i0; i <= R.i1; ++i) f(i, j, z);
  ex.launches++;
}
#endif

template <class S>
void run(Exec& ex, int mode, const S& s0, const Ctx& c);
#ifdef FV3LM_HOST_EMUL
template <class S>
void run_multi(Exec& ex, int mode, const S* s0, int n, const Ctx& c) { for (int r = 0; r < n; ++r) run(ex, mode, s0[r], c); }
#endif
template <class S>
void run(Exec& ex, int mode, const S& s0, const Ctx& c) {
  S s = s0;
  if (ex.no_wmask) s.wmask = 0;
  for (int m = 0; m < S::NIN; ++m) s.in[m] = ex.sh(s.in[m]);
  for (int n = 0; n < S::NOUT; ++n) s.out[n] = ex.sh(s.out[n]);
  if (mode == MODE_NL) run_nl(ex, s, c);
  else if (mode == MODE_TL) run_tl(ex, s, c);
  else run_ad(ex, s, c);
}

}  // namespace fv3
