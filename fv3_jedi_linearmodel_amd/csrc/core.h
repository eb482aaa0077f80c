// fv3lm-hip core: scalar types, field/geometry descriptors and the generic stage launchers.
//
// Design: the TL/AD dycore is expressed as a sequence of *stages*.  A stage is a
// small struct that computes its outputs at one grid point (i,j,k) from its inputs at fixed stencil
// offsets — the nonlinear formula only, written once on a generic scalar T.  Three generic kernels
// run a stage over all levels and tiles:
//   nonlinear  T = double          (trajectory recompute of the adjoint sweep; FV_DYNAMICS_FWD role)
//   tangent    T = Dual            (value + tangent in registers; the *_TLM role)
//   adjoint    gather form: the thread that owns input point p sums dOut(o)/dIn(p) * out_ad(o) over
//              the outputs o in the stage's declared stencil box, each local derivative obtained by
//              seeding a Dual at p — no atomics, no tape (the *_BWD role; replaces adStack.c).
// Branches test trajectory values only, like the Tapenade code (`IF (c(i,j) .GT. 0.)`).
//
// This header compiles two ways: with hipcc for gfx950 (the product), and with g++ under
// -DFV3LM_HOST_EMUL (a test-only build that runs the same stage code in host loops so that the
// stage logic can be checked against the oracle on machines without a GPU; it is never loaded by
// the package).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifdef FV3LM_HOST_EMUL
#define HD inline
#define DEV inline
#else
#include <hip/hip_runtime.h>
#define HD __host__ __device__ inline
#define DEV __device__ inline
#endif

// The HIP build is split into translation units that hipcc compiles side by side (__graft_entry__.build(): -DFV3LM_SPLIT_BUILD; tu_*.cpp
// each define one FV3LM_IMPL_* and hold that module's kernels): a module's launch functions are then ordinary external functions.  The
// test-only host-emulation build is one translation unit and keeps them inline.
#ifdef FV3LM_SPLIT_BUILD
#define FV3LM_LINK
#else
#define FV3LM_LINK inline
#endif

namespace fv3 {

// ------------------------------------------------------------------ scalars
struct Dual {
  double v, d;
  HD Dual() : v(0.0), d(0.0) {}
  HD Dual(double v_) : v(v_), d(0.0) {}
  HD Dual(double v_, double d_) : v(v_), d(d_) {}
};
HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
HD Dual operator/(Dual a, Dual b) { double r = 1.0 / b.v, q = a.v * r; return Dual(q, (a.d - q * b.d) * r); }
HD Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
HD Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
HD Dual operator+(double a, Dual b) { return Dual(a + b.v, b.d); }
HD Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
HD Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.d); }
HD Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
HD Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
HD Dual operator/(Dual a, double b) { double r = 1.0 / b; return Dual(a.v * r, a.d * r); }
HD Dual operator/(double a, Dual b) { double q = a / b.v; return Dual(q, -q * b.d / b.v); }
HD Dual dlog(Dual a) { return Dual(log(a.v), a.d / a.v); }
HD Dual dexp(Dual a) { double e = exp(a.v); return Dual(e, a.d * e); }
HD Dual dsqrt(Dual a) { double s = sqrt(a.v); return Dual(s, a.v == 0.0 ? 0.0 : a.d / (2.0 * s)); }
HD double dlog(double a) { return log(a); }
HD double dexp(double a) { return exp(a); }
HD double dsqrt(double a) { return sqrt(a); }
HD double val(double a) { return a; }
HD double val(const Dual& a) { return a.v; }
// value of a, tangent of b (trajectory and perturbation evaluated with different coefficients,
// sw_core_tlm.F90:2436-2452)
HD double combine(double a, double) { return a; }
HD Dual combine(const Dual& a, const Dual& b) { return Dual(a.v, b.d); }
// the value replaced, the tangents kept (split schemes: the nonlinear routine's result over the tangent routine's, sw_core_tlm.F90:1987-2002)
HD void set_val(double& a, double x) { a = x; }
HD void set_val(Dual& a, double x) { a.v = x; }

// Forward-mode scalar with N tangent directions: the joint adjoint launcher (exec.h) seeds one direction per stage
// input, so one evaluation of an output point yields its derivatives with respect to every input at once.  All loops
// have constant trip counts; directions that are never seeded are compile-time zeros and fold away.
template <int N>
struct DualV {
  double v; double d[N];
  HD DualV() : v(0.0) { for (int n = 0; n < N; ++n) d[n] = 0.0; }
  HD DualV(double v_) : v(v_) { for (int n = 0; n < N; ++n) d[n] = 0.0; }
};
template <int N> HD DualV<N> operator+(const DualV<N>& a, const DualV<N>& b) { DualV<N> r; r.v = a.v + b.v; for (int n = 0; n < N; ++n) r.d[n] = a.d[n] + b.d[n]; return r; }
template <int N> HD DualV<N> operator-(const DualV<N>& a, const DualV<N>& b) { DualV<N> r; r.v = a.v - b.v; for (int n = 0; n < N; ++n) r.d[n] = a.d[n] - b.d[n]; return r; }
template <int N> HD DualV<N> operator*(const DualV<N>& a, const DualV<N>& b) { DualV<N> r; r.v = a.v * b.v; for (int n = 0; n < N; ++n) r.d[n] = a.d[n] * b.v + a.v * b.d[n]; return r; }
template <int N> HD DualV<N> operator/(const DualV<N>& a, const DualV<N>& b) { DualV<N> r; const double ib = 1.0 / b.v, q = a.v * ib; r.v = q; for (int n = 0; n < N; ++n) r.d[n] = (a.d[n] - q * b.d[n]) * ib; return r; }
template <int N> HD DualV<N> operator-(const DualV<N>& a) { DualV<N> r; r.v = -a.v; for (int n = 0; n < N; ++n) r.d[n] = -a.d[n]; return r; }
template <int N> HD DualV<N> operator+(const DualV<N>& a, double b) { DualV<N> r = a; r.v = a.v + b; return r; }
template <int N> HD DualV<N> operator+(double a, const DualV<N>& b) { DualV<N> r = b; r.v = a + b.v; return r; }
template <int N> HD DualV<N> operator-(const DualV<N>& a, double b) { DualV<N> r = a; r.v = a.v - b; return r; }
template <int N> HD DualV<N> operator-(double a, const DualV<N>& b) { DualV<N> r; r.v = a - b.v; for (int n = 0; n < N; ++n) r.d[n] = -b.d[n]; return r; }
template <int N> HD DualV<N> operator*(const DualV<N>& a, double b) { DualV<N> r; r.v = a.v * b; for (int n = 0; n < N; ++n) r.d[n] = a.d[n] * b; return r; }
template <int N> HD DualV<N> operator*(double a, const DualV<N>& b) { DualV<N> r; r.v = a * b.v; for (int n = 0; n < N; ++n) r.d[n] = a * b.d[n]; return r; }
template <int N> HD DualV<N> operator/(const DualV<N>& a, double b) { const double ib = 1.0 / b; DualV<N> r; r.v = a.v * ib; for (int n = 0; n < N; ++n) r.d[n] = a.d[n] * ib; return r; }
template <int N> HD DualV<N> operator/(double a, const DualV<N>& b) { DualV<N> r; const double q = a / b.v; r.v = q; for (int n = 0; n < N; ++n) r.d[n] = -q * b.d[n] / b.v; return r; }
template <int N> HD DualV<N> dlog(const DualV<N>& a) { DualV<N> r; r.v = log(a.v); for (int n = 0; n < N; ++n) r.d[n] = a.d[n] / a.v; return r; }
template <int N> HD DualV<N> dexp(const DualV<N>& a) { DualV<N> r; const double e = exp(a.v); r.v = e; for (int n = 0; n < N; ++n) r.d[n] = a.d[n] * e; return r; }
template <int N> HD DualV<N> dsqrt(const DualV<N>& a) { DualV<N> r; const double sq = sqrt(a.v); r.v = sq; for (int n = 0; n < N; ++n) r.d[n] = a.v == 0.0 ? 0.0 : a.d[n] / (2.0 * sq); return r; }
template <int N> HD double val(const DualV<N>& a) { return a.v; }
template <int N> HD DualV<N> combine(const DualV<N>& a, const DualV<N>& b) { DualV<N> r = b; r.v = a.v; return r; }
template <int N> HD void set_val(DualV<N>& a, double x) { a.v = x; }

// ------------------------------------------------------------------ geometry
// All 2-D planes share one padded index map: (isd:ied+1, jsd:jed+1), isd = is-ng, ng = 3.
// Indices are the reference's GLOBAL face indices: a tile is the window is..ie x js..je of a face of nx x ny cells (the whole face:
// is = js = 1, ie = nx; a sub-face tile of `layout` > 1x1, fv_control_nlm.F90:556: any window).  Every `is .EQ. 1`, `i .EQ. npx`, corner
// test of the reference compares global indices with nx, ny; storage and loops use the window.
struct Geom {
  int nx, ny;          // cells per FACE edge (npx-1, npy-1); the doubly-periodic tile: the tile itself
  int ng, npz, ntile;
  int face;            // 1: tiles of cube faces (edge and corner branches where a tile touches a cube edge)
  int i0 = 1, j0 = 1;  // is, js of the tile(s) the launch covers (one class of resident tiles at a time: dycore.h set_class)
  int tx = 0, ty = 0;  // cells of a tile (ie-is+1, je-js+1)
  int pi, pj;          // padded plane dims
  int plane;           // pi*pj
  HD int idx(int i, int j) const { return (j - j0 + ng) * pi + (i - i0 + ng); }   // global i, j
  HD int is() const { return i0; }
  HD int ie() const { return i0 + tx - 1; }
  HD int js() const { return j0; }
  HD int je() const { return j0 + ty - 1; }
  HD int isd() const { return i0 - ng; }
  HD int ied() const { return i0 + tx - 1 + ng; }
  HD int jsd() const { return j0 - ng; }
  HD int jed() const { return j0 + ty - 1 + ng; }
  HD bool in_plane(int i, int j) const { return i >= isd() && i <= ied() + 1 && j >= jsd() && j <= jed() + 1; }
  HD bool whole_face() const { return i0 == 1 && j0 == 1 && tx == nx && ty == ny; }
};

// A 3-D field: trajectory and perturbation(TL)/adjoint(AD) buffers, [ntile*nk][pj][pi].
struct Fld {
  double* t = nullptr;
  double* p = nullptr;
  int nk = 0;          // levels per tile (npz or npz+1, or 1 for a 2-D field)
};

struct Rect { int i0, i1, j0, j1; HD bool has(int i, int j) const { return i >= i0 && i <= i1 && j >= j0 && j <= j1; } };
struct Box { int di0, di1, dj0, dj1, dk0, dk1; };

// Metric terms, each [ntile][pj][pi] (NLM/fv_arrays_nlm.F90:115-234).
struct Metrics {
  const double *area, *rarea, *rarea_c, *dx, *dy, *dxa, *dya, *dxc, *dyc, *rdx, *rdy, *rdxa, *rdya, *rdxc, *rdyc;
  const double *cosa, *sina, *rsina, *cosa_u, *cosa_v, *cosa_s, *sina_u, *sina_v, *rsin_u, *rsin_v, *rsin2;
  const double *f0, *fC, *del6_u, *del6_v, *divg_u, *divg_v;
  const double* sin_sg[10];
  const double* cos_sg[10];
  const double* edge;      // a2b_ord4 edge weights [ntile][4: w,e,s,n][pj] (face mode)
  const double* ecorner;   // extrap_corner coefficients [ntile][4: sw,se,ne,nw][3] (face mode)
  double da_min, da_min_c;
};
constexpr int NMETRIC = 32 + 18;

// Per-level resolved options (dyn_core_tlm.F90:741-921), device array of npz entries.
struct LevelParams {
  int hord_mt, hord_vt, hord_tm, hord_dp, hord_tr;       // the schemes the tangent / adjoint is taken of (the *_pert flags where they differ)
  int hord_tm_g;            // flagstruct%hord_tm as passed to UPDATE_DZ_D for all npz+1 interfaces (dyn_core_tlm.F90:1091): no sponge override
  // trajectory schemes (split_hord): where one differs from the scheme above, that transport's VALUES come from a second, values-only
  // pass with this scheme and the trajectory damping (sw_core_tlm.F90:1664-1682); may be monotone (8, 10)
  int hord_mt_t, hord_vt_t, hord_tm_t, hord_dp_t, hord_tr_t, hord_tm_g_t;
  int nord, nord_v, nord_w, nord_t, nord_v_pert;
  double d2_divg, damp_vt, damp_w, damp_t, d_con, damp_vt_pert;
  // split_damp (fv_arrays_tlmadm.F90:76): the divergence damping the tangent / adjoint is taken of -- the perturbation's own nord_k_pert,
  // d2_divg_pert, dddmp_pert, d4_bg_pert (dyn_core_tlm.F90:835-921, sw_core_tlm.F90:2358-2366) where the flag is set, the trajectory's
  // otherwise -- and the perturbation's damping pair of the heat transport, taken BEFORE the perturbation sponge rules (:856-859)
  int split_damp = 0, nord_p = 0, nord_t_pert = 0;
  double d2_divg_p = 0., damp_t_pert = 0., dddmp = 0., d4_bg = 0., dddmp_p = 0., d4_bg_p = 0.;
  // tracer_2d sub-cycling of the current call (fv_tracer2d_tlm.F90:1306-1345): sub-steps this level takes and 1/that
  int tr_ksplt = 1; double tr_frac = 1.0;
  double dp_ref = 0.;       // ak(k+1)-ak(k) + (bk(k+1)-bk(k))*1e5 (dyn_core_tlm.F90:1704-1706), non-hydrostatic interface weights
};

// Point context handed to every stage evaluation.
struct Ctx {
  Geom g;
  Metrics m;
  const LevelParams* lev;   // [npz]
  int nlev;                 // levels per tile in this launch
  int tr_it = 1;            // tracer_2d: current sub-step (levels with tr_ksplt < tr_it are done)
  HD int mi(int tile, int i, int j) const { return tile * g.plane + g.idx(i, j); }
};

// ------------------------------------------------------------------ accessors
// z = tile*nlev + (k-1) is the launch's plane index; a field with nk levels per tile maps level
// k+dk of tile to plane tile*nk + (k-1) + dk.
template <class S>
struct AccNL {
  const S& s; const Ctx& c; int tile, k;
  static constexpr unsigned want = ~0u;
  template <int M> HD double in(int i, int j, int dk = 0) const {
    const Fld& f = s.in[M];
    return f.t[(size_t)(tile * f.nk + k - 1 + dk) * c.g.plane + c.g.idx(i, j)];
  }
};
template <class S>
struct AccTL {
  const S& s; const Ctx& c; int tile, k;
  static constexpr unsigned want = ~0u;
  template <int M> HD Dual in(int i, int j, int dk = 0) const {
    const Fld& f = s.in[M];
    size_t n = (size_t)(tile * f.nk + k - 1 + dk) * c.g.plane + c.g.idx(i, j);
    return Dual(f.t[n], f.p ? f.p[n] : 0.0);
  }
};
template <class S, int MS>
struct AccAD {
  const S& s; const Ctx& c; int tile, k; int si, sj, sk;   // seed point (absolute i, j, level)
  unsigned want;                                             // outputs that depend on input MS (bit n = output n)
  template <int M> HD Dual in(int i, int j, int dk = 0) const {
    const Fld& f = s.in[M];
    double t = f.t[(size_t)(tile * f.nk + k - 1 + dk) * c.g.plane + c.g.idx(i, j)];
    return Dual(t, (M == MS && i == si && j == sj && k + dk == sk) ? 1.0 : 0.0);
  }
};
// Joint adjoint accessor: direction M is seeded where input M is read at the thread's own point — for the inputs in
// `mask` (those whose stencil contains the current output offset) only, which keeps the bookkeeping of the per-input
// launcher (and of the corner-alias launch that complements it) exact.
template <class S>
struct AccADV {
  const S& s; const Ctx& c; int tile, k; int si, sj, sk;
  unsigned mask;       // inputs seeded at this offset (bit m = input m)
  unsigned want;       // outputs needed
  template <int M> HD DualV<S::NIN> in(int i, int j, int dk = 0) const {
    const Fld& f = s.in[M];
    DualV<S::NIN> r(f.t[(size_t)(tile * f.nk + k - 1 + dk) * c.g.plane + c.g.idx(i, j)]);
    r.d[M] = (((mask >> M) & 1u) && i == si && j == sj && k + dk == sk) ? 1.0 : 0.0;
    return r;
  }
};
#ifdef FV3LM_HOST_EMUL
// Debug accessor: checks every access against the stage's declared stencil box.
template <class S>
struct AccChk {
  const S& s; const Ctx& c; int tile, k; int ei, ej;   // evaluation point
  static constexpr unsigned want = ~0u;
  template <int M> Dual in(int i, int j, int dk = 0) const {
    Box b = S::box(M);
    bool ok = !(i - ei < b.di0 || i - ei > b.di1 || j - ej < b.dj0 || j - ej > b.dj1 || dk < b.dk0 || dk > b.dk1 ||
                !S::uses(M, i - ei, j - ej, dk));
    for (int n = 0; !ok && n < S::NALIAS; ++n) {   // a corner-halo read redirected to (i,j): check the virtual location
      int ai, aj;
      if (!s.alias(c, M, i, j, n, ai, aj)) continue;
      Box ba = S::box(S::alias_box(M));
      ok = !(ai - ei < ba.di0 || ai - ei > ba.di1 || aj - ej < ba.dj0 || aj - ej > ba.dj1 || dk < ba.dk0 || dk > ba.dk1 ||
             !S::uses(S::alias_box(M), ai - ei, aj - ej, dk));
    }
    if (!ok) {
      std::fprintf(stderr, "stage %s: input %d accessed at offset (%d,%d,%d) outside declared box\n", S::name(), M,
                   i - ei, j - ej, dk);
      std::abort();
    }
    const Fld& f = s.in[M];
    size_t n = (size_t)(tile * f.nk + k - 1 + dk) * c.g.plane + c.g.idx(i, j);
    return Dual(f.t[n], f.p ? f.p[n] : 0.0);
  }
};
#endif

}  // namespace fv3
