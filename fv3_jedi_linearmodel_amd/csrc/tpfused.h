// fv3lm-hip: fv_tp_2d (tp_core_tlm.F90:83-236, _TLM :2123-2324) as ONE LDS-tiled kernel for the nonlinear and tangent-linear modes.
//
// The staged form (dycore.h build_tp) runs the routine as seven launches + edge strips -- inner y-sweep, q_i, outer x-sweep, inner
// x-sweep, q_j, outer y-sweep, flux assembly -- with every intermediate (fy2, q_i, fxo, fx2, q_j, fyo) making a round trip through
// HBM.  Here one workgroup owns a 64 x 16 block of cells of one level: it loads the block of q plus a 3-cell halo into LDS once
// (tangent mode: value and tangent planes), runs the four 1-D PPM sweeps and the two intermediate updates against LDS, and writes
// only the two fluxes.  Cube-face edges are the same four-cell edge values with other weights (edges.h ppm_w), the corner halo of
// the inner sweeps is read through the copy_corners view of the sweep direction (corner_map) inside the tile -- no strip launches.
// The arithmetic is the staged stages' own (ppm_flux, the TpQi/TpQj/TpFlux formulas in their order), so both forms agree bit for
// bit; the adjoint keeps the staged gather launches, and the nonlinear mode stores the six intermediates they read (STORE).
//
// Per output point the staged tangent chain moves 64 doubles, this kernel 22 (9 inputs + 2 outputs, value and tangent).
#pragma once
#include "stages.h"

namespace fv3 {

struct TpFusedArgs {
  Fld q, crx, cry, xfx, yfx, rax, ray, mx, my;   // inputs
  Fld mass, d2b;                                 // damping inputs (t == nullptr when unused)
  Fld d2b_t;                                     // the damping Laplacian of the trajectory pass of split levels (TpD2 with traj = 1)
  Fld fx, fy;                                    // outputs
  Fld fy2, q_i, fxo, fx2, q_j, fyo;              // trajectory intermediates (nonlinear mode stores them for the staged adjoint)
  int hsel, dsel, use_mass, nk;
  // flux capacitors of d_sw (sw_core_tlm.F90:3000-3016): cx += crx, cy += cry, mfx += fx, mfy += fy folded into this launch
  // (t == nullptr: none).  do_acc is cleared for the adjoint's trajectory recompute, which must leave the accumulators alone.
  Fld acx, acy, amfx, amfy; int do_acc;
  int store_fo = 1;                              // fxo, fyo are arrays of their own (0: the names alias fx2 / fy2 and nothing may write them)
  int exp = 0;                                   // tp2.h timing experiments (never set in the product build)
  int store_mid = 1;                             // fy2, q_i, fx2, q_j likewise (0: the fused adjoint of tpad.h recomputes them; the nonlinear launch stores nothing)
};
// nonlinear / tangent-linear launch (the adjoint runs the staged launches of build_tp); traj: the values-only trajectory pass of
// the levels with split schemes, in the nonlinear and the tangent mode alike
FV3LM_LINK void run_tp_fused(Exec& ex, int mode, const TpFusedArgs& a0, const Ctx& c, bool traj = false);
FV3LM_LINK void run_tp_outer_ad(Exec& ex, const TpFusedArgs& a0, const Ctx& c);
// the same routine laid out for the wavefront (tp2.h): takes the nonlinear and tangent launches that store nothing for the staged adjoint
FV3LM_LINK void run_tp2(Exec& ex, int mode, const TpFusedArgs& a0, const Ctx& c);
inline bool tp2_enabled() { const char* e = std::getenv("FV3LM_TP2"); return !(e && e[0] == '0'); }      // read per launch: a test switches it between two models
}  // namespace fv3

#ifdef FV3LM_HOST_EMUL
#define TPF_SYNC() ((void)0)
#define TPF_LOOP(e, n) for (int e = tid; e < (n); e += nth)
#define TPF_LOOPU(e, u, n, E) for (int u = 0, e = 0; e < (n); ++e, ++u)      /* one "thread": slot u of a per-thread array is element e */
#else
#define TPF_SYNC() __syncthreads()
// constant trip count, unrolled: the global loads of a thread's elements of a phase are issued together
#ifdef FV3LM_TPF_NOUNROLL      /* experiment: real loops, a quarter of the code */
#define TPF_LOOP(e, n) _Pragma("nounroll") for (int e = tid; e < (n); e += NTH)
#else
#define TPF_LOOP(e, n) _Pragma("unroll") for (int e = tid; e < (n); e += NTH)
#endif
// the same with the slot index u of a per-thread register array (constant trip count E = ceil(n / NTH): static indices after unrolling)
#define TPF_LOOPU(e, u, n, E) _Pragma("unroll") for (int u = 0; u < (E); ++u) if (const int e = tid + u * NTH; e < (n))
#endif

namespace fv3 {
template <class T> struct TpfIO;
template <> struct TpfIO<double> {
  static constexpr int NC = 1;
  DEV static double ld(const Fld& f, size_t n) { return f.t[n]; }
  DEV static void st(const Fld& f, size_t n, double x) { f.t[n] = x; }
  DEV static double lget(const double* v, int, int e) { return v[e]; }
  DEV static void lset(double* v, int, int e, double x) { v[e] = x; }
};
template <> struct TpfIO<Dual> {
  static constexpr int NC = 2;
  DEV static Dual ld(const Fld& f, size_t n) { return Dual(f.t[n], f.p ? f.p[n] : 0.0); }
  DEV static void st(const Fld& f, size_t n, const Dual& x) { f.t[n] = x.v; f.p[n] = x.d; }
  DEV static Dual lget(const double* v, int nt, int e) { return Dual(v[e], v[nt + e]); }
  DEV static void lset(double* v, int nt, int e, const Dual& x) { v[e] = x.v; v[nt + e] = x.d; }
};
// algorithmic bytes of one launch: 9 inputs (+ d2b, mass) read and 2 outputs written per cell (x2 in the tangent mode), + the six stored
// trajectory intermediates of the nonlinear mode
inline double tpf_bytes(const TpFusedArgs& a, const Geom& g, int mode) {
  const double cells = double(g.tx) * g.ty * g.ntile * a.nk, nin = 9. + (a.d2b.t ? 1. : 0.) + (a.mass.t ? 1. : 0.);
  return 8. * cells * (mode == MODE_TL ? 2. * (nin + 2.) : (nin + 2.) + (a.store_mid ? (a.store_fo ? 6. : 4.) : 0.));
}

}  // namespace fv3

#if !defined(FV3LM_SPLIT_BUILD) || defined(FV3LM_IMPL_TPFUSED)
namespace fv3 {
#ifndef FV3LM_TPF_H
#define FV3LM_TPF_H 16
#endif
// threads per block, per mode (measured on MI355X, C192L127 six faces, ms per launch): nonlinear 512: 1.43, 1024: 1.85;
// tangent (107 KB of LDS, one block per CU) 512: 3.11, 1024: 2.22; 64 x 8 blocks were slower in both modes
#ifndef FV3LM_TPF_THREADS_NL
#define FV3LM_TPF_THREADS_NL 512
#endif
#ifndef FV3LM_TPF_THREADS_TL
#define FV3LM_TPF_THREADS_TL 1024
#endif
constexpr int TPF_W = 64, TPF_H = FV3LM_TPF_H;   // cells per block
constexpr int TPF_QW = TPF_W + 6, TPF_QH = TPF_H + 6;
constexpr int TPF_NQ = TPF_QW * TPF_QH, TPF_NFY2 = TPF_QW * (TPF_H + 1), TPF_NQI = TPF_QW * TPF_H, TPF_NFX2 = (TPF_W + 1) * TPF_QH,
              TPF_NQJ = TPF_W * TPF_QH, TPF_NT = TPF_NQ + TPF_NFY2 + TPF_NQI + TPF_NFX2 + TPF_NQJ;   // doubles per component
constexpr int TPF_THREADS_NL = FV3LM_TPF_THREADS_NL, TPF_THREADS_TL = FV3LM_TPF_THREADS_TL;


// an LDS array of T over the rectangle [i0, i0+w) x [j0, ...)
template <class T>
struct TpfTile {
  double* v; int i0, j0, w;
  DEV T get(int i, int j) const { return TpfIO<T>::lget(v, TPF_NT, (j - j0) * w + (i - i0)); }
  DEV void set(int i, int j, const T& x) const { TpfIO<T>::lset(v, TPF_NT, (j - j0) * w + (i - i0), x); }
};

// One block: cells I0..I1 x J0..J1 of level k of one tile.  tid / nth: this thread and the number of threads sharing the block
// (host emulation: 0 / 1).
// TRAJ: the values-only second pass of a level whose trajectory scheme differs from the scheme the tangent / adjoint is taken of
// (split_hord, sw_core_tlm.F90:1664-1682): the whole chain again with the trajectory scheme (inner sweeps: 8 where it is 10,
// tp_core_tlm.F90:135-136) and the trajectory damping, fx.t / fy.t overwritten; levels without a split return at once.
template <class T, bool STORE, int NTH, bool TRAJ = false>
DEV void tp_fused_block(const TpFusedArgs& a, const Ctx& c, int tile, int k, int bx, int by, double* lds, int tid, int nth) {
  typedef TpfIO<T> IO;
  static_assert(!TRAJ || (std::is_same<T, double>::value && !STORE), "the trajectory pass computes values only");
  const bool split = level_split(c.lev[k - 1], a.hsel, a.dsel);
  if (TRAJ && !split) return;
  const Geom& g = c.g;
  const int nx = g.nx, ny = g.ny;
  const bool face = g.face != 0;
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je();      // the tile's window of the face (nx, ny: the FACE, for the edge formulas)
  const int I0 = is + bx * TPF_W, I1 = (I0 + TPF_W - 1 < ie) ? I0 + TPF_W - 1 : ie;
  const int J0 = js + by * TPF_H, J1 = (J0 + TPF_H - 1 < je) ? J0 + TPF_H - 1 : je;
  const bool firstx = bx == 0, lastx = I1 == ie, firsty = by == 0, lasty = J1 == je;
  const size_t base = (size_t)(tile * a.nk + k - 1) * g.plane;
  auto at = [&](int i, int j) -> size_t { return base + g.idx(i, j); };
  // constant row pitches (the full block's), whatever the block's own width: index arithmetic by compile-time constants
  const TpfTile<T> q{lds, I0 - 3, J0 - 3, TPF_QW}, fy2{lds + TPF_NQ, I0 - 3, J0, TPF_QW}, qi{lds + TPF_NQ + TPF_NFY2, I0 - 3, J0, TPF_QW},
      fx2{lds + TPF_NQ + TPF_NFY2 + TPF_NQI, I0, J0 - 3, TPF_W + 1}, qj{lds + TPF_NQ + TPF_NFY2 + TPF_NQI + TPF_NFX2, I0, J0 - 3, TPF_W};
  const int iord = TRAJ ? hord_traj_of(c.lev[k - 1], a.hsel) : hord_of(c.lev[k - 1], a.hsel), iord_in = (TRAJ && iord == 10) ? 8 : iord;
  auto flux1d = [&](int io, int m, int n1, const auto& line, const auto& da, const T& cc) -> T {
    if constexpr (TRAJ) return ppm_flux_traj(io, face, m, n1, line, da, cc); else return ppm_flux<T>(io, face, m, n1, line, da, cc);
  };
  // flux capacitors on a split level: the first pass adds the tangents, the trajectory pass the values
  auto acc_flux = [&](const Fld& acc, size_t n, const T& f) {
    if constexpr (TRAJ) acc.t[n] += f;
    else if (!split) IO::st(acc, n, IO::ld(acc, n) + f);
    else if constexpr (!std::is_same<T, double>::value) acc.p[n] += f.d;
  };
  // ownership of the stored intermediates: every element of their full regions belongs to exactly one block
  auto own_i = [&](int i) { return (i >= I0 && i <= I1) || (firstx && i < I0) || (lastx && i > I1); };
  auto own_j = [&](int j) { return (j >= J0 && j <= J1) || (firsty && j < J0) || (lasty && j > J1); };

  // Tangent mode, device build: operands of the NEXT phase are fetched into registers before the barrier that ends the current one --
  // with one 1024-thread block per CU (107 KB of LDS) nothing else hides the HBM latency behind a barrier.
#ifdef FV3LM_TPF_NOUNROLL
  constexpr bool PRE = false;
#else
  constexpr bool PRE = NTH > 1 && !TRAJ && !std::is_same<T, double>::value;
#endif      // the nonlinear kernel runs three blocks per CU: they hide each other's latency, and the registers would cost it one
  constexpr int N2A = TPF_QW * (TPF_H + 1), N2B = (TPF_W + 1) * TPF_QH, N3A = TPF_QW * TPF_H, N3B = TPF_W * TPF_QH;
  constexpr int E2A = PRE ? (N2A + NTH - 1) / NTH : 1, E2B = PRE ? (N2B + NTH - 1) / NTH : 1, E3A = PRE ? (N3A + NTH - 1) / NTH : 1, E3B = PRE ? (N3B + NTH - 1) / NTH : 1;
  T cy_pre[E2A], cx_pre[E2B], y0_pre[E3A], y1_pre[E3A], ry_pre[E3A], x0_pre[E3B], x1_pre[E3B], rx_pre[E3B];
  if constexpr (PRE) {
    { int u = 0; TPF_LOOP(e, N2A) { const int i = I0 - 3 + e % TPF_QW, j = J0 + e / TPF_QW; cy_pre[u++] = (i <= I1 + 3 && j <= J1 + 1) ? IO::ld(a.cry, at(i, j)) : T(0.); } }
    { int u = 0; TPF_LOOP(e, N2B) { const int i = I0 + e % (TPF_W + 1), j = J0 - 3 + e / (TPF_W + 1); cx_pre[u++] = (i <= I1 + 1 && j <= J1 + 3) ? IO::ld(a.crx, at(i, j)) : T(0.); } }
  }
  // ---- the block of q and its halo
  { constexpr int w = TPF_QW, n = w * TPF_QH;
    TPF_LOOP(e, n) { const int i = I0 - 3 + e % w, j = J0 - 3 + e / w; if (i <= I1 + 3 && j <= J1 + 3) q.set(i, j, IO::ld(a.q, at(i, j))); } }
  TPF_SYNC();
  if constexpr (PRE) {
    { int u = 0; TPF_LOOP(e, N3A) { const int i = I0 - 3 + e % TPF_QW, j = J0 + e / TPF_QW; const bool in = i <= I1 + 3 && j <= J1;
        y0_pre[u] = in ? IO::ld(a.yfx, at(i, j)) : T(0.); y1_pre[u] = in ? IO::ld(a.yfx, at(i, j + 1)) : T(0.); ry_pre[u] = in ? IO::ld(a.ray, at(i, j)) : T(1.); ++u; } }
    { int u = 0; TPF_LOOP(e, N3B) { const int i = I0 + e % TPF_W, j = J0 - 3 + e / TPF_W; const bool in = i <= I1 && j <= J1 + 3;
        x0_pre[u] = in ? IO::ld(a.xfx, at(i, j)) : T(0.); x1_pre[u] = in ? IO::ld(a.xfx, at(i + 1, j)) : T(0.); rx_pre[u] = in ? IO::ld(a.rax, at(i, j)) : T(1.); ++u; } }
  }
  // ---- inner sweeps: fy2 = yppm(q) on the halo'd columns (copy_corners view 2), fx2 = xppm(q) on the halo'd rows (view 1)
  { constexpr int w = TPF_QW, n = w * (TPF_H + 1);
    int u = 0;
    TPF_LOOP(e, n) {
      const int i = I0 - 3 + e % w, j = J0 + e / w;
      const int uu = PRE ? u++ : 0;
      if (i > I1 + 3 || j > J1 + 1) continue;
      auto line = [&](int jj) -> T { int ii = i, j2 = jj; if (face) corner_map(g, 2, ii, j2); return q.get(ii, j2); };
      const MetY da{c.m.dya, c, tile, i};
      const T cc = PRE ? cy_pre[uu] : IO::ld(a.cry, at(i, j));
      const T f = flux1d(iord_in, j, ny + 1, line, da, cc);
      fy2.set(i, j, f);
      if (STORE && own_i(i) && (j <= J1 || lasty)) a.fy2.t[at(i, j)] = val(f);
      if (!TRAJ && a.do_acc && own_i(i) && (j <= J1 || lasty)) IO::st(a.acy, at(i, j), IO::ld(a.acy, at(i, j)) + cc);
    } }
  { constexpr int w = TPF_W + 1, n = w * TPF_QH;
    int u = 0;
    TPF_LOOP(e, n) {
      const int i = I0 + e % w, j = J0 - 3 + e / w;
      const int uu = PRE ? u++ : 0;
      if (i > I1 + 1 || j > J1 + 3) continue;
      auto line = [&](int ii) -> T { int i2 = ii, jj = j; if (face) corner_map(g, 1, i2, jj); return q.get(i2, jj); };
      const MetX da{c.m.dxa, c, tile, j};
      const T cc = PRE ? cx_pre[uu] : IO::ld(a.crx, at(i, j));
      const T f = flux1d(iord_in, i, nx + 1, line, da, cc);
      fx2.set(i, j, f);
      if (STORE && own_j(j) && (i <= I1 || lastx)) a.fx2.t[at(i, j)] = val(f);
      if (!TRAJ && a.do_acc && own_j(j) && (i <= I1 || lastx)) IO::st(a.acx, at(i, j), IO::ld(a.acx, at(i, j)) + cc);
    } }
  TPF_SYNC();
  // ---- q_i, q_j: the field advanced by the inner fluxes (tp_core_tlm.F90:149-159, :173-181)
  { constexpr int w = TPF_QW, n = w * TPF_H;
    int u = 0;
    TPF_LOOP(e, n) {
      const int i = I0 - 3 + e % w, j = J0 + e / w;
      const int uu = PRE ? u++ : 0;
      if (i > I1 + 3 || j > J1) continue;
      const T f0 = (PRE ? y0_pre[uu] : IO::ld(a.yfx, at(i, j))) * fy2.get(i, j), f1 = (PRE ? y1_pre[uu] : IO::ld(a.yfx, at(i, j + 1))) * fy2.get(i, j + 1);
      const T x = (q.get(i, j) * MET(area, i, j) + f0 - f1) / (PRE ? ry_pre[uu] : IO::ld(a.ray, at(i, j)));
      qi.set(i, j, x);
      if (STORE && own_i(i)) a.q_i.t[at(i, j)] = val(x);
    } }
  { constexpr int w = TPF_W, n = w * TPF_QH;
    int u = 0;
    TPF_LOOP(e, n) {
      const int i = I0 + e % w, j = J0 - 3 + e / w;
      const int uu = PRE ? u++ : 0;
      if (i > I1 || j > J1 + 3) continue;
      const T f0 = (PRE ? x0_pre[uu] : IO::ld(a.xfx, at(i, j))) * fx2.get(i, j), f1 = (PRE ? x1_pre[uu] : IO::ld(a.xfx, at(i + 1, j))) * fx2.get(i + 1, j);
      const T x = (q.get(i, j) * MET(area, i, j) + f0 - f1) / (PRE ? rx_pre[uu] : IO::ld(a.rax, at(i, j)));
      qj.set(i, j, x);
      if (STORE && own_j(j)) a.q_j.t[at(i, j)] = val(x);
    } }
  TPF_SYNC();
  // ---- outer sweeps and flux assembly (tp_core_tlm.F90:187-234; deln_flux :1918-2043 as in stages.h TpFlux)
  int nord; double dc; damp_of(c.lev[k - 1], a.dsel, nord, dc, split && !TRAJ);
  const bool dmp = (a.dsel != DAMP_NONE) && (dc > 1.e-4);
  double damp = 0.;
  if (dmp) damp = damp_pow(dc * c.m.da_min, nord);
  const Fld& d2 = TRAJ ? a.d2b_t : a.d2b;
  { constexpr int w = TPF_W + 1, n = w * (TPF_H + 1);
    TPF_LOOP(e, n) {
      const int i = I0 + e % w, j = J0 + e / w;
      if (i > I1 + 1 || j > J1 + 1) continue;
      if (j <= J1 && (i <= I1 || lastx)) {          // fx(i,j)
        auto line = [&](int ii) -> T { return qi.get(ii, j); };
        const MetX da{c.m.dxa, c, tile, j};
        const T fo = flux1d(iord, i, nx + 1, line, da, IO::ld(a.crx, at(i, j)));
        if (STORE && a.store_fo) a.fxo.t[at(i, j)] = val(fo);
        T f = 0.5 * (fo + fx2.get(i, j)) * IO::ld(a.mx, at(i, j));
        if (dmp) {
          T f2;
          if (nord == 0) { f2 = MET(del6_v, i, j) * (q.get(i - 1, j) - q.get(i, j)); if (!a.use_mass) f2 = damp * f2; }
          else f2 = MET(del6_v, i, j) * (IO::ld(d2, at(i, j)) - IO::ld(d2, at(i - 1, j)));
          if (a.use_mass) f = f + (0.5 * damp) * (IO::ld(a.mass, at(i - 1, j)) + IO::ld(a.mass, at(i, j))) * f2;
          else f = f + f2;
        }
        IO::st(a.fx, at(i, j), f);
        if (a.do_acc) acc_flux(a.amfx, at(i, j), f);
      }
      if (i <= I1 && (j <= J1 || lasty)) {          // fy(i,j)
        auto line = [&](int jj) -> T { return qj.get(i, jj); };
        const MetY da{c.m.dya, c, tile, i};
        const T fo = flux1d(iord, j, ny + 1, line, da, IO::ld(a.cry, at(i, j)));
        if (STORE && a.store_fo) a.fyo.t[at(i, j)] = val(fo);
        T f = 0.5 * (fo + fy2.get(i, j)) * IO::ld(a.my, at(i, j));
        if (dmp) {
          T f2;
          if (nord == 0) { f2 = MET(del6_u, i, j) * (q.get(i, j - 1) - q.get(i, j)); if (!a.use_mass) f2 = damp * f2; }
          else f2 = MET(del6_u, i, j) * (IO::ld(d2, at(i, j)) - IO::ld(d2, at(i, j - 1)));
          if (a.use_mass) f = f + (0.5 * damp) * (IO::ld(a.mass, at(i, j - 1)) + IO::ld(a.mass, at(i, j))) * f2;
          else f = f + f2;
        }
        IO::st(a.fy, at(i, j), f);
        if (a.do_acc) acc_flux(a.amfy, at(i, j), f);
      }
    } }
}

inline void tpf_grid(const Geom& g, int& nbx, int& nby) { nbx = (g.tx + TPF_W - 1) / TPF_W; nby = (g.ty + TPF_H - 1) / TPF_H; }
#ifndef FV3LM_HOST_EMUL
template <class T, bool STORE, int NTH, bool TRAJ = false>
__global__ void __launch_bounds__(NTH) k_tp_fused(TpFusedArgs a, Ctx c) {
  extern __shared__ double tpf_lds[];
  int bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by);
  tp_fused_block<T, STORE, NTH, TRAJ>(a, c, blockIdx.z / a.nk, 1 + blockIdx.z % a.nk, bx, by, tpf_lds, threadIdx.x, NTH);
}
#endif

FV3LM_LINK void run_tp_fused(Exec& ex, int mode, const TpFusedArgs& a0, const Ctx& c, bool traj) {
  if (!traj && (mode == MODE_TL || !a0.store_mid) && tp2_enabled() && !(mode == MODE_TL && std::getenv("FV3LM_TP2_NL_ONLY"))) { run_tp2(ex, mode, a0, c); return; }
  TpFusedArgs a = a0;
  for (Fld* f : {&a.q, &a.crx, &a.cry, &a.xfx, &a.yfx, &a.rax, &a.ray, &a.mx, &a.my, &a.mass, &a.d2b, &a.d2b_t, &a.fx, &a.fy, &a.fy2, &a.q_i, &a.fxo, &a.fx2, &a.q_j, &a.fyo, &a.acx, &a.acy, &a.amfx, &a.amfy}) *f = ex.sh(*f);
  a.do_acc = (a.acx.t && !ex.skip_accum) ? 1 : 0;
  int nbx, nby; tpf_grid(c.g, nbx, nby);
  ex.mark_begin(traj ? "TpFusedTraj" : "TpFused", traj ? ".nl" : mode == MODE_TL ? ".tl" : ".nl", traj ? 0. : tpf_bytes(a, c.g, mode));
#ifdef FV3LM_HOST_EMUL
  std::vector<double> lds((size_t)TPF_NT * 2);
  for (int z = 0; z < c.g.ntile * a.nk; ++z)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) {
        if (traj) tp_fused_block<double, false, 1, true>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0, 1);
        else if (mode == MODE_TL) tp_fused_block<Dual, false, 1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0, 1);
        else if (a.store_mid) tp_fused_block<double, true, 1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0, 1);
        else tp_fused_block<double, false, 1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0, 1);
      }
#else
  const dim3 grid(nbx, nby, c.g.ntile * a.nk);
  if (traj) {
    hipLaunchKernelGGL((k_tp_fused<double, false, TPF_THREADS_NL, true>), grid, dim3(TPF_THREADS_NL), TPF_NT * 8, ex.stream, a, c);
  } else if (mode == MODE_TL) {
    static bool attr = false;      // 2 x 53.5 KB of LDS per block: above the 64 KB default limit of a launch
    if (!attr) { if (hipFuncSetAttribute((const void*)k_tp_fused<Dual, false, TPF_THREADS_TL>, hipFuncAttributeMaxDynamicSharedMemorySize, TPF_NT * 16) != hipSuccess) set_sticky("hipFuncSetAttribute(k_tp_fused) failed"); attr = true; }
    hipLaunchKernelGGL((k_tp_fused<Dual, false, TPF_THREADS_TL>), grid, dim3(TPF_THREADS_TL), TPF_NT * 16, ex.stream, a, c);
  } else if (a.store_mid) {
    hipLaunchKernelGGL((k_tp_fused<double, true, TPF_THREADS_NL>), grid, dim3(TPF_THREADS_NL), TPF_NT * 8, ex.stream, a, c);
  } else {
    hipLaunchKernelGGL((k_tp_fused<double, false, TPF_THREADS_NL>), grid, dim3(TPF_THREADS_NL), TPF_NT * 8, ex.stream, a, c);
  }
#endif
  ex.mark_end();
  ex.launches++;
}

// =====================================================================================================================
// Hand-written adjoint of the OUTER half of fv_tp_2d: flux assembly + outer x-sweep (q_i -> fxo) + outer y-sweep (q_j -> fyo), one
// LDS-tiled launch instead of three gather launches + four strip launches.
//   fxo_ad = 0.5 mx fx_ad               fx2_ad  = 0.5 mx fx_ad  (stored)            mx_ad += 0.5 (fxo + fx2) fx_ad
//   q_i_ad(k) = sum_{m = k-2 .. k+3} d flux(m)/d q_i(k) fxo_ad(m)   (stored)       crx_ad(m) += d flux(m)/d c  fxo_ad(m)
// and the same in y.  The schemes are linear in q for a given Courant number, so d flux/d q is a closed form of c and the edge-aware
// weights (stages.h ppm_dq) -- no dual-number evaluation per offset as in the generic gather; d flux/d c is one ppm_flux<Dual> with
// only c seeded.  The damping part of the flux is a stage of its own in the adjoint (stages.h TpDamp).  The launch is the first
// one of the backward pass through the routine: it STORES the adjoints of q_i, q_j, fx2, fy2 over the rectangles the staged launches
// that follow (TpQi, TpQj, the inner sweeps) read, zeros included, and accumulates into crx, cry, mx, my.
// =====================================================================================================================
constexpr int TPA_AW = TPF_W + 6;                                   // pitch of the x-halo'd tiles
constexpr int TPA_NA = TPA_AW * TPF_H, TPA_NC = TPF_W * (TPF_H + 6);
constexpr int TPA_NT = 3 * TPA_NA + 3 * TPA_NC;                     // fxo_ad, crx, q_i | fyo_ad, cry, q_j
constexpr int TPA_THREADS = 512;

template <int NTH>
DEV void tp_outer_ad_block(const TpFusedArgs& a, const Ctx& c, int tile, int k, int bx, int by, double* lds, int tid, int nth) {
  (void)nth;
  const Geom& g = c.g;
  const int nx = g.nx, ny = g.ny;
  const bool face = g.face != 0;
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je();      // the tile's window of the face (nx, ny: the FACE, for the edge formulas)
  const int I0 = is + bx * TPF_W, I1 = (I0 + TPF_W - 1 < ie) ? I0 + TPF_W - 1 : ie;
  const int J0 = js + by * TPF_H, J1 = (J0 + TPF_H - 1 < je) ? J0 + TPF_H - 1 : je;
  const bool firstx = bx == 0, lastx = I1 == ie, firsty = by == 0, lasty = J1 == je;
  const size_t base = (size_t)(tile * a.nk + k - 1) * g.plane;
  auto at = [&](int i, int j) -> size_t { return base + g.idx(i, j); };
  const int iord = hord_of(c.lev[k - 1], a.hsel);
  // tiles: x-halo'd [I0-3, I1+3] x [J0, J1] and y-halo'd [I0, I1] x [J0-3, J1+3]
  double* fxa = lds; double* cxt = lds + TPA_NA; double* qit = lds + 2 * TPA_NA;
  double* fya = lds + 3 * TPA_NA; double* cyt = fya + TPA_NC; double* qjt = fya + 2 * TPA_NC;
  auto ex_ = [&](int i, int j) { return (j - J0) * TPA_AW + (i - (I0 - 3)); };
  auto ey_ = [&](int i, int j) { return (j - (J0 - 3)) * TPF_W + (i - I0); };
  { constexpr int n = TPA_NA;
    TPF_LOOP(e, n) {
      const int i = I0 - 3 + e % TPA_AW, j = J0 + e / TPA_AW;
      double f = 0., cc = 0., q = 0.;
      if (i <= I1 + 3 && j <= J1) {
        q = a.q_i.t[at(i, j)];
        if (i >= is && i <= ie + 1) { cc = a.crx.t[at(i, j)]; f = 0.5 * a.mx.t[at(i, j)] * a.fx.p[at(i, j)]; }
      }
      fxa[e] = f; cxt[e] = cc; qit[e] = q;
    } }
  { constexpr int n = TPA_NC;
    TPF_LOOP(e, n) {
      const int i = I0 + e % TPF_W, j = J0 - 3 + e / TPF_W;
      double f = 0., cc = 0., q = 0.;
      if (i <= I1 && j <= J1 + 3) {
        q = a.q_j.t[at(i, j)];
        if (j >= js && j <= je + 1) { cc = a.cry.t[at(i, j)]; f = 0.5 * a.my.t[at(i, j)] * a.fy.p[at(i, j)]; }
      }
      fya[e] = f; cyt[e] = cc; qjt[e] = q;
    } }
  TPF_SYNC();
  // ---- flux points of the block: Courant-number and mass-flux adjoints, the stored inner-flux adjoints
  { constexpr int w = TPF_W + 1, n = w * (TPF_H + 1);
    TPF_LOOP(e, n) {
      const int i = I0 + e % w, j = J0 + e / w;
      if (i > I1 + 1 || j > J1 + 1) continue;
      if (j <= J1 && (i <= I1 || lastx)) {
        const double fo = fxa[ex_(i, j)], cc = cxt[ex_(i, j)];
        auto line = [&](int ii) -> Dual { return Dual(qit[ex_(ii, j)], 0.); };
        const MetX da{c.m.dxa, c, tile, j};
        const Dual fl = ppm_flux<Dual>(iord, face, i, nx + 1, line, da, Dual(cc, 1.));
        a.crx.p[at(i, j)] += fl.d * fo;
        a.mx.p[at(i, j)] += 0.5 * (fl.v + a.fx2.t[at(i, j)]) * a.fx.p[at(i, j)];      // fl.v = the outer flux fxo, re-evaluated from q_i
        a.fx2.p[at(i, j)] = fo;
      }
      if (i <= I1 && (j <= J1 || lasty)) {
        const double fo = fya[ey_(i, j)], cc = cyt[ey_(i, j)];
        auto line = [&](int jj) -> Dual { return Dual(qjt[ey_(i, jj)], 0.); };
        const MetY da{c.m.dya, c, tile, i};
        const Dual fl = ppm_flux<Dual>(iord, face, j, ny + 1, line, da, Dual(cc, 1.));
        a.cry.p[at(i, j)] += fl.d * fo;
        a.my.p[at(i, j)] += 0.5 * (fl.v + a.fy2.t[at(i, j)]) * a.fy.p[at(i, j)];
        a.fy2.p[at(i, j)] = fo;
      }
    } }
  // the stored inner-flux adjoints are zero where the flux assembly does not reach: fx2 on the halo rows, fy2 on the halo columns
  if (firsty || lasty) {
    constexpr int w = TPF_W + 1, n = w * 6;
    TPF_LOOP(e, n) {
      const int i = I0 + e % w, r = e / w;
      if (i > I1 + 1 || !(i <= I1 || lastx)) continue;
      if (firsty && r < 3) a.fx2.p[at(i, js + r - 3)] = 0.;        // rows js-3 .. js-1
      if (lasty && r >= 3) a.fx2.p[at(i, je + r - 2)] = 0.;        // rows je+1 .. je+3
    } }
  if (firstx || lastx) {
    constexpr int n = 6 * (TPF_H + 1);
    TPF_LOOP(e, n) {
      const int r = e % 6, j = J0 + e / 6;
      if (j > J1 + 1 || !(j <= J1 || lasty)) continue;
      if (firstx && r < 3) a.fy2.p[at(is + r - 3, j)] = 0.;
      if (lastx && r >= 3) a.fy2.p[at(ie + r - 2, j)] = 0.;
    } }
  // ---- transposed outer sweeps: q_i_ad on the halo'd columns, q_j_ad on the halo'd rows (stored)
  { constexpr int n = TPA_NA;
    TPF_LOOP(e, n) {
      const int i = I0 - 3 + e % TPA_AW, j = J0 + e / TPA_AW;
      if (i > I1 + 3 || j > J1) continue;
      if (!((i >= I0 && i <= I1) || (firstx && i < I0) || (lastx && i > I1))) continue;
      const MetX da{c.m.dxa, c, tile, j};
      double s = 0.;
#pragma unroll
      for (int m = i - 2; m <= i + 3; ++m) {
        if (m < is || m > ie + 1 || m < I0 - 3 || m > I1 + 3) continue;
        const double f = fxa[ex_(m, j)];
        if (f != 0.) s += ppm_dq(iord, face, m, nx + 1, i, da, cxt[ex_(m, j)]) * f;
      }
      a.q_i.p[at(i, j)] = s;
    } }
  { constexpr int n = TPA_NC;
    TPF_LOOP(e, n) {
      const int i = I0 + e % TPF_W, j = J0 - 3 + e / TPF_W;
      if (i > I1 || j > J1 + 3) continue;
      if (!((j >= J0 && j <= J1) || (firsty && j < J0) || (lasty && j > J1))) continue;
      const MetY da{c.m.dya, c, tile, i};
      double s = 0.;
#pragma unroll
      for (int m = j - 2; m <= j + 3; ++m) {
        if (m < js || m > je + 1 || m < J0 - 3 || m > J1 + 3) continue;
        const double f = fya[ey_(i, m)];
        if (f != 0.) s += ppm_dq(iord, face, m, ny + 1, j, da, cyt[ey_(i, m)]) * f;
      }
      a.q_j.p[at(i, j)] = s;
    } }
}

#ifndef FV3LM_HOST_EMUL
__global__ void __launch_bounds__(TPA_THREADS) k_tp_outer_ad(TpFusedArgs a, Ctx c) {
  extern __shared__ double tpa_lds[];
  int bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by);
  tp_outer_ad_block<TPA_THREADS>(a, c, blockIdx.z / a.nk, 1 + blockIdx.z % a.nk, bx, by, tpa_lds, threadIdx.x, TPA_THREADS);
}
#endif

FV3LM_LINK void run_tp_outer_ad(Exec& ex, const TpFusedArgs& a0, const Ctx& c) {
  TpFusedArgs a = a0;
  for (Fld* f : {&a.q, &a.crx, &a.cry, &a.xfx, &a.yfx, &a.rax, &a.ray, &a.mx, &a.my, &a.mass, &a.d2b, &a.fx, &a.fy, &a.fy2, &a.q_i, &a.fxo, &a.fx2, &a.q_j, &a.fyo}) *f = ex.sh(*f);
  int nbx, nby; tpf_grid(c.g, nbx, nby);
  // algorithmic bytes: trajectory q_i, q_j, crx, cry, mx, my, fx2, fy2 and the adjoints fx, fy read; q_i, q_j, fx2, fy2 adjoints written;
  // crx, cry, mx, my adjoints read-modify-written
  const double cells = double(c.g.tx) * c.g.ty * c.g.ntile * a.nk;
  ex.mark_begin("TpOuter", ".ad", 8. * cells * (8. + 2. + 4. + 8.));
#ifdef FV3LM_HOST_EMUL
  std::vector<double> lds((size_t)TPA_NT);
  for (int z = 0; z < c.g.ntile * a.nk; ++z)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) tp_outer_ad_block<1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0, 1);
#else
  hipLaunchKernelGGL(k_tp_outer_ad, dim3(nbx, nby, c.g.ntile * a.nk), dim3(TPA_THREADS), TPA_NT * 8, ex.stream, a, c);
#endif
  ex.mark_end();
  ex.launches++;
}

}  // namespace fv3
#endif
