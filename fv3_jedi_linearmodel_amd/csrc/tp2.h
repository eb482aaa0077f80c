// fv3lm-hip: fv_tp_2d (tp_core_tlm.F90:83-236, _TLM :2123-2324), nonlinear and tangent-linear, as one LDS-tiled kernel laid out for the
// wavefront.  Same routine, same LDS tiles and the same arithmetic as tpfused.h's tp_fused_block (which stays for the trajectory pass of
// split schemes and for the launches that store the intermediates of the staged adjoint); what differs is the data flow:
//   * a wave owns a ROW of the block and a lane a COLUMN: no flattened element index, no division, the row of every element is a scalar,
//     so the face-edge tests of the y-sweeps are scalar branches; the few halo columns beyond the 64 lanes are shared out afterwards;
//   * an interface flux reads the five cells around its UPWIND cell (address selected by the sign of the Courant number) and forms the
//     two edge values of that cell -- two 4-point edge values per flux instead of three, no divergent branch on the sign;
//   * the edge-value weights are constants except within two cells of a cube edge, where they come from a per-block table (the
//     metric-dependent two-sided weights of tp_core_tlm.F90:2402-2429 are evaluated once per row / column, not once per flux);
//   * the copy_corners views (tp_core_tlm.F90:2046-2118) are taken only by the few elements of a corner block that read corner cells.
// The measured instruction mix of the first form was 17 % fp64 arithmetic, the rest index arithmetic, scalar bookkeeping and branches.
#pragma once
#include "tpfused.h"

namespace fv3 {
FV3LM_LINK void run_tp2(Exec& ex, int mode, const TpFusedArgs& a0, const Ctx& c);
}

#if !defined(FV3LM_SPLIT_BUILD) || defined(FV3LM_IMPL_TP2)
namespace fv3 {
#ifndef FV3LM_TP2_H
#define FV3LM_TP2_H 16
#endif
#ifndef FV3LM_TP2_EXP
#define FV3LM_TP2_EXP 0
#endif
// operands of the next phase fetched into registers before the barrier that ends the current one (0: every load where it is used)
#ifndef FV3LM_TP2_PRE_NL
#define FV3LM_TP2_PRE_NL 1
#endif
#ifndef FV3LM_TP2_PRE_TL
#define FV3LM_TP2_PRE_TL 1
#endif
#ifndef FV3LM_TP2_WAVES_NL
#define FV3LM_TP2_WAVES_NL 8
#endif
#ifndef FV3LM_TP2_WAVES_TL
#define FV3LM_TP2_WAVES_TL 16
#endif
constexpr int TP2_W = 64, TP2_H = FV3LM_TP2_H;
constexpr int TP2_PQ = TP2_W + 6, TP2_PX = TP2_W + 1, TP2_RQ = TP2_H + 6;      // pitches of the x-halo'd arrays / of fx2, rows of the y-halo'd ones
// LDS, in elements of T: q | fx2 | fy2 | q_i | q_j, then (doubles) the edge-weight tables
constexpr int TP2_OQ = 0, TP2_OFX2 = TP2_OQ + TP2_PQ * TP2_RQ, TP2_OFY2 = TP2_OFX2 + TP2_PX * TP2_RQ, TP2_OQI = TP2_OFY2 + TP2_PQ * (TP2_H + 1),
              TP2_OQJ = TP2_OQI + TP2_PQ * TP2_H, TP2_NT = TP2_OQJ + TP2_W * TP2_RQ;
constexpr int TP2_NEW = 2 * 4 * (TP2_RQ + TP2_PQ);            // weights at m = 1 and m = n1: per row (x-sweeps) and per column (y-sweeps)
template <class T> constexpr size_t tp2_lds_bytes() { return (size_t)TP2_NT * sizeof(T) + (size_t)TP2_NEW * 8; }

#ifdef FV3LM_HOST_EMUL
#define TP2_INLINE
#else
#define TP2_INLINE __attribute__((always_inline))
#endif

// NC columns x NR rows of elements: the first 64 columns by lane with the rows dealt to the waves (row index uniform in the wave), the
// remaining columns shared out one element per thread.  f(x, r, it): it = the thread's iteration over its rows (a compile-time constant
// after unrolling: the slot of the operands prefetched for that element), -1 for an element of the shared-out columns and in the host
// emulation, where one "thread" runs them all.
template <int NWV, int NC, int NR, class F>
DEV void tp2_sweep(int tid, const F& f) {
#ifdef FV3LM_HOST_EMUL
  (void)tid;
  for (int r = 0; r < NR; ++r) for (int x = 0; x < NC; ++x) f(x, r, -1);
#else
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NIT = (NR + NWV - 1) / NWV;
#pragma unroll
  for (int it = 0; it < NIT; ++it) { const int r = wv + it * NWV; if (r < NR) f(lane, r, it); }
  if constexpr (NC > 64) {
    constexpr int HC = NC - 64, N = HC * NR;
    for (int e = tid; e < N; e += 64 * NWV) f(64 + e % HC, e / HC, -1);
  }
#endif
}
// the lane-mapped elements only: issues the global loads of a later phase (prefetch into registers)
template <int NWV, int NR, class F>
DEV void tp2_fetch(int tid, const F& f) {
#ifndef FV3LM_HOST_EMUL
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NIT = (NR + NWV - 1) / NWV;
#pragma unroll
  for (int it = 0; it < NIT; ++it) { const int r = wv + it * NWV; if (r < NR) f(lane, r, it); }
#else
  (void)tid; (void)f;
#endif
}

// edge-value weights at position pos of a line whose face has n1 - 1 cells (edges.h ppm_w), the metric-dependent ones from the tables
DEV void tp2_weights(int pos, int n1, const double* tlo, const double* thi, double* w) {
  w[0] = w[3] = -1. / 12.; w[1] = w[2] = 7. / 12.;
  if (pos == 0 || pos == n1 - 1) { w[0] = EC1; w[1] = EC2; w[2] = EC3; w[3] = 0.; }
  else if (pos == 2 || pos == n1 + 1) { w[0] = 0.; w[1] = EC3; w[2] = EC2; w[3] = EC1; }
  else if (pos == 1) { w[0] = tlo[0]; w[1] = tlo[1]; w[2] = tlo[2]; w[3] = tlo[3]; }
  else if (pos == n1) { w[0] = thi[0]; w[1] = thi[1]; w[2] = thi[2]; w[3] = thi[3]; }
}
template <class T> DEV T tp2_sel(bool s, const T& a, const T& b);
template <> DEV double tp2_sel<double>(bool s, const double& a, const double& b) { return s ? a : b; }
template <> DEV Dual tp2_sel<Dual>(bool s, const Dual& a, const Dual& b) { return Dual(s ? a.v : b.v, s ? a.d : b.d); }
// scheme 2 (xppm / yppm iord 2, tp_core_tlm.F90:2431-2487) from the five cells around the upwind cell, p[-2S] .. p[2S]: w0 / w1 are the
// weights of the edge values on the upwind cell's low / high side.  up: the flow comes from the low side (Courant number > 0).  The
// expression is the two branches of stages.h ppm_flux written once (al0 is the edge value ON the interface, B the one behind it).
template <class T, int S>
DEV T tp2_flux2(const T* p, const T& cc, bool up, const double* w0, const double* w1) {
  const T qa = p[-2 * S], qb = p[-S], qt = p[0], qd = p[S], qe = p[2 * S];
  const T a0 = w0[0] * qa + w0[1] * qb + w0[2] * qt + w0[3] * qd;
  const T a1 = w1[0] * qb + w1[1] * qt + w1[2] * qd + w1[3] * qe;
  const T al0 = tp2_sel<T>(up, a1, a0), B = tp2_sel<T>(up, a0, a1);
  const T sc = tp2_sel<T>(up, cc, -cc);
  return qt + (1. - sc) * (al0 - qt - sc * (B + al0 - (qt + qt)));
}

// One block: cells I0..I1 x J0..J1 of level k of one tile.  NWV waves share it (host emulation: NWV = 1, tid = 0).
template <class T, int NWV>
DEV void tp2_block(const TpFusedArgs& a, const Ctx& c, int tile, int k, int bx, int by, T* lds, int tid) {
  // timing experiments (tools/kbench.py --group, a library built with -DFV3LM_TP2_EXP=1; results are garbage): a.exp bit 0 = fluxes without
  // arithmetic, bit 1 = no global loads after the q tile, bit 2 = no global stores
  struct IO {
    DEV static T ld(const Fld& f, size_t n, int ex = 0) { if (FV3LM_TP2_EXP && (ex & 2)) return T(1.e-3 * double(n & 7)); return TpfIO<T>::ld(f, n); }
    DEV static void st(const Fld& f, size_t n, const T& x, int ex = 0) { if (FV3LM_TP2_EXP && (ex & 4)) return; TpfIO<T>::st(f, n, x); }
  };
#ifdef FV3LM_HOST_EMUL
  constexpr int NTH = 1;           // one "thread" runs every element
#else
  constexpr int NTH = 64 * NWV;
#endif
  const LevelParams& lev = c.lev[k - 1];
  const bool split = level_split(lev, a.hsel, a.dsel);
  const Geom& g = c.g;
  const int nx = g.nx, ny = g.ny, pi = g.pi;
  const bool face = g.face != 0;
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je();      // the tile's window of the face (nx, ny: the FACE, for the edge formulas)
  const int I0 = is + bx * TP2_W, I1 = (I0 + TP2_W - 1 < ie) ? I0 + TP2_W - 1 : ie;
  const int J0 = js + by * TP2_H, J1 = (J0 + TP2_H - 1 < je) ? J0 + TP2_H - 1 : je;
  const bool firstx = bx == 0, lastx = I1 == ie, firsty = by == 0, lasty = J1 == je;
  const size_t base = (size_t)(tile * a.nk + k - 1) * g.plane + (size_t)g.idx(I0 - 3, J0 - 3);     // element (x, r) of the halo'd tile: base + r * pi + x
  const size_t mbase = (size_t)tile * g.plane + (size_t)g.idx(I0 - 3, J0 - 3);
  T* const Q = lds + TP2_OQ; T* const FX2 = lds + TP2_OFX2; T* const FY2 = lds + TP2_OFY2; T* const QI = lds + TP2_OQI; T* const QJ = lds + TP2_OQJ;
  double* const EWX = reinterpret_cast<double*>(lds + TP2_NT);      // [2][TP2_RQ][4]: rows, m = 1 and m = nx + 1
  double* const EWY = EWX + 2 * 4 * TP2_RQ;                         // [2][TP2_PQ][4]: columns, m = 1 and m = ny + 1
  const int iord = hord_of(lev, a.hsel);
  const bool fast = iord == 1 || iord == 2;
  // positions of the edge values the x-sweeps of this block form: I0-1 .. I1+2 (y: J0-1 .. J1+2); special within two cells of a cube edge
  const bool nearx = face && (I0 - 1 <= 2 || I1 + 2 >= nx), neary = face && (J0 - 1 <= 2 || J1 + 2 >= ny);
  auto own_i = [&](int i) { return (i >= I0 && i <= I1) || (firstx && i < I0) || (lastx && i > I1); };
  auto own_j = [&](int j) { return (j >= J0 && j <= J1) || (firsty && j < J0) || (lasty && j > J1); };
  auto acc_flux = [&](const Fld& acc, size_t n, const T& f) {
    if (!split) IO::st(acc, n, IO::ld(acc, n, a.exp) + f, a.exp);
    else if constexpr (!std::is_same<T, double>::value) acc.p[n] += f.d;
  };
  // Global operands of the next phase are fetched into registers before the barrier that ends the current one: with one or a few blocks
  // per CU nothing else hides the HBM latency behind a barrier (measured without: 1.0 of 2.4 ms per tangent launch waiting on these loads).
#ifdef FV3LM_HOST_EMUL
  constexpr bool PRE = false;
#else
  constexpr bool PRE = std::is_same<T, double>::value ? (FV3LM_TP2_PRE_NL != 0) : (FV3LM_TP2_PRE_TL != 0);
#endif
  constexpr int NIH = PRE ? (TP2_H + NWV - 1) / NWV : 1, NIH1 = PRE ? (TP2_H + 1 + NWV - 1) / NWV : 1, NIQ = PRE ? (TP2_RQ + NWV - 1) / NWV : 1;
  T cy1[NIH1], cx1[NIQ];                                                                            // inner sweeps
  T y0_2[NIH], y1_2[NIH], ry_2[NIH], x0_2[NIQ], x1_2[NIQ], rx_2[NIQ]; double ai_2[NIH], aj_2[NIQ];   // q_i, q_j
  T cx3[NIH], mx3[NIH], cy3[NIH1], my3[NIH1];                                                       // outer sweeps
  if constexpr (PRE) {
    tp2_fetch<NWV, TP2_H + 1>(tid, [&](int x, int r, int it) TP2_INLINE {
      cy1[it] = (I0 - 3 + x <= I1 + 3 && J0 + r <= J1 + 1) ? IO::ld(a.cry, base + (size_t)((r + 3) * pi + x), a.exp) : T(0.); });
    tp2_fetch<NWV, TP2_RQ>(tid, [&](int x, int r, int it) TP2_INLINE {
      cx1[it] = (I0 + x <= I1 + 1 && J0 - 3 + r <= J1 + 3) ? IO::ld(a.crx, base + (size_t)(r * pi + x + 3), a.exp) : T(0.); });
  }
  // ---- the block of q and its halo; the metric-dependent edge weights of the rows / columns of this block
  tp2_sweep<NWV, TP2_PQ, TP2_RQ>(tid, [&](int x, int r, int it) TP2_INLINE {
    if (I0 - 3 + x <= I1 + 3 && J0 - 3 + r <= J1 + 3) Q[r * TP2_PQ + x] = TpfIO<T>::ld(a.q, base + (size_t)(r * pi + x));
  });
  if (nearx) {
    const bool lo = I0 - 1 <= 1, hi = I1 + 2 >= nx + 1;
    for (int e = tid; e < 2 * TP2_RQ; e += NTH) {
      const int s = e / TP2_RQ, r = e % TP2_RQ, j = J0 - 3 + r;
      if (j > J1 + 3 || !(s ? hi : lo)) continue;
      const MetX da{c.m.dxa, c, tile, j};
      ppm_w(true, s ? nx + 1 : 1, nx + 1, da, EWX + (s * TP2_RQ + r) * 4);
    }
  }
  if (neary) {
    const bool lo = J0 - 1 <= 1, hi = J1 + 2 >= ny + 1;
    for (int e = tid; e < 2 * TP2_PQ; e += NTH) {
      const int s = e / TP2_PQ, x = e % TP2_PQ, i = I0 - 3 + x;
      if (i > I1 + 3 || !(s ? hi : lo)) continue;
      const MetY da{c.m.dya, c, tile, i};
      ppm_w(true, s ? ny + 1 : 1, ny + 1, da, EWY + (s * TP2_PQ + x) * 4);
    }
  }
  TPF_SYNC();
  if constexpr (PRE) {
    tp2_fetch<NWV, TP2_H>(tid, [&](int x, int r, int it) TP2_INLINE {
      const bool in = I0 - 3 + x <= I1 + 3 && J0 + r <= J1;
      const size_t o = (size_t)((r + 3) * pi + x), n = base + o;
      y0_2[it] = in ? IO::ld(a.yfx, n, a.exp) : T(0.); y1_2[it] = in ? IO::ld(a.yfx, n + pi, a.exp) : T(0.); ry_2[it] = in ? IO::ld(a.ray, n, a.exp) : T(1.);
      ai_2[it] = in ? c.m.area[mbase + o] : 0.; });
    tp2_fetch<NWV, TP2_RQ>(tid, [&](int x, int r, int it) TP2_INLINE {
      const bool in = I0 + x <= I1 && J0 - 3 + r <= J1 + 3;
      const size_t o = (size_t)(r * pi + x + 3), n = base + o;
      x0_2[it] = in ? IO::ld(a.xfx, n, a.exp) : T(0.); x1_2[it] = in ? IO::ld(a.xfx, n + 1, a.exp) : T(0.); rx_2[it] = in ? IO::ld(a.rax, n, a.exp) : T(1.);
      aj_2[it] = in ? c.m.area[mbase + o] : 0.; });
  }
  const double W0[4] = {-1. / 12., 7. / 12., 7. / 12., -1. / 12.};
  // flux of a y-sweep at interface j of the line with LDS column pointer col (element of row J0-3 of that line), pitch S
  // x: column index into EWY.  r3: row index of interface row j in the line's array (j - first row of the array).
  auto yflux = [&](const T* line0, auto S_, int r3, int j, int x, const T& cc) TP2_INLINE -> T {
    constexpr int S = decltype(S_)::value;
    const bool up = val(cc) > 0.;
    const T* p = line0 + (r3 - (up ? 1 : 0)) * S;
    if (iord == 1) return p[0];
    if (FV3LM_TP2_EXP && (a.exp & 1)) return p[0] * cc;
    if (neary && (j <= 3 || j >= ny - 1)) {
      double w0[4], w1[4];
      const int u = j - (up ? 1 : 0);
      tp2_weights(u, ny + 1, EWY + x * 4, EWY + (TP2_PQ + x) * 4, w0); tp2_weights(u + 1, ny + 1, EWY + x * 4, EWY + (TP2_PQ + x) * 4, w1);
      return tp2_flux2<T, S>(p, cc, up, w0, w1);
    }
    return tp2_flux2<T, S>(p, cc, up, W0, W0);
  };
  auto xflux = [&](const T* row0, int c3, int i, int r, const T& cc) TP2_INLINE -> T {      // r: row index into EWX; c3: index of cell i in the row
    const bool up = val(cc) > 0.;
    const T* p = row0 + (c3 - (up ? 1 : 0));
    if (iord == 1) return p[0];
    if (FV3LM_TP2_EXP && (a.exp & 1)) return p[0] * cc;
    if (nearx) {
      double w0[4], w1[4];
      const int u = i - (up ? 1 : 0);
      tp2_weights(u, nx + 1, EWX + r * 4, EWX + (TP2_RQ + r) * 4, w0); tp2_weights(u + 1, nx + 1, EWX + r * 4, EWX + (TP2_RQ + r) * 4, w1);
      return tp2_flux2<T, 1>(p, cc, up, w0, w1);
    }
    return tp2_flux2<T, 1>(p, cc, up, W0, W0);
  };
  // ---- inner sweeps: fy2 = yppm(q) on the halo'd columns (copy_corners view 2), fx2 = xppm(q) on the halo'd rows (view 1)
  tp2_sweep<NWV, TP2_PQ, TP2_H + 1>(tid, [&](int x, int r, int it) TP2_INLINE {
    const int i = I0 - 3 + x, j = J0 + r;
    if (i > I1 + 3 || j > J1 + 1) return;
    const size_t n = base + (size_t)((r + 3) * pi + x);
    const T cc = (PRE && it >= 0) ? cy1[it < 0 ? 0 : it] : IO::ld(a.cry, n, a.exp);
    T f;
    if (!fast || (face && (i < 1 || i > nx) && (j <= 3 || j >= ny - 1))) {       // corner columns next to a face edge: the rotated view
      auto line = [&](int jj) -> T { int ii = i, j2 = jj; if (face) corner_map(g, 2, ii, j2); return Q[(j2 - (J0 - 3)) * TP2_PQ + (ii - (I0 - 3))]; };
      const MetY da{c.m.dya, c, tile, i};
      f = ppm_flux<T>(iord, face, j, ny + 1, line, da, cc);
    } else f = yflux(Q + x, std::integral_constant<int, TP2_PQ>(), r + 3, j, x, cc);
#ifdef TP2_DEBUG
    { auto line = [&](int jj) -> T { int ii = i, j2 = jj; if (face) corner_map(g, 2, ii, j2); return Q[(j2 - (J0 - 3)) * TP2_PQ + (ii - (I0 - 3))]; };
      const MetY da{c.m.dya, c, tile, i};
      T f_ = ppm_flux<T>(iord, face, j, ny + 1, line, da, cc);
      if (val(f_) != val(f)) std::fprintf(stderr, "FY2 i %d j %d: %.17g vs %.17g (c %g)\n", i, j, val(f), val(f_), val(cc)); }
#endif
    FY2[r * TP2_PQ + x] = f;
    if (a.do_acc && own_i(i) && (j <= J1 || lasty)) IO::st(a.acy, n, IO::ld(a.acy, n, a.exp) + cc, a.exp);
  });
  tp2_sweep<NWV, TP2_PX, TP2_RQ>(tid, [&](int x, int r, int it) TP2_INLINE {
    const int i = I0 + x, j = J0 - 3 + r;
    if (i > I1 + 1 || j > J1 + 3) return;
    const size_t n = base + (size_t)(r * pi + x + 3);
    const T cc = (PRE && it >= 0) ? cx1[it < 0 ? 0 : it] : IO::ld(a.crx, n, a.exp);
    T f;
    if (!fast || (face && (j < 1 || j > ny) && (i <= 3 || i >= nx - 1))) {
      auto line = [&](int ii) -> T { int i2 = ii, jj = j; if (face) corner_map(g, 1, i2, jj); return Q[(jj - (J0 - 3)) * TP2_PQ + (i2 - (I0 - 3))]; };
      const MetX da{c.m.dxa, c, tile, j};
      f = ppm_flux<T>(iord, face, i, nx + 1, line, da, cc);
    } else f = xflux(Q + r * TP2_PQ, x + 3, i, r, cc);
#ifdef TP2_DEBUG
    { auto line = [&](int ii) -> T { int i2 = ii, jj = j; if (face) corner_map(g, 1, i2, jj); return Q[(jj - (J0 - 3)) * TP2_PQ + (i2 - (I0 - 3))]; };
      const MetX da{c.m.dxa, c, tile, j};
      T f_ = ppm_flux<T>(iord, face, i, nx + 1, line, da, cc);
      if (val(f_) != val(f)) std::fprintf(stderr, "FX2 i %d j %d: %.17g vs %.17g (c %g)\n", i, j, val(f), val(f_), val(cc)); }
#endif
    FX2[r * TP2_PX + x] = f;
    if (a.do_acc && own_j(j) && (i <= I1 || lastx)) IO::st(a.acx, n, IO::ld(a.acx, n, a.exp) + cc, a.exp);
  });
  TPF_SYNC();
  if constexpr (PRE) {
    tp2_fetch<NWV, TP2_H>(tid, [&](int x, int r, int it) TP2_INLINE {
      const int i = I0 + x; const bool in = J0 + r <= J1 && (i <= I1 || (lastx && i == I1 + 1));
      const size_t n = base + (size_t)((r + 3) * pi + x + 3);
      cx3[it] = in ? IO::ld(a.crx, n, a.exp) : T(0.); mx3[it] = in ? IO::ld(a.mx, n, a.exp) : T(0.); });
    tp2_fetch<NWV, TP2_H + 1>(tid, [&](int x, int r, int it) TP2_INLINE {
      const int j = J0 + r; const bool in = I0 + x <= I1 && (j <= J1 || (lasty && j == J1 + 1));
      const size_t n = base + (size_t)((r + 3) * pi + x + 3);
      cy3[it] = in ? IO::ld(a.cry, n, a.exp) : T(0.); my3[it] = in ? IO::ld(a.my, n, a.exp) : T(0.); });
  }
  // ---- q_i, q_j: the field advanced by the inner fluxes (tp_core_tlm.F90:149-159, :173-181)
  tp2_sweep<NWV, TP2_PQ, TP2_H>(tid, [&](int x, int r, int it) TP2_INLINE {
    const int i = I0 - 3 + x, j = J0 + r;
    if (i > I1 + 3 || j > J1) return;
    const size_t o = (size_t)((r + 3) * pi + x), n = base + o;
    const bool pf = PRE && it >= 0; const int u = it < 0 ? 0 : it;
    const T f0 = (pf ? y0_2[u] : IO::ld(a.yfx, n, a.exp)) * FY2[r * TP2_PQ + x], f1 = (pf ? y1_2[u] : IO::ld(a.yfx, n + pi, a.exp)) * FY2[(r + 1) * TP2_PQ + x];
    QI[r * TP2_PQ + x] = (Q[(r + 3) * TP2_PQ + x] * (pf ? ai_2[u] : c.m.area[mbase + o]) + f0 - f1) / (pf ? ry_2[u] : IO::ld(a.ray, n, a.exp));
  });
  tp2_sweep<NWV, TP2_W, TP2_RQ>(tid, [&](int x, int r, int it) TP2_INLINE {
    const int i = I0 + x, j = J0 - 3 + r;
    if (i > I1 || j > J1 + 3) return;
    const size_t o = (size_t)(r * pi + x + 3), n = base + o;
    const bool pf = PRE && it >= 0; const int u = it < 0 ? 0 : it;
    const T f0 = (pf ? x0_2[u] : IO::ld(a.xfx, n, a.exp)) * FX2[r * TP2_PX + x], f1 = (pf ? x1_2[u] : IO::ld(a.xfx, n + 1, a.exp)) * FX2[r * TP2_PX + x + 1];
    QJ[r * TP2_W + x] = (Q[r * TP2_PQ + x + 3] * (pf ? aj_2[u] : c.m.area[mbase + o]) + f0 - f1) / (pf ? rx_2[u] : IO::ld(a.rax, n, a.exp));
  });
  TPF_SYNC();
  // ---- outer sweeps and flux assembly (tp_core_tlm.F90:187-234; deln_flux :1918-2043 as in stages.h TpFlux)
  int nord; double dc; damp_of(lev, a.dsel, nord, dc, split);
  const bool dmp = (a.dsel != DAMP_NONE) && (dc > 1.e-4);
  double damp = 0.;
  if (dmp) damp = damp_pow(dc * c.m.da_min, nord);
  const Fld& d2 = a.d2b;
  tp2_sweep<NWV, TP2_PX, TP2_H>(tid, [&](int x, int r, int it) TP2_INLINE {        // fx(i, j)
    const int i = I0 + x, j = J0 + r;
    if (j > J1 || i > I1 + 1 || !(i <= I1 || lastx)) return;
    const size_t o = (size_t)((r + 3) * pi + x + 3), n = base + o;
    const bool pf = PRE && it >= 0; const int u = it < 0 ? 0 : it;
    const T cc = pf ? cx3[u] : IO::ld(a.crx, n, a.exp);
    T fo;
    if (!fast) {
      auto line = [&](int ii) -> T { return QI[r * TP2_PQ + (ii - (I0 - 3))]; };
      const MetX da{c.m.dxa, c, tile, j};
      fo = ppm_flux<T>(iord, face, i, nx + 1, line, da, cc);
    } else fo = xflux(QI + r * TP2_PQ, x + 3, i, r + 3, cc);
    T f = 0.5 * (fo + FX2[(r + 3) * TP2_PX + x]) * (pf ? mx3[u] : IO::ld(a.mx, n, a.exp));
    if (dmp) {
      T f2;
      if (nord == 0) { f2 = c.m.del6_v[mbase + o] * (Q[(r + 3) * TP2_PQ + x + 2] - Q[(r + 3) * TP2_PQ + x + 3]); if (!a.use_mass) f2 = damp * f2; }
      else f2 = c.m.del6_v[mbase + o] * (IO::ld(d2, n, a.exp) - IO::ld(d2, n - 1, a.exp));
      if (a.use_mass) f = f + (0.5 * damp) * (IO::ld(a.mass, n - 1, a.exp) + IO::ld(a.mass, n, a.exp)) * f2;
      else f = f + f2;
    }
    IO::st(a.fx, n, f, a.exp);
    if (a.do_acc) acc_flux(a.amfx, n, f);
  });
  tp2_sweep<NWV, TP2_W, TP2_H + 1>(tid, [&](int x, int r, int it) TP2_INLINE {       // fy(i, j)
    const int i = I0 + x, j = J0 + r;
    if (i > I1 || j > J1 + 1 || !(j <= J1 || lasty)) return;
    const size_t o = (size_t)((r + 3) * pi + x + 3), n = base + o;
    const bool pf = PRE && it >= 0; const int u = it < 0 ? 0 : it;
    const T cc = pf ? cy3[u] : IO::ld(a.cry, n, a.exp);
    T fo;
    if (!fast) {
      auto line = [&](int jj) -> T { return QJ[(jj - (J0 - 3)) * TP2_W + x]; };
      const MetY da{c.m.dya, c, tile, i};
      fo = ppm_flux<T>(iord, face, j, ny + 1, line, da, cc);
    } else fo = yflux(QJ + x, std::integral_constant<int, TP2_W>(), r + 3, j, x + 3, cc);
    T f = 0.5 * (fo + FY2[r * TP2_PQ + x + 3]) * (pf ? my3[u] : IO::ld(a.my, n, a.exp));
    if (dmp) {
      T f2;
      if (nord == 0) { f2 = c.m.del6_u[mbase + o] * (Q[(r + 2) * TP2_PQ + x + 3] - Q[(r + 3) * TP2_PQ + x + 3]); if (!a.use_mass) f2 = damp * f2; }
      else f2 = c.m.del6_u[mbase + o] * (IO::ld(d2, n, a.exp) - IO::ld(d2, n - pi, a.exp));
      if (a.use_mass) f = f + (0.5 * damp) * (IO::ld(a.mass, n - pi, a.exp) + IO::ld(a.mass, n, a.exp)) * f2;
      else f = f + f2;
    }
    IO::st(a.fy, n, f, a.exp);
    if (a.do_acc) acc_flux(a.amfy, n, f);
  });
}

inline void tp2_grid(const Geom& g, int& nbx, int& nby) { nbx = (g.tx + TP2_W - 1) / TP2_W; nby = (g.ty + TP2_H - 1) / TP2_H; }

#ifndef FV3LM_HOST_EMUL
template <class T, int NWV>
__global__ void __launch_bounds__(64 * NWV) k_tp2(TpFusedArgs a, Ctx c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tp2_lds[];
  int bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by);
  tp2_block<T, NWV>(a, c, blockIdx.z / a.nk, 1 + blockIdx.z % a.nk, bx, by, reinterpret_cast<T*>(tp2_lds), threadIdx.x);
}
#endif

// nonlinear (nothing stored for the staged adjoint) / tangent-linear launch; profile names as tpfused.h: it is the same routine
FV3LM_LINK void run_tp2(Exec& ex, int mode, const TpFusedArgs& a0, const Ctx& c) {
  TpFusedArgs a = a0;
  for (Fld* f : {&a.q, &a.crx, &a.cry, &a.xfx, &a.yfx, &a.rax, &a.ray, &a.mx, &a.my, &a.mass, &a.d2b, &a.d2b_t, &a.fx, &a.fy, &a.acx, &a.acy, &a.amfx, &a.amfy}) *f = ex.sh(*f);
  a.do_acc = (a.acx.t && !ex.skip_accum) ? 1 : 0;
  a.store_mid = 0;
#if FV3LM_TP2_EXP
  if (const char* e = std::getenv("FV3LM_TP2_EXPERIMENT")) a.exp = std::atoi(e);
#endif
  int nbx, nby; tp2_grid(c.g, nbx, nby);
  ex.mark_begin("TpFused", mode == MODE_TL ? ".tl" : ".nl", tpf_bytes(a, c.g, mode));
#ifdef FV3LM_HOST_EMUL
  std::vector<double> lds((tp2_lds_bytes<Dual>() + 7) / 8);
  for (int z = 0; z < c.g.ntile * a.nk; ++z)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) {
        if (mode == MODE_TL) tp2_block<Dual, 1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, reinterpret_cast<Dual*>(lds.data()), 0);
        else tp2_block<double, 1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0);
      }
#else
  const dim3 grid(nbx, nby, c.g.ntile * a.nk);
  static bool attr = false;      // more LDS per block than the 64 KB default limit of a launch
  if (!attr) {
    if ((tp2_lds_bytes<Dual>() <= 160 * 1024 && hipFuncSetAttribute((const void*)k_tp2<Dual, FV3LM_TP2_WAVES_TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp2_lds_bytes<Dual>()) != hipSuccess) ||
        hipFuncSetAttribute((const void*)k_tp2<double, FV3LM_TP2_WAVES_NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tp2_lds_bytes<double>()) != hipSuccess) set_sticky("hipFuncSetAttribute(k_tp2) failed");
    attr = true;
  }
  if (mode == MODE_TL) hipLaunchKernelGGL((k_tp2<Dual, FV3LM_TP2_WAVES_TL>), grid, dim3(64 * FV3LM_TP2_WAVES_TL), tp2_lds_bytes<Dual>(), ex.stream, a, c);
  else hipLaunchKernelGGL((k_tp2<double, FV3LM_TP2_WAVES_NL>), grid, dim3(64 * FV3LM_TP2_WAVES_NL), tp2_lds_bytes<double>(), ex.stream, a, c);
#endif
  ex.mark_end();
  ex.launches++;
}

}  // namespace fv3
#endif
