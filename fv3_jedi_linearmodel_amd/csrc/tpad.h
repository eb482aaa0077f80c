// fv3lm-hip: the ADJOINT of fv_tp_2d (FV_TP_2D_FWD / _BWD, tp_core_adm.F90:7026 / :7263; joint form FV_TP_2D_ADM :109) as ONE LDS-tiled
// kernel -- flux assembly, the two outer sweeps, the two intermediate updates and the two inner sweeps, transposed in closed form.
//
// Round 2 ran it as one hand-written launch for the outer half (tp_outer_ad_block) + six staged gather launches, their strips and corner
// launches for the inner half, with the adjoints of q_i, q_j, fx2, fy2 and the trajectory of the same four making round trips through
// HBM.  Here a workgroup owns a 64 x 16 block of cells of one level and GATHERS everything that block owns:
//
//   values (as the forward kernel, recomputed, never stored):  q tile -> fy2 -> q_i          (branch A)        q tile -> fx2 -> q_j   (branch B)
//   A: fxo_ad = fx2_ad' = 0.5 mx fx_ad                      q_i_ad(k)  = sum_m dfxo(m)/dq_i(k) fxo_ad(m)       gi = q_i_ad / ra_y
//      fy2_ad(m) = 0.5 my fy_ad(m) + yfx(m) (gi(m) - gi(m-1))                 q_ad(k) += area gi(k) + sum_m dfy2(m)/dq(k) fy2_ad(m)
//      crx_ad += dfxo/dc fxo_ad      mx_ad += 0.5 fxo fx_ad     ra_y_ad -= q_i gi     yfx_ad += fy2 (gi(m) - gi(m-1))
//      cry_ad += dfy2/dc fy2_ad      my_ad += 0.5 fy2 fy_ad
//   B: the same with x and y exchanged.
//
// The schemes the adjoint is taken of (iord 1, 2, 333) are linear in q for given Courant numbers, so dflux/dq is the closed form ppm_dq
// (stages.h; the four-cell cube-edge values included) and dflux/dc one ppm_flux<Dual> with only c seeded.  Every output element is
// owned by exactly one thread: no atomics, fixed summation order.  The outputs ACCUMULATE (the launch never stores: plan_adjoint clears
// what it touches first).  The damping part of the flux keeps its stage (stages.h TpDamp).
//
// Corner halo (cube faces): the inner sweeps read corner-halo cells through the copy_corners view of their direction, i.e. a source
// cell next to a face corner is ALSO read as a virtual corner cell.  Those few extra contributions (9 source cells per corner and
// direction) are added by a second, tiny launch that re-derives what it needs from the same global inputs (k_tp_ad_corner).
#pragma once
#include "tpfused.h"

namespace fv3 {
FV3LM_LINK void run_tp_ad(Exec& ex, const TpFusedArgs& a0, const Ctx& c);
}

#if !defined(FV3LM_SPLIT_BUILD) || defined(FV3LM_IMPL_TPAD)
namespace fv3 {

#ifndef FV3LM_TPD_H
#define FV3LM_TPD_H 32
#endif
// FV3LM_TPD_NOUNROLL: the element loops of a phase as real loops (code size / 4: the unrolled kernel is 350 KB of straight-line code);
// the area term of q_ad then goes to q_ad at once instead of waiting in a register for the inner sweep's term
#if defined(FV3LM_TPD_NOUNROLL) && !defined(FV3LM_HOST_EMUL)
#define TPD_LOOP(e, n) _Pragma("nounroll") for (int e = tid; e < (n); e += NTH)
#define TPD_LOOPU(e, u, n, E) _Pragma("nounroll") for (int e = tid, u = 0; e < (n); e += NTH)
#define TPD_QACC_DIRECT 1
#else
#define TPD_LOOP(e, n) TPF_LOOP(e, n)
#define TPD_LOOPU(e, u, n, E) TPF_LOOPU(e, u, n, E)
#define TPD_QACC_DIRECT 0
#endif
// operands of the next phase fetched into registers before the barrier that ends the current one (0: every load where it is used)
#ifndef FV3LM_TPD_PREFETCH
#define FV3LM_TPD_PREFETCH 1
#endif
#ifndef FV3LM_TPD_EXP          /* timing experiments (tools/kbench.py --group; results are garbage): a.exp bits switch phases off */
#define FV3LM_TPD_EXP 0
#endif
#ifndef FV3LM_TPD_THREADS
#define FV3LM_TPD_THREADS 1024
#endif
constexpr int TPD_W = 64, TPD_H = FV3LM_TPD_H;
constexpr int TPD_QW = TPD_W + 6, TPD_QH = TPD_H + 6;        // halo'd tile
constexpr int TPD_AW = TPD_W + 12, TPD_AH = TPD_H + 12;      // flux-adjoint tile: three more either side along the sweep
constexpr int TPD_NQ = TPD_QW * TPD_QH;                      // q; q_i / q_j values, later gi / gj
constexpr int TPD_NF2 = TPD_QW * (TPD_QH + 1);               // inner flux values, later (in place) their adjoints
constexpr int TPD_NFA = (TPD_AW * TPD_QH > TPD_AH * TPD_QW ? TPD_AW * TPD_QH : TPD_AH * TPD_QW);    // outer-flux adjoint tile (either orientation)
constexpr int TPD_NC = TPD_NFA;                              // Courant numbers of the sweep being transposed (layout of the flux-adjoint tile)
constexpr int TPD_NT = 2 * TPD_NQ + TPD_NF2 + 2 * TPD_NFA + TPD_NC;   // 64 x 32 cells: 14,210 doubles = 114 KB, one 1024-thread block per CU (measured: 4.55 ms per launch at C192L127 x 6; 64 x 16 / 512 threads / two blocks: 4.81)
constexpr int TPD_THREADS = FV3LM_TPD_THREADS;

// Transposed 1-D sweep in two stages.  flux(m) of iord 2 is  q_up c'(3 -+ 2c) + al(m) (1 -+ c)^2 -+ al(m -+ 1) c (1 -+ c)  (upper signs: c > 0,
// upwind cell m-1; c' = |c|), al(x) = sum_e w_e(x) q(x-2+e): first the adjoint of the edge values al_ad(x) from the (at most three) fluxes
// that read al(x), then q_ad(k) from the four edge values that read q(k) + the upwind terms.  The flux adjoints f come from LDS tiles that
// hold zeros outside the range of the fluxes, the Courant numbers c are loaded by the caller (all of a thread's global loads first, then
// the arithmetic: nothing here waits on memory).
DEV double tps_al(double fm, double f0, double fp, double cm, double c0, double cp) {      // flux x-1, x, x+1
  double s = (c0 > 0. ? (1. - c0) * (1. - c0) : (1. + c0) * (1. + c0)) * f0;
  if (cp > 0.) s -= cp * (1. - cp) * fp;
  if (!(cm > 0.)) s += cm * (1. + cm) * fm;
  return s;
}
template <class D>
DEV double tps_w(bool face, int x, int n1, const D& da, int k) {      // coefficient of q(k) in al(x) (stages.h ppm_al_coef; constants away from the cube edges)
  const int e = k - x + 2;
  if (e < 0 || e > 3) return 0.;
  if (!face || (x >= 3 && x <= n1 - 2)) return (e == 0 || e == 3) ? P2 : P1;
  double w[4]; ppm_w(face, x, n1, da, w);
  return w[e];
}
// q_ad(k): f0, fp = adjoints of flux k, k+1; c0, cp their Courant numbers; al[0..3] = al_ad(k-1 .. k+2); fm2 .. fp2 (iord 333 only) fluxes k-1 .. k+2
template <class D>
DEV double tps_q(int iord, bool face, int k, int n1, double f0, double fp, double c0, double cp, const double* al, const D& da) {
  double s = 0.;
  if (cp > 0.) s += (iord == 1 ? 1. : cp * (3. - 2. * cp)) * fp;
  if (!(c0 > 0.)) s += (iord == 1 ? 1. : -c0 * (3. + 2. * c0)) * f0;
  if (iord == 1) return s;
  if (!face || (k >= 4 && k <= n1 - 4)) return s + P2 * al[0] + P1 * al[1] + P1 * al[2] + P2 * al[3];      // the four edge values k-1 .. k+2 away from the cube edges: constant weights
#pragma unroll
  for (int n = 0; n < 4; ++n) if (al[n] != 0.) s += tps_w(face, k - 1 + n, n1, da, k) * al[n];
  return s;
}
// iord 333 has no edge values: three cells per flux (stages.h ppm_dq), fluxes k-1 .. k+2
template <class D>
DEV double tps_q333(bool face, int k, int n1, const double* f, const double* cc, const D& da) {
  double s = 0.;
#pragma unroll
  for (int n = 0; n < 4; ++n) { const int m = k - 1 + n; if (m >= 1 && m <= n1 && f[n] != 0.) s += ppm_dq(333, face, m, n1, k, da, cc[n]) * f[n]; }
  return s;
}

// Scheme-2 flux at interface m of a line held in LDS (cell0 = the element of cell m, S doubles between neighbours): value and derivative
// with respect to the Courant number from the five cells around the UPWIND cell -- two edge values instead of the three of the generic
// ppm_flux<Dual>, no dual arithmetic, the cube-edge weights (edges.h ppm_w) only within two cells of an edge.  Same expression as tp2.h.
template <int S, class D>
DEV void tpd_flux2(bool face, int m, int n1, const double* cell0, double c, const D& da, double& v, double& dc) {
  const bool up = c > 0.;
  const int u = m - (up ? 1 : 0);
  const double* p = cell0 - (up ? S : 0);
  double w0[4] = {P2, P1, P1, P2}, w1[4] = {P2, P1, P1, P2};
  if (face && (u <= 2 || u >= n1 - 2)) { ppm_w(true, u, n1, da, w0); ppm_w(true, u + 1, n1, da, w1); }
  const double qa = p[-2 * S], qb = p[-S], qt = p[0], qd = p[S], qe = p[2 * S];
  const double a0 = w0[0] * qa + w0[1] * qb + w0[2] * qt + w0[3] * qd, a1 = w1[0] * qb + w1[1] * qt + w1[2] * qd + w1[3] * qe;
  const double al0 = up ? a1 : a0, B = up ? a0 : a1, sc = up ? c : -c;
  const double X = al0 - qt, Y = B + al0 - (qt + qt), Z = X - sc * Y;
  v = qt + (1. - sc) * Z;
  const double dsc = -Z - (1. - sc) * Y;
  dc = up ? dsc : -dsc;
}

// One block: cells I0..I1 x J0..J1 of level k of one tile (+ the halo cells next to them in the first / last block row and column).
template <int NTH>
DEV void tp_ad_block(const TpFusedArgs& a, const Ctx& c, int tile, int k, int bx, int by, double* lds, int tid, int nth) {
  (void)nth;
  const Geom& g = c.g;
  const int nx = g.nx, ny = g.ny;                                     // the FACE: cube-edge weights and corner views test global indices against these
  const int is = g.is(), ie = g.ie(), js = g.js(), je = g.je();       // the tile's window of the face: ranges of cells and fluxes
  const bool face = g.face != 0;
  const int I0 = is + bx * TPD_W, I1 = (I0 + TPD_W - 1 < ie) ? I0 + TPD_W - 1 : ie;
  const int J0 = js + by * TPD_H, J1 = (J0 + TPD_H - 1 < je) ? J0 + TPD_H - 1 : je;
  const bool firstx = bx == 0, lastx = I1 == ie, firsty = by == 0, lasty = J1 == je;
  const int GI0 = firstx ? I0 - 3 : I0, GI1 = lastx ? I1 + 3 : I1, GJ0 = firsty ? J0 - 3 : J0, GJ1 = lasty ? J1 + 3 : J1;     // owned cells incl. the halo
  const size_t base = (size_t)(tile * a.nk + k - 1) * g.plane;
  auto at = [&](int i, int j) -> size_t { return base + g.idx(i, j); };
  const int iord = hord_of(c.lev[k - 1], a.hsel);
  double* Q = lds;                       // q values                          (I0-3 .. I1+3) x (J0-3 .. J1+3)
  double* QG = lds + TPD_NQ;             // q_i / q_j values, then gi / gj    same
  double* F2 = lds + 2 * TPD_NQ;         // inner flux values, then their adjoints in place
  double* FA = F2 + TPD_NF2;             // outer-flux adjoint 0.5 m f_ad, three more cells either side along the sweep
  double* AL = FA + TPD_NFA;             // adjoint of the edge values of the sweep being transposed
  // Courant numbers of the sweep being transposed, loaded once per sweep instead of once per use (three phases read them, with neighbours):
  //   crx, x-faces I0-6 .. I1+6 of rows J0-3 .. J1+3, index cxe (= branch A's fae)      -- outer sweep of branch A, inner sweep of branch B
  //   cry, y-faces J0-6 .. J1+6 of columns I0-3 .. I1+3, index cye (= branch B's fae)   -- inner sweep of branch A, outer sweep of branch B
  // zero outside the range of the fluxes (where the flux adjoints are zero as well)
  double* CT = AL + TPD_NFA;
  auto cxe = [&](int i, int j) { return (j - (J0 - 3)) * TPD_AW + (i - (I0 - 6)); };
  auto cye = [&](int i, int j) { return (j - (J0 - 6)) * TPD_QW + (i - (I0 - 3)); };
  auto load_cx = [&](int e) -> double { const int i = I0 - 6 + e % TPD_AW, j = J0 - 3 + e / TPD_AW; return (i >= is && i <= ie + 1 && i <= I1 + 6 && j <= J1 + 3) ? a.crx.t[at(i, j)] : 0.; };
  auto load_cy = [&](int e) -> double { const int i = I0 - 3 + e % TPD_QW, j = J0 - 6 + e / TPD_QW; return (j >= js && j <= je + 1 && j <= J1 + 6 && i <= I1 + 3) ? a.cry.t[at(i, j)] : 0.; };
  // Prefetch: the global operands of a phase are loaded into registers during the phase before it, ahead of the barrier between them.
#if defined(FV3LM_HOST_EMUL) || defined(FV3LM_TPD_NOUNROLL)
  constexpr bool PRE = false;
#else
  constexpr bool PRE = FV3LM_TPD_PREFETCH != 0;
#endif
#define TPD_PF(arr, u, expr) (PRE ? arr[u] : (expr))
  // elements per thread of each phase (trip counts of the unrolled loops; the prefetch arrays have that many slots)
  constexpr int E_AQ = (TPD_AW * TPD_QH + NTH - 1) / NTH, E_QA = (TPD_QW * TPD_AH + NTH - 1) / NTH, E_Q = (TPD_NQ + NTH - 1) / NTH,
                E_V = (TPD_QW * (TPD_H + 1) + NTH - 1) / NTH, E_F = ((TPD_W + 1) * TPD_H + NTH - 1) / NTH,
                E_FB = (TPD_W * (TPD_H + 1) + NTH - 1) / NTH, E_I = (TPD_QW * TPD_H + NTH - 1) / NTH, E_J = (TPD_W * TPD_QH + NTH - 1) / NTH,
                E_G = (TPD_QW * (TPD_QH - 1) + NTH - 1) / NTH, E_GB = ((TPD_QW - 1) * TPD_QH + NTH - 1) / NTH;
#define TPD_SLOTS(E) (PRE ? (E) : 1)
  auto qe = [&](int i, int j) { return (j - (J0 - 3)) * TPD_QW + (i - (I0 - 3)); };
  auto corner_cell = [&](int i, int j) { return face && (i < 1 || i > nx) && (j < 1 || j > ny); };
  // q_ad of the owned cells: area term kept by the thread that also finishes the cell (same element loop in both phases)
  constexpr int EQ = (TPD_NQ + NTH - 1) / NTH;
  double qacc[EQ];

  auto cl = [](int m, int n1) { return m < 1 ? 1 : (m > n1 ? n1 : m); };      // face index clamped into its range (used by the iord 333 path only)
  (void)cl;
  { constexpr int n = TPD_NQ;
    TPD_LOOP(e, n) { const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW; Q[e] = (i <= I1 + 3 && j <= J1 + 3) ? a.q.t[at(i, j)] : 0.; } }

  // ============================== branch A: outer sweep in x, inner sweep in y ==============================
  {
    auto f2e = [&](int i, int j) { return (j - (J0 - 2)) * TPD_QW + (i - (I0 - 3)); };          // rows J0-2 .. J1+3
    auto fae = [&](int i, int j) { return (j - (J0 - 3)) * TPD_AW + (i - (I0 - 6)); };          // columns I0-6 .. I1+6, rows J0-3 .. J1+3
    auto ale = [&](int i, int j) { return (j - (J0 - 5)) * TPD_QW + (i - (I0 - 3)); };          // inner sweep: rows J0-5 .. J1+6
    { constexpr int n = TPD_AW * TPD_QH;            // outer-flux adjoints and the Courant numbers of the outer sweep
      TPD_LOOP(e, n) {
        const int i = I0 - 6 + e % TPD_AW, j = J0 - 3 + e / TPD_AW;
        FA[e] = (i >= is && i <= ie + 1 && j >= js && j <= je && i <= I1 + 6 && j <= J1 + 3) ? 0.5 * a.mx.t[at(i, j)] * a.fx.p[at(i, j)] : 0.;
        CT[e] = load_cx(e);
      } }
    double cv[TPD_SLOTS(E_V)];                                 // prefetch: Courant numbers of the inner flux values
    if constexpr (PRE) { constexpr int n = TPD_QW * (TPD_H + 1);
      TPD_LOOPU(e, u, n, E_V) { const int i = I0 - 3 + e % TPD_QW, j = J0 + e / TPD_QW; cv[u] = (i <= I1 + 3 && j <= J1 + 1) ? a.cry.t[at(i, j)] : 0.; } }
    TPF_SYNC();
    double ar2[TPD_SLOTS(E_I)], y02[TPD_SLOTS(E_I)], y12[TPD_SLOTS(E_I)], ry2[TPD_SLOTS(E_I)];  // prefetch: q_i values
    if constexpr (PRE) { constexpr int n = TPD_QW * TPD_H;
      TPD_LOOPU(e, u, n, E_I) { const int i = I0 - 3 + e % TPD_QW, j = J0 + e / TPD_QW; const bool in = i <= I1 + 3 && j <= J1;
        ar2[u] = in ? MET(area, i, j) : 0.; y02[u] = in ? a.yfx.t[at(i, j)] : 0.; y12[u] = in ? a.yfx.t[at(i, j + 1)] : 0.; ry2[u] = in ? a.ray.t[at(i, j)] : 1.; } }
    { constexpr int n = TPD_QW * (TPD_H + 1);       // inner flux values fy2 on rows J0 .. J1+1
      TPD_LOOPU(e, u, n, E_V) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 + e / TPD_QW;
        if (i > I1 + 3 || j > J1 + 1) continue;
        auto line = [&](int jj) -> double { int ii = i, j2 = jj; if (face) corner_map(g, 2, ii, j2); return Q[qe(ii, j2)]; };
        const MetY da{c.m.dya, c, tile, i};
        if (FV3LM_TPD_EXP && (a.exp & 1)) { F2[f2e(i, j)] = line(j) * TPD_PF(cv, u, a.cry.t[at(i, j)]); continue; }
        const double cc = TPD_PF(cv, u, a.cry.t[at(i, j)]);
        if (iord == 2 && !(face && (i < 1 || i > nx) && (j <= 3 || j >= ny - 1))) { double v_, d_; tpd_flux2<TPD_QW>(face, j, ny + 1, Q + qe(i, j), cc, da, v_, d_); F2[f2e(i, j)] = v_; }
        else F2[f2e(i, j)] = ppm_flux<double>(iord, face, j, ny + 1, line, da, cc);
      } }
    TPF_SYNC();
    double fx3[TPD_SLOTS(E_F)];                                // prefetch: own x-faces
    if constexpr (PRE) { constexpr int w = TPD_W + 1, n = w * TPD_H;
      TPD_LOOPU(e, u, n, E_F) { const int i = I0 + e % w, j = J0 + e / w; fx3[u] = (j <= J1 && (i <= I1 || (lastx && i == I1 + 1))) ? a.fx.p[at(i, j)] : 0.; } }
    { constexpr int n = TPD_QW * TPD_H;             // q_i values
      TPD_LOOPU(e, u, n, E_I) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 + e / TPD_QW;
        if (i > I1 + 3 || j > J1) continue;
        QG[qe(i, j)] = (Q[qe(i, j)] * TPD_PF(ar2, u, MET(area, i, j)) + TPD_PF(y02, u, a.yfx.t[at(i, j)]) * F2[f2e(i, j)] - TPD_PF(y12, u, a.yfx.t[at(i, j + 1)]) * F2[f2e(i, j + 1)]) / TPD_PF(ry2, u, a.ray.t[at(i, j)]);
      } }
    TPF_SYNC();
    double ry4[TPD_SLOTS(E_Q)], ar4[TPD_SLOTS(E_Q)], ro4[TPD_SLOTS(E_Q)];            // prefetch: gi
    if constexpr (PRE) { constexpr int n = TPD_NQ;
      TPD_LOOPU(e, u, n, E_Q) { const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
        const bool in = i >= GI0 && i <= GI1 && j >= js && j <= je && j <= J1 + 3, own = in && j >= J0 && j <= J1;
        ry4[u] = in ? a.ray.t[at(i, j)] : 1.; ar4[u] = in ? MET(area, i, j) : 0.; ro4[u] = own ? a.ray.p[at(i, j)] : 0.; } }
    auto FX = [&](int m, int j) { return FA[fae(m, j)]; };
    { constexpr int w = TPD_W + 1, n = w * TPD_H;   // own x-faces: Courant-number and mass-flux adjoints of the outer sweep
      TPD_LOOPU(e, u, n, E_F) {
        const int i = I0 + e % w, j = J0 + e / w;
        if (j > J1 || !(i <= I1 || (lastx && i == I1 + 1))) continue;
        const double fo = FX(i, j), fxad = TPD_PF(fx3, u, a.fx.p[at(i, j)]);
        if (fo == 0. && fxad == 0.) continue;
        if (FV3LM_TPD_EXP && (a.exp & 2)) continue;
        auto line = [&](int ii) -> Dual { return Dual(QG[qe(ii, j)], 0.); };
        const MetX da{c.m.dxa, c, tile, j};
        Dual fl;
        if (iord == 2) tpd_flux2<1>(face, i, nx + 1, QG + qe(i, j), CT[cxe(i, j)], da, fl.v, fl.d);
        else fl = ppm_flux<Dual>(iord, face, i, nx + 1, line, da, Dual(CT[cxe(i, j)], 1.));
        a.crx.p[at(i, j)] += fl.d * fo;
        a.mx.p[at(i, j)] += 0.5 * fl.v * fxad;
      } }
    if (iord == 2 && !(FV3LM_TPD_EXP && (a.exp & 4))) {                                // edge-value adjoints of the outer sweep, columns I0-4 .. I1+5
      constexpr int n = TPD_AW * TPD_QH;
      TPD_LOOP(e, n) {
        const int x = I0 - 6 + e % TPD_AW, j = J0 - 3 + e / TPD_AW;
        double s = 0.;
        if (x >= I0 - 4 && x <= I1 + 5 && j >= js && j <= je && j <= J1 + 3) s = tps_al(FA[e - 1], FA[e], FA[e + 1], CT[e - 1], CT[e], CT[e + 1]);
        AL[e] = s;
      } }
    TPF_SYNC();
    double yf5[TPD_SLOTS(E_G)], fy5[TPD_SLOTS(E_G)], mm5[TPD_SLOTS(E_G)];            // prefetch: fy2_ad
    if constexpr (PRE) { constexpr int n = TPD_QW * (TPD_QH - 1);
      TPD_LOOPU(e, u, n, E_G) { const int i = I0 - 3 + e % TPD_QW, m = J0 - 2 + e / TPD_QW;
        const bool on = i >= GI0 && i <= GI1 && m >= js && m <= je + 1 && m <= J1 + 3, in = on && i >= is && i <= ie;
        yf5[u] = on ? a.yfx.t[at(i, m)] : 0.; fy5[u] = in ? a.fy.p[at(i, m)] : 0.; mm5[u] = in ? a.my.t[at(i, m)] : 0.; } }
    { constexpr int n = TPD_NQ;                     // gi = q_i_ad / ra_y on the owned columns; own cells: ra_y_ad, the area term of q_ad
      TPD_LOOPU(e, u, n, EQ) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
        double s = 0., acc = 0.;
        if (i >= GI0 && i <= GI1 && j >= js && j <= je && j <= J1 + 3) {
          const bool own = j >= J0 && j <= J1;
          const double ry = TPD_PF(ry4, u, a.ray.t[at(i, j)]), ar = TPD_PF(ar4, u, MET(area, i, j)), rold = TPD_PF(ro4, u, own ? a.ray.p[at(i, j)] : 0.);
          const MetX da{c.m.dxa, c, tile, j};
          const int b0 = fae(i, j);
          if (iord == 333) {
            const double f[4] = {FA[b0 - 1], FA[b0], FA[b0 + 1], FA[b0 + 2]};
            const double cc[4] = {CT[b0 - 1], CT[b0], CT[b0 + 1], CT[b0 + 2]};
            s = tps_q333(face, i, nx + 1, f, cc, da);
          } else {
            const double al[4] = {AL[b0 - 1], AL[b0], AL[b0 + 1], AL[b0 + 2]};
            if (!(FV3LM_TPD_EXP && (a.exp & 8))) s = tps_q(iord, face, i, nx + 1, FA[b0], FA[b0 + 1], CT[b0], CT[b0 + 1], al, da);
          }
          s = s / ry;
          if (own) { a.ray.p[at(i, j)] = rold - QG[e] * s; acc = ar * s; }
        }
        if (TPD_QACC_DIRECT) { if (acc != 0.) a.q.p[at(i, j)] += acc; } else qacc[u] = acc;
        QG[e] = s;
      } }
    TPF_SYNC();
    double cyt[TPD_SLOTS(E_QA)], qo7[TPD_SLOTS(E_Q)];                     // the Courant numbers of the inner sweep (cry) on their way into CT; prefetch: q_ad
    if constexpr (PRE) {
      { constexpr int n = TPD_QW * TPD_AH; TPD_LOOPU(e, u, n, E_QA) cyt[u] = load_cy(e); }
      { constexpr int n = TPD_NQ; TPD_LOOPU(e, u, n, E_Q) { const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
          qo7[u] = (i < GI0 || i > GI1 || j < GJ0 || j > GJ1 || corner_cell(i, j) || i > I1 + 3 || j > J1 + 3) ? 0. : a.q.p[at(i, j)]; } }
    }
    { constexpr int n = TPD_QW * (TPD_QH - 1);      // fy2_ad on rows J0-2 .. J1+3 of the owned columns, in place of the values; own y-faces: yfx_ad, my_ad (inner part)
      TPD_LOOPU(e, u, n, E_G) {
        const int i = I0 - 3 + e % TPD_QW, m = J0 - 2 + e / TPD_QW;
        double s = 0.;
        if (i >= GI0 && i <= GI1 && m >= js && m <= je + 1 && m <= J1 + 3) {
          const bool in = i >= is && i <= ie, own = m >= J0 && (m <= J1 || (lasty && m == J1 + 1));
          const double yf = TPD_PF(yf5, u, a.yfx.t[at(i, m)]), fyad = TPD_PF(fy5, u, in ? a.fy.p[at(i, m)] : 0.), mm = TPD_PF(mm5, u, in ? a.my.t[at(i, m)] : 0.);
          const double dg = (m <= je ? QG[qe(i, m)] : 0.) - (m >= js + 1 ? QG[qe(i, m - 1)] : 0.);
          s = yf * dg + 0.5 * mm * fyad;
          if (own) {          // yfx and my may be one array (the mass transport): two accumulates in program order
            const double f2v = F2[e];
            a.yfx.p[at(i, m)] += f2v * dg;
            if (in) a.my.p[at(i, m)] += 0.5 * f2v * fyad;
          }
        }
        F2[e] = s;
      } }
    { constexpr int n = TPD_QW * TPD_AH;            // CT <- cry (the outer sweep's crx was last read before the barrier above)
      TPD_LOOPU(e, u, n, E_QA) CT[e] = TPD_PF(cyt, u, load_cy(e)); }
    TPF_SYNC();
    auto FY2 = [&](int i, int m) { return (m >= J0 - 2 && m <= J1 + 3) ? F2[f2e(i, m)] : 0.; };
    if (iord == 2 && !(FV3LM_TPD_EXP && (a.exp & 4))) {                                // edge-value adjoints of the inner sweep, rows J0-4 .. J1+5 of the owned columns
      constexpr int n = TPD_QW * TPD_AH;
      TPD_LOOP(e, n) {
        const int i = I0 - 3 + e % TPD_QW, x = J0 - 5 + e / TPD_QW;
        double s = 0.;
        if (i >= GI0 && i <= GI1 && x >= J0 - 4 && x <= J1 + 5) s = tps_al(FY2(i, x - 1), FY2(i, x), FY2(i, x + 1), CT[e], CT[e + TPD_QW], CT[e + 2 * TPD_QW]);
        AL[e] = s;
      } }
    TPF_SYNC();
    { constexpr int n = TPD_NQ;                     // transposed inner sweep: q_ad of the owned cells
      TPD_LOOPU(e, u, n, EQ) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
        const double acc = TPD_QACC_DIRECT ? 0. : qacc[u];
        if (i < GI0 || i > GI1 || j < GJ0 || j > GJ1 || corner_cell(i, j)) continue;
        const double qold = TPD_PF(qo7, u, a.q.p[at(i, j)]);
        const MetY da{c.m.dya, c, tile, i};
        const int cb = cye(i, j);
        double s;
        if (iord == 333) {
          const double f[4] = {FY2(i, j - 1), FY2(i, j), FY2(i, j + 1), FY2(i, j + 2)};
          const double cc[4] = {CT[cb - TPD_QW], CT[cb], CT[cb + TPD_QW], CT[cb + 2 * TPD_QW]};
          s = tps_q333(face, j, ny + 1, f, cc, da);
        } else {
          const int b0 = ale(i, j);
          const double al[4] = {AL[b0 - TPD_QW], AL[b0], AL[b0 + TPD_QW], AL[b0 + 2 * TPD_QW]};
          if (!(FV3LM_TPD_EXP && (a.exp & 8))) s = tps_q(iord, face, j, ny + 1, FY2(i, j), FY2(i, j + 1), CT[cb], CT[cb + TPD_QW], al, da);
        }
        a.q.p[at(i, j)] = qold + (acc + s);
      } }
    { constexpr int n = TPD_QW * (TPD_H + 1);       // own y-faces: cry_ad of the inner sweep
      TPD_LOOP(e, n) {
        const int i = I0 - 3 + e % TPD_QW, m = J0 + e / TPD_QW;
        if (i < GI0 || i > GI1 || !(m <= J1 || (lasty && m == J1 + 1))) continue;
        const double f = F2[f2e(i, m)];
        if (f == 0.) continue;
        if (FV3LM_TPD_EXP && (a.exp & 2)) continue;
        auto line = [&](int jj) -> Dual { int ii = i, j2 = jj; if (face) corner_map(g, 2, ii, j2); return Dual(Q[qe(ii, j2)], 0.); };
        const MetY da{c.m.dya, c, tile, i};
        double v_, d_;
        if (iord == 2 && !(face && (i < 1 || i > nx) && (m <= 3 || m >= ny - 1))) tpd_flux2<TPD_QW>(face, m, ny + 1, Q + qe(i, m), CT[cye(i, m)], da, v_, d_);
        else d_ = ppm_flux<Dual>(iord, face, m, ny + 1, line, da, Dual(CT[cye(i, m)], 1.)).d;
        a.cry.p[at(i, m)] += d_ * f;
      } }
    TPF_SYNC();
  }

  // ============================== branch B: outer sweep in y, inner sweep in x ==============================
  // (CT holds cry from branch A's inner sweep: the Courant numbers of this branch's outer sweep)
  {
    auto f2e = [&](int i, int j) { return (j - (J0 - 3)) * (TPD_QW + 1) + (i - (I0 - 2)); };    // columns I0-2 .. I1+3 (pitch TPD_QW+1), rows J0-3 .. J1+3
    auto fae = [&](int i, int j) { return (j - (J0 - 6)) * TPD_QW + (i - (I0 - 3)); };          // rows J0-6 .. J1+6, columns I0-3 .. I1+3
    auto ale = [&](int i, int j) { return (j - (J0 - 3)) * TPD_AW + (i - (I0 - 5)); };          // inner sweep: columns I0-5 .. I1+6 (pitch TPD_AW)
    { constexpr int n = TPD_QW * TPD_AH;
      TPD_LOOP(e, n) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 - 6 + e / TPD_QW;
        FA[e] = (j >= js && j <= je + 1 && i >= is && i <= ie && j <= J1 + 6 && i <= I1 + 3) ? 0.5 * a.my.t[at(i, j)] * a.fy.p[at(i, j)] : 0.;
      } }
    double ar2[TPD_SLOTS(E_J)], x02[TPD_SLOTS(E_J)], x12[TPD_SLOTS(E_J)], rx2[TPD_SLOTS(E_J)];  // prefetch: q_j values
    if constexpr (PRE) { constexpr int n = TPD_W * TPD_QH;
      TPD_LOOPU(e, u, n, E_J) { const int i = I0 + e % TPD_W, j = J0 - 3 + e / TPD_W; const bool in = i <= I1 && j <= J1 + 3;
        ar2[u] = in ? MET(area, i, j) : 0.; x02[u] = in ? a.xfx.t[at(i, j)] : 0.; x12[u] = in ? a.xfx.t[at(i + 1, j)] : 0.; rx2[u] = in ? a.rax.t[at(i, j)] : 1.; } }
    { constexpr int w = TPD_W + 1, n = w * TPD_QH;  // inner flux values fx2 on columns I0 .. I1+1
      TPD_LOOP(e, n) {
        const int i = I0 + e % w, j = J0 - 3 + e / w;
        if (i > I1 + 1 || j > J1 + 3) continue;
        auto line = [&](int ii) -> double { int i2 = ii, jj = j; if (face) corner_map(g, 1, i2, jj); return Q[qe(i2, jj)]; };
        const MetX da{c.m.dxa, c, tile, j};
        if (FV3LM_TPD_EXP && (a.exp & 1)) { F2[f2e(i, j)] = line(i) * a.crx.t[at(i, j)]; continue; }
        const double cc = a.crx.t[at(i, j)];
        if (iord == 2 && !(face && (j < 1 || j > ny) && (i <= 3 || i >= nx - 1))) { double v_, d_; tpd_flux2<1>(face, i, nx + 1, Q + qe(i, j), cc, da, v_, d_); F2[f2e(i, j)] = v_; }
        else F2[f2e(i, j)] = ppm_flux<double>(iord, face, i, nx + 1, line, da, cc);
      } }
    TPF_SYNC();
    double fy3[TPD_SLOTS(E_FB)];                               // prefetch: own y-faces
    if constexpr (PRE) { constexpr int n = TPD_W * (TPD_H + 1);
      TPD_LOOPU(e, u, n, E_FB) { const int i = I0 + e % TPD_W, j = J0 + e / TPD_W; fy3[u] = (i <= I1 && (j <= J1 || (lasty && j == J1 + 1))) ? a.fy.p[at(i, j)] : 0.; } }
    { constexpr int n = TPD_W * TPD_QH;             // q_j values
      TPD_LOOPU(e, u, n, E_J) {
        const int i = I0 + e % TPD_W, j = J0 - 3 + e / TPD_W;
        if (i > I1 || j > J1 + 3) continue;
        QG[qe(i, j)] = (Q[qe(i, j)] * TPD_PF(ar2, u, MET(area, i, j)) + TPD_PF(x02, u, a.xfx.t[at(i, j)]) * F2[f2e(i, j)] - TPD_PF(x12, u, a.xfx.t[at(i + 1, j)]) * F2[f2e(i + 1, j)]) / TPD_PF(rx2, u, a.rax.t[at(i, j)]);
      } }
    TPF_SYNC();
    double rx4[TPD_SLOTS(E_Q)], ar4[TPD_SLOTS(E_Q)], ro4[TPD_SLOTS(E_Q)];            // prefetch: gj
    if constexpr (PRE) { constexpr int n = TPD_NQ;
      TPD_LOOPU(e, u, n, E_Q) { const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
        const bool in = j >= GJ0 && j <= GJ1 && i >= is && i <= ie && i <= I1 + 3, own = in && i >= I0 && i <= I1;
        rx4[u] = in ? a.rax.t[at(i, j)] : 1.; ar4[u] = in ? MET(area, i, j) : 0.; ro4[u] = own ? a.rax.p[at(i, j)] : 0.; } }
    auto FY = [&](int i, int m) { return FA[fae(i, m)]; };
    { constexpr int n = TPD_W * (TPD_H + 1);        // own y-faces: Courant-number and mass-flux adjoints of the outer sweep
      TPD_LOOPU(e, u, n, E_FB) {
        const int i = I0 + e % TPD_W, j = J0 + e / TPD_W;
        if (i > I1 || !(j <= J1 || (lasty && j == J1 + 1))) continue;
        const double fo = FY(i, j), fyad = TPD_PF(fy3, u, a.fy.p[at(i, j)]);
        if (fo == 0. && fyad == 0.) continue;
        if (FV3LM_TPD_EXP && (a.exp & 2)) continue;
        auto line = [&](int jj) -> Dual { return Dual(QG[qe(i, jj)], 0.); };
        const MetY da{c.m.dya, c, tile, i};
        Dual fl;
        if (iord == 2) tpd_flux2<TPD_QW>(face, j, ny + 1, QG + qe(i, j), CT[cye(i, j)], da, fl.v, fl.d);
        else fl = ppm_flux<Dual>(iord, face, j, ny + 1, line, da, Dual(CT[cye(i, j)], 1.));
        a.cry.p[at(i, j)] += fl.d * fo;
        a.my.p[at(i, j)] += 0.5 * fl.v * fyad;
      } }
    if (iord == 2 && !(FV3LM_TPD_EXP && (a.exp & 4))) {                                // edge-value adjoints of the outer sweep, rows J0-4 .. J1+5
      constexpr int n = TPD_QW * TPD_AH;
      TPD_LOOP(e, n) {
        const int i = I0 - 3 + e % TPD_QW, x = J0 - 6 + e / TPD_QW;
        double s = 0.;
        if (x >= J0 - 4 && x <= J1 + 5 && i >= is && i <= ie && i <= I1 + 3) s = tps_al(FA[e - TPD_QW], FA[e], FA[e + TPD_QW], CT[e - TPD_QW], CT[e], CT[e + TPD_QW]);
        AL[e] = s;
      } }
    TPF_SYNC();
    double xf5[TPD_SLOTS(E_GB)], fx5[TPD_SLOTS(E_GB)], mm5[TPD_SLOTS(E_GB)];         // prefetch: fx2_ad
    if constexpr (PRE) { constexpr int w = TPD_QW - 1, n = w * TPD_QH;
      TPD_LOOPU(e, u, n, E_GB) { const int m = I0 - 2 + e % w, j = J0 - 3 + e / w;
        const bool on = j >= GJ0 && j <= GJ1 && m >= is && m <= ie + 1 && m <= I1 + 3, in = on && j >= js && j <= je;
        xf5[u] = on ? a.xfx.t[at(m, j)] : 0.; fx5[u] = in ? a.fx.p[at(m, j)] : 0.; mm5[u] = in ? a.mx.t[at(m, j)] : 0.; } }
    { constexpr int n = TPD_NQ;                     // gj = q_j_ad / ra_x on the owned rows; own cells: ra_x_ad, the area term of q_ad
      TPD_LOOPU(e, u, n, EQ) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
        double s = 0., acc = 0.;
        if (j >= GJ0 && j <= GJ1 && i >= is && i <= ie && i <= I1 + 3) {
          const bool own = i >= I0 && i <= I1;
          const double rx = TPD_PF(rx4, u, a.rax.t[at(i, j)]), ar = TPD_PF(ar4, u, MET(area, i, j)), rold = TPD_PF(ro4, u, own ? a.rax.p[at(i, j)] : 0.);
          const MetY da{c.m.dya, c, tile, i};
          const int b0 = fae(i, j);
          if (iord == 333) {
            const double f[4] = {FA[b0 - TPD_QW], FA[b0], FA[b0 + TPD_QW], FA[b0 + 2 * TPD_QW]};
            const double cc[4] = {CT[b0 - TPD_QW], CT[b0], CT[b0 + TPD_QW], CT[b0 + 2 * TPD_QW]};
            s = tps_q333(face, j, ny + 1, f, cc, da);
          } else {
            const double al[4] = {AL[b0 - TPD_QW], AL[b0], AL[b0 + TPD_QW], AL[b0 + 2 * TPD_QW]};
            if (!(FV3LM_TPD_EXP && (a.exp & 8))) s = tps_q(iord, face, j, ny + 1, FA[b0], FA[b0 + TPD_QW], CT[b0], CT[b0 + TPD_QW], al, da);
          }
          s = s / rx;
          if (own) { a.rax.p[at(i, j)] = rold - QG[e] * s; acc = ar * s; }
        }
        if (TPD_QACC_DIRECT) { if (acc != 0.) a.q.p[at(i, j)] += acc; } else qacc[u] = acc;
        QG[e] = s;
      } }
    TPF_SYNC();
    double cxt[TPD_SLOTS(E_AQ)], qo7[TPD_SLOTS(E_Q)];                     // the Courant numbers of the inner sweep (crx) on their way into CT; prefetch: q_ad
    if constexpr (PRE) {
      { constexpr int n = TPD_AW * TPD_QH; TPD_LOOPU(e, u, n, E_AQ) cxt[u] = load_cx(e); }
      { constexpr int n = TPD_NQ; TPD_LOOPU(e, u, n, E_Q) { const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
          qo7[u] = (i < GI0 || i > GI1 || j < GJ0 || j > GJ1 || corner_cell(i, j) || i > I1 + 3 || j > J1 + 3) ? 0. : a.q.p[at(i, j)]; } }
    }
    { constexpr int w = TPD_QW - 1, n = w * TPD_QH; // fx2_ad on columns I0-2 .. I1+3 of the owned rows, in place of the values; own x-faces: xfx_ad, mx_ad (inner part)
      TPD_LOOPU(e, u, n, E_GB) {
        const int m = I0 - 2 + e % w, j = J0 - 3 + e / w;
        double s = 0.;
        if (j >= GJ0 && j <= GJ1 && m >= is && m <= ie + 1 && m <= I1 + 3) {
          const bool in = j >= js && j <= je, own = m >= I0 && (m <= I1 || (lastx && m == I1 + 1));
          const double xf = TPD_PF(xf5, u, a.xfx.t[at(m, j)]), fxad = TPD_PF(fx5, u, in ? a.fx.p[at(m, j)] : 0.), mm = TPD_PF(mm5, u, in ? a.mx.t[at(m, j)] : 0.);
          const double dg = (m <= ie ? QG[qe(m, j)] : 0.) - (m >= is + 1 ? QG[qe(m - 1, j)] : 0.);
          s = xf * dg + 0.5 * mm * fxad;
          if (own) {          // xfx and mx may be one array (the mass transport): two accumulates in program order
            const double f2v = F2[f2e(m, j)];
            a.xfx.p[at(m, j)] += f2v * dg;
            if (in) a.mx.p[at(m, j)] += 0.5 * f2v * fxad;
          }
        }
        F2[f2e(m, j)] = s;
      } }
    { constexpr int n = TPD_AW * TPD_QH;            // CT <- crx (the outer sweep's cry was last read before the barrier above)
      TPD_LOOPU(e, u, n, E_AQ) CT[e] = TPD_PF(cxt, u, load_cx(e)); }
    TPF_SYNC();
    auto FX2 = [&](int m, int j) { return (m >= I0 - 2 && m <= I1 + 3) ? F2[f2e(m, j)] : 0.; };
    if (iord == 2 && !(FV3LM_TPD_EXP && (a.exp & 4))) {                                // edge-value adjoints of the inner sweep, columns I0-4 .. I1+5 of the owned rows
      constexpr int n = TPD_AW * TPD_QH;
      TPD_LOOP(e, n) {
        const int x = I0 - 5 + e % TPD_AW, j = J0 - 3 + e / TPD_AW;
        double s = 0.;
        if (j >= GJ0 && j <= GJ1 && x >= I0 - 4 && x <= I1 + 5) s = tps_al(FX2(x - 1, j), FX2(x, j), FX2(x + 1, j), CT[e], CT[e + 1], CT[e + 2]);
        AL[e] = s;
      } }
    TPF_SYNC();
    { constexpr int n = TPD_NQ;                     // transposed inner sweep: q_ad of the owned cells
      TPD_LOOPU(e, u, n, EQ) {
        const int i = I0 - 3 + e % TPD_QW, j = J0 - 3 + e / TPD_QW;
        const double acc = TPD_QACC_DIRECT ? 0. : qacc[u];
        if (i < GI0 || i > GI1 || j < GJ0 || j > GJ1 || corner_cell(i, j)) continue;
        const double qold = TPD_PF(qo7, u, a.q.p[at(i, j)]);
        const MetX da{c.m.dxa, c, tile, j};
        const int cb = cxe(i, j);
        double s;
        if (iord == 333) {
          const double f[4] = {FX2(i - 1, j), FX2(i, j), FX2(i + 1, j), FX2(i + 2, j)};
          const double cc[4] = {CT[cb - 1], CT[cb], CT[cb + 1], CT[cb + 2]};
          s = tps_q333(face, i, nx + 1, f, cc, da);
        } else {
          const int b0 = ale(i, j);
          const double al[4] = {AL[b0 - 1], AL[b0], AL[b0 + 1], AL[b0 + 2]};
          if (!(FV3LM_TPD_EXP && (a.exp & 8))) s = tps_q(iord, face, i, nx + 1, FX2(i, j), FX2(i + 1, j), CT[cb], CT[cb + 1], al, da);
        }
        a.q.p[at(i, j)] = qold + (acc + s);
      } }
    { constexpr int w = TPD_W + 1, n = w * TPD_QH;  // own x-faces: crx_ad of the inner sweep
      TPD_LOOP(e, n) {
        const int m = I0 + e % w, j = J0 - 3 + e / w;
        if (j < GJ0 || j > GJ1 || j > J1 + 3 || !(m <= I1 || (lastx && m == I1 + 1))) continue;
        const double f = F2[f2e(m, j)];
        if (f == 0.) continue;
        if (FV3LM_TPD_EXP && (a.exp & 2)) continue;
        auto line = [&](int ii) -> Dual { int i2 = ii, jj = j; if (face) corner_map(g, 1, i2, jj); return Dual(Q[qe(i2, jj)], 0.); };
        const MetX da{c.m.dxa, c, tile, j};
        double v_, d_;
        if (iord == 2 && !(face && (j < 1 || j > ny) && (m <= 3 || m >= nx - 1))) tpd_flux2<1>(face, m, nx + 1, Q + qe(m, j), CT[cxe(m, j)], da, v_, d_);
        else d_ = ppm_flux<Dual>(iord, face, m, nx + 1, line, da, Dual(CT[cxe(m, j)], 1.)).d;
        a.crx.p[at(m, j)] += d_ * f;
      } }
  }
#undef TPD_PF
#undef TPD_SLOTS
}

// The corner-alias contributions: source cell s next to a face corner is also read, by the inner sweep of direction dir, as the
// virtual corner cell v = corner_alias(dir, s).  q_ad(s) += sum over the inner fluxes that read v of dflux/dq(v) * flux_ad, with the
// flux adjoints of the virtual line re-derived from the global inputs (no flux of a virtual line takes part in the flux assembly, so
// its adjoint is the q_i / q_j part alone).  n = 0 .. 8: the 3 x 3 sources of one corner; dir 2: inner y sweep (branch A), 1: inner x.
DEV void tp_ad_corner_point(const TpFusedArgs& a, const Ctx& c, int tile, int k, int cn, int n, int dir) {
  const Geom& g = c.g;
  if (!g.face) return;
  const int nx = g.nx, ny = g.ny, ng = g.ng;
  const size_t base = (size_t)(tile * a.nk + k - 1) * g.plane;
  auto at = [&](int i, int j) -> size_t { return base + g.idx(i, j); };
  const int iord = hord_of(c.lev[k - 1], a.hsel);
  // the 3 x 3 sources of corner cn for this direction: dir 2 reads them from the south / north halo rows next to the corner columns,
  // dir 1 from the west / east halo columns next to the corner rows (edges.h corner_alias)
  const bool west = (cn == 0 || cn == 3), south = (cn < 2);
  if ((west ? g.is() != 1 : g.ie() != nx) || (south ? g.js() != 1 : g.je() != ny)) return;      // a sub-face tile holds at most one face corner
  int si, sj;
  if (dir == 2) { si = (west ? 1 : nx - ng + 1) + n % ng; sj = (south ? 1 - ng : ny + 1) + n / ng; }
  else { si = (west ? 1 - ng : nx + 1) + n % ng; sj = (south ? 1 : ny - ng + 1) + n / ng; }
  int vi, vj;
  if (!corner_alias(g, dir, si, sj, vi, vj)) return;
  double s = 0.;
  if (dir == 2) {      // virtual cell (vi, vj) in a halo column vi: fluxes fy2(vi, m), m = vj-2 .. vj+3 within 1 .. ny+1
    const MetY da{c.m.dya, c, tile, vi};
    auto gi = [&](int r) -> double {       // q_i_ad(vi, r) / ra_y(vi, r)
      if (r < 1 || r > ny) return 0.;
      const MetX dx_{c.m.dxa, c, tile, r};
      double t = 0.;
      for (int mm = vi - 2; mm <= vi + 3; ++mm) {
        if (mm < 1 || mm > nx + 1) continue;
        const double f = 0.5 * a.mx.t[at(mm, r)] * a.fx.p[at(mm, r)];
        if (f != 0.) t += ppm_dq(iord, true, mm, nx + 1, vi, dx_, a.crx.t[at(mm, r)]) * f;
      }
      return t / a.ray.t[at(vi, r)];
    };
    for (int m = vj - 2; m <= vj + 3; ++m) {
      if (m < 1 || m > ny + 1) continue;
      const double f = a.yfx.t[at(vi, m)] * (gi(m) - gi(m - 1));
      if (f != 0.) s += ppm_dq(iord, true, m, ny + 1, vj, da, a.cry.t[at(vi, m)]) * f;
    }
  } else {
    const MetX da{c.m.dxa, c, tile, vj};
    auto gj = [&](int r) -> double {       // q_j_ad(r, vj) / ra_x(r, vj)
      if (r < 1 || r > nx) return 0.;
      const MetY dy_{c.m.dya, c, tile, r};
      double t = 0.;
      for (int mm = vj - 2; mm <= vj + 3; ++mm) {
        if (mm < 1 || mm > ny + 1) continue;
        const double f = 0.5 * a.my.t[at(r, mm)] * a.fy.p[at(r, mm)];
        if (f != 0.) t += ppm_dq(iord, true, mm, ny + 1, vj, dy_, a.cry.t[at(r, mm)]) * f;
      }
      return t / a.rax.t[at(r, vj)];
    };
    for (int m = vi - 2; m <= vi + 3; ++m) {
      if (m < 1 || m > nx + 1) continue;
      const double f = a.xfx.t[at(m, vj)] * (gj(m) - gj(m - 1));
      if (f != 0.) s += ppm_dq(iord, true, m, nx + 1, vi, da, a.crx.t[at(m, vj)]) * f;
    }
  }
  if (s != 0.) a.q.p[at(si, sj)] += s;
}

#ifndef FV3LM_HOST_EMUL
__global__ void __launch_bounds__(TPD_THREADS) k_tp_ad(TpFusedArgs a, Ctx c) {
  extern __shared__ double tpd_lds[];
  int bx = blockIdx.x, by = blockIdx.y; xcd_block(gridDim.x, gridDim.y, bx, by);
  tp_ad_block<TPD_THREADS>(a, c, blockIdx.z / a.nk, 1 + blockIdx.z % a.nk, bx, by, tpd_lds, threadIdx.x, TPD_THREADS);
}
__global__ void __launch_bounds__(128) k_tp_ad_corner(TpFusedArgs a, Ctx c) {
  // 4 corners x 9 sources per direction: the first wave takes direction 2, the second direction 1 (no source belongs to both: the
  // direction-2 sources lie in the south / north halo rows, the direction-1 sources in the west / east halo columns)
  const int t = threadIdx.x & 63, dir = threadIdx.x < 64 ? 2 : 1;
  if (t >= 36) return;
  const int tile = blockIdx.x / a.nk, k = 1 + blockIdx.x % a.nk;
  tp_ad_corner_point(a, c, tile, k, t / 9, t % 9, dir);
}
#endif

FV3LM_LINK void run_tp_ad(Exec& ex, const TpFusedArgs& a0, const Ctx& c) {
  TpFusedArgs a = a0;
  for (Fld* f : {&a.q, &a.crx, &a.cry, &a.xfx, &a.yfx, &a.rax, &a.ray, &a.mx, &a.my, &a.fx, &a.fy}) *f = ex.sh(*f);
  const int nbx = (c.g.tx + TPD_W - 1) / TPD_W, nby = (c.g.ty + TPD_H - 1) / TPD_H;
#if FV3LM_TPD_EXP
  if (const char* e = std::getenv("FV3LM_TP2_EXPERIMENT")) a.exp = std::atoi(e);
#endif
  // algorithmic bytes: trajectory q, crx, cry, xfx, yfx, ra_x, ra_y, mx, my and the adjoints fx, fy read; the nine input adjoints read-modify-written
  const double cells = double(c.g.tx) * c.g.ty * c.g.ntile * a.nk;
  ex.mark_begin("TpAd", ".ad", 8. * cells * (9. + 2. + 18.));
#ifdef FV3LM_HOST_EMUL
  std::vector<double> lds((size_t)TPD_NT);
  for (int z = 0; z < c.g.ntile * a.nk; ++z)
    for (int by = 0; by < nby; ++by)
      for (int bx = 0; bx < nbx; ++bx) tp_ad_block<1>(a, c, z / a.nk, 1 + z % a.nk, bx, by, lds.data(), 0, 1);
  if (c.g.face)
    for (int z = 0; z < c.g.ntile * a.nk; ++z)
      for (int t = 0; t < 36; ++t) { tp_ad_corner_point(a, c, z / a.nk, 1 + z % a.nk, t / 9, t % 9, 2); tp_ad_corner_point(a, c, z / a.nk, 1 + z % a.nk, t / 9, t % 9, 1); }
#else
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)k_tp_ad, hipFuncAttributeMaxDynamicSharedMemorySize, TPD_NT * 8) != hipSuccess) set_sticky("hipFuncSetAttribute(k_tp_ad) failed"); attr = true; }
  hipLaunchKernelGGL(k_tp_ad, dim3(nbx, nby, c.g.ntile * a.nk), dim3(TPD_THREADS), TPD_NT * 8, ex.stream, a, c);
#endif
  ex.mark_end();
  ex.launches++;
#ifndef FV3LM_HOST_EMUL
  if (c.g.face) {
    ex.mark_begin("TpAd", ".ad_corner", 0.);
    hipLaunchKernelGGL(k_tp_ad_corner, dim3(c.g.ntile * a.nk), dim3(128), 0, ex.stream, a, c);
    ex.mark_end();
    ex.launches++;
  }
#endif
}

}  // namespace fv3
#endif
