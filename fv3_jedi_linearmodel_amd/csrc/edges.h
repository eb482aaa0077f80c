// fv3lm-hip: cube-face edge and corner helpers used by the stages when every tile is a whole face
// (is = 1, ie = npx-1): corner-halo index maps of copy_corners (tp_core_tlm.F90:2046-2118) and
// fill2_4corners (sw_core_tlm.F90:7061-7134), their inverses (needed by the gather-form adjoint: a
// point that is the source of a corner copy is also "read" at the corner location), one-sided PPM
// edge values (tp_core_tlm.F90:2402-2429) and the xtp_u/ytp_v edge slopes (sw_core_tlm.F90:7381-7460).
//
// The reference fills the corner halo of a field in place, twice per routine with different
// rotations (x-sweep / y-sweep).  Here nothing is mutated: a stage that sweeps in x reads corner-halo
// points through corner_map(dir=1), one that sweeps in y through dir=2.
#pragma once
#include "core.h"

namespace fv3 {

// copy_corners: where does the value at corner-halo point (i,j) come from.  No-op elsewhere.
HD void corner_map(const Geom& g, int dir, int& i, int& j) {
  if (!g.face) return;
  const int n = g.nx, npx = n + 1, npy = g.ny + 1;
  const bool w = i < 1, e = i > n, s = j < 1, nn = j > g.ny;
  if (!((w || e) && (s || nn))) return;
  const int i0 = i, j0 = j;
  if (dir == 1) {
    if (w && s) { i = j0; j = 1 - i0; }
    else if (e && s) { i = npy - j0; j = i0 - npx + 1; }
    else if (e && nn) { i = j0; j = 2 * npx - 1 - i0; }
    else { i = npy - j0; j = i0 - 1 + npx; }
  } else {
    if (w && s) { i = 1 - j0; j = i0; }
    else if (e && s) { i = npy + j0 - 1; j = npx - i0; }
    else if (e && nn) { i = 2 * npy - 1 - j0; j = i0; }
    else { i = j0 + 1 - npx; j = npy - i0; }
  }
}
// inverse: the corner-halo point (ai,aj) whose dir-view reads (i,j); false if none
HD bool corner_alias(const Geom& g, int dir, int i, int j, int& ai, int& aj) {
  if (!g.face) return false;
  const int n = g.nx, npx = n + 1, npy = g.ny + 1, ng = g.ng;
  const bool ilo = i >= 1 - ng && i <= 0, ihi = i >= npx && i <= npx + ng - 1;
  const bool jlo = j >= 1 - ng && j <= 0, jhi = j >= npy && j <= npy + ng - 1;
  const bool i13 = i >= 1 && i <= ng, in3 = i >= npx - ng && i <= npx - 1;
  const bool j13 = j >= 1 && j <= ng, jn3 = j >= npy - ng && j <= npy - 1;
  if (dir == 1) {
    if (ilo && j13) { ai = 1 - j; aj = i; return true; }                    // sw
    if (ihi && j13) { ai = j + npx - 1; aj = npy - i; return true; }        // se
    if (ihi && jn3) { ai = 2 * npx - 1 - j; aj = i; return true; }          // ne
    if (ilo && jn3) { ai = j + 1 - npx; aj = npy - i; return true; }        // nw
  } else {
    if (i13 && jlo) { ai = j; aj = 1 - i; return true; }                    // sw
    if (in3 && jlo) { ai = npx - j; aj = i + 1 - npy; return true; }        // se
    if (in3 && jhi) { ai = j; aj = 2 * npy - 1 - i; return true; }          // ne
    if (i13 && jhi) { ai = npy - j; aj = i - 1 + npx; return true; }        // nw
  }
  return false;
}

// fill2_4corners (c_sw): only two points per corner and direction
HD void fill2_map(const Geom& g, int dir, int& i, int& j) {
  if (!g.face) return;
  const int npx = g.nx + 1, npy = g.ny + 1;
  if (dir == 1) {
    if (j != 0 && j != npy) return;
    const int near = (j == 0) ? 1 : npy - 1, far = (j == 0) ? 2 : npy - 2;
    if (i == 0) { j = near; }
    else if (i == -1) { i = 0; j = far; }
    else if (i == npx) { j = near; }
    else if (i == npx + 1) { i = npx; j = far; }
  } else {
    if (i != 0 && i != npx) return;
    const int near = (i == 0) ? 1 : npx - 1, far = (i == 0) ? 2 : npx - 2;
    if (j == 0) { i = near; }
    else if (j == -1) { i = far; j = 0; }
    else if (j == npy) { i = near; }
    else if (j == npy + 1) { i = far; j = npy; }
  }
}
HD bool fill2_alias(const Geom& g, int dir, int i, int j, int& ai, int& aj) {
  if (!g.face) return false;
  const int npx = g.nx + 1, npy = g.ny + 1;
  if (dir == 1) {
    if (i != 0 && i != npx) return false;
    const int off = (i == 0) ? -1 : 1;       // far point sits one further out
    if (j == 1) { ai = i; aj = 0; return true; }
    if (j == 2) { ai = i + off; aj = 0; return true; }
    if (j == npy - 1) { ai = i; aj = npy; return true; }
    if (j == npy - 2) { ai = i + off; aj = npy; return true; }
  } else {
    if (j != 0 && j != npy) return false;
    const int off = (j == 0) ? -1 : 1;
    if (i == 1) { ai = 0; aj = j; return true; }
    if (i == 2) { ai = 0; aj = j + off; return true; }
    if (i == npx - 1) { ai = npx; aj = j; return true; }
    if (i == npx - 2) { ai = npx; aj = j + off; return true; }
  }
  return false;
}

// d2a2c_vect corner fixes (sw_core_tlm.F90:6617-6640, :6662-6677, :6726-6760): in the x-direction pass
// the A-grid winds utmp/ua on the rows j = 0, npy beyond the face corner are the rotated vtmp/va of the
// adjacent edge halo, in the y-direction pass vtmp/va on the columns i = 0, npx are the rotated utmp/ua.
// xview: (i,j) of utmp/ua -> location in vtmp/va and sign; false if (i,j) is an ordinary point.
HD bool d2a2c_xview(const Geom& g, int i, int j, int& oi, int& oj, double& sg) {
  const int npx = g.nx + 1, npy = g.ny + 1;
  if (j == 0) {
    if (i <= 0) { oi = 0; oj = 1 - i; sg = -1.; return true; }
    if (i >= npx) { oi = npx; oj = i - npx + 1; sg = 1.; return true; }
  } else if (j == npy) {
    if (i <= 0) { oi = 0; oj = npy - 1 + i; sg = 1.; return true; }
    if (i >= npx) { oi = npx; oj = npy - 1 - (i - npx); sg = -1.; return true; }
  }
  return false;
}
HD bool d2a2c_yview(const Geom& g, int i, int j, int& oi, int& oj, double& sg) {
  const int npx = g.nx + 1, npy = g.ny + 1;
  if (i == 0) {
    if (j <= 0) { oi = 1 - j; oj = 0; sg = -1.; return true; }
    if (j >= npy) { oi = j - npy + 1; oj = npy; sg = 1.; return true; }
  } else if (i == npx) {
    if (j <= 0) { oi = npx - 1 + j; oj = 0; sg = 1.; return true; }
    if (j >= npy) { oi = npx - 1 - (j - npy); oj = npy; sg = -1.; return true; }
  }
  return false;
}
// inverses: the x-view location that reads vtmp/va(i,j); the y-view location that reads utmp/ua(i,j)
HD bool d2a2c_xalias(const Geom& g, int i, int j, int& ai, int& aj) {
  if (!g.face) return false;
  const int npx = g.nx + 1, npy = g.ny + 1, ng = g.ng;
  if (i != 0 && i != npx) return false;
  if (j >= 1 && j <= ng) { ai = (i == 0) ? 1 - j : j + npx - 1; aj = 0; return true; }
  if (j >= npy - ng && j <= npy - 1) { ai = (i == 0) ? j - npy + 1 : npx + npy - 1 - j; aj = npy; return true; }
  return false;
}
HD bool d2a2c_yalias(const Geom& g, int i, int j, int& ai, int& aj) {
  if (!g.face) return false;
  const int npx = g.nx + 1, npy = g.ny + 1, ng = g.ng;
  if (j != 0 && j != npy) return false;
  if (i >= 1 && i <= ng) { ai = 0; aj = (j == 0) ? 1 - i : i + npy - 1; return true; }
  if (i >= npx - ng && i <= npx - 1) { ai = npx; aj = (j == 0) ? i - npx + 1 : npy + npx - 1 - i; return true; }
  return false;
}

constexpr double EC1 = -2. / 14., EC2 = 11. / 14., EC3 = 5. / 14.;   // c1,c2,c3 tp_core_tlm.F90:62-64

// PPM edge value al(m) = sum_d w[d] * q(m-2+d), d = 0..3, on a line with absolute cell index m; metric dxa/dya
// (or dx/dy for xtp_u/ytp_v) by absolute index, n1 = npx (or npy) of the face.  Away from a face edge the weights
// are (p2, p1, p1, p2); within one cell of it they are the one-sided / two-sided extrapolations of
// tp_core_tlm.F90:2402-2429 (sw_core_tlm.F90:7381-7460 for the winds) — all of which happen to use the same four
// cells, so every lane of a wave runs the same loads and multiply-adds and only the weights differ.
template <class D>
HD void ppm_w(bool face, int m, int n1, const D& da, double* w) {
  w[0] = w[3] = -1. / 12.; w[1] = w[2] = 7. / 12.;
  if (face) {
    if (m == 0 || m == n1 - 1) { w[0] = EC1; w[1] = EC2; w[2] = EC3; w[3] = 0.; }
    else if (m == 2 || m == n1 + 1) { w[0] = 0.; w[1] = EC3; w[2] = EC2; w[3] = EC1; }
    else if (m == 1 || m == n1) {
      const double a = da(m - 2), b = da(m - 1), c = da(m), d = da(m + 1);
      w[0] = -0.5 * b / (a + b); w[1] = 0.5 * (2. * b + a) / (a + b); w[2] = 0.5 * (2. * c + d) / (c + d); w[3] = -0.5 * c / (c + d);
    }
  }
}
template <class T, class Q, class D>
HD T ppm_al(bool face, int m, int n1, const Q& q, const D& da) {
  double w[4];
  ppm_w(face, m, n1, da, w);
  return w[0] * q(m - 2) + w[1] * q(m - 1) + w[2] * q(m) + w[3] * q(m + 1);
}

// xtp_u / ytp_v slopes bl(m), br(m) of cell m on a line (sw_core_tlm.F90:7363-7460, :7585-7690): the same edge
// values with the D-grid metric.  row_edge: the line itself lies on a face edge of the other direction
// (j == 1 or j == npy for xtp_u), where the two cells next to the corner get zero slopes.
template <class T, class Q, class D>
HD void uv_blbr(bool face, int m, int n1, bool row_edge, const Q& q, const D& dd, T& bl, T& br) {
  if (face && row_edge && (m == 0 || m == 1 || m == n1 - 1 || m == n1)) { bl = T(0.); br = T(0.); return; }
  bl = ppm_al<T>(face, m, n1, q, dd) - q(m);
  br = ppm_al<T>(face, m + 1, n1, q, dd) - q(m);
}

// edge_interpolate4, sw_core_tlm.F90:6822-6832
template <class T>
HD T edge_interp4(const T& u1, const T& u2, const T& u3, const T& u4, double d1, double d2, double d3, double d4) {
  const double t1 = d1 + d2, t2 = d3 + d4;
  return 0.5 * (((t1 + d2) * u2 - d2 * u1) / t1 + ((t2 + d3) * u3 - d3 * u4) / t2);
}

}  // namespace fv3
