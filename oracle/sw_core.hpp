// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Restatement of model_tlmadm/sw_core_tlm.F90 (per-level shallow-water stencils):
//   c_sw (:646-1038 / _TLM :87-645), d2a2c_vect (:6399-6806), divergence_corner (:3966-4082),
//   fill2_4corners (:7061-7134), d_sw (:2533-3617 / _TLM :1047-2531), xtp_u/ytp_v (_TLM :7272-7486 /
//   :7490-7759), del6_vt_flux (:3719-3801), compute_divergence_damping (:7760-8072 / _TLM :8178-8598),
// and a2b_ord4 (a2b_edge_tlm.F90:48-542).  grid_type < 3 (gnomonic cubed sphere), not nested.
// Two tile kinds: a tile that touches no cube edge (all `is .EQ. 1` / `j .EQ. npy` / corner branches
// skipped — an MPI rank in the middle of a face) and a whole face (is=1, ie=npx-1, all edges and the
// four corners).  Hydrostatic, inline_q=.false., d_con <= 1e-5.
#pragma once
#include "tp_core.hpp"

namespace orc {

static const double a2b_a1 = 0.5625, a2b_a2 = -0.0625;  // sw_core_tlm.F90:56-57 / a2b_edge_tlm.F90:35-38
static const double a2b_b1 = 7. / 12., a2b_b2 = -1. / 12.;

// a2b_ord4, a2b_edge_tlm.F90:48-542.  replace=true copies qout back into qin on is..ie+1, js..je+1.
template <class T>
void a2b_ord4(Arr2<T>& qin, Arr2<T>& qout, const Grid& g, const Bounds& bd, bool replace) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, npx = bd.npx, npy = bd.npy;
  const double c1 = 2. / 3., c2 = -(1. / 6.), r3 = 1. / 3.;
  Arr2<T> qx(bd), qy(bd), qxx(bd), qyy(bd);
  if (!bd.any_edge()) {
    for (int j = js - 2; j <= je + 2; ++j)
      for (int i = is; i <= ie + 1; ++i)
        qx(i, j) = a2b_b2 * (qin(i - 2, j) + qin(i + 1, j)) + a2b_b1 * (qin(i - 1, j) + qin(i, j));
    for (int j = js; j <= je + 1; ++j)
      for (int i = is - 2; i <= ie + 2; ++i)
        qy(i, j) = a2b_b2 * (qin(i, j - 2) + qin(i, j + 1)) + a2b_b1 * (qin(i, j - 1) + qin(i, j));
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        T xx = a2b_a2 * (qx(i, j - 2) + qx(i, j + 1)) + a2b_a1 * (qx(i, j - 1) + qx(i, j));
        T yy = a2b_a2 * (qy(i - 2, j) + qy(i + 1, j)) + a2b_a1 * (qy(i - 1, j) + qy(i, j));
        qout(i, j) = 0.5 * (xx + yy);
      }
  } else {
    const int is1 = std::max(1, is - 1), js1 = std::max(1, js - 1), is2 = std::max(2, is), js2 = std::max(2, js);
    const int ie1 = std::min(npx - 1, ie + 1), je1 = std::min(npy - 1, je + 1);
    auto ext = [&](int cn, int n, const T& q1, const T& q2) { return q1 + g.ecorner[cn][n] * (q1 - q2); };   // extrap_corner :1478
    if (bd.sw_corner) qout(1, 1) = (ext(0, 0, qin(1, 1), qin(2, 2)) + ext(0, 1, qin(0, 1), qin(-1, 2)) + ext(0, 2, qin(1, 0), qin(2, -1))) * r3;
    if (bd.se_corner) qout(npx, 1) = (ext(1, 0, qin(npx - 1, 1), qin(npx - 2, 2)) + ext(1, 1, qin(npx - 1, 0), qin(npx - 2, -1)) +
                                     ext(1, 2, qin(npx, 1), qin(npx + 1, 2))) * r3;
    if (bd.ne_corner) qout(npx, npy) = (ext(2, 0, qin(npx - 1, npy - 1), qin(npx - 2, npy - 2)) + ext(2, 1, qin(npx, npy - 1), qin(npx + 1, npy - 2)) +
                                       ext(2, 2, qin(npx - 1, npy), qin(npx - 2, npy + 1))) * r3;
    if (bd.nw_corner) qout(1, npy) = (ext(3, 0, qin(1, npy - 1), qin(2, npy - 2)) + ext(3, 1, qin(0, npy - 1), qin(-1, npy - 2)) +
                                     ext(3, 2, qin(1, npy), qin(2, npy + 1))) * r3;
    // X-interior (:163-176)
    for (int j = std::max(1, js - 2); j <= std::min(npy - 1, je + 2); ++j)
      for (int i = std::max(3, is); i <= std::min(npx - 2, ie + 1); ++i)
        qx(i, j) = a2b_b2 * (qin(i - 2, j) + qin(i + 1, j)) + a2b_b1 * (qin(i - 1, j) + qin(i, j));
    if (bd.edge_w) {   // :178-207
      std::vector<T> q2(npy + 2);
      for (int j = js1; j <= je1; ++j) q2[j] = (qin(0, j) * g.dxa(1, j) + qin(1, j) * g.dxa(0, j)) / (g.dxa(0, j) + g.dxa(1, j));
      for (int j = js2; j <= je1; ++j) qout(1, j) = g.edge_w[j] * q2[j - 1] + (1. - g.edge_w[j]) * q2[j];
      for (int j = std::max(1, js - 2); j <= std::min(npy - 1, je + 2); ++j) {
        const double g_in = g.dxa(2, j) / g.dxa(1, j), g_ou = g.dxa(-1, j) / g.dxa(0, j);
        qx(1, j) = 0.5 * (((2. + g_in) * qin(1, j) - qin(2, j)) / (1. + g_in) + ((2. + g_ou) * qin(0, j) - qin(-1, j)) / (1. + g_ou));
        qx(2, j) = (3. * (g_in * qin(1, j) + qin(2, j)) - (g_in * qx(1, j) + qx(3, j))) / (2. + 2. * g_in);
      }
    }
    if (bd.edge_e) {   // :209-240
      std::vector<T> q2(npy + 2);
      for (int j = js1; j <= je1; ++j) q2[j] = (qin(npx - 1, j) * g.dxa(npx, j) + qin(npx, j) * g.dxa(npx - 1, j)) / (g.dxa(npx - 1, j) + g.dxa(npx, j));
      for (int j = js2; j <= je1; ++j) qout(npx, j) = g.edge_e[j] * q2[j - 1] + (1. - g.edge_e[j]) * q2[j];
      for (int j = std::max(1, js - 2); j <= std::min(npy - 1, je + 2); ++j) {
        const double g_in = g.dxa(npx - 2, j) / g.dxa(npx - 1, j), g_ou = g.dxa(npx + 1, j) / g.dxa(npx, j);
        qx(npx, j) = 0.5 * (((2. + g_in) * qin(npx - 1, j) - qin(npx - 2, j)) / (1. + g_in) + ((2. + g_ou) * qin(npx, j) - qin(npx + 1, j)) / (1. + g_ou));
        qx(npx - 1, j) = (3. * (qin(npx - 2, j) + g_in * qin(npx - 1, j)) - (g_in * qx(npx, j) + qx(npx - 2, j))) / (2. + 2. * g_in);
      }
    }
    // Y-interior (:268-291)
    for (int j = std::max(3, js); j <= std::min(npy - 2, je + 1); ++j)
      for (int i = std::max(1, is - 2); i <= std::min(npx - 1, ie + 2); ++i)
        qy(i, j) = a2b_b2 * (qin(i, j - 2) + qin(i, j + 1)) + a2b_b1 * (qin(i, j - 1) + qin(i, j));
    if (bd.edge_s) {   // :293-322
      std::vector<T> q1(npx + 2);
      for (int i = is1; i <= ie1; ++i) q1[i] = (qin(i, 0) * g.dya(i, 1) + qin(i, 1) * g.dya(i, 0)) / (g.dya(i, 0) + g.dya(i, 1));
      for (int i = is2; i <= ie1; ++i) qout(i, 1) = g.edge_s[i] * q1[i - 1] + (1. - g.edge_s[i]) * q1[i];
      for (int i = std::max(1, is - 2); i <= std::min(npx - 1, ie + 2); ++i) {
        const double g_in = g.dya(i, 2) / g.dya(i, 1), g_ou = g.dya(i, -1) / g.dya(i, 0);
        qy(i, 1) = 0.5 * (((2. + g_in) * qin(i, 1) - qin(i, 2)) / (1. + g_in) + ((2. + g_ou) * qin(i, 0) - qin(i, -1)) / (1. + g_ou));
        qy(i, 2) = (3. * (g_in * qin(i, 1) + qin(i, 2)) - (g_in * qy(i, 1) + qy(i, 3))) / (2. + 2. * g_in);
      }
    }
    if (bd.edge_n) {   // :324-355
      std::vector<T> q1(npx + 2);
      for (int i = is1; i <= ie1; ++i) q1[i] = (qin(i, npy - 1) * g.dya(i, npy) + qin(i, npy) * g.dya(i, npy - 1)) / (g.dya(i, npy - 1) + g.dya(i, npy));
      for (int i = is2; i <= ie1; ++i) qout(i, npy) = g.edge_n[i] * q1[i - 1] + (1. - g.edge_n[i]) * q1[i];
      for (int i = std::max(1, is - 2); i <= std::min(npx - 1, ie + 2); ++i) {
        const double g_in = g.dya(i, npy - 2) / g.dya(i, npy - 1), g_ou = g.dya(i, npy + 1) / g.dya(i, npy);
        qy(i, npy) = 0.5 * (((2. + g_in) * qin(i, npy - 1) - qin(i, npy - 2)) / (1. + g_in) + ((2. + g_ou) * qin(i, npy) - qin(i, npy + 1)) / (1. + g_ou));
        qy(i, npy - 1) = (3. * (qin(i, npy - 2) + g_in * qin(i, npy - 1)) - (g_in * qy(i, npy) + qy(i, npy - 2))) / (2. + 2. * g_in);
      }
    }
    // :365-505
    for (int j = std::max(3, js); j <= std::min(npy - 2, je + 1); ++j)
      for (int i = std::max(2, is); i <= std::min(npx - 1, ie + 1); ++i)
        qxx(i, j) = a2b_a2 * (qx(i, j - 2) + qx(i, j + 1)) + a2b_a1 * (qx(i, j - 1) + qx(i, j));
    if (bd.edge_s)
      for (int i = std::max(2, is); i <= std::min(npx - 1, ie + 1); ++i) qxx(i, 2) = c1 * (qx(i, 1) + qx(i, 2)) + c2 * (qout(i, 1) + qxx(i, 3));
    if (bd.edge_n)
      for (int i = std::max(2, is); i <= std::min(npx - 1, ie + 1); ++i)
        qxx(i, npy - 1) = c1 * (qx(i, npy - 2) + qx(i, npy - 1)) + c2 * (qout(i, npy) + qxx(i, npy - 2));
    for (int j = std::max(2, js); j <= std::min(npy - 1, je + 1); ++j) {
      for (int i = std::max(3, is); i <= std::min(npx - 2, ie + 1); ++i)
        qyy(i, j) = a2b_a2 * (qy(i - 2, j) + qy(i + 1, j)) + a2b_a1 * (qy(i - 1, j) + qy(i, j));
      if (bd.edge_w) qyy(2, j) = c1 * (qy(1, j) + qy(2, j)) + c2 * (qout(1, j) + qyy(3, j));
      if (bd.edge_e) qyy(npx - 1, j) = c1 * (qy(npx - 2, j) + qy(npx - 1, j)) + c2 * (qout(npx, j) + qyy(npx - 2, j));
      for (int i = std::max(2, is); i <= std::min(npx - 1, ie + 1); ++i) qout(i, j) = 0.5 * (qxx(i, j) + qyy(i, j));
    }
  }
  if (replace)
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) qin(i, j) = qout(i, j);
}

// edge_interpolate4, sw_core_tlm.F90:6822-6832
template <class T>
T edge_interpolate4(const T& u1, const T& u2, const T& u3, const T& u4, double d1, double d2, double d3, double d4) {
  const double t1 = d1 + d2, t2 = d3 + d4;
  return 0.5 * (((t1 + d2) * u2 - d2 * u1) / t1 + ((t2 + d3) * u3 - d3 * u4) / t2);
}

// d2a2c_vect (dord4=.true.), sw_core_tlm.F90:6399-6806.
template <class T>
void d2a2c_vect(const Arr2<T>& u, const Arr2<T>& v, Arr2<T>& ua, Arr2<T>& va, Arr2<T>& uc, Arr2<T>& vc, Arr2<T>& ut,
                Arr2<T>& vt, const Grid& g, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  const int npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  const int npt = face ? 4 : -(1 << 20);       // interior rank: the max/min clamps never bind
  const double c1 = ppm_c1, c2 = ppm_c2, c3 = ppm_c3;   // sw_core_tlm.F90:60-62 (same values)
  Arr2<T> utmp(bd), vtmp(bd);
  utmp.fill(T(1.e30)); vtmp.fill(T(1.e30));  // big_number, :6463-6464
  {
    const int j0 = face ? std::max(npt, js - 1) : js - 1, j1 = face ? std::min(npy - npt, je + 1) : je + 1;
    const int i0 = face ? std::max(npt, isd) : isd, i1 = face ? std::min(npx - npt, ied) : ied;
    for (int j = j0; j <= j1; ++j)       // :6505-6519
      for (int i = i0; i <= i1; ++i) utmp(i, j) = a2b_a2 * (u(i, j - 1) + u(i, j + 2)) + a2b_a1 * (u(i, j) + u(i, j + 1));
    const int jj0 = face ? std::max(npt, jsd) : jsd, jj1 = face ? std::min(npy - npt, jed) : jed;
    const int ii0 = face ? std::max(npt, is - 1) : is - 1, ii1 = face ? std::min(npx - npt, ie + 1) : ie + 1;
    for (int j = jj0; j <= jj1; ++j)     // :6530-6544
      for (int i = ii0; i <= ii1; ++i) vtmp(i, j) = a2b_a2 * (v(i - 1, j) + v(i + 2, j)) + a2b_a1 * (v(i, j) + v(i + 1, j));
  }
  if (face) {   // edges: 2-point averages (:6548-6602)
    auto avg = [&](int i, int j) { utmp(i, j) = 0.5 * (u(i, j) + u(i, j + 1)); vtmp(i, j) = 0.5 * (v(i, j) + v(i + 1, j)); };
    for (int j = jsd; j <= npt - 1; ++j) for (int i = isd; i <= ied; ++i) avg(i, j);
    for (int j = npy - npt + 1; j <= jed; ++j) for (int i = isd; i <= ied; ++i) avg(i, j);
    for (int j = std::max(npt, jsd); j <= std::min(npy - npt, jed); ++j) {
      for (int i = isd; i <= npt - 1; ++i) avg(i, j);
      for (int i = npx - npt + 1; i <= ied; ++i) avg(i, j);
    }
  }
  const int id = 1;   // dord4
  const int ja0 = face ? js - 1 - id : js - 1, ja1 = face ? je + 1 + id : je + 1, ia0 = face ? is - 1 - id : is - 1, ia1 = face ? ie + 1 + id : ie + 1;
  for (int j = ja0; j <= ja1; ++j)     // :6605-6611 (interior rank: only is-1..ie+1 carries defined data)
    for (int i = ia0; i <= ia1; ++i) {
      ua(i, j) = (utmp(i, j) - vtmp(i, j) * g.cosa_s(i, j)) * g.rsin2(i, j);
      va(i, j) = (vtmp(i, j) - utmp(i, j) * g.cosa_s(i, j)) * g.rsin2(i, j);
    }
  // ---- A -> C, x direction (:6617-6722)
  if (bd.sw_corner) for (int i = -2; i <= 0; ++i) utmp(i, 0) = -vtmp(0, 1 - i);
  if (bd.se_corner) for (int i = 0; i <= 2; ++i) utmp(npx + i, 0) = vtmp(npx, i + 1);
  if (bd.ne_corner) for (int i = 0; i <= 2; ++i) utmp(npx + i, npy) = -vtmp(npx, je - i);
  if (bd.nw_corner) for (int i = -2; i <= 0; ++i) utmp(i, npy) = vtmp(0, je + i);
  const int ifirst = face ? std::max(3, is - 1) : is - 1, ilast = face ? std::min(npx - 2, ie + 2) : ie + 2;
  for (int j = js - 1; j <= je + 1; ++j)
    for (int i = ifirst; i <= ilast; ++i) {
      uc(i, j) = a2b_a2 * (utmp(i - 2, j) + utmp(i + 1, j)) + a2b_a1 * (utmp(i - 1, j) + utmp(i, j));
      ut(i, j) = (uc(i, j) - v(i, j) * g.cosa_u(i, j)) * g.rsin_u(i, j);
    }
  if (bd.sw_corner) { ua(-1, 0) = -va(0, 2); ua(0, 0) = -va(0, 1); }
  if (bd.se_corner) { ua(npx, 0) = va(npx, 1); ua(npx + 1, 0) = va(npx, 2); }
  if (bd.ne_corner) { ua(npx, npy) = -va(npx, npy - 1); ua(npx + 1, npy) = -va(npx, npy - 2); }
  if (bd.nw_corner) { ua(-1, npy) = va(0, npy - 2); ua(0, npy) = va(0, npy - 1); }
  if (bd.edge_w)
    for (int j = js - 1; j <= je + 1; ++j) {
      uc(0, j) = c1 * utmp(-2, j) + c2 * utmp(-1, j) + c3 * utmp(0, j);
      ut(1, j) = edge_interpolate4(ua(-1, j), ua(0, j), ua(1, j), ua(2, j), g.dxa(-1, j), g.dxa(0, j), g.dxa(1, j), g.dxa(2, j));
      uc(1, j) = (val(ut(1, j)) > 0.) ? ut(1, j) * g.sin_sg[3](0, j) : ut(1, j) * g.sin_sg[1](1, j);
      uc(2, j) = c1 * utmp(3, j) + c2 * utmp(2, j) + c3 * utmp(1, j);
      ut(0, j) = (uc(0, j) - v(0, j) * g.cosa_u(0, j)) * g.rsin_u(0, j);
      ut(2, j) = (uc(2, j) - v(2, j) * g.cosa_u(2, j)) * g.rsin_u(2, j);
    }
  if (bd.edge_e)
    for (int j = js - 1; j <= je + 1; ++j) {
      uc(npx - 1, j) = c1 * utmp(npx - 3, j) + c2 * utmp(npx - 2, j) + c3 * utmp(npx - 1, j);
      ut(npx, j) = edge_interpolate4(ua(npx - 2, j), ua(npx - 1, j), ua(npx, j), ua(npx + 1, j), g.dxa(npx - 2, j), g.dxa(npx - 1, j),
                                     g.dxa(npx, j), g.dxa(npx + 1, j));
      uc(npx, j) = (val(ut(npx, j)) > 0.) ? ut(npx, j) * g.sin_sg[3](npx - 1, j) : ut(npx, j) * g.sin_sg[1](npx, j);
      uc(npx + 1, j) = c3 * utmp(npx, j) + c2 * utmp(npx + 1, j) + c1 * utmp(npx + 2, j);
      ut(npx - 1, j) = (uc(npx - 1, j) - v(npx - 1, j) * g.cosa_u(npx - 1, j)) * g.rsin_u(npx - 1, j);
      ut(npx + 1, j) = (uc(npx + 1, j) - v(npx + 1, j) * g.cosa_u(npx + 1, j)) * g.rsin_u(npx + 1, j);
    }
  // ---- y direction (:6726-6803)
  if (bd.sw_corner) for (int j = -2; j <= 0; ++j) vtmp(0, j) = -utmp(1 - j, 0);
  if (bd.nw_corner) for (int j = 0; j <= 2; ++j) vtmp(0, npy + j) = utmp(j + 1, npy);
  if (bd.se_corner) for (int j = -2; j <= 0; ++j) vtmp(npx, j) = utmp(ie + j, 0);
  if (bd.ne_corner) for (int j = 0; j <= 2; ++j) vtmp(npx, npy + j) = -utmp(ie - j, npy);
  if (bd.sw_corner) { va(0, -1) = -ua(2, 0); va(0, 0) = -ua(1, 0); }
  if (bd.se_corner) { va(npx, 0) = ua(npx - 1, 0); va(npx, -1) = ua(npx - 2, 0); }
  if (bd.ne_corner) { va(npx, npy) = -ua(npx - 1, npy); va(npx, npy + 1) = -ua(npx - 2, npy); }
  if (bd.nw_corner) { va(0, npy) = ua(1, npy); va(0, npy + 1) = ua(2, npy); }
  for (int j = js - 1; j <= je + 2; ++j) {
    if (face && j == 1) {
      for (int i = is - 1; i <= ie + 1; ++i) {
        vt(i, j) = edge_interpolate4(va(i, -1), va(i, 0), va(i, 1), va(i, 2), g.dya(i, -1), g.dya(i, 0), g.dya(i, 1), g.dya(i, 2));
        vc(i, j) = (val(vt(i, j)) > 0.) ? vt(i, j) * g.sin_sg[4](i, j - 1) : vt(i, j) * g.sin_sg[2](i, j);
      }
    } else if (face && (j == 0 || j == npy - 1)) {
      for (int i = is - 1; i <= ie + 1; ++i) {
        vc(i, j) = c1 * vtmp(i, j - 2) + c2 * vtmp(i, j - 1) + c3 * vtmp(i, j);
        vt(i, j) = (vc(i, j) - u(i, j) * g.cosa_v(i, j)) * g.rsin_v(i, j);
      }
    } else if (face && (j == 2 || j == npy + 1)) {
      for (int i = is - 1; i <= ie + 1; ++i) {
        vc(i, j) = c1 * vtmp(i, j + 1) + c2 * vtmp(i, j) + c3 * vtmp(i, j - 1);
        vt(i, j) = (vc(i, j) - u(i, j) * g.cosa_v(i, j)) * g.rsin_v(i, j);
      }
    } else if (face && j == npy) {
      for (int i = is - 1; i <= ie + 1; ++i) {
        vt(i, j) = edge_interpolate4(va(i, j - 2), va(i, j - 1), va(i, j), va(i, j + 1), g.dya(i, j - 2), g.dya(i, j - 1), g.dya(i, j), g.dya(i, j + 1));
        vc(i, j) = (val(vt(i, j)) > 0.) ? vt(i, j) * g.sin_sg[4](i, j - 1) : vt(i, j) * g.sin_sg[2](i, j);
      }
    } else {
      for (int i = is - 1; i <= ie + 1; ++i) {
        vc(i, j) = a2b_a2 * (vtmp(i, j - 2) + vtmp(i, j + 1)) + a2b_a1 * (vtmp(i, j - 1) + vtmp(i, j));
        vt(i, j) = (vc(i, j) - u(i, j) * g.cosa_v(i, j)) * g.rsin_v(i, j);
      }
    }
  }
}

// divergence_corner, sw_core_tlm.F90:3966-4082 (grid_type<3 branch).
template <class T>
void divergence_corner(const Arr2<T>& u, const Arr2<T>& v, const Arr2<T>& ua, const Arr2<T>& va, Arr2<T>& divg_d,
                       const Grid& g, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  const int is2 = face ? std::max(2, is) : is, ie1 = face ? std::min(npx - 1, ie + 1) : ie + 1;
  Arr2<T> uf(bd), vf(bd);
  for (int j = js; j <= je + 1; ++j) {
    if (face && (j == 1 || j == npy)) {
      for (int i = is - 1; i <= ie + 1; ++i) uf(i, j) = u(i, j) * g.dyc(i, j) * 0.5 * (g.sin_sg[4](i, j - 1) + g.sin_sg[2](i, j));
    } else {
      for (int i = is - 1; i <= ie + 1; ++i)
        uf(i, j) = (u(i, j) - 0.25 * (va(i, j - 1) + va(i, j)) * (g.cos_sg[4](i, j - 1) + g.cos_sg[2](i, j))) * g.dyc(i, j) * 0.5 *
                   (g.sin_sg[4](i, j - 1) + g.sin_sg[2](i, j));
    }
  }
  for (int j = js - 1; j <= je + 1; ++j) {
    for (int i = is2; i <= ie1; ++i)
      vf(i, j) = (v(i, j) - 0.25 * (ua(i - 1, j) + ua(i, j)) * (g.cos_sg[3](i - 1, j) + g.cos_sg[1](i, j))) * g.dxc(i, j) * 0.5 *
                 (g.sin_sg[3](i - 1, j) + g.sin_sg[1](i, j));
    if (bd.edge_w) vf(1, j) = v(1, j) * g.dxc(1, j) * 0.5 * (g.sin_sg[3](0, j) + g.sin_sg[1](1, j));
    if (bd.edge_e) vf(npx, j) = v(npx, j) * g.dxc(npx, j) * 0.5 * (g.sin_sg[3](npx - 1, j) + g.sin_sg[1](npx, j));
  }
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) divg_d(i, j) = vf(i, j - 1) - vf(i, j) + (uf(i - 1, j) - uf(i, j));
  if (bd.sw_corner) divg_d(1, 1) = divg_d(1, 1) - vf(1, 0);
  if (bd.se_corner) divg_d(npx, 1) = divg_d(npx, 1) - vf(npx, 0);
  if (bd.ne_corner) divg_d(npx, npy) = divg_d(npx, npy) + vf(npx, npy);
  if (bd.nw_corner) divg_d(1, npy) = divg_d(1, npy) + vf(1, npy);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) divg_d(i, j) = g.rarea_c(i, j) * divg_d(i, j);
}

// fill2_4corners, sw_core_tlm.F90:7061-7134
template <class T>
void fill2_4corners(Arr2<T>& q1, Arr2<T>& q2, int dir, const Bounds& bd) {
  const int npx = bd.npx, npy = bd.npy;
  auto both = [&](int di, int dj, int si, int sj) { q1(di, dj) = q1(si, sj); q2(di, dj) = q2(si, sj); };
  if (dir == 1) {
    if (bd.sw_corner) { both(-1, 0, 0, 2); both(0, 0, 0, 1); }
    if (bd.se_corner) { both(npx + 1, 0, npx, 2); both(npx, 0, npx, 1); }
    if (bd.nw_corner) { both(0, npy, 0, npy - 1); both(-1, npy, 0, npy - 2); }
    if (bd.ne_corner) { both(npx, npy, npx, npy - 1); both(npx + 1, npy, npx, npy - 2); }
  } else {
    if (bd.sw_corner) { both(0, 0, 1, 0); both(0, -1, 2, 0); }
    if (bd.se_corner) { both(npx, 0, npx - 1, 0); both(npx, -1, npx - 2, 0); }
    if (bd.nw_corner) { both(0, npy, 1, npy); both(0, npy + 1, 2, npy); }
    if (bd.ne_corner) { both(npx, npy, npx - 1, npy); both(npx, npy + 1, npx - 2, npy); }
  }
}

// fill_4corners (one field; the same two-point pattern as fill2_4corners), sw_core_tlm.F90:6953-7060
template <class T>
void fill_4corners(Arr2<T>& q, int dir, const Bounds& bd) { Arr2<T> other = q; fill2_4corners(q, other, dir, bd); }

// c_sw, sw_core_tlm.F90:646-1038.  Non-hydrostatic (w_in, wc given): w is carried by the same upwind fluxes (:758-778, :812-838).  delp/pt corner halos are filled in place by the
// reference (intent(inout)); the oracle works on copies.
template <class T>
void c_sw(Arr2<T>& delpc, const Arr2<T>& delp_in, Arr2<T>& ptc, const Arr2<T>& pt_in, const Arr2<T>& u, const Arr2<T>& v,
          Arr2<T>& uc, Arr2<T>& vc, Arr2<T>& ua, Arr2<T>& va, Arr2<T>& ut, Arr2<T>& vt, Arr2<T>& divg_d, int nord,
          double dt2, const Grid& g, const Bounds& bd, const Arr2<T>* w_in = nullptr, Arr2<T>* wc = nullptr) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, iep1 = ie + 1, jep1 = je + 1, npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  Arr2<T> delp = delp_in, pt = pt_in;
  d2a2c_vect(u, v, ua, va, uc, vc, ut, vt, g, bd);
  if (nord > 0) divergence_corner(u, v, ua, va, divg_d, g, bd);
  for (int j = js - 1; j <= jep1; ++j)        // :713-721
    for (int i = is - 1; i <= iep1 + 1; ++i) {
      if (val(ut(i, j)) > 0.) ut(i, j) = dt2 * ut(i, j) * g.dy(i, j) * g.sin_sg[3](i - 1, j);
      else                    ut(i, j) = dt2 * ut(i, j) * g.dy(i, j) * g.sin_sg[1](i, j);
    }
  for (int j = js - 1; j <= je + 2; ++j)      // :722-733
    for (int i = is - 1; i <= iep1; ++i) {
      if (val(vt(i, j)) > 0.) vt(i, j) = dt2 * vt(i, j) * g.dx(i, j) * g.sin_sg[4](i, j - 1);
      else                    vt(i, j) = dt2 * vt(i, j) * g.dx(i, j) * g.sin_sg[2](i, j);
    }
  Arr2<T> fx(bd), fx1(bd), fy(bd), fy1(bd), ke(bd), vort(bd), fx2(bd), fy2(bd), w(bd);
  if (w_in) w = *w_in;
  if (face) fill2_4corners(delp, pt, 1, bd);  // :739-741
  if (w_in && face) fill_4corners(w, 1, bd);
  for (int j = js - 1; j <= jep1; ++j)        // :744-757
    for (int i = is - 1; i <= ie + 2; ++i) {
      if (val(ut(i, j)) > 0.) { fx1(i, j) = delp(i - 1, j); fx(i, j) = pt(i - 1, j); }
      else                    { fx1(i, j) = delp(i, j);     fx(i, j) = pt(i, j); }
      fx1(i, j) = ut(i, j) * fx1(i, j);
      fx(i, j) = fx1(i, j) * fx(i, j);
      if (w_in) fx2(i, j) = fx1(i, j) * ((val(ut(i, j)) > 0.) ? w(i - 1, j) : w(i, j));
    }
  if (face) fill2_4corners(delp, pt, 2, bd);  // :784-786
  if (w_in && face) fill_4corners(w, 2, bd);
  for (int j = js - 1; j <= jep1 + 1; ++j)    // :788-800
    for (int i = is - 1; i <= iep1; ++i) {
      if (val(vt(i, j)) > 0.) { fy1(i, j) = delp(i, j - 1); fy(i, j) = pt(i, j - 1); }
      else                    { fy1(i, j) = delp(i, j);     fy(i, j) = pt(i, j); }
      fy1(i, j) = vt(i, j) * fy1(i, j);
      fy(i, j) = fy1(i, j) * fy(i, j);
      if (w_in) fy2(i, j) = fy1(i, j) * ((val(vt(i, j)) > 0.) ? w(i, j - 1) : w(i, j));
    }
  for (int j = js - 1; j <= jep1; ++j)        // :801-808
    for (int i = is - 1; i <= iep1; ++i) {
      delpc(i, j) = delp(i, j) + (fx1(i, j) - fx1(i + 1, j) + (fy1(i, j) - fy1(i, j + 1))) * g.rarea(i, j);
      ptc(i, j) = (pt(i, j) * delp(i, j) + (fx(i, j) - fx(i + 1, j) + (fy(i, j) - fy(i, j + 1))) * g.rarea(i, j)) / delpc(i, j);
      if (w_in) (*wc)(i, j) = (w(i, j) * delp(i, j) + (fx2(i, j) - fx2(i + 1, j) + (fy2(i, j) - fy2(i, j + 1))) * g.rarea(i, j)) / delpc(i, j);
    }
  // KE: upstream C-grid wind, true covariant wind at face edges (:853-917)
  for (int j = js - 1; j <= jep1; ++j)
    for (int i = is - 1; i <= iep1; ++i) {
      if (val(ua(i, j)) > 0.) {
        if (face && i == 1) ke(1, j) = uc(1, j) * g.sin_sg[1](1, j) + v(1, j) * g.cos_sg[1](1, j);
        else if (face && i == npx) ke(i, j) = uc(npx, j) * g.sin_sg[1](npx, j) + v(npx, j) * g.cos_sg[1](npx, j);
        else ke(i, j) = uc(i, j);
      } else {
        if (face && i == 0) ke(0, j) = uc(1, j) * g.sin_sg[3](0, j) + v(1, j) * g.cos_sg[3](0, j);
        else if (face && i == npx - 1) ke(i, j) = uc(npx, j) * g.sin_sg[3](npx - 1, j) + v(npx, j) * g.cos_sg[3](npx - 1, j);
        else ke(i, j) = uc(i + 1, j);
      }
      if (val(va(i, j)) > 0.) {
        if (face && j == 1) vort(i, 1) = vc(i, 1) * g.sin_sg[2](i, 1) + u(i, 1) * g.cos_sg[2](i, 1);
        else if (face && j == npy) vort(i, j) = vc(i, npy) * g.sin_sg[2](i, npy) + u(i, npy) * g.cos_sg[2](i, npy);
        else vort(i, j) = vc(i, j);
      } else {
        if (face && j == 0) vort(i, 0) = vc(i, 1) * g.sin_sg[4](i, 0) + u(i, 1) * g.cos_sg[4](i, 0);
        else if (face && j == npy - 1) vort(i, j) = vc(i, npy) * g.sin_sg[4](i, npy - 1) + u(i, npy) * g.cos_sg[4](i, npy - 1);
        else vort(i, j) = vc(i, j + 1);
      }
    }
  const double dt4 = 0.5 * dt2;
  for (int j = js - 1; j <= jep1; ++j)
    for (int i = is - 1; i <= iep1; ++i) ke(i, j) = dt4 * (ua(i, j) * ke(i, j) + va(i, j) * vort(i, j));
  for (int j = js - 1; j <= je + 1; ++j)      // :929-943
    for (int i = is; i <= ie + 1; ++i) fx(i, j) = uc(i, j) * g.dxc(i, j);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is - 1; i <= ie + 1; ++i) fy(i, j) = vc(i, j) * g.dyc(i, j);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) vort(i, j) = fx(i, j - 1) - fx(i, j) + (fy(i, j) - fy(i - 1, j));
  if (bd.sw_corner) vort(1, 1) = vort(1, 1) + fy(0, 1);           // :945-948
  if (bd.se_corner) vort(npx, 1) = vort(npx, 1) - fy(npx, 1);
  if (bd.ne_corner) vort(npx, npy) = vort(npx, npy) - fy(npx, npy);
  if (bd.nw_corner) vort(1, npy) = vort(1, npy) + fy(0, npy);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) vort(i, j) = g.fC(i, j) + g.rarea_c(i, j) * vort(i, j);   // :952-957
  for (int j = js; j <= je; ++j)              // :985-1003
    for (int i = is; i <= iep1; ++i) {
      if (face && (i == 1 || i == npx)) fy1(i, j) = dt2 * v(i, j);
      else fy1(i, j) = dt2 * (v(i, j) - uc(i, j) * g.cosa_u(i, j)) / g.sina_u(i, j);
      fy(i, j) = (val(fy1(i, j)) > 0.) ? vort(i, j) : vort(i, j + 1);
    }
  for (int j = js; j <= jep1; ++j)            // :1004-1025
    for (int i = is; i <= ie; ++i) {
      if (face && (j == 1 || j == npy)) fx1(i, j) = dt2 * u(i, j);
      else fx1(i, j) = dt2 * (u(i, j) - vc(i, j) * g.cosa_v(i, j)) / g.sina_v(i, j);
      fx(i, j) = (val(fx1(i, j)) > 0.) ? vort(i, j) : vort(i + 1, j);
    }
  for (int j = js; j <= je; ++j)              // :1027-1031
    for (int i = is; i <= iep1; ++i) uc(i, j) = uc(i, j) + fy1(i, j) * fy(i, j) + g.rdxc(i, j) * (ke(i - 1, j) - ke(i, j));
  for (int j = js; j <= jep1; ++j)            // :1032-1037
    for (int i = is; i <= ie; ++i) vc(i, j) = vc(i, j) - fx1(i, j) * fx(i, j) + g.rdyc(i, j) * (ke(i, j - 1) - ke(i, j));
}

// xtp_u: x-transport of u on B-grid points, flux on (is..ie+1, js..je+1).  _TLM :7272-7486.
template <class T>
void xtp_u(const Arr2<T>& c, const Arr2<T>& u, Arr2<T>& flux, int iord, const Grid& g, const Bounds& bd) {
  assert(iord == 1 || iord == 2 || iord == 333);
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  const double c1 = ppm_c1, c2 = ppm_c2, c3 = ppm_c3;
  for (int j = js; j <= je + 1; ++j) {
    if (iord == 1) {
      for (int i = is; i <= ie + 1; ++i) flux(i, j) = (val(c(i, j)) > 0.) ? u(i - 1, j) : u(i, j);
    } else if (iord == 333) {
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.)
          flux(i, j) = (2.0 * u(i, j) + 5.0 * u(i - 1, j) - u(i - 2, j)) / 6.0 - 0.5 * c(i, j) * g.rdx(i - 1, j) * (u(i, j) - u(i - 1, j)) +
                       c(i, j) * g.rdx(i - 1, j) * c(i, j) * g.rdx(i - 1, j) / 6.0 * (u(i, j) - 2.0 * u(i - 1, j) + u(i - 2, j));
        else
          flux(i, j) = (2.0 * u(i - 1, j) + 5.0 * u(i, j) - u(i + 1, j)) / 6.0 - 0.5 * c(i, j) * g.rdx(i, j) * (u(i, j) - u(i - 1, j)) +
                       c(i, j) * g.rdx(i, j) * c(i, j) * g.rdx(i, j) / 6.0 * (u(i + 1, j) - 2.0 * u(i, j) + u(i - 1, j));
      }
    } else {
      const int is3 = face ? std::max(3, is - 1) : is - 1, ie3 = face ? std::min(npx - 3, ie + 1) : ie + 1;
      const int lo = is - 2;
      std::vector<T> al(ie + 6 - lo), bl(ie + 6 - lo), br(ie + 6 - lo), b0(ie + 6 - lo);
      auto AL = [&](int i) -> T& { return al[i - lo]; }; auto BL = [&](int i) -> T& { return bl[i - lo]; };
      auto BR = [&](int i) -> T& { return br[i - lo]; }; auto B0 = [&](int i) -> T& { return b0[i - lo]; };
      for (int i = is3; i <= ie3 + 1; ++i) AL(i) = ppm_p1 * (u(i - 1, j) + u(i, j)) + ppm_p2 * (u(i - 2, j) + u(i + 1, j));
      for (int i = is3; i <= ie3; ++i) { BL(i) = AL(i) - u(i, j); BR(i) = AL(i + 1) - u(i, j); }
      if (bd.edge_w) {   // :7381-7413
        T xt = c3 * u(1, j) + c2 * u(2, j) + c1 * u(3, j);
        BR(1) = xt - u(1, j); BL(2) = xt - u(2, j); BR(2) = AL(3) - u(2, j);
        if (j == 1 || j == npy) { BL(0) = T(0.); BR(0) = T(0.); BL(1) = T(0.); BR(1) = T(0.); }
        else {
          BL(0) = c1 * u(-2, j) + c2 * u(-1, j) + c3 * u(0, j) - u(0, j);
          xt = 0.5 * (((2. * g.dx(0, j) + g.dx(-1, j)) * u(0, j) - g.dx(0, j) * u(-1, j)) / (g.dx(0, j) + g.dx(-1, j)) +
                      ((2. * g.dx(1, j) + g.dx(2, j)) * u(1, j) - g.dx(1, j) * u(2, j)) / (g.dx(1, j) + g.dx(2, j)));
          BR(0) = xt - u(0, j); BL(1) = xt - u(1, j);
        }
      }
      if (bd.edge_e) {   // :7415-7460
        BL(npx - 2) = AL(npx - 2) - u(npx - 2, j);
        T xt = c1 * u(npx - 3, j) + c2 * u(npx - 2, j) + c3 * u(npx - 1, j);
        BR(npx - 2) = xt - u(npx - 2, j); BL(npx - 1) = xt - u(npx - 1, j);
        if (j == 1 || j == npy) { BL(npx - 1) = T(0.); BR(npx - 1) = T(0.); BL(npx) = T(0.); BR(npx) = T(0.); }
        else {
          xt = 0.5 * (((2. * g.dx(npx - 1, j) + g.dx(npx - 2, j)) * u(npx - 1, j) - g.dx(npx - 1, j) * u(npx - 2, j)) / (g.dx(npx - 1, j) + g.dx(npx - 2, j)) +
                      ((2. * g.dx(npx, j) + g.dx(npx + 1, j)) * u(npx, j) - g.dx(npx, j) * u(npx + 1, j)) / (g.dx(npx, j) + g.dx(npx + 1, j)));
          BR(npx - 1) = xt - u(npx - 1, j); BL(npx) = xt - u(npx, j);
          BR(npx) = c3 * u(npx, j) + c2 * u(npx + 1, j) + c1 * u(npx + 2, j) - u(npx, j);
        }
      }
      for (int i = is - 1; i <= ie + 1; ++i) B0(i) = BL(i) + BR(i);
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.) {
          T cfl = c(i, j) * g.rdx(i - 1, j);
          flux(i, j) = u(i - 1, j) + (1. - cfl) * (BR(i - 1) - cfl * B0(i - 1));
        } else {
          T cfl = c(i, j) * g.rdx(i, j);
          flux(i, j) = u(i, j) + (1. + cfl) * (BL(i) + cfl * B0(i));
        }
      }
    }
  }
}

// ytp_v: _TLM :7490-7759.
template <class T>
void ytp_v(const Arr2<T>& c, const Arr2<T>& v, Arr2<T>& flux, int jord, const Grid& g, const Bounds& bd) {
  assert(jord == 1 || jord == 2 || jord == 333);
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  const double c1 = ppm_c1, c2 = ppm_c2, c3 = ppm_c3;
  if (jord == 1) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) flux(i, j) = (val(c(i, j)) > 0.) ? v(i, j - 1) : v(i, j);
  } else if (jord == 333) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.)
          flux(i, j) = (2.0 * v(i, j) + 5.0 * v(i, j - 1) - v(i, j - 2)) / 6.0 - 0.5 * c(i, j) * g.rdy(i, j - 1) * (v(i, j) - v(i, j - 1)) +
                       c(i, j) * g.rdy(i, j - 1) * c(i, j) * g.rdy(i, j - 1) / 6.0 * (v(i, j) - 2.0 * v(i, j - 1) + v(i, j - 2));
        else
          flux(i, j) = (2.0 * v(i, j - 1) + 5.0 * v(i, j) - v(i, j + 1)) / 6.0 - 0.5 * c(i, j) * g.rdy(i, j) * (v(i, j) - v(i, j - 1)) +
                       c(i, j) * g.rdy(i, j) * c(i, j) * g.rdy(i, j) / 6.0 * (v(i, j + 1) - 2.0 * v(i, j) + v(i, j - 1));
      }
  } else {
    const int js3 = face ? std::max(3, js - 1) : js - 1, je3 = face ? std::min(npy - 3, je + 1) : je + 1;
    Arr2<T> al(bd), bl(bd), br(bd), b0(bd);
    for (int j = js3; j <= je3 + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) al(i, j) = ppm_p1 * (v(i, j - 1) + v(i, j)) + ppm_p2 * (v(i, j - 2) + v(i, j + 1));
    for (int j = js3; j <= je3; ++j)
      for (int i = is; i <= ie + 1; ++i) { bl(i, j) = al(i, j) - v(i, j); br(i, j) = al(i, j + 1) - v(i, j); }
    if (bd.edge_s) {   // :7601-7636
      for (int i = is; i <= ie + 1; ++i) {
        bl(i, 0) = c1 * v(i, -2) + c2 * v(i, -1) + c3 * v(i, 0) - v(i, 0);
        T xt = 0.5 * (((2. * g.dy(i, 0) + g.dy(i, -1)) * v(i, 0) - g.dy(i, 0) * v(i, -1)) / (g.dy(i, 0) + g.dy(i, -1)) +
                      ((2. * g.dy(i, 1) + g.dy(i, 2)) * v(i, 1) - g.dy(i, 1) * v(i, 2)) / (g.dy(i, 1) + g.dy(i, 2)));
        br(i, 0) = xt - v(i, 0); bl(i, 1) = xt - v(i, 1);
        xt = c3 * v(i, 1) + c2 * v(i, 2) + c1 * v(i, 3);
        br(i, 1) = xt - v(i, 1); bl(i, 2) = xt - v(i, 2); br(i, 2) = al(i, 3) - v(i, 2);
      }
      if (bd.edge_w) { bl(1, 0) = T(0.); br(1, 0) = T(0.); bl(1, 1) = T(0.); br(1, 1) = T(0.); }
      if (bd.edge_e) { bl(npx, 0) = T(0.); br(npx, 0) = T(0.); bl(npx, 1) = T(0.); br(npx, 1) = T(0.); }
    }
    if (bd.edge_n) {   // :7640-7685
      for (int i = is; i <= ie + 1; ++i) {
        bl(i, npy - 2) = al(i, npy - 2) - v(i, npy - 2);
        T xt = c1 * v(i, npy - 3) + c2 * v(i, npy - 2) + c3 * v(i, npy - 1);
        br(i, npy - 2) = xt - v(i, npy - 2); bl(i, npy - 1) = xt - v(i, npy - 1);
        xt = 0.5 * (((2. * g.dy(i, npy - 1) + g.dy(i, npy - 2)) * v(i, npy - 1) - g.dy(i, npy - 1) * v(i, npy - 2)) / (g.dy(i, npy - 1) + g.dy(i, npy - 2)) +
                    ((2. * g.dy(i, npy) + g.dy(i, npy + 1)) * v(i, npy) - g.dy(i, npy) * v(i, npy + 1)) / (g.dy(i, npy) + g.dy(i, npy + 1)));
        br(i, npy - 1) = xt - v(i, npy - 1); bl(i, npy) = xt - v(i, npy);
        br(i, npy) = c3 * v(i, npy) + c2 * v(i, npy + 1) + c1 * v(i, npy + 2) - v(i, npy);
      }
      if (bd.edge_w) { bl(1, npy - 1) = T(0.); br(1, npy - 1) = T(0.); bl(1, npy) = T(0.); br(1, npy) = T(0.); }
      if (bd.edge_e) { bl(npx, npy - 1) = T(0.); br(npx, npy - 1) = T(0.); bl(npx, npy) = T(0.); br(npx, npy) = T(0.); }
    }
    for (int j = js - 1; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) b0(i, j) = bl(i, j) + br(i, j);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.) {
          T cfl = c(i, j) * g.rdy(i, j - 1);
          flux(i, j) = v(i, j - 1) + (1. - cfl) * (br(i, j - 1) - cfl * b0(i, j - 1));
        } else {
          T cfl = c(i, j) * g.rdy(i, j);
          flux(i, j) = v(i, j) + (1. + cfl) * (bl(i, j) + cfl * b0(i, j));
        }
      }
  }
}

// del6_vt_flux, sw_core_tlm.F90:3719-3801 (nord <= 2).
template <class T>
void del6_vt_flux(int nord, double damp, const Arr2<T>& q, Arr2<T>& d2, Arr2<T>& fx2, Arr2<T>& fy2, const Grid& g,
                  const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  for (int j = js - 1 - nord; j <= je + 1 + nord; ++j)
    for (int i = is - 1 - nord; i <= ie + 1 + nord; ++i) d2(i, j) = damp * q(i, j);
  if (nord > 0) copy_corners(d2, 1, bd);
  for (int j = js - nord; j <= je + nord; ++j)
    for (int i = is - nord; i <= ie + nord + 1; ++i) fx2(i, j) = g.del6_v(i, j) * (d2(i - 1, j) - d2(i, j));
  if (nord > 0) copy_corners(d2, 2, bd);
  for (int j = js - nord; j <= je + nord + 1; ++j)
    for (int i = is - nord; i <= ie + nord; ++i) fy2(i, j) = g.del6_u(i, j) * (d2(i, j - 1) - d2(i, j));
  for (int n = 1; n <= nord; ++n) {
    const int nt = nord - n;
    for (int j = js - nt - 1; j <= je + nt + 1; ++j)
      for (int i = is - nt - 1; i <= ie + nt + 1; ++i)
        d2(i, j) = (fx2(i, j) - fx2(i + 1, j) + (fy2(i, j) - fy2(i, j + 1))) * g.rarea(i, j);
    copy_corners(d2, 1, bd);
    for (int j = js - nt; j <= je + nt; ++j)
      for (int i = is - nt; i <= ie + nt + 1; ++i) fx2(i, j) = g.del6_v(i, j) * (d2(i, j) - d2(i - 1, j));
    copy_corners(d2, 2, bd);
    for (int j = js - nt; j <= je + nt + 1; ++j)
      for (int i = is - nt; i <= ie + nt; ++i) fy2(i, j) = g.del6_u(i, j) * (d2(i, j) - d2(i, j - 1));
  }
}

// fill_corners of a B-grid (corner-point) scalar, tools/fv_mp_nlm_mod.F90:1055-1084 (FILL = XDir / YDir, BGRID), and of a D-grid
// vector pair, fill_corners_dgrid :1271-1303 with mySign = -1 (VECTOR): the corner squares of the halo of a whole cube face.
template <class T>
void fill_corners_bgrid(Arr2<T>& q, int dir, const Bounds& bd) {
  const int npx = bd.npx, npy = bd.npy, ng = bd.ng;
  for (int j = 1; j <= ng; ++j)
    for (int i = 1; i <= ng; ++i) {
      if (dir == 1) {
        if (bd.sw_corner) q(1 - i, 1 - j) = q(1 - j, i + 1);
        if (bd.nw_corner) q(1 - i, npy + j) = q(1 - j, npy - i);
        if (bd.se_corner) q(npx + i, 1 - j) = q(npx + j, i + 1);
        if (bd.ne_corner) q(npx + i, npy + j) = q(npx + j, npy - i);
      } else {
        if (bd.sw_corner) q(1 - j, 1 - i) = q(i + 1, 1 - j);
        if (bd.nw_corner) q(1 - j, npy + i) = q(i + 1, npy + j);
        if (bd.se_corner) q(npx + j, 1 - i) = q(npx - i, 1 - j);
        if (bd.ne_corner) q(npx + j, npy + i) = q(npx - i, npy + j);
      }
    }
}
template <class T>
void fill_corners_dgrid_vector(Arr2<T>& x, Arr2<T>& y, const Bounds& bd) {
  const int npx = bd.npx, npy = bd.npy, ng = bd.ng; const double sgn = -1.0;
  for (int j = 1; j <= ng; ++j)
    for (int i = 1; i <= ng; ++i) {
      if (bd.sw_corner) x(1 - i, 1 - j) = sgn * y(1 - j, i);
      if (bd.nw_corner) x(1 - i, npy + j) = y(1 - j, npy - i);
      if (bd.se_corner) x(npx - 1 + i, 1 - j) = y(npx + j, i);
      if (bd.ne_corner) x(npx - 1 + i, npy + j) = sgn * y(npx + j, npy - i);
    }
  for (int j = 1; j <= ng; ++j)
    for (int i = 1; i <= ng; ++i) {
      if (bd.sw_corner) y(1 - i, 1 - j) = sgn * x(j, 1 - i);
      if (bd.nw_corner) y(1 - i, npy - 1 + j) = x(j, npy + i);
      if (bd.se_corner) y(npx + i, 1 - j) = x(npx - j, 1 - i);
      if (bd.ne_corner) y(npx + i, npy - 1 + j) = sgn * x(npx - j, npy + i);
    }
}

// compute_divergence_damping, sw_core_tlm.F90:7760-8072 (_TLM :8178-8598), grid_type<3, not
// stretched; nord > 1 fills the corner squares before each difference (fill_c, :8463-8530).
template <class T>
void compute_divergence_damping(int nord, double d2_bg, double d4_bg, double dddmp, double dt, Arr2<T>& vort,
                                Arr2<T>& ptc, Arr2<T>& delpc, Arr2<T>& ke, const Arr2<T>& u, const Arr2<T>& v,
                                Arr2<T>& uc, Arr2<T>& vc, const Arr2<T>& ua, const Arr2<T>& va, Arr2<T>& divg_d,
                                Arr2<T>& wk, const Grid& g, const Bounds& bd) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  const int is2 = face ? std::max(2, is) : is, ie1 = face ? std::min(npx - 1, ie + 1) : ie + 1;
  const double absdt = dt >= 0. ? dt : -dt;
  assert(nord <= 3);
  if (nord == 0) {   // :7874-7957
    for (int j = js; j <= je + 1; ++j) {
      if (face && (j == 1 || j == npy)) {
        for (int i = is - 1; i <= ie + 1; ++i)
          ptc(i, j) = (val(vc(i, j)) > 0) ? u(i, j) * g.dyc(i, j) * g.sin_sg[4](i, j - 1) : u(i, j) * g.dyc(i, j) * g.sin_sg[2](i, j);
      } else {
        for (int i = is - 1; i <= ie + 1; ++i)
          ptc(i, j) = (u(i, j) - 0.5 * (va(i, j - 1) + va(i, j)) * g.cosa_v(i, j)) * g.dyc(i, j) * g.sina_v(i, j);
      }
    }
    for (int j = js - 1; j <= je + 1; ++j) {
      for (int i = is2; i <= ie1; ++i)
        vort(i, j) = (v(i, j) - 0.5 * (ua(i - 1, j) + ua(i, j)) * g.cosa_u(i, j)) * g.dxc(i, j) * g.sina_u(i, j);
      if (bd.edge_w) vort(1, j) = (val(uc(1, j)) > 0) ? v(1, j) * g.dxc(1, j) * g.sin_sg[3](0, j) : v(1, j) * g.dxc(1, j) * g.sin_sg[1](1, j);
      if (bd.edge_e) vort(npx, j) = (val(uc(npx, j)) > 0) ? v(npx, j) * g.dxc(npx, j) * g.sin_sg[3](npx - 1, j) : v(npx, j) * g.dxc(npx, j) * g.sin_sg[1](npx, j);
    }
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) delpc(i, j) = vort(i, j - 1) - vort(i, j) + ptc(i - 1, j) - ptc(i, j);
    if (bd.sw_corner) delpc(1, 1) = delpc(1, 1) - vort(1, 0);
    if (bd.se_corner) delpc(npx, 1) = delpc(npx, 1) - vort(npx, 0);
    if (bd.ne_corner) delpc(npx, npy) = delpc(npx, npy) + vort(npx, npy);
    if (bd.nw_corner) delpc(1, npy) = delpc(1, npy) + vort(1, npy);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        delpc(i, j) = g.rarea_c(i, j) * delpc(i, j);
        T abs2 = (val(delpc(i, j)) * dt >= 0.) ? delpc(i, j) * dt : -(delpc(i, j) * dt);
        T y3 = dddmp * abs2;
        T y1 = (0.20 > val(y3)) ? y3 : T(0.20);
        T max1 = (d2_bg < val(y1)) ? y1 : T(d2_bg);
        T damp = g.da_min_c * max1;
        vort(i, j) = damp * delpc(i, j);
        ke(i, j) = ke(i, j) + vort(i, j);
      }
  } else {           // :7958-8071
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) delpc(i, j) = divg_d(i, j);
    const int n2 = nord + 1;
    for (int n = 1; n <= nord; ++n) {
      const int nt = nord - n;
      const bool fill_c = nt != 0 && (bd.sw_corner || bd.se_corner || bd.ne_corner || bd.nw_corner);
      if (fill_c) fill_corners_bgrid(divg_d, 1, bd);
      for (int j = js - nt; j <= je + 1 + nt; ++j)
        for (int i = is - 1 - nt; i <= ie + 1 + nt; ++i) vc(i, j) = (divg_d(i + 1, j) - divg_d(i, j)) * g.divg_u(i, j);
      if (fill_c) fill_corners_bgrid(divg_d, 2, bd);
      for (int j = js - 1 - nt; j <= je + 1 + nt; ++j)
        for (int i = is - nt; i <= ie + 1 + nt; ++i) uc(i, j) = (divg_d(i, j + 1) - divg_d(i, j)) * g.divg_v(i, j);
      if (fill_c) fill_corners_dgrid_vector(vc, uc, bd);
      for (int j = js - nt; j <= je + 1 + nt; ++j)
        for (int i = is - nt; i <= ie + 1 + nt; ++i) divg_d(i, j) = uc(i, j - 1) - uc(i, j) + vc(i - 1, j) - vc(i, j);
      if (bd.sw_corner) divg_d(1, 1) = divg_d(1, 1) - uc(1, 0);
      if (bd.se_corner) divg_d(npx, 1) = divg_d(npx, 1) - uc(npx, 0);
      if (bd.ne_corner) divg_d(npx, npy) = divg_d(npx, npy) + uc(npx, npy);
      if (bd.nw_corner) divg_d(1, npy) = divg_d(1, npy) + uc(1, npy);
      for (int j = js - nt; j <= je + 1 + nt; ++j)
        for (int i = is - nt; i <= ie + 1 + nt; ++i) divg_d(i, j) = divg_d(i, j) * g.rarea_c(i, j);
    }
    if (dddmp < 1.e-5) {
      vort.fill(T(0.0));
    } else {
      a2b_ord4(wk, vort, g, bd, false);
      for (int j = js; j <= je + 1; ++j)
        for (int i = is; i <= ie + 1; ++i) {
          T arg1 = delpc(i, j) * delpc(i, j) + vort(i, j) * vort(i, j);
          vort(i, j) = absdt * sqrt(arg1);
        }
    }
    const double dd8 = std::pow(g.da_min_c * d4_bg, n2);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        T y2 = (0.20 > dddmp * val(vort(i, j))) ? dddmp * vort(i, j) : T(0.20);
        T max2 = (d2_bg < val(y2)) ? y2 : T(d2_bg);
        T damp2 = g.da_min_c * max2;
        vort(i, j) = damp2 * delpc(i, j) + dd8 * divg_d(i, j);
        ke(i, j) = ke(i, j) + vort(i, j);
      }
  }
}

// split_hord for the momentum fluxes (sw_core_tlm.F90:1987-2002 ytp_v, :2059-2072 xtp_u): the _TLM routine with the perturbation scheme,
// then the nonlinear routine with the trajectory scheme -- which may be the monotone 8 / 10 (tp_mono.hpp uv_line_mono) -- for the values.
inline void xtp_u_traj(const Arr2<double>& c, const Arr2<double>& u, Arr2<double>& flux, int iord, const Grid& g, const Bounds& bd) {
  if (!(iord >= 3 && iord <= 13)) { xtp_u<double>(c, u, flux, iord, g, bd); return; }
  for (int j = bd.js; j <= bd.je + 1; ++j)
    if (iord <= 7)
      uv_line_low(iord, bd.is, bd.ie, bd.npx, bd.any_edge(), bd.edge_w, bd.edge_e, bd.any_edge() && (j == 1 || j == bd.npy), [&](int i) { return u(i, j); },
                  [&](int i) { return c(i, j); }, [&](int i) { return g.dx(i, j); }, [&](int i) { return g.rdx(i, j); }, [&](int i, double f) { flux(i, j) = f; });
    else
    uv_line_mono(iord, bd.is, bd.ie, bd.npx, bd.any_edge(), bd.edge_w, bd.edge_e, bd.any_edge() && (j == 1 || j == bd.npy), [&](int i) { return u(i, j); },
                 [&](int i) { return c(i, j); }, [&](int i) { return g.dx(i, j); }, [&](int i) { return g.rdx(i, j); }, [&](int i, double f) { flux(i, j) = f; });
}
inline void ytp_v_traj(const Arr2<double>& c, const Arr2<double>& v, Arr2<double>& flux, int jord, const Grid& g, const Bounds& bd) {
  if (!(jord >= 3 && jord <= 13)) { ytp_v<double>(c, v, flux, jord, g, bd); return; }
  for (int i = bd.is; i <= bd.ie + 1; ++i)
    if (jord <= 7)
      uv_line_low(jord, bd.js, bd.je, bd.npy, bd.any_edge(), bd.edge_s, bd.edge_n, bd.any_edge() && (i == 1 || i == bd.npx), [&](int j) { return v(i, j); },
                  [&](int j) { return c(i, j); }, [&](int j) { return g.dy(i, j); }, [&](int j) { return g.rdy(i, j); }, [&](int j, double f) { flux(i, j) = f; });
    else
    uv_line_mono(jord, bd.js, bd.je, bd.npy, bd.any_edge(), bd.edge_s, bd.edge_n, bd.any_edge() && (i == 1 || i == bd.npx), [&](int j) { return v(i, j); },
                 [&](int j) { return c(i, j); }, [&](int j) { return g.dy(i, j); }, [&](int j) { return g.rdy(i, j); }, [&](int j, double f) { flux(i, j) = f; });
}
template <class T>
void xtp_u_split(const Arr2<T>& c, const Arr2<T>& u, Arr2<T>& flux, int iord, int iord_pert, const Grid& g, const Bounds& bd) {
  if (iord == iord_pert) { xtp_u(c, u, flux, iord, g, bd); return; }
  xtp_u(c, u, flux, iord_pert, g, bd);
  const Arr2<double> cd = values_of(c), ud = values_of(u);
  Arr2<double> fd(bd);
  xtp_u_traj(cd, ud, fd, iord, g, bd);
  for (int j = bd.js; j <= bd.je + 1; ++j) for (int i = bd.is; i <= bd.ie + 1; ++i) set_val(flux(i, j), fd(i, j));
}
template <class T>
void ytp_v_split(const Arr2<T>& c, const Arr2<T>& v, Arr2<T>& flux, int jord, int jord_pert, const Grid& g, const Bounds& bd) {
  if (jord == jord_pert) { ytp_v(c, v, flux, jord, g, bd); return; }
  ytp_v(c, v, flux, jord_pert, g, bd);
  const Arr2<double> cd = values_of(c), vd = values_of(v);
  Arr2<double> fd(bd);
  ytp_v_traj(cd, vd, fd, jord, g, bd);
  for (int j = bd.js; j <= bd.je + 1; ++j) for (int i = bd.is; i <= bd.ie + 1; ++i) set_val(flux(i, j), fd(i, j));
}

// d_sw, hydrostatic, inline_q=.false., d_con<=1e-5, grid_type<3.
// sw_core_tlm.F90:2533-3617 (_TLM :1047-2531).  In/out: delp, pt, u, v (updated on the compute
// domain), xflux,yflux,cx,cy accumulators; in: uc, vc, ua, va, divg_d; out: crx_adv.. yfx_adv.
template <class T>
void d_sw(Arr2<T>& delp, Arr2<T>& pt, Arr2<T>& u, Arr2<T>& v, Arr2<T>& uc, Arr2<T>& vc, const Arr2<T>& ua,
          const Arr2<T>& va, Arr2<T>& divg_d, Arr2<T>& xflux, Arr2<T>& yflux, Arr2<T>& cx, Arr2<T>& cy,
          Arr2<T>& crx_adv, Arr2<T>& cry_adv, Arr2<T>& xfx_adv, Arr2<T>& yfx_adv, double dt, const LevelParams& lp,
          double dddmp, double d4_bg, const Grid& g, const Bounds& bd, Arr2<T>* w = nullptr) {
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  const int npx = bd.npx, npy = bd.npy;
  const bool face = bd.any_edge();
  Arr2<T> ut(bd), vt(bd), ra_x(bd), ra_y(bd), fx(bd), fy(bd), gx(bd), gy(bd), ub(bd), vb(bd), ke(bd), wk(bd),
      vort(bd), ptc(bd), delpc(bd), fx2(bd), fy2(bd);
  // contra-variant winds (:2717-2738): rows/columns next to a face edge are set by the edge code below
  for (int j = jsd; j <= jed; ++j) {
    if (face && (j == 0 || j == 1 || j == npy - 1 || j == npy)) continue;
    for (int i = is - 1; i <= ie + 2; ++i)
      ut(i, j) = (uc(i, j) - 0.25 * g.cosa_u(i, j) * (vc(i - 1, j) + vc(i, j) + vc(i - 1, j + 1) + vc(i, j + 1))) * g.rsin_u(i, j);
  }
  for (int j = js - 1; j <= je + 2; ++j) {
    if (face && (j == 1 || j == npy)) continue;
    for (int i = isd; i <= ied; ++i)
      vt(i, j) = (vc(i, j) - 0.25 * g.cosa_v(i, j) * (uc(i, j - 1) + uc(i + 1, j - 1) + uc(i, j) + uc(i + 1, j))) * g.rsin_v(i, j);
  }
  if (bd.edge_w) {   // :2742-2766
    for (int j = jsd; j <= jed; ++j) ut(1, j) = (val(uc(1, j)) * dt > 0.) ? uc(1, j) / g.sin_sg[3](0, j) : uc(1, j) / g.sin_sg[1](1, j);
    for (int j = std::max(3, js); j <= std::min(npy - 2, je + 1); ++j) {
      vt(0, j) = vc(0, j) - 0.25 * g.cosa_v(0, j) * (ut(0, j - 1) + ut(1, j - 1) + ut(0, j) + ut(1, j));
      vt(1, j) = vc(1, j) - 0.25 * g.cosa_v(1, j) * (ut(1, j - 1) + ut(2, j - 1) + ut(1, j) + ut(2, j));
    }
  }
  if (bd.edge_e) {   // :2768-2793
    for (int j = jsd; j <= jed; ++j)
      ut(npx, j) = (val(uc(npx, j)) * dt > 0.) ? uc(npx, j) / g.sin_sg[3](npx - 1, j) : uc(npx, j) / g.sin_sg[1](npx, j);
    for (int j = std::max(3, js); j <= std::min(npy - 2, je + 1); ++j) {
      vt(npx - 1, j) = vc(npx - 1, j) - 0.25 * g.cosa_v(npx - 1, j) * (ut(npx - 1, j - 1) + ut(npx, j - 1) + ut(npx - 1, j) + ut(npx, j));
      vt(npx, j) = vc(npx, j) - 0.25 * g.cosa_v(npx, j) * (ut(npx, j - 1) + ut(npx + 1, j - 1) + ut(npx, j) + ut(npx + 1, j));
    }
  }
  if (bd.edge_s) {   // :2795-2819
    for (int i = isd; i <= ied; ++i) vt(i, 1) = (val(vc(i, 1)) * dt > 0.) ? vc(i, 1) / g.sin_sg[4](i, 0) : vc(i, 1) / g.sin_sg[2](i, 1);
    for (int i = std::max(3, is); i <= std::min(npx - 2, ie + 1); ++i) {
      ut(i, 0) = uc(i, 0) - 0.25 * g.cosa_u(i, 0) * (vt(i - 1, 0) + vt(i, 0) + vt(i - 1, 1) + vt(i, 1));
      ut(i, 1) = uc(i, 1) - 0.25 * g.cosa_u(i, 1) * (vt(i - 1, 1) + vt(i, 1) + vt(i - 1, 2) + vt(i, 2));
    }
  }
  if (bd.edge_n) {   // :2821-2846
    for (int i = isd; i <= ied; ++i)
      vt(i, npy) = (val(vc(i, npy)) * dt > 0.) ? vc(i, npy) / g.sin_sg[4](i, npy - 1) : vc(i, npy) / g.sin_sg[2](i, npy);
    for (int i = std::max(3, is); i <= std::min(npx - 2, ie + 1); ++i) {
      ut(i, npy - 1) = uc(i, npy - 1) - 0.25 * g.cosa_u(i, npy - 1) * (vt(i - 1, npy - 1) + vt(i, npy - 1) + vt(i - 1, npy) + vt(i, npy));
      ut(i, npy) = uc(i, npy) - 0.25 * g.cosa_u(i, npy) * (vt(i - 1, npy) + vt(i, npy) + vt(i - 1, npy + 1) + vt(i, npy + 1));
    }
  }
  // 2x2 systems at the four corners (:2856-2919)
  if (bd.sw_corner) {
    double damp = 1. / (1. - 0.0625 * g.cosa_u(2, 0) * g.cosa_v(1, 0));
    ut(2, 0) = (uc(2, 0) - 0.25 * g.cosa_u(2, 0) * (vt(1, 1) + vt(2, 1) + vt(2, 0) + vc(1, 0) - 0.25 * g.cosa_v(1, 0) * (ut(1, 0) + ut(1, -1) + ut(2, -1)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(0, 1) * g.cosa_v(0, 2));
    vt(0, 2) = (vc(0, 2) - 0.25 * g.cosa_v(0, 2) * (ut(1, 1) + ut(1, 2) + ut(0, 2) + uc(0, 1) - 0.25 * g.cosa_u(0, 1) * (vt(0, 1) + vt(-1, 1) + vt(-1, 2)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(2, 1) * g.cosa_v(1, 2));
    ut(2, 1) = (uc(2, 1) - 0.25 * g.cosa_u(2, 1) * (vt(1, 1) + vt(2, 1) + vt(2, 2) + vc(1, 2) - 0.25 * g.cosa_v(1, 2) * (ut(1, 1) + ut(1, 2) + ut(2, 2)))) * damp;
    vt(1, 2) = (vc(1, 2) - 0.25 * g.cosa_v(1, 2) * (ut(1, 1) + ut(1, 2) + ut(2, 2) + uc(2, 1) - 0.25 * g.cosa_u(2, 1) * (vt(1, 1) + vt(2, 1) + vt(2, 2)))) * damp;
  }
  if (bd.se_corner) {
    double damp = 1. / (1. - 0.0625 * g.cosa_u(npx - 1, 0) * g.cosa_v(npx - 1, 0));
    ut(npx - 1, 0) = (uc(npx - 1, 0) - 0.25 * g.cosa_u(npx - 1, 0) * (vt(npx - 1, 1) + vt(npx - 2, 1) + vt(npx - 2, 0) + vc(npx - 1, 0) -
                      0.25 * g.cosa_v(npx - 1, 0) * (ut(npx, 0) + ut(npx, -1) + ut(npx - 1, -1)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(npx + 1, 1) * g.cosa_v(npx, 2));
    vt(npx, 2) = (vc(npx, 2) - 0.25 * g.cosa_v(npx, 2) * (ut(npx, 1) + ut(npx, 2) + ut(npx + 1, 2) + uc(npx + 1, 1) -
                  0.25 * g.cosa_u(npx + 1, 1) * (vt(npx, 1) + vt(npx + 1, 1) + vt(npx + 1, 2)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(npx - 1, 1) * g.cosa_v(npx - 1, 2));
    ut(npx - 1, 1) = (uc(npx - 1, 1) - 0.25 * g.cosa_u(npx - 1, 1) * (vt(npx - 1, 1) + vt(npx - 2, 1) + vt(npx - 2, 2) + vc(npx - 1, 2) -
                      0.25 * g.cosa_v(npx - 1, 2) * (ut(npx, 1) + ut(npx, 2) + ut(npx - 1, 2)))) * damp;
    vt(npx - 1, 2) = (vc(npx - 1, 2) - 0.25 * g.cosa_v(npx - 1, 2) * (ut(npx, 1) + ut(npx, 2) + ut(npx - 1, 2) + uc(npx - 1, 1) -
                      0.25 * g.cosa_u(npx - 1, 1) * (vt(npx - 1, 1) + vt(npx - 2, 1) + vt(npx - 2, 2)))) * damp;
  }
  if (bd.ne_corner) {
    double damp = 1. / (1. - 0.0625 * g.cosa_u(npx - 1, npy) * g.cosa_v(npx - 1, npy + 1));
    ut(npx - 1, npy) = (uc(npx - 1, npy) - 0.25 * g.cosa_u(npx - 1, npy) * (vt(npx - 1, npy) + vt(npx - 2, npy) + vt(npx - 2, npy + 1) + vc(npx - 1, npy + 1) -
                        0.25 * g.cosa_v(npx - 1, npy + 1) * (ut(npx, npy) + ut(npx, npy + 1) + ut(npx - 1, npy + 1)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(npx + 1, npy - 1) * g.cosa_v(npx, npy - 1));
    vt(npx, npy - 1) = (vc(npx, npy - 1) - 0.25 * g.cosa_v(npx, npy - 1) * (ut(npx, npy - 1) + ut(npx, npy - 2) + ut(npx + 1, npy - 2) + uc(npx + 1, npy - 1) -
                        0.25 * g.cosa_u(npx + 1, npy - 1) * (vt(npx, npy) + vt(npx + 1, npy) + vt(npx + 1, npy - 1)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(npx - 1, npy - 1) * g.cosa_v(npx - 1, npy - 1));
    ut(npx - 1, npy - 1) = (uc(npx - 1, npy - 1) - 0.25 * g.cosa_u(npx - 1, npy - 1) * (vt(npx - 1, npy) + vt(npx - 2, npy) + vt(npx - 2, npy - 1) + vc(npx - 1, npy - 1) -
                            0.25 * g.cosa_v(npx - 1, npy - 1) * (ut(npx, npy - 1) + ut(npx, npy - 2) + ut(npx - 1, npy - 2)))) * damp;
    vt(npx - 1, npy - 1) = (vc(npx - 1, npy - 1) - 0.25 * g.cosa_v(npx - 1, npy - 1) * (ut(npx, npy - 1) + ut(npx, npy - 2) + ut(npx - 1, npy - 2) + uc(npx - 1, npy - 1) -
                            0.25 * g.cosa_u(npx - 1, npy - 1) * (vt(npx - 1, npy) + vt(npx - 2, npy) + vt(npx - 2, npy - 1)))) * damp;
  }
  if (bd.nw_corner) {
    double damp = 1. / (1. - 0.0625 * g.cosa_u(2, npy) * g.cosa_v(1, npy + 1));
    ut(2, npy) = (uc(2, npy) - 0.25 * g.cosa_u(2, npy) * (vt(1, npy) + vt(2, npy) + vt(2, npy + 1) + vc(1, npy + 1) -
                  0.25 * g.cosa_v(1, npy + 1) * (ut(1, npy) + ut(1, npy + 1) + ut(2, npy + 1)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(0, npy - 1) * g.cosa_v(0, npy - 1));
    vt(0, npy - 1) = (vc(0, npy - 1) - 0.25 * g.cosa_v(0, npy - 1) * (ut(1, npy - 1) + ut(1, npy - 2) + ut(0, npy - 2) + uc(0, npy - 1) -
                      0.25 * g.cosa_u(0, npy - 1) * (vt(0, npy) + vt(-1, npy) + vt(-1, npy - 1)))) * damp;
    damp = 1. / (1. - 0.0625 * g.cosa_u(2, npy - 1) * g.cosa_v(1, npy - 1));
    ut(2, npy - 1) = (uc(2, npy - 1) - 0.25 * g.cosa_u(2, npy - 1) * (vt(1, npy) + vt(2, npy) + vt(2, npy - 1) + vc(1, npy - 1) -
                      0.25 * g.cosa_v(1, npy - 1) * (ut(1, npy - 1) + ut(1, npy - 2) + ut(2, npy - 2)))) * damp;
    vt(1, npy - 1) = (vc(1, npy - 1) - 0.25 * g.cosa_v(1, npy - 1) * (ut(1, npy - 1) + ut(1, npy - 2) + ut(2, npy - 2) + uc(2, npy - 1) -
                      0.25 * g.cosa_u(2, npy - 1) * (vt(1, npy) + vt(2, npy) + vt(2, npy - 1)))) * damp;
  }
  for (int j = jsd; j <= jed; ++j)            // :2932-2936
    for (int i = is; i <= ie + 1; ++i) xfx_adv(i, j) = dt * ut(i, j);
  for (int j = js; j <= je + 1; ++j)
    for (int i = isd; i <= ied; ++i) yfx_adv(i, j) = dt * vt(i, j);
  for (int j = jsd; j <= jed; ++j)            // :2945-2956
    for (int i = is; i <= ie + 1; ++i) {
      if (val(xfx_adv(i, j)) > 0.) {
        crx_adv(i, j) = xfx_adv(i, j) * g.rdxa(i - 1, j);
        xfx_adv(i, j) = g.dy(i, j) * xfx_adv(i, j) * g.sin_sg[3](i - 1, j);
      } else {
        crx_adv(i, j) = xfx_adv(i, j) * g.rdxa(i, j);
        xfx_adv(i, j) = g.dy(i, j) * xfx_adv(i, j) * g.sin_sg[1](i, j);
      }
    }
  for (int j = js; j <= je + 1; ++j)          // :2957-2968
    for (int i = isd; i <= ied; ++i) {
      if (val(yfx_adv(i, j)) > 0.) {
        cry_adv(i, j) = yfx_adv(i, j) * g.rdya(i, j - 1);
        yfx_adv(i, j) = g.dx(i, j) * yfx_adv(i, j) * g.sin_sg[4](i, j - 1);
      } else {
        cry_adv(i, j) = yfx_adv(i, j) * g.rdya(i, j);
        yfx_adv(i, j) = g.dx(i, j) * yfx_adv(i, j) * g.sin_sg[2](i, j);
      }
    }
  for (int j = jsd; j <= jed; ++j)            // :2969-2978
    for (int i = is; i <= ie; ++i) ra_x(i, j) = g.area(i, j) + (xfx_adv(i, j) - xfx_adv(i + 1, j));
  for (int j = js; j <= je; ++j)
    for (int i = isd; i <= ied; ++i) ra_y(i, j) = g.area(i, j) + (yfx_adv(i, j) - yfx_adv(i, j + 1));
  // delp transport (:2979-2986): nord=nord_v, damp_c=damp_v, no mass
  fv_tp_2d_split<T>(delp, crx_adv, cry_adv, lp.hord_dp, lp.hord_dp_pert, fx, fy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, nullptr, nullptr,
                    nullptr, lp.nord_v, lp.damp_vt, lp.nord_v_pert, lp.damp_vt_pert, lp.split_damp);
  for (int j = jsd; j <= jed; ++j)            // flux capacitors :2988-3006
    for (int i = is; i <= ie + 1; ++i) cx(i, j) = cx(i, j) + crx_adv(i, j);
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie + 1; ++i) xflux(i, j) = xflux(i, j) + fx(i, j);
  for (int j = js; j <= je + 1; ++j) {
    for (int i = isd; i <= ied; ++i) cy(i, j) = cy(i, j) + cry_adv(i, j);
    for (int i = is; i <= ie; ++i) yflux(i, j) = yflux(i, j) + fy(i, j);
  }
  // non-hydrostatic: del-2/4 damping increment of w and transport of w with the mass fluxes (:3020-3062)
  Arr2<T> dw(bd);
  if (w) {
    if (lp.damp_w > 1.e-5) {
      const double damp4 = std::pow(lp.damp_w * g.da_min_c, lp.nord_w + 1);
      del6_vt_flux(lp.nord_w, damp4, *w, wk, fx2, fy2, g, bd);
      for (int j = js; j <= je; ++j)
        for (int i = is; i <= ie; ++i) dw(i, j) = (fx2(i, j) - fx2(i + 1, j) + (fy2(i, j) - fy2(i, j + 1))) * g.rarea(i, j);
    }
    fv_tp_2d_split<T>(*w, crx_adv, cry_adv, lp.hord_vt, lp.hord_vt_pert, gx, gy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, &fx, &fy, nullptr, -1, 0.0, -1, 0.0);
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) (*w)(i, j) = delp(i, j) * (*w)(i, j) + (gx(i, j) - gx(i + 1, j) + (gy(i, j) - gy(i, j + 1))) * g.rarea(i, j);
  }
  // pt transport (:3064-3072): mass=delp, nord_t, damp_t
  // perturbation damping of the split call: nord_t_pert, damp_t_pert = nord_v_pert(k), damp_vt_pert(k) as they stand BEFORE the
  // perturbation sponge overrides the latter two (dyn_core_tlm.F90:856-859, :913-917)
  fv_tp_2d_split<T>(pt, crx_adv, cry_adv, lp.hord_tm, lp.hord_tm_pert, gx, gy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, &fx, &fy, &delp,
                    lp.nord_t, lp.damp_t, lp.nord_t_pert, lp.damp_t_pert, lp.split_damp);
  for (int j = js; j <= je; ++j)              // :3107-3116
    for (int i = is; i <= ie; ++i) {
      pt(i, j) = pt(i, j) * delp(i, j) + (gx(i, j) - gx(i + 1, j) + (gy(i, j) - gy(i, j + 1))) * g.rarea(i, j);
      delp(i, j) = delp(i, j) + (fx(i, j) - fx(i + 1, j) + (fy(i, j) - fy(i, j + 1))) * g.rarea(i, j);
      pt(i, j) = pt(i, j) / delp(i, j);
    }
  // kinetic-energy fluxes (:3126-3254)
  const double dt5 = 0.5 * dt, dt4 = 0.25 * dt;
  const int is2 = face ? std::max(2, is) : is, ie1 = face ? std::min(npx - 1, ie + 1) : ie + 1;
  const int js2 = face ? std::max(2, js) : js, je1 = face ? std::min(npy - 1, je + 1) : je + 1;
  if (bd.edge_s) for (int i = is; i <= ie + 1; ++i) vb(i, 1) = dt5 * (vt(i - 1, 1) + vt(i, 1));
  for (int j = js2; j <= je1; ++j) {
    for (int i = is2; i <= ie1; ++i)
      vb(i, j) = dt5 * (vc(i - 1, j) + vc(i, j) - (uc(i, j - 1) + uc(i, j)) * g.cosa(i, j)) * g.rsina(i, j);
    if (bd.edge_w) vb(1, j) = dt4 * (-vt(-1, j) + 3. * (vt(0, j) + vt(1, j)) - vt(2, j));
    if (bd.edge_e) vb(npx, j) = dt4 * (-vt(npx - 2, j) + 3. * (vt(npx - 1, j) + vt(npx, j)) - vt(npx + 1, j));
  }
  if (bd.edge_n) for (int i = is; i <= ie + 1; ++i) vb(i, npy) = dt5 * (vt(i - 1, npy) + vt(i, npy));
  ytp_v_split(vb, v, ub, lp.hord_mt, lp.hord_mt_pert, g, bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) ke(i, j) = vb(i, j) * ub(i, j);
  if (bd.edge_w) for (int j = js; j <= je + 1; ++j) ub(1, j) = dt5 * (ut(1, j - 1) + ut(1, j));
  for (int j = js; j <= je + 1; ++j) {
    if (face && (j == 1 || j == npy)) {
      for (int i = is2; i <= ie1; ++i) ub(i, j) = dt4 * (-ut(i, j - 2) + 3. * (ut(i, j - 1) + ut(i, j)) - ut(i, j + 1));
    } else {
      for (int i = is2; i <= ie1; ++i)
        ub(i, j) = dt5 * (uc(i, j - 1) + uc(i, j) - (vc(i - 1, j) + vc(i, j)) * g.cosa(i, j)) * g.rsina(i, j);
    }
  }
  if (bd.edge_e) for (int j = js; j <= je + 1; ++j) ub(npx, j) = dt5 * (ut(npx, j - 1) + ut(npx, j));
  xtp_u_split(ub, u, vb, lp.hord_mt, lp.hord_mt_pert, g, bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) ke(i, j) = 0.5 * (ke(i, j) + ub(i, j) * vb(i, j));
  {   // KE at the 4 corners of the face (:3258-3273)
    const double dt6 = dt / 6.;
    if (bd.sw_corner) ke(1, 1) = dt6 * ((ut(1, 1) + ut(1, 0)) * u(1, 1) + (vt(1, 1) + vt(0, 1)) * v(1, 1) + (ut(1, 1) + vt(1, 1)) * u(0, 1));
    if (bd.se_corner) ke(npx, 1) = dt6 * ((ut(npx, 1) + ut(npx, 0)) * u(npx - 1, 1) + (vt(npx, 1) + vt(npx - 1, 1)) * v(npx, 1) + (ut(npx, 1) - vt(npx - 1, 1)) * u(npx, 1));
    if (bd.ne_corner) ke(npx, npy) = dt6 * ((ut(npx, npy) + ut(npx, npy - 1)) * u(npx - 1, npy) + (vt(npx, npy) + vt(npx - 1, npy)) * v(npx, npy - 1) +
                                           (ut(npx, npy - 1) + vt(npx - 1, npy)) * u(npx, npy));
    if (bd.nw_corner) ke(1, npy) = dt6 * ((ut(1, npy) + ut(1, npy - 1)) * u(1, npy) + (vt(1, npy) + vt(0, npy)) * v(1, npy - 1) + (ut(1, npy - 1) - vt(1, npy)) * u(0, npy));
  }
  // vorticity (:3275-3293); vt/ut are reused as u*dx, v*dy like the reference
  for (int j = jsd; j <= jed + 1; ++j)
    for (int i = isd; i <= ied; ++i) vt(i, j) = u(i, j) * g.dx(i, j);
  for (int j = jsd; j <= jed; ++j)
    for (int i = isd; i <= ied + 1; ++i) ut(i, j) = v(i, j) * g.dy(i, j);
  for (int j = jsd; j <= jed; ++j)
    for (int i = isd; i <= ied; ++i) wk(i, j) = g.rarea(i, j) * (vt(i, j) - vt(i, j + 1) + (ut(i + 1, j) - ut(i, j)));
  if (w) {   // :3305-3330 (new delp)
    for (int j = js; j <= je; ++j)
      for (int i = is; i <= ie; ++i) {
        (*w)(i, j) = (*w)(i, j) / delp(i, j);
        if (lp.damp_w > 1.e-5) (*w)(i, j) = (*w)(i, j) + dw(i, j);
      }
  }
  if (!lp.split_damp) {
    compute_divergence_damping(lp.nord, lp.d2_divg, d4_bg, dddmp, dt, vort, ptc, delpc, ke, u, v, uc, vc, ua, va,
                               divg_d, wk, g, bd);
  } else {
    // sw_core_tlm.F90:2350-2369: the _TLM routine with the perturbation's coefficients on copies (*_tj) -- its tangents are kept --
    // then the nonlinear routine with the trajectory's coefficients, whose values are kept
    Arr2<T> wk_tj = wk, vort_tj = vort, delpc_tj = delpc, ptc_tj = ptc, ke_tj = ke, vc_tj = vc, uc_tj = uc, divg_d_tj = divg_d;
    compute_divergence_damping(lp.nord_pert, lp.d2_divg_pert, lp.d4_bg_pert, lp.dddmp_pert, dt, vort_tj, ptc_tj, delpc_tj, ke_tj, u, v,
                               uc_tj, vc_tj, ua, va, divg_d_tj, wk_tj, g, bd);
    Arr2<double> wkd = values_of(wk), vortd = values_of(vort), delpcd = values_of(delpc), ptcd = values_of(ptc), ked = values_of(ke),
                 ud = values_of(u), vd = values_of(v), ucd = values_of(uc), vcd = values_of(vc), uad = values_of(ua), vad = values_of(va),
                 dgd = values_of(divg_d);
    compute_divergence_damping<double>(lp.nord, lp.d2_divg, d4_bg, dddmp, dt, vortd, ptcd, delpcd, ked, ud, vd, ucd, vcd, uad, vad, dgd, wkd, g, bd);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) { ke(i, j) = ke_tj(i, j); set_val(ke(i, j), ked(i, j)); }
  }
  for (int j = jsd; j <= jed; ++j)            // :3535-3540 hydrostatic
    for (int i = isd; i <= ied; ++i) vort(i, j) = wk(i, j) + g.f0(i, j);
  fv_tp_2d_split<T>(vort, crx_adv, cry_adv, lp.hord_vt, lp.hord_vt_pert, fx, fy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, nullptr, nullptr,
                    nullptr, -1, 0.0, -1, 0.0);
  for (int j = js; j <= je + 1; ++j)          // :3555-3564
    for (int i = is; i <= ie; ++i) u(i, j) = vt(i, j) + (ke(i, j) - ke(i + 1, j)) + fy(i, j);
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie + 1; ++i) v(i, j) = ut(i, j) + (ke(i, j) - ke(i, j + 1)) - fx(i, j);
  // Vorticity damping: trajectory with (nord_v, damp_vt), perturbation with the _pert pair
  // (sw_core_tlm.F90:2436-2452, :2502-2530).
  Arr2<T> ut_p(bd), vt_p(bd), d2_p(bd);
  ut.fill(T(0.0)); vt.fill(T(0.0));
  if (lp.damp_vt > 1.e-5) {
    double damp4 = std::pow(lp.damp_vt * g.da_min_c, lp.nord_v + 1);
    del6_vt_flux(lp.nord_v, damp4, wk, vort, ut, vt, g, bd);
  }
  if (lp.damp_vt_pert > 1.e-5) {
    double damp4 = std::pow(lp.damp_vt_pert * g.da_min_c, lp.nord_v_pert + 1);
    del6_vt_flux(lp.nord_v_pert, damp4, wk, d2_p, ut_p, vt_p, g, bd);
  }
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie; ++i) u(i, j) = u(i, j) + combine(vt(i, j), vt_p(i, j));
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie + 1; ++i) v(i, j) = v(i, j) - combine(ut(i, j), ut_p(i, j));
}

}  // namespace orc
