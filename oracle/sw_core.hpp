// TEST INFRASTRUCTURE — CPU oracle (see scalar.hpp).
// Restatement of model_tlmadm/sw_core_tlm.F90 (per-level shallow-water stencils):
//   c_sw (:646-1038 / _TLM :87-645), d2a2c_vect (:6399-6806), divergence_corner (:3966-4082),
//   d_sw (:2533-3617 / _TLM :1047-2531), xtp_u/ytp_v (_TLM :7272-7486 / :7490-7759),
//   del6_vt_flux (:3719-3801), compute_divergence_damping (:7760-8072 / _TLM :8178-8598),
// and a2b_ord4 (a2b_edge_tlm.F90:48-542).  Interior-rank path (no cube edge in the tile):
// every `is .EQ. 1`, `j .EQ. npy`, corner and `nested` branch of the reference is skipped, exactly
// as it is for an MPI rank in the middle of a face; grid_type < 3 formulas (cosa/sina metrics).
#pragma once
#include "tp_core.hpp"

namespace orc {

static const double a2b_a1 = 0.5625, a2b_a2 = -0.0625;  // sw_core_tlm.F90:56-57 / a2b_edge_tlm.F90:37-38
static const double a2b_b1 = 7. / 12., a2b_b2 = -1. / 12.;

// a2b_ord4, interior rank: a2b_edge_tlm.F90:163-176 (qx), :268-291 (qy), :365-420, :441-505.
// replace=true copies qout back into qin on is..ie+1, js..je+1 (:531-541).
template <class T>
void a2b_ord4(Arr2<T>& qin, Arr2<T>& qout, const Grid& g, const Bounds& bd, bool replace) {
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  Arr2<T> qx(bd), qy(bd);
  for (int j = js - 2; j <= je + 2; ++j)
    for (int i = is; i <= ie + 1; ++i)
      qx(i, j) = a2b_b2 * (qin(i - 2, j) + qin(i + 1, j)) + a2b_b1 * (qin(i - 1, j) + qin(i, j));
  for (int j = js; j <= je + 1; ++j)
    for (int i = is - 2; i <= ie + 2; ++i)
      qy(i, j) = a2b_b2 * (qin(i, j - 2) + qin(i, j + 1)) + a2b_b1 * (qin(i, j - 1) + qin(i, j));
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) {
      T qxx = a2b_a2 * (qx(i, j - 2) + qx(i, j + 1)) + a2b_a1 * (qx(i, j - 1) + qx(i, j));
      T qyy = a2b_a2 * (qy(i - 2, j) + qy(i + 1, j)) + a2b_a1 * (qy(i - 1, j) + qy(i, j));
      qout(i, j) = 0.5 * (qxx + qyy);
    }
  if (replace)
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) qin(i, j) = qout(i, j);
}

// d2a2c_vect (dord4=.true. call from c_sw), sw_core_tlm.F90:6399-6806, interior rank.
template <class T>
void d2a2c_vect(const Arr2<T>& u, const Arr2<T>& v, Arr2<T>& ua, Arr2<T>& va, Arr2<T>& uc, Arr2<T>& vc, Arr2<T>& ut,
                Arr2<T>& vt, const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  Arr2<T> utmp(bd), vtmp(bd);
  utmp.fill(T(1.e30)); vtmp.fill(T(1.e30));  // big_number, :6463-6464
  for (int j = js - 1; j <= je + 1; ++j)       // :6505-6519
    for (int i = isd; i <= ied; ++i)
      utmp(i, j) = a2b_a2 * (u(i, j - 1) + u(i, j + 2)) + a2b_a1 * (u(i, j) + u(i, j + 1));
  for (int j = jsd; j <= jed; ++j)             // :6530-6544
    for (int i = is - 1; i <= ie + 1; ++i)
      vtmp(i, j) = a2b_a2 * (v(i - 1, j) + v(i + 2, j)) + a2b_a1 * (v(i, j) + v(i + 1, j));
  // contra-variant components at cell centres (:6605-6611); only is-1..ie+1 carries defined data
  for (int j = js - 1; j <= je + 1; ++j)
    for (int i = is - 1; i <= ie + 1; ++i) {
      ua(i, j) = (utmp(i, j) - vtmp(i, j) * g.cosa_s(i, j)) * g.rsin2(i, j);
      va(i, j) = (vtmp(i, j) - utmp(i, j) * g.cosa_s(i, j)) * g.rsin2(i, j);
    }
  for (int j = js - 1; j <= je + 1; ++j)       // :6655-6661
    for (int i = is - 1; i <= ie + 2; ++i) {
      uc(i, j) = a2b_a2 * (utmp(i - 2, j) + utmp(i + 1, j)) + a2b_a1 * (utmp(i - 1, j) + utmp(i, j));
      ut(i, j) = (uc(i, j) - v(i, j) * g.cosa_u(i, j)) * g.rsin_u(i, j);
    }
  for (int j = js - 1; j <= je + 2; ++j)       // :6786-6792
    for (int i = is - 1; i <= ie + 1; ++i) {
      vc(i, j) = a2b_a2 * (vtmp(i, j - 2) + vtmp(i, j + 1)) + a2b_a1 * (vtmp(i, j - 1) + vtmp(i, j));
      vt(i, j) = (vc(i, j) - u(i, j) * g.cosa_v(i, j)) * g.rsin_v(i, j);
    }
}

// divergence_corner, sw_core_tlm.F90:3966-4082 (grid_type<3 branch, interior rank).
template <class T>
void divergence_corner(const Arr2<T>& u, const Arr2<T>& v, const Arr2<T>& ua, const Arr2<T>& va, Arr2<T>& divg_d,
                       const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  Arr2<T> uf(bd), vf(bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is - 1; i <= ie + 1; ++i)
      uf(i, j) = (u(i, j) - 0.25 * (va(i, j - 1) + va(i, j)) * (g.cos_sg[4](i, j - 1) + g.cos_sg[2](i, j))) *
                 g.dyc(i, j) * 0.5 * (g.sin_sg[4](i, j - 1) + g.sin_sg[2](i, j));
  for (int j = js - 1; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i)
      vf(i, j) = (v(i, j) - 0.25 * (ua(i - 1, j) + ua(i, j)) * (g.cos_sg[3](i - 1, j) + g.cos_sg[1](i, j))) *
                 g.dxc(i, j) * 0.5 * (g.sin_sg[3](i - 1, j) + g.sin_sg[1](i, j));
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) {
      divg_d(i, j) = vf(i, j - 1) - vf(i, j) + (uf(i - 1, j) - uf(i, j));
      divg_d(i, j) = g.rarea_c(i, j) * divg_d(i, j);
    }
}

// c_sw, hydrostatic, sw_core_tlm.F90:646-1038.  Outputs: delpc, ptc (is-1..ie+1), uc, vc (updated),
// ua, va, ut, vt (flux form after :713-733), divg_d (nord>0).
template <class T>
void c_sw(Arr2<T>& delpc, const Arr2<T>& delp, Arr2<T>& ptc, const Arr2<T>& pt, const Arr2<T>& u, const Arr2<T>& v,
          Arr2<T>& uc, Arr2<T>& vc, Arr2<T>& ua, Arr2<T>& va, Arr2<T>& ut, Arr2<T>& vt, Arr2<T>& divg_d, int nord,
          double dt2, const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, iep1 = ie + 1, jep1 = je + 1;
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
  Arr2<T> fx(bd), fx1(bd), fy(bd), fy1(bd), ke(bd), vort(bd);
  for (int j = js - 1; j <= jep1; ++j)        // :744-757
    for (int i = is - 1; i <= ie + 2; ++i) {
      if (val(ut(i, j)) > 0.) { fx1(i, j) = delp(i - 1, j); fx(i, j) = pt(i - 1, j); }
      else                    { fx1(i, j) = delp(i, j);     fx(i, j) = pt(i, j); }
      fx1(i, j) = ut(i, j) * fx1(i, j);
      fx(i, j) = fx1(i, j) * fx(i, j);
    }
  for (int j = js - 1; j <= jep1 + 1; ++j)    // :788-800
    for (int i = is - 1; i <= iep1; ++i) {
      if (val(vt(i, j)) > 0.) { fy1(i, j) = delp(i, j - 1); fy(i, j) = pt(i, j - 1); }
      else                    { fy1(i, j) = delp(i, j);     fy(i, j) = pt(i, j); }
      fy1(i, j) = vt(i, j) * fy1(i, j);
      fy(i, j) = fy1(i, j) * fy(i, j);
    }
  for (int j = js - 1; j <= jep1; ++j)        // :801-808
    for (int i = is - 1; i <= iep1; ++i) {
      delpc(i, j) = delp(i, j) + (fx1(i, j) - fx1(i + 1, j) + (fy1(i, j) - fy1(i, j + 1))) * g.rarea(i, j);
      ptc(i, j) = (pt(i, j) * delp(i, j) + (fx(i, j) - fx(i + 1, j) + (fy(i, j) - fy(i, j + 1))) * g.rarea(i, j)) /
                  delpc(i, j);
    }
  // KE: upstream C-grid wind; interior branches of :870-917
  for (int j = js - 1; j <= jep1; ++j)
    for (int i = is - 1; i <= iep1; ++i) {
      ke(i, j) = (val(ua(i, j)) > 0.) ? uc(i, j) : uc(i + 1, j);
      vort(i, j) = (val(va(i, j)) > 0.) ? vc(i, j) : vc(i, j + 1);
    }
  const double dt4 = 0.5 * dt2;
  for (int j = js - 1; j <= jep1; ++j)
    for (int i = is - 1; i <= iep1; ++i) ke(i, j) = dt4 * (ua(i, j) * ke(i, j) + va(i, j) * vort(i, j));
  for (int j = js - 1; j <= je + 1; ++j)      // :929-943
    for (int i = is; i <= ie + 1; ++i) fx(i, j) = uc(i, j) * g.dxc(i, j);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is - 1; i <= ie + 1; ++i) fy(i, j) = vc(i, j) * g.dyc(i, j);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) {
      vort(i, j) = fx(i, j - 1) - fx(i, j) + (fy(i, j) - fy(i - 1, j));
      vort(i, j) = g.fC(i, j) + g.rarea_c(i, j) * vort(i, j);   // :952-957
    }
  for (int j = js; j <= je; ++j)              // :991-1003 (interior: i != 1, npx)
    for (int i = is; i <= iep1; ++i) {
      fy1(i, j) = dt2 * (v(i, j) - uc(i, j) * g.cosa_u(i, j)) / g.sina_u(i, j);
      fy(i, j) = (val(fy1(i, j)) > 0.) ? vort(i, j) : vort(i, j + 1);
    }
  for (int j = js; j <= jep1; ++j)            // :1015-1023
    for (int i = is; i <= ie; ++i) {
      fx1(i, j) = dt2 * (u(i, j) - vc(i, j) * g.cosa_v(i, j)) / g.sina_v(i, j);
      fx(i, j) = (val(fx1(i, j)) > 0.) ? vort(i, j) : vort(i + 1, j);
    }
  for (int j = js; j <= je; ++j)              // :1027-1031
    for (int i = is; i <= iep1; ++i)
      uc(i, j) = uc(i, j) + fy1(i, j) * fy(i, j) + g.rdxc(i, j) * (ke(i - 1, j) - ke(i, j));
  for (int j = js; j <= jep1; ++j)            // :1032-1037
    for (int i = is; i <= ie; ++i)
      vc(i, j) = vc(i, j) - fx1(i, j) * fx(i, j) + g.rdyc(i, j) * (ke(i, j - 1) - ke(i, j));
}

// xtp_u: x-transport of u on B-grid points, flux on (is..ie+1, js..je+1).  _TLM :7272-7486.
template <class T>
void xtp_u(const Arr2<T>& c, const Arr2<T>& u, Arr2<T>& flux, int iord, const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  assert(iord == 1 || iord == 2 || iord == 333);
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  for (int j = js; j <= je + 1; ++j) {
    if (iord == 1) {
      for (int i = is; i <= ie + 1; ++i) flux(i, j) = (val(c(i, j)) > 0.) ? u(i - 1, j) : u(i, j);
    } else if (iord == 333) {
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.)
          flux(i, j) = (2.0 * u(i, j) + 5.0 * u(i - 1, j) - u(i - 2, j)) / 6.0 -
                       0.5 * c(i, j) * g.rdx(i - 1, j) * (u(i, j) - u(i - 1, j)) +
                       c(i, j) * g.rdx(i - 1, j) * c(i, j) * g.rdx(i - 1, j) / 6.0 * (u(i, j) - 2.0 * u(i - 1, j) + u(i - 2, j));
        else
          flux(i, j) = (2.0 * u(i - 1, j) + 5.0 * u(i, j) - u(i + 1, j)) / 6.0 -
                       0.5 * c(i, j) * g.rdx(i, j) * (u(i, j) - u(i - 1, j)) +
                       c(i, j) * g.rdx(i, j) * c(i, j) * g.rdx(i, j) / 6.0 * (u(i + 1, j) - 2.0 * u(i, j) + u(i - 1, j));
      }
    } else {
      const int is3 = is - 1, ie3 = ie + 1;
      std::vector<T> al(ie3 + 1 - is3 + 1), bl(ie3 - is3 + 1), br(ie3 - is3 + 1), b0(ie3 - is3 + 1);
      for (int i = is3; i <= ie3 + 1; ++i)
        al[i - is3] = ppm_p1 * (u(i - 1, j) + u(i, j)) + ppm_p2 * (u(i - 2, j) + u(i + 1, j));
      for (int i = is3; i <= ie3; ++i) {
        bl[i - is3] = al[i - is3] - u(i, j);
        br[i - is3] = al[i + 1 - is3] - u(i, j);
        b0[i - is3] = bl[i - is3] + br[i - is3];
      }
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.) {
          T cfl = c(i, j) * g.rdx(i - 1, j);
          flux(i, j) = u(i - 1, j) + (1. - cfl) * (br[i - 1 - is3] - cfl * b0[i - 1 - is3]);
        } else {
          T cfl = c(i, j) * g.rdx(i, j);
          flux(i, j) = u(i, j) + (1. + cfl) * (bl[i - is3] + cfl * b0[i - is3]);
        }
      }
    }
  }
}

// ytp_v: _TLM :7490-7759.
template <class T>
void ytp_v(const Arr2<T>& c, const Arr2<T>& v, Arr2<T>& flux, int jord, const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  assert(jord == 1 || jord == 2 || jord == 333);
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  if (jord == 1) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) flux(i, j) = (val(c(i, j)) > 0.) ? v(i, j - 1) : v(i, j);
  } else if (jord == 333) {
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        if (val(c(i, j)) > 0.)
          flux(i, j) = (2.0 * v(i, j) + 5.0 * v(i, j - 1) - v(i, j - 2)) / 6.0 -
                       0.5 * c(i, j) * g.rdy(i, j - 1) * (v(i, j) - v(i, j - 1)) +
                       c(i, j) * g.rdy(i, j - 1) * c(i, j) * g.rdy(i, j - 1) / 6.0 * (v(i, j) - 2.0 * v(i, j - 1) + v(i, j - 2));
        else
          flux(i, j) = (2.0 * v(i, j - 1) + 5.0 * v(i, j) - v(i, j + 1)) / 6.0 -
                       0.5 * c(i, j) * g.rdy(i, j) * (v(i, j) - v(i, j - 1)) +
                       c(i, j) * g.rdy(i, j) * c(i, j) * g.rdy(i, j) / 6.0 * (v(i, j + 1) - 2.0 * v(i, j) + v(i, j - 1));
      }
  } else {
    const int js3 = js - 1, je3 = je + 1;
    Arr2<T> al(bd), bl(bd), br(bd), b0(bd);
    for (int j = js3; j <= je3 + 1; ++j)
      for (int i = is; i <= ie + 1; ++i)
        al(i, j) = ppm_p1 * (v(i, j - 1) + v(i, j)) + ppm_p2 * (v(i, j - 2) + v(i, j + 1));
    for (int j = js3; j <= je3; ++j)
      for (int i = is; i <= ie + 1; ++i) {
        bl(i, j) = al(i, j) - v(i, j);
        br(i, j) = al(i, j + 1) - v(i, j);
        b0(i, j) = bl(i, j) + br(i, j);
      }
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
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  for (int j = js - 1 - nord; j <= je + 1 + nord; ++j)
    for (int i = is - 1 - nord; i <= ie + 1 + nord; ++i) d2(i, j) = damp * q(i, j);
  for (int j = js - nord; j <= je + nord; ++j)
    for (int i = is - nord; i <= ie + nord + 1; ++i) fx2(i, j) = g.del6_v(i, j) * (d2(i - 1, j) - d2(i, j));
  for (int j = js - nord; j <= je + nord + 1; ++j)
    for (int i = is - nord; i <= ie + nord; ++i) fy2(i, j) = g.del6_u(i, j) * (d2(i, j - 1) - d2(i, j));
  for (int n = 1; n <= nord; ++n) {
    const int nt = nord - n;
    for (int j = js - nt - 1; j <= je + nt + 1; ++j)
      for (int i = is - nt - 1; i <= ie + nt + 1; ++i)
        d2(i, j) = (fx2(i, j) - fx2(i + 1, j) + (fy2(i, j) - fy2(i, j + 1))) * g.rarea(i, j);
    for (int j = js - nt; j <= je + nt; ++j)
      for (int i = is - nt; i <= ie + nt + 1; ++i) fx2(i, j) = g.del6_v(i, j) * (d2(i, j) - d2(i - 1, j));
    for (int j = js - nt; j <= je + nt + 1; ++j)
      for (int i = is - nt; i <= ie + nt; ++i) fy2(i, j) = g.del6_u(i, j) * (d2(i, j) - d2(i, j - 1));
  }
}

// compute_divergence_damping, sw_core_tlm.F90:7760-8072 (_TLM :8178-8598), grid_type<3, not
// stretched.  In: u,v,ua,va,divg_d (halo'd corner field, modified in place when nord>0),
// wk (relative vorticity).  Out: vort (damping term), ke += vort; delpc, ptc, uc, vc are work.
template <class T>
void compute_divergence_damping(int nord, double d2_bg, double d4_bg, double dddmp, double dt, Arr2<T>& vort,
                                Arr2<T>& ptc, Arr2<T>& delpc, Arr2<T>& ke, const Arr2<T>& u, const Arr2<T>& v,
                                Arr2<T>& uc, Arr2<T>& vc, const Arr2<T>& ua, const Arr2<T>& va, Arr2<T>& divg_d,
                                Arr2<T>& wk, const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je;
  const double absdt = dt >= 0. ? dt : -dt;
  if (nord == 0) {   // :7874-7957
    for (int j = js; j <= je + 1; ++j)
      for (int i = is - 1; i <= ie + 1; ++i)
        ptc(i, j) = (u(i, j) - 0.5 * (va(i, j - 1) + va(i, j)) * g.cosa_v(i, j)) * g.dyc(i, j) * g.sina_v(i, j);
    for (int j = js - 1; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i)
        vort(i, j) = (v(i, j) - 0.5 * (ua(i - 1, j) + ua(i, j)) * g.cosa_u(i, j)) * g.dxc(i, j) * g.sina_u(i, j);
    for (int j = js; j <= je + 1; ++j)
      for (int i = is; i <= ie + 1; ++i) delpc(i, j) = vort(i, j - 1) - vort(i, j) + ptc(i - 1, j) - ptc(i, j);
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
      for (int j = js - nt; j <= je + 1 + nt; ++j)
        for (int i = is - 1 - nt; i <= ie + 1 + nt; ++i) vc(i, j) = (divg_d(i + 1, j) - divg_d(i, j)) * g.divg_u(i, j);
      for (int j = js - 1 - nt; j <= je + 1 + nt; ++j)
        for (int i = is - nt; i <= ie + 1 + nt; ++i) uc(i, j) = (divg_d(i, j + 1) - divg_d(i, j)) * g.divg_v(i, j);
      for (int j = js - nt; j <= je + 1 + nt; ++j)
        for (int i = is - nt; i <= ie + 1 + nt; ++i) {
          divg_d(i, j) = uc(i, j - 1) - uc(i, j) + vc(i - 1, j) - vc(i, j);
          divg_d(i, j) = divg_d(i, j) * g.rarea_c(i, j);
        }
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

// d_sw, hydrostatic, inline_q=.false., d_con<=1e-5, grid_type<3, interior rank.
// sw_core_tlm.F90:2533-3617 (_TLM :1047-2531).  In/out: delp, pt, u, v (updated on the compute
// domain), xflux,yflux,cx,cy accumulators; in: uc, vc, ua, va, divg_d; out: crx_adv.. yfx_adv.
template <class T>
void d_sw(Arr2<T>& delp, Arr2<T>& pt, Arr2<T>& u, Arr2<T>& v, Arr2<T>& uc, Arr2<T>& vc, const Arr2<T>& ua,
          const Arr2<T>& va, Arr2<T>& divg_d, Arr2<T>& xflux, Arr2<T>& yflux, Arr2<T>& cx, Arr2<T>& cy,
          Arr2<T>& crx_adv, Arr2<T>& cry_adv, Arr2<T>& xfx_adv, Arr2<T>& yfx_adv, double dt, const LevelParams& lp,
          double dddmp, double d4_bg, const Grid& g, const Bounds& bd) {
  assert(!bd.any_edge());
  const int is = bd.is, ie = bd.ie, js = bd.js, je = bd.je, isd = bd.isd, ied = bd.ied, jsd = bd.jsd, jed = bd.jed;
  Arr2<T> ut(bd), vt(bd), ra_x(bd), ra_y(bd), fx(bd), fy(bd), gx(bd), gy(bd), ub(bd), vb(bd), ke(bd), wk(bd),
      vort(bd), ptc(bd), delpc(bd), fx2(bd), fy2(bd);
  // contra-variant winds, interior rows (:2722-2738)
  for (int j = jsd; j <= jed; ++j)
    for (int i = is - 1; i <= ie + 2; ++i)
      ut(i, j) = (uc(i, j) - 0.25 * g.cosa_u(i, j) * (vc(i - 1, j) + vc(i, j) + vc(i - 1, j + 1) + vc(i, j + 1))) *
                 g.rsin_u(i, j);
  for (int j = js - 1; j <= je + 2; ++j)
    for (int i = isd; i <= ied; ++i)
      vt(i, j) = (vc(i, j) - 0.25 * g.cosa_v(i, j) * (uc(i, j - 1) + uc(i + 1, j - 1) + uc(i, j) + uc(i + 1, j))) *
                 g.rsin_v(i, j);
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
  fv_tp_2d<T>(delp, crx_adv, cry_adv, lp.hord_dp, fx, fy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, nullptr, nullptr,
              nullptr, lp.nord_v, lp.damp_vt);
  for (int j = jsd; j <= jed; ++j)            // flux capacitors :2988-3006
    for (int i = is; i <= ie + 1; ++i) cx(i, j) = cx(i, j) + crx_adv(i, j);
  for (int j = js; j <= je; ++j)
    for (int i = is; i <= ie + 1; ++i) xflux(i, j) = xflux(i, j) + fx(i, j);
  for (int j = js; j <= je + 1; ++j) {
    for (int i = isd; i <= ied; ++i) cy(i, j) = cy(i, j) + cry_adv(i, j);
    for (int i = is; i <= ie; ++i) yflux(i, j) = yflux(i, j) + fy(i, j);
  }
  // pt transport (:3064-3072): mass=delp, nord_t, damp_t
  fv_tp_2d<T>(pt, crx_adv, cry_adv, lp.hord_tm, gx, gy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, &fx, &fy, &delp,
              lp.nord_t, lp.damp_t);
  for (int j = js; j <= je; ++j)              // :3107-3116
    for (int i = is; i <= ie; ++i) {
      pt(i, j) = pt(i, j) * delp(i, j) + (gx(i, j) - gx(i + 1, j) + (gy(i, j) - gy(i, j + 1))) * g.rarea(i, j);
      delp(i, j) = delp(i, j) + (fx(i, j) - fx(i + 1, j) + (fy(i, j) - fy(i, j + 1))) * g.rarea(i, j);
      pt(i, j) = pt(i, j) / delp(i, j);
    }
  // kinetic-energy fluxes (:3126-3254)
  const double dt5 = 0.5 * dt;
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i)
      vb(i, j) = dt5 * (vc(i - 1, j) + vc(i, j) - (uc(i, j - 1) + uc(i, j)) * g.cosa(i, j)) * g.rsina(i, j);
  ytp_v(vb, v, ub, lp.hord_mt, g, bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) ke(i, j) = vb(i, j) * ub(i, j);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i)
      ub(i, j) = dt5 * (uc(i, j - 1) + uc(i, j) - (vc(i - 1, j) + vc(i, j)) * g.cosa(i, j)) * g.rsina(i, j);
  xtp_u(ub, u, vb, lp.hord_mt, g, bd);
  for (int j = js; j <= je + 1; ++j)
    for (int i = is; i <= ie + 1; ++i) ke(i, j) = 0.5 * (ke(i, j) + ub(i, j) * vb(i, j));
  // vorticity (:3275-3293); vt/ut are reused as u*dx, v*dy like the reference
  for (int j = jsd; j <= jed + 1; ++j)
    for (int i = isd; i <= ied; ++i) vt(i, j) = u(i, j) * g.dx(i, j);
  for (int j = jsd; j <= jed; ++j)
    for (int i = isd; i <= ied + 1; ++i) ut(i, j) = v(i, j) * g.dy(i, j);
  for (int j = jsd; j <= jed; ++j)
    for (int i = isd; i <= ied; ++i)
      wk(i, j) = g.rarea(i, j) * (vt(i, j) - vt(i, j + 1) + (ut(i + 1, j) - ut(i, j)));
  compute_divergence_damping(lp.nord, lp.d2_divg, d4_bg, dddmp, dt, vort, ptc, delpc, ke, u, v, uc, vc, ua, va,
                             divg_d, wk, g, bd);
  for (int j = jsd; j <= jed; ++j)            // :3535-3540 hydrostatic
    for (int i = isd; i <= ied; ++i) vort(i, j) = wk(i, j) + g.f0(i, j);
  fv_tp_2d<T>(vort, crx_adv, cry_adv, lp.hord_vt, fx, fy, xfx_adv, yfx_adv, g, bd, ra_x, ra_y, nullptr, nullptr,
              nullptr, -1, 0.0);
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
