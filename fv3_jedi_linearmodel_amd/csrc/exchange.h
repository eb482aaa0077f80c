// fv3lm-hip: halo exchange between the resident cube faces — the device-side stand-in for
// mpp_update_domains / mpp_get_boundary (called at dyn_core_tlm.F90:1744-1790, :1960-1990, :2280-2300,
// :2418-2434 and fv_dynamics_tlm.F90:646-651, :708-712; the data motion itself lives inside FMS).
//
// An exchange is table driven.  A row (dst_field, dst_tile, dst_index, src_field, src_tile, src_index, sign)
// says: halo element dst takes sign * the value of element src — one or two co-located fields per
// exchange (a scalar, or the two components of a staggered vector, which swap roles and change sign
// across rotated face contacts).  The tables come from the host (fv3lm_set_exchange; cube.py derives them
// from the face frames), so the kernels know nothing about the topology.
//   nonlinear / tangent: one thread per (row, level), pure gather.
//   adjoint:             rows grouped by source element (CSR built once on the host): one thread per
//                        (source, level) sums sign * halo adjoints in a fixed order and clears them —
//                        no atomics, bitwise reproducible.
#pragma once
#include "exec.h"
#include <algorithm>
#include <vector>

namespace fv3 {

enum HaloKind { H_CELL = 0, H_DVEC = 1, H_CVEC = 2, H_CORNER = 3, H_DEDGE = 4, H_NKIND = 5 };

struct ExTable {
  int n = 0;            // rows
  int* rows = nullptr;  // device [n][7]
  int ns = 0;           // distinct sources
  int* src = nullptr;   // device [ns][3]  field, tile, index
  int* ptr = nullptr;   // device [ns+1]
  int* dst = nullptr;   // device [n][4]   field, tile, index, sign
};

struct ExFwdFn {
  Fld f0, f1; const int* rows; int plane, mode;
  HD void operator()(int r, int, int k) const {
    const int* w = rows + 7 * (size_t)r;
    const Fld& fd = w[0] ? f1 : f0; const Fld& fs = w[3] ? f1 : f0;
    const size_t d = ((size_t)w[1] * fd.nk + k) * plane + w[2], s = ((size_t)w[4] * fs.nk + k) * plane + w[5];
    const double sg = (double)w[6];
    fd.t[d] = sg * fs.t[s];
    if (mode == MODE_TL) fd.p[d] = sg * fs.p[s];
  }
};
struct ExAdFn {
  Fld f0, f1; const int* src; const int* ptr; const int* dst; int plane;
  HD void operator()(int n, int, int k) const {
    const int* s = src + 3 * (size_t)n;
    const Fld& fs = s[0] ? f1 : f0;
    double acc = 0.;
    for (int m = ptr[n]; m < ptr[n + 1]; ++m) {
      const int* w = dst + 4 * (size_t)m;
      const Fld& fd = w[0] ? f1 : f0;
      const size_t d = ((size_t)w[1] * fd.nk + k) * plane + w[2];
      acc += (double)w[3] * fd.p[d];
      fd.p[d] = 0.;
    }
    fs.p[((size_t)s[1] * fs.nk + k) * plane + s[2]] += acc;
  }
};

inline void run_exchange(Exec& ex, int mode, const Geom& g, const ExTable& t, const Fld& f0, const Fld& f1) {
  if (t.n == 0) return;
  const double bytes = 8. * 2. * t.n * f0.nk * (mode == MODE_NL ? 1. : mode == MODE_TL ? 2. : 1.);
  if (mode == MODE_AD) for_points(ex, Rect{0, t.ns - 1, 0, 0}, f0.nk, ExAdFn{f0, f1, t.src, t.ptr, t.dst, g.plane}, "exchange.ad", bytes);
  else for_points(ex, Rect{0, t.n - 1, 0, 0}, f0.nk, ExFwdFn{f0, f1, t.rows, g.plane, mode}, mode == MODE_TL ? "exchange.tl" : "exchange.nl", bytes);
}

// host side: validate and group by source
inline bool build_extable_host(const int* rows, int n, int ntile, int plane, std::vector<int>& src, std::vector<int>& ptr, std::vector<int>& dst,
                               const char** why) {
  std::vector<int> order(n);
  for (int r = 0; r < n; ++r) {
    const int* w = rows + 7 * (size_t)r;
    if ((w[0] | 1) != 1 || (w[3] | 1) != 1 || w[1] < 0 || w[1] >= ntile || w[4] < 0 || w[4] >= ntile || w[2] < 0 || w[2] >= plane || w[5] < 0 ||
        w[5] >= plane || (w[6] != 1 && w[6] != -1)) { *why = "exchange table row out of range"; return false; }
    order[r] = r;
  }
  auto key = [&](int r) { const int* w = rows + 7 * (size_t)r; return ((long long)w[3] << 48) | ((long long)w[4] << 32) | (long long)w[5]; };
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key(a) < key(b); });
  src.clear(); ptr.clear(); dst.clear();
  long long last = -1;
  for (int m = 0; m < n; ++m) {
    const int* w = rows + 7 * (size_t)order[m];
    if (key(order[m]) != last) { last = key(order[m]); ptr.push_back(m); src.push_back(w[3]); src.push_back(w[4]); src.push_back(w[5]); }
    dst.push_back(w[0]); dst.push_back(w[1]); dst.push_back(w[2]); dst.push_back(w[6]);
  }
  ptr.push_back(n);
  return true;
}

}  // namespace fv3
