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
//
// Faces on other GPUs (one process per GPU): the rows whose source face lives on another rank go through per-peer
// send/receive lists built by the host from the same global table (cube.split_tables).  Forward: pack kernel ->
// grouped ncclSend/ncclRecv on the library's own stream (RCCL over xGMI, point-to-point, no collective) -> unpack
// kernel.  Adjoint: the halo adjoints travel the other way and are summed per source element in a fixed order.
// RCCL is reached through dlopen of the library the host names (the one torch already loaded, so that a process
// never runs two RCCL runtimes); the test-only host emulation uses a caller-supplied transport callback instead.
#pragma once
#include "exec.h"
#include <algorithm>
#include <string>
#include <vector>
#ifndef FV3LM_HOST_EMUL
#include <dlfcn.h>
#include <rccl/rccl.h>
#endif

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

// ---------------------------------------------------------------- faces on other ranks
typedef void (*fv3lm_transport_fn)(void* user, int npeers, const int* peer_ranks, double* const* sendbufs, const long* send_counts,
                                   double* const* recvbufs, const long* recv_counts);
struct Transport {
  fv3lm_transport_fn cb = nullptr; void* user = nullptr;   // host-emulation tests (gloo) or a host-staged transport
  int rank = 0, nranks = 1;
#ifndef FV3LM_HOST_EMUL
  void* lib = nullptr; ncclComm_t comm = nullptr;
  ncclResult_t (*pGetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*pCommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*pCommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*pSend)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*pRecv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*pGroupStart)() = nullptr;
  ncclResult_t (*pGroupEnd)() = nullptr;
  ncclResult_t (*pAllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  bool load(const char* path, std::string& err) {
    if (lib) return true;
    lib = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { err = std::string("dlopen of RCCL failed: ") + dlerror(); return false; }
    pGetUniqueId = (decltype(pGetUniqueId))dlsym(lib, "ncclGetUniqueId"); pCommInitRank = (decltype(pCommInitRank))dlsym(lib, "ncclCommInitRank");
    pCommDestroy = (decltype(pCommDestroy))dlsym(lib, "ncclCommDestroy"); pSend = (decltype(pSend))dlsym(lib, "ncclSend");
    pRecv = (decltype(pRecv))dlsym(lib, "ncclRecv"); pGroupStart = (decltype(pGroupStart))dlsym(lib, "ncclGroupStart");
    pGroupEnd = (decltype(pGroupEnd))dlsym(lib, "ncclGroupEnd"); pAllReduce = (decltype(pAllReduce))dlsym(lib, "ncclAllReduce");
    if (!pGetUniqueId || !pCommInitRank || !pSend || !pRecv || !pGroupStart || !pGroupEnd || !pAllReduce) { err = "RCCL symbols missing"; return false; }
    return true;
  }
#endif
};
inline Transport& transport() { static Transport t; return t; }

struct ExRemote {
  bool active = false;
  std::vector<int> peer, nsend, nrecv;      // per peer: rank, rows this rank sends / receives in the forward direction
  std::vector<long> soff, roff;             // row offsets of each peer in the concatenated lists
  int nsend_tot = 0, nrecv_tot = 0;
  int* send_rows = nullptr;                 // device [nsend_tot][3]  field, local tile, index   (forward sources)
  int* recv_rows = nullptr;                 // device [nrecv_tot][4]  field, local tile, index, sign  (forward halo targets)
  int ns = 0; int* asrc = nullptr; int* aptr = nullptr; int* apos = nullptr;   // adjoint: per distinct source, the send rows that carry it
  double *sendbuf = nullptr, *recvbuf = nullptr; size_t cap = 0;              // doubles per buffer
};

struct ExPackFn {      // forward: gather the sources other ranks need
  Fld f0, f1; const int* rows; double* buf; int plane, nk, c;
  HD void operator()(int r, int, int k) const {
    const int* w = rows + 3 * (size_t)r; const Fld& f = w[0] ? f1 : f0;
    const size_t s = ((size_t)w[1] * f.nk + k) * plane + w[2], b = ((size_t)r * nk + k) * c;
    buf[b] = f.t[s];
    if (c == 2) buf[b + 1] = f.p[s];
  }
};
struct ExUnpackFn {    // forward: received values into the halo
  Fld f0, f1; const int* rows; const double* buf; int plane, nk, c;
  HD void operator()(int r, int, int k) const {
    const int* w = rows + 4 * (size_t)r; const Fld& f = w[0] ? f1 : f0;
    const size_t d = ((size_t)w[1] * f.nk + k) * plane + w[2], b = ((size_t)r * nk + k) * c;
    f.t[d] = (double)w[3] * buf[b];
    if (c == 2) f.p[d] = (double)w[3] * buf[b + 1];
  }
};
struct ExPackAdFn {    // adjoint: halo adjoints leave (and are cleared)
  Fld f0, f1; const int* rows; double* buf; int plane, nk;
  HD void operator()(int r, int, int k) const {
    const int* w = rows + 4 * (size_t)r; const Fld& f = w[0] ? f1 : f0;
    const size_t d = ((size_t)w[1] * f.nk + k) * plane + w[2];
    buf[(size_t)r * nk + k] = (double)w[3] * f.p[d];
    f.p[d] = 0.;
  }
};
struct ExUnpackAdFn {  // adjoint: every source element sums what came back for it, fixed order
  Fld f0, f1; const int* src; const int* ptr; const int* pos; const double* buf; int plane, nk;
  HD void operator()(int n, int, int k) const {
    const int* s = src + 3 * (size_t)n; const Fld& f = s[0] ? f1 : f0;
    double acc = 0.;
    for (int m = ptr[n]; m < ptr[n + 1]; ++m) acc += buf[(size_t)pos[m] * nk + k];
    f.p[((size_t)s[1] * f.nk + k) * plane + s[2]] += acc;
  }
};

// one message pair per peer: rows * per_row doubles each way
inline bool run_transport(Exec& ex, const ExRemote& x, bool reverse, size_t per_row, std::string& err) {
  Transport& T = transport();
  const int np = (int)x.peer.size();
  std::vector<double*> sb(np), rb(np); std::vector<long> sc(np), rc(np);
  for (int p = 0; p < np; ++p) {
    if (!reverse) { sb[p] = x.sendbuf + x.soff[p] * per_row; sc[p] = (long)x.nsend[p] * per_row; rb[p] = x.recvbuf + x.roff[p] * per_row; rc[p] = (long)x.nrecv[p] * per_row; }
    else          { sb[p] = x.recvbuf + x.roff[p] * per_row; sc[p] = (long)x.nrecv[p] * per_row; rb[p] = x.sendbuf + x.soff[p] * per_row; rc[p] = (long)x.nsend[p] * per_row; }
  }
  if (T.cb) {
#ifndef FV3LM_HOST_EMUL
    (void)hipStreamSynchronize(ex.stream);
#endif
    T.cb(T.user, np, x.peer.data(), sb.data(), sc.data(), rb.data(), rc.data());
    return true;
  }
#ifndef FV3LM_HOST_EMUL
  if (!T.comm) { err = "halo exchange with another rank needed before fv3lm_comm_init"; return false; }
  // every RCCL result is checked: a failed send or receive must not leave the halo silently stale
  ncclResult_t rc_ = T.pGroupStart();
  if (rc_ != ncclSuccess) { err = "ncclGroupStart failed (code " + std::to_string((int)rc_) + ")"; return false; }
  bool ok = true; int bad = 0;
  for (int p = 0; p < np; ++p) {
    if (sc[p]) { const ncclResult_t r = T.pSend(sb[p], (size_t)sc[p], ncclDouble, x.peer[p], T.comm, ex.stream); if (r != ncclSuccess) { ok = false; bad = (int)r; } }
    if (rc[p]) { const ncclResult_t r = T.pRecv(rb[p], (size_t)rc[p], ncclDouble, x.peer[p], T.comm, ex.stream); if (r != ncclSuccess) { ok = false; bad = (int)r; } }
  }
  rc_ = T.pGroupEnd();        // always closed, also after a failed call inside the group
  if (!ok) { err = "ncclSend/ncclRecv failed (code " + std::to_string(bad) + ")"; return false; }
  if (rc_ != ncclSuccess) { err = "ncclGroupEnd failed (code " + std::to_string((int)rc_) + ")"; return false; }
  return true;
#else
  err = "halo exchange with another rank needed but no transport callback is set";
  return false;
#endif
}

inline bool run_exchange_remote(Exec& ex, int mode, const Geom& g, ExRemote& x, const Fld& f0, const Fld& f1, std::string& err) {
  if (!x.active) return true;
  const int nk = f0.nk, c = (mode == MODE_TL) ? 2 : 1;
  const size_t need = (size_t)std::max(x.nsend_tot, x.nrecv_tot) * nk * c;
  if (need > x.cap) { err = "exchange buffers too small"; return false; }
  if (mode != MODE_AD) {
    if (x.nsend_tot) for_points(ex, Rect{0, x.nsend_tot - 1, 0, 0}, nk, ExPackFn{f0, f1, x.send_rows, x.sendbuf, g.plane, nk, c}, "exchange.pack", 8. * 2 * x.nsend_tot * nk * c);
    if (!run_transport(ex, x, false, (size_t)nk * c, err)) return false;
    if (x.nrecv_tot) for_points(ex, Rect{0, x.nrecv_tot - 1, 0, 0}, nk, ExUnpackFn{f0, f1, x.recv_rows, x.recvbuf, g.plane, nk, c}, "exchange.unpack", 8. * 2 * x.nrecv_tot * nk * c);
  } else {
    if (x.nrecv_tot) for_points(ex, Rect{0, x.nrecv_tot - 1, 0, 0}, nk, ExPackAdFn{f0, f1, x.recv_rows, x.recvbuf, g.plane, nk}, "exchange.pack_ad", 8. * 2 * x.nrecv_tot * nk);
    if (!run_transport(ex, x, true, (size_t)nk, err)) return false;
    if (x.ns) for_points(ex, Rect{0, x.ns - 1, 0, 0}, nk, ExUnpackAdFn{f0, f1, x.asrc, x.aptr, x.apos, x.sendbuf, g.plane, nk}, "exchange.unpack_ad", 8. * 2 * x.nsend_tot * nk);
  }
  return true;
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
