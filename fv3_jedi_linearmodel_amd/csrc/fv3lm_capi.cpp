// fv3lm-hip: implementation of the C-ABI declared in include/fv3lm.h.
// Built with `hipcc -x hip --offload-arch=gfx950` into libfv3lm_hip.so (the product), and with
// `g++ -DFV3LM_HOST_EMUL` into tests/_emul/libfv3lm_emul.so (test-only host emulation, never
// loaded by the package).
#include "dynamics.h"
#include <array>
#include <string>

using namespace fv3;

struct fv3lm_handle { Dynamics d; };

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return 1; }
static int g_live = 0;      // handles alive in this process (one per GPU / MPI rank in production; tests create several)

extern "C" {

const char* fv3lm_last_error(void) { return g_err.c_str(); }

const char* fv3lm_metric_names(void) {
  return "area,rarea,rarea_c,dx,dy,dxa,dya,dxc,dyc,rdx,rdy,rdxa,rdya,rdxc,rdyc,cosa,sina,rsina,cosa_u,cosa_v,cosa_s,"
         "sina_u,sina_v,rsin_u,rsin_v,rsin2,f0,fC,del6_u,del6_v,divg_u,divg_v,"
         "sin_sg1,sin_sg2,sin_sg3,sin_sg4,sin_sg5,sin_sg6,sin_sg7,sin_sg8,sin_sg9,"
         "cos_sg1,cos_sg2,cos_sg3,cos_sg4,cos_sg5,cos_sg6,cos_sg7,cos_sg8,cos_sg9";
}

int fv3lm_create(fv3lm_handle** out, const fv3lm_dims* dm, const fv3lm_options* opt, const double* const* metrics,
                 double da_min, double da_min_c, const double* phis, const double* ak, const double* bk) {
  if (!out || !dm || !opt || !metrics) return fail("fv3lm_create: null argument");
#ifndef FV3LM_HOST_EMUL
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail("fv3lm_create: no HIP device visible (this library has no CPU fallback)");
#endif
  // The failure record is per process (exec.h sticky_error): a fault of a LIVE handle that no call has reported yet must not be erased by
  // the creation of another one -- that create fails with it instead; with no handle alive there is nobody left to report it to.
  if (!sticky_error().empty()) {
    if (g_live > 0) return fail("fv3lm_create: an earlier failure of a live handle is pending: " + sticky_error());
    sticky_error().clear();
  }
  fv3lm_handle* h = new fv3lm_handle;
  // every failure path releases what was allocated so far (destroy2 / destroy accept a partly built object); the failure is reported here, once
  auto bail = [&](std::string e) { h->d.destroy2(); h->d.destroy(); delete h; sticky_error().clear(); return fail("fv3lm_create: " + e); };   // e by value: it may live in *h
  if (!h->d.init(dm->nx, dm->ny, dm->npz, dm->ntile, dm->face, dm->nq, dm->dt, dm->n_split, dm->k_split, *opt, metrics, da_min,
                 da_min_c, phis, dm->nface, dm->tile_ij0)) return bail(h->d.err);
  if (!sticky_error().empty()) return bail(sticky_error());
  if (!h->d.init2(ak, bk)) return bail(h->d.err);
  if (!sticky_error().empty()) return bail(sticky_error());
  *out = h;
  ++g_live;
  return 0;
}

int fv3lm_destroy(fv3lm_handle* h) {
  if (!h) return 0;
  h->d.destroy2(); h->d.destroy(); delete h;
  if (--g_live <= 0) { g_live = 0; sticky_error().clear(); }      // the last handle takes its unreported failures with it
  return 0;
}

static bool find(fv3lm_handle* h, const char* name, Fld& f) {
  if (std::string(name) == "phis") { f = Fld{}; f.t = h->d.hs_dev; f.p = h->d.hs_dev; f.nk = 1; return true; }      // surface geopotential, one padded plane per tile (trajectory only)
  auto it = h->d.F.find(name);
  if (it == h->d.F.end()) { g_err = std::string("unknown field ") + name; return false; }
  f = it->second; return true;
}
int fv3lm_field_levels(fv3lm_handle* h, const char* name) { Fld f; return find(h, name, f) ? f.nk : -1; }
int fv3lm_field_put(fv3lm_handle* h, const char* name, int which, const double* host) {
  Fld f; if (!find(h, name, f)) return 1;
  h2d(h->d.ex, which ? f.p : f.t, host, (size_t)h->d.g.ntile * f.nk * h->d.g.plane * 8);
  return sticky_error().empty() ? 0 : fail(sticky_error());
}
int fv3lm_field_get(fv3lm_handle* h, const char* name, int which, double* host) {
  Fld f; if (!find(h, name, f)) return 1;
  d2h(h->d.ex, host, which ? f.p : f.t, (size_t)h->d.g.ntile * f.nk * h->d.g.plane * 8);
  return sticky_error().empty() ? 0 : fail(sticky_error());
}
// sticky conditions a sweep may have run into
static int status(fv3lm_handle* h) {
#ifndef FV3LM_HOST_EMUL
  { const hipError_t e = hipGetLastError();      // launch-time failures (bad configuration, out of resources) of the kernels queued so far
    if (e != hipSuccess) set_sticky(std::string("HIP kernel launch failed: ") + hipGetErrorString(e)); }
  if (const char* dbg = std::getenv("FV3LM_DEBUG_SYNC")) if (dbg[0] == '1') {   // debug runs: execution-time faults surface at the call that caused them
    const hipError_t e = hipStreamSynchronize(h->d.ex.stream);
    if (e != hipSuccess) set_sticky(std::string("HIP stream failed: ") + hipGetErrorString(e));
  }
#endif
  if (!sticky_error().empty()) return fail(sticky_error());
  if (!h->d.err.empty() && h->d.halo_missing) return fail(h->d.err);
  if (h->d.halo_missing) return fail("halo exchange needed before fv3lm_set_exchange provided its table (face mode)");
  if (h->d.tracer_subcycle_error) return fail("tracer_2d: accumulated Courant number > 60: trajectory is not usable");
  if (h->d.nh_overflow()) return fail("non-hydrostatic column solver: reverse-mode tape overflow (internal sizing error)");
  return 0;
}
int fv3lm_set_face_data(fv3lm_handle* h, const double* edge, const double* ecorner) {
  if (!edge || !ecorner) return fail("fv3lm_set_face_data: null argument");
  return h->d.set_face_data(edge, ecorner) ? 0 : fail(h->d.err);
}
int fv3lm_set_exchange(fv3lm_handle* h, int kind, const int* rows, int nrows) {
  if (!rows && nrows > 0) return fail("fv3lm_set_exchange: null table");
  return h->d.set_exchange(kind, rows, nrows) ? 0 : fail(h->d.err);
}
int fv3lm_set_exchange_remote(fv3lm_handle* h, int kind, int npeers, const int* peers, const int* nsend, const int* send_rows, const int* nrecv,
                              const int* recv_rows) {
  return h->d.set_exchange_remote(kind, npeers, peers, nsend, send_rows, nrecv, recv_rows) ? 0 : fail(h->d.err);
}
// Transport between ranks.  Product: RCCL point-to-point on the library stream (library path = the RCCL the process already
// uses, e.g. torch's; the 128-byte unique id comes from rank 0 and is distributed by the host).  Callback: host-emulation tests.
int fv3lm_comm_unique_id(const char* rccl_path, void* id128) {
#ifndef FV3LM_HOST_EMUL
  std::string e; Transport& T = transport();
  if (!T.load(rccl_path, e)) return fail(e);
  ncclUniqueId id;
  if (T.pGetUniqueId(&id) != ncclSuccess) return fail("ncclGetUniqueId failed");
  std::memcpy(id128, &id, sizeof id);
  return 0;
#else
  (void)rccl_path; (void)id128; return fail("host emulation has no RCCL; use fv3lm_set_transport_callback");
#endif
}
int fv3lm_comm_init(const char* rccl_path, const void* id128, int nranks, int rank) {
#ifndef FV3LM_HOST_EMUL
  std::string e; Transport& T = transport();
  if (!T.load(rccl_path, e)) return fail(e);
  ncclUniqueId id; std::memcpy(&id, id128, sizeof id);
  if (T.pCommInitRank(&T.comm, nranks, id, rank) != ncclSuccess) return fail("ncclCommInitRank failed");
  T.rank = rank; T.nranks = nranks;
  return 0;
#else
  (void)rccl_path; (void)id128; (void)nranks; (void)rank; return fail("host emulation has no RCCL; use fv3lm_set_transport_callback");
#endif
}
int fv3lm_comm_destroy(void) {
#ifndef FV3LM_HOST_EMUL
  Transport& T = transport();
  if (T.comm && T.pCommDestroy) T.pCommDestroy(T.comm);
  T.comm = nullptr;
#endif
  return 0;
}
int fv3lm_set_allreduce_callback(fv3lm_allreduce_fn fn, void* user) { allreduce_max_hook().cb = fn; allreduce_max_hook().user = user; return 0; }
int fv3lm_set_transport_callback(fv3lm_transport_fn fn, void* user) { transport().cb = fn; transport().user = user; return 0; }
int fv3lm_halo(fv3lm_handle* h, int kind, const char* name0, const char* name1, int mode) {
  if (mode < 0 || mode > 2 || kind < 0 || kind >= H_NKIND) return fail("bad mode or kind");
  Fld f0, f1{};
  auto it = h->d.F.find(name0);
  if (it == h->d.F.end()) return fail(std::string("unknown field ") + name0);
  f0 = it->second;
  if (name1 && name1[0]) { it = h->d.F.find(name1); if (it == h->d.F.end()) return fail(std::string("unknown field ") + name1); f1 = it->second; }
  h->d.halo(mode, kind, f0, f1);
  return status(h);
}
int fv3lm_run_group(fv3lm_handle* h, const char* group, int mode) {
  if (mode < 0 || mode > 2) return fail("bad mode");
  // one group on its own: every adjoint launch accumulates (the stored-first-write plan of Dycore::plan_adjoint belongs to the whole sweep)
  h->d.ex.no_wmask = group && group[0];
  h->d.run_group(h->d.acoustic, group, mode);
  h->d.ex.no_wmask = false;
  return status(h);
}
int fv3lm_dyn_core(fv3lm_handle* h, int mode) {
  if (mode < 0 || mode > 2) return fail("bad mode");
  h->d.dyn_core(mode);
  return status(h);
}
int fv3lm_pressures(fv3lm_handle* h, int mode) { h->d.pressures(mode); return status(h); }
int fv3lm_tracer_2d(fv3lm_handle* h, int mode) {
  if (mode == MODE_AD) h->d.tracer_ad(); else h->d.tracer_fwd(mode);
  return status(h);
}
int fv3lm_traj_slots(fv3lm_handle* h) { return (int)h->d.traj_slot.size(); }   /* acoustic steps whose intermediates stay resident */
int fv3lm_tracer_nsplt(fv3lm_handle* h) { return h->d.nsplt_max; }   /* largest sub-step count tracer_2d has used so far */
int fv3lm_remap(fv3lm_handle* h, int mode, int last_step) { h->d.each_class([&]() { run_remap(h->d.ex, mode, h->d.remap_args(last_step != 0)); }); return status(h); }
int fv3lm_fv_dynamics(fv3lm_handle* h, int mode) { h->d.fv_dynamics(mode); return status(h); }
int fv3lm_step_tl(fv3lm_handle* h) { h->d.step_tl(); return status(h); }
int fv3lm_step_nl(fv3lm_handle* h) { h->d.step_nl(); return status(h); }
int fv3lm_step_ad(fv3lm_handle* h) { h->d.step_ad(); return status(h); }
int fv3lm_traj_to_fv3(fv3lm_handle* h, const double* u, const double* v, const double* t, const double* delp, const double* const* q,
                      const double* w, const double* delz, const double* phis) {
  if (!h->d.traj_to_fv3(u, v, t, delp, q, w, delz, phis)) return fail(h->d.err);
  return status(h);
}
int fv3lm_pert_to_fv3(fv3lm_handle* h, const double* u, const double* v, const double* t, const double* delp, const double* const* q,
                      const double* w, const double* delz) {
  if (!h->d.pert_to_fv3(u, v, t, delp, q, w, delz)) return fail(h->d.err);
  return status(h);
}
int fv3lm_fv3_to_pert(fv3lm_handle* h, double* u, double* v, double* t, double* delp, double* const* q, double* w, double* delz) {
  if (!h->d.fv3_to_pert(u, v, t, delp, q, w, delz)) return fail(h->d.err);
  return status(h);
}
int fv3lm_zero_work_adjoint(fv3lm_handle* h) { h->d.zero_work_adjoint(); return 0; }
int fv3lm_sync(fv3lm_handle* h) { dev_sync(h->d.ex); return status(h); }
long fv3lm_launch_count(fv3lm_handle* h) { return h->d.ex.launches; }
int fv3lm_level_params(fv3lm_handle* h, int k, int* ip, double* rp) {
  if (k < 1 || k > h->d.g.npz) return fail("level out of range");
  const LevelParams& l = h->d.lev_host[k - 1];
  ip[0] = l.hord_mt; ip[1] = l.hord_vt; ip[2] = l.hord_tm; ip[3] = l.hord_dp; ip[4] = l.hord_tr;
  ip[5] = l.nord; ip[6] = l.nord_v; ip[7] = l.nord_w; ip[8] = l.nord_t; ip[9] = l.nord_v_pert;
  rp[0] = l.d2_divg; rp[1] = l.damp_vt; rp[2] = l.damp_w; rp[3] = l.damp_t; rp[4] = l.d_con; rp[5] = l.damp_vt_pert;
  return 0;
}
// Device selection for one-process-per-GPU launches (call before fv3lm_create).
int fv3lm_set_device(int dev) {
#ifndef FV3LM_HOST_EMUL
  if (hipSetDevice(dev) != hipSuccess) return fail("hipSetDevice failed");
#endif
  (void)dev; return 0;
}
// Device-side snapshot / restore of the prognostic state (trajectory and perturbation of u v pt delp q*).
int fv3lm_state_save(fv3lm_handle* h) {
  Dynamics& d = h->d; std::vector<Fld> fs{d.f("u"), d.f("v"), d.f("pt"), d.f("delp")};
  if (d.nh) { fs.push_back(d.f("w")); fs.push_back(d.f("delz")); }
  for (auto& q : d.q) fs.push_back(q);
  if (d.snap.size() != fs.size() * 2) { for (double* p : d.snap) dev_free(p); d.snap.clear(); for (size_t n = 0; n < fs.size() * 2; ++n) d.snap.push_back((double*)dev_alloc(d.n3 * 8)); }
  for (size_t n = 0; n < fs.size(); ++n) { dev_copy(d.ex, d.snap[2 * n], fs[n].t, d.n3 * 8); dev_copy(d.ex, d.snap[2 * n + 1], fs[n].p, d.n3 * 8); }
  return 0;
}
int fv3lm_state_restore(fv3lm_handle* h) {
  Dynamics& d = h->d; std::vector<Fld> fs{d.f("u"), d.f("v"), d.f("pt"), d.f("delp")};
  if (d.nh) { fs.push_back(d.f("w")); fs.push_back(d.f("delz")); }
  for (auto& q : d.q) fs.push_back(q);
  if (d.snap.size() != fs.size() * 2) return fail("fv3lm_state_restore: no snapshot");
  for (size_t n = 0; n < fs.size(); ++n) { dev_copy(d.ex, fs[n].t, d.snap[2 * n], d.n3 * 8); dev_copy(d.ex, fs[n].p, d.snap[2 * n + 1], d.n3 * 8); }
  return 0;
}
// Per-kernel HIP-event profile of everything launched between begin and end (bench.py roofline leg).
// end() writes lines "name count total_ms total_algorithmic_bytes" into buf and returns the length needed.
int fv3lm_profile_begin(fv3lm_handle* h) { h->d.ex.recs.clear(); h->d.ex.profiling = true; return 0; }
int fv3lm_profile_end(fv3lm_handle* h, char* buf, int buflen) {
  Exec& ex = h->d.ex; ex.profiling = false; dev_sync(ex);
  std::map<std::string, std::array<double, 3>> agg;
  for (ProfRec& r : ex.recs) {
    float ms = 0.f;
#ifndef FV3LM_HOST_EMUL
    (void)hipEventElapsedTime(&ms, r.e0, r.e1); (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
#endif
    auto& a = agg[r.name]; a[0] += 1.; a[1] += ms; a[2] += r.bytes;
  }
  ex.recs.clear();
  std::string out;
  for (auto& kv : agg) { char line[256]; std::snprintf(line, sizeof line, "%s %.0f %.6f %.0f\n", kv.first.c_str(), kv.second[0], kv.second[1], kv.second[2]); out += line; }
  if (buf && buflen > 0) { std::strncpy(buf, out.c_str(), buflen - 1); buf[buflen - 1] = 0; }
  return (int)out.size() + 1;
}
#ifdef FV3LM_HOST_EMUL
int fv3lm_emul_check_boxes(fv3lm_handle* h, int on) { h->d.ex.check_boxes = on != 0; return 0; }
#endif

}  // extern "C"
