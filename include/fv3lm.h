/* fv3lm.h — C-ABI of the MI355X-native FV3 tangent-linear / adjoint dynamical core.
 *
 * Drop-in boundary: the body of fv3jedi_lm_dynamics_type%step_tl / %step_ad between the
 * traj/pert -> FV_Atm copies and the copy back (reference src/dynamics/fv3jedi_lm_dynamics_mod.F90:
 * step_tl :347-456 calls FV_DYNAMICS_TLM at :421-438; step_ad :460-689 calls FV_DYNAMICS_FWD :507 and
 * FV_DYNAMICS_BWD :615).  A Fortran host binds these entry points with ISO_C_BINDING
 * (fortran/fv3lm_hip_mod.F90).
 *
 * Conventions: every pointer is a host pointer to fp64 data unless a function says "device".
 * All functions return 0 on success, nonzero on failure; fv3lm_last_error() gives the message
 * (the reference has no status returns — it calls mpp_error(FATAL)/exit(1),
 * src/fv3jedi_lm_mod.F90:93 — the Fortran shim turns a nonzero status into that).
 *
 * Field layout ("padded plane"): a 3-D field is [ntile][nk][pj][pi] doubles, i fastest,
 *   pi = nx + 2*ng + 1, pj = ny + 2*ng + 1, ng = 3 (TOOLS/fv_mp_nlm_mod.F90:67);
 *   Fortran element (i,j,k), i in isd..ied+1, j in jsd..jed+1 (isd = 1-ng), sits at
 *   ((k-1)*pj + (j-jsd))*pi + (i-isd).  This one index map holds the A-, C-, D- and corner-
 *   staggered arrays of the reference (u(isd:ied,jsd:jed+1), v(isd:ied+1,jsd:jed), ...).
 */
#ifndef FV3LM_H
#define FV3LM_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fv3lm_handle fv3lm_handle;

/* fv_flags_type subset (NLM/fv_arrays_nlm.F90:236-506) and fv_flags_pert_type
 * (TLM/fv_arrays_tlmadm.F90:37-92).  The trajectory-side values are those in force after
 * run_setup_pert (TLM/fv_control_tlmadm.F90:219-252).  Two sets of physical constants are mixed on
 * the path: the JEDI set (utils/fv3jedi_lm_const_mod.F90:17-41, passed at
 * DYN/fv3jedi_lm_dynamics_mod.F90:423-424) and FMS constants_mod — both are runtime inputs. */
typedef struct fv3lm_options {
  int hord_mt, hord_vt, hord_tm, hord_dp, hord_tr;                 /* trajectory advection schemes: 1, 2, 333 (the differentiated ones) or, beside a different
                                                                      perturbation scheme (split_hord), any of the nonlinear routines' 3 .. 13 -- values only */
  int nord, do_vort_damp, n_sponge;
  int hord_mt_pert, hord_vt_pert, hord_tm_pert, hord_dp_pert, hord_tr_pert;      /* perturbation schemes: 1, 2 or 333 (tp_core_tlm.F90:2393-2487) */
  int nord_pert, do_vort_damp_pert, n_sponge_pert, hord_ks_traj, hord_ks_pert;
  int hord_mt_ks_traj, hord_vt_ks_traj, hord_tm_ks_traj, hord_dp_ks_traj, hord_tr_ks_traj;
  int hord_mt_ks_pert, hord_vt_ks_pert, hord_tm_ks_pert, hord_dp_ks_pert, hord_tr_ks_pert;
  int kord_tm, kord_mt, kord_wz, kord_tr;                          /* trajectory remap profiles: |kord| > 16 (linear) or the limited 8 .. 15 of cs_profile / scalar_profile (split_kord) */
  int kord_tm_pert, kord_mt_pert, kord_wz_pert, kord_tr_pert;      /* perturbation remap profiles (fv_flags_pert_type): |kord| > 16, the linear one */
  int hydrostatic;
  int split_damp;   /* fv_flags_pert_type%split_damp (fv_arrays_tlmadm.F90:76; the reference's default is .true.).  1: the perturbation takes its own damping
                       (nord_pert, dddmp_pert, d2_bg_pert, d4_bg_pert, vtdm4_pert ... with the perturbation sponge rules, dyn_core_tlm.F90:835-921), the
                       trajectory values the trajectory's (sw_core_tlm.F90:1664-1682, :1787-1803, :2341-2369); the trajectory's nord may then be 2 or 3.
                       0: run_setup_pert has copied the perturbation's damping options onto the trajectory (fv_control_tlmadm.F90:220-229): the two sets
                       handed over must then be equal, anything else is refused. */
  double dddmp, d2_bg, d4_bg, vtdm4, d2_bg_k1, d2_bg_k2, d_con, ke_bg;
  double dddmp_pert, d2_bg_pert, d4_bg_pert, vtdm4_pert, d2_bg_k1_pert, d2_bg_k2_pert, d2_bg_ks_pert;
  double akap, cp, zvir, grav_jedi;                          /* JEDI constants */
  double cp_air, rdgas, rvgas, grav, radius, omega, hlv;     /* FMS constants_mod */
  double ptop;
  double a_imp, p_fac, scale_z;                              /* non-hydrostatic solver (hydrostatic = 0): off-centering, pressure floor factor, w damping */
} fv3lm_options;

typedef struct fv3lm_dims {
  int nx, ny, npz;      /* cells per tile edge (npx-1, npy-1), levels */
  int ntile;            /* tiles resident on this GPU */
  int nq;               /* tracers (qv, ql, qi, o3: DYN/fv3jedi_lm_dynamics_mod.F90:158-167) */
  int n_split, k_split; /* acoustic / remap sub-steps (fv_flags_type) */
  int face;             /* 0: one doubly-periodic tile with no cube edge in it (the code path of an MPI rank in the middle of a
                           face: every `is .EQ. 1` / `j .EQ. npy` / sw_corner branch of the reference off); ntile must be 1.
                           1: every resident tile is a whole cube face (is = 1, ie = npx-1; all edge and corner branches on);
                           needs fv3lm_set_face_data and, for anything that exchanges halos, fv3lm_set_exchange. */
  double dt;            /* create(self,dt,...) src/fv3jedi_lm_mod.F90:44 */
  /* Sub-face tiles (fv_flags_type%layout > 1 x 1, NLM/fv_control_nlm.F90:556; rank -> tile map tools/fv_mp_nlm_mod.F90:452-453).  face = 1 only.
     nface: cells per edge of a whole face (npx-1); 0 or nx: every resident tile is a whole face.  tile_ij0: for each resident tile its (is, js)
     in the global indices of its face, 2*ntile ints (NULL: (1, 1) for all).  nx, ny are then the cells of a tile (ie-is+1); all resident
     tiles have the same size.  Exchange tables (fv3lm_set_exchange*) index the padded plane of a TILE. */
  int nface, pad_;
  const int* tile_ij0;
} fv3lm_dims;

/* Number and order of the metric planes handed to fv3lm_create (fv_grid_type,
 * NLM/fv_arrays_nlm.F90:115-234), each [ntile][pj][pi]:
 *  area rarea rarea_c dx dy dxa dya dxc dyc rdx rdy rdxa rdya rdxc rdyc cosa sina rsina cosa_u cosa_v
 *  cosa_s sina_u sina_v rsin_u rsin_v rsin2 f0 fC del6_u del6_v divg_u divg_v sin_sg(1..9) cos_sg(1..9) */
#define FV3LM_NMETRIC 50
const char* fv3lm_metric_names(void);

/* Replaces fv3jedi_lm_dynamics_type%create (DYN/fv3jedi_lm_dynamics_mod.F90:69-264) for the device
 * side: uploads metric terms once, resolves the per-level scheme table, allocates device state. */
int fv3lm_create(fv3lm_handle** h, const fv3lm_dims* dims, const fv3lm_options* opt, const double* const* metrics,
                 double da_min, double da_min_c, const double* phis, const double* ak, const double* bk);
/* Face mode only.  edge: the a2b_ord4 edge weights edge_w, edge_e, edge_s, edge_n of fv_grid_type
 * (NLM/fv_arrays_nlm.F90:150-151) as [ntile][4][pj], entry (j - jsd) / (i - isd) of each; ecorner: the
 * extrap_corner factors x1/(x2-x1) of a2b_ord4's four corners (a2b_edge_tlm.F90:101-139, :1478-1487) as
 * [ntile][4: sw se ne nw][3], in the order the reference evaluates them. */
int fv3lm_set_face_data(fv3lm_handle* h, const double* edge, const double* ecorner);
/* Halo exchange between the resident faces: replaces the FMS calls mpp_update_domains / mpp_get_boundary that the
 * reference issues from DYN_CORE_TLM (dyn_core_tlm.F90:1744-1790, :1960-1990, :2280-2300, :2418-2434) and
 * FV_DYNAMICS_TLM (fv_dynamics_tlm.F90:646-651, :708-712).  One table per kind of exchange:
 *   0 cell-centred scalar (delp, pt, q)        1 D-grid wind pair (u, v)        2 C-grid wind pair (uc, vc)
 *   3 corner scalar (divg_d)                   4 shared edge rows of (u, v): u(:,npy), v(npx,:)  (mpp_get_boundary)
 * rows[nrows][7] = dst_field(0|1), dst_tile, dst_index, src_field(0|1), src_tile, src_index, sign(+1|-1): plane
 * element dst_index of the halo takes sign * element src_index of the neighbour face (indices into the padded
 * plane, tiles 0-based among the resident ones; field 0/1 = first/second component of a pair). */
int fv3lm_set_exchange(fv3lm_handle* h, int kind, const int* rows, int nrows);
int fv3lm_halo(fv3lm_handle* h, int kind, const char* field0, const char* field1, int mode);   /* one exchange (tests) */
/* Faces held by other ranks (one process per GPU).  The rows of a table whose source face lives on another rank become, per
 * peer, a send list (send_rows[.][3] = field, local tile, index of the source element) and a receive list (recv_rows[.][4] =
 * field, local tile, index of the halo element, sign), concatenated over the peers in peers[] order; both ends must list the
 * rows in the order of the global table (cube.split_table).  nsend[p] / nrecv[p] = rows for peer p. */
int fv3lm_set_exchange_remote(fv3lm_handle* h, int kind, int npeers, const int* peers, const int* nsend, const int* send_rows,
                              const int* nrecv, const int* recv_rows);
/* Transport between ranks: RCCL point-to-point (ncclSend / ncclRecv, grouped per exchange) on the library's own stream.
 * rccl_path: the RCCL shared library the process already uses (e.g. the one bundled with torch), opened with dlopen so that a
 * process never runs two RCCL runtimes; id128: 128-byte ncclUniqueId made by fv3lm_comm_unique_id on rank 0 and distributed by
 * the host (mpp_broadcast / MPI_Bcast / torch.distributed); every rank that holds faces then calls fv3lm_comm_init. */
int fv3lm_comm_unique_id(const char* rccl_path, void* id128);
int fv3lm_comm_init(const char* rccl_path, const void* id128, int nranks, int rank);
int fv3lm_comm_destroy(void);
/* Host hooks.  Transport callback: replaces RCCL (the test-only host-emulation build has no other transport): called once per
 * exchange with one send and one receive buffer per peer, counts in doubles.  All-reduce callback: in-place element-wise max
 * over the ranks that hold faces, for tracer_2d's per-level Courant maxima (mp_reduce_max, fv_tracer2d_tlm.F90:1306). */
typedef void (*fv3lm_transport_fn)(void* user, int npeers, const int* peer_ranks, double* const* sendbufs, const long* send_counts,
                                   double* const* recvbufs, const long* recv_counts);
typedef void (*fv3lm_allreduce_fn)(void* user, double* buf, int n);
int fv3lm_set_transport_callback(fv3lm_transport_fn fn, void* user);
int fv3lm_set_allreduce_callback(fv3lm_allreduce_fn fn, void* user);
int fv3lm_destroy(fv3lm_handle* h);     /* %delete, DYN/fv3jedi_lm_dynamics_mod.F90:693-713 */
const char* fv3lm_last_error(void);

/* Device-resident named fields (padded-plane layout).  which: 0 = trajectory, 1 = perturbation /
 * adjoint.  State fields: u v delp pt (+ q1.. for tracers), diagnostics pe peln pk pkz,
 * accumulators mfx mfy cx cy; every work array of the step is also addressable (tests). */
int fv3lm_field_put(fv3lm_handle* h, const char* name, int which, const double* host);
int fv3lm_field_get(fv3lm_handle* h, const char* name, int which, double* host);
int fv3lm_field_levels(fv3lm_handle* h, const char* name);

/* Execution.  mode: 0 nonlinear, 1 tangent linear, 2 adjoint. */
int fv3lm_run_group(fv3lm_handle* h, const char* group, int mode);  /* one kernel group of the acoustic step:
     "c_sw" (C_SW_TLM sw_core_tlm.F90:87), "geopk_c"/"geopk_d" (GEOPK_TLM dyn_core_tlm.F90:4578),
     "p_grad_c" (:3194), "d_sw" (D_SW_TLM sw_core_tlm.F90:1047), "one_grad_p" (:3867), "halo_*" */
int fv3lm_dyn_core(fv3lm_handle* h, int mode);   /* DYN_CORE_TLM dyn_core_tlm.F90:93 / DYN_CORE_FWD+BWD dyn_core_adm.F90:115,1686 */
/* The operator itself.  State fields u v pt(=temperature) delp q1..qn — and w, delz when hydrostatic = 0 (nh_core_tlm.F90 /
 * nh_utils_tlm.F90 path: RIEM_SOLVER_C, RIEM_SOLVER3 with a_imp in (0.5, 0.999], UPDATE_DZ_C/D) — are device-resident (fv3lm_field_put /
 * _get; compute domain is..ie x js..je, D-grid winds incl. the far edge row/column).
 *  fv3lm_step_tl : replaces compute_fv3_pressures_tlm + FV_DYNAMICS_TLM in %step_tl
 *                  (DYN/fv3jedi_lm_dynamics_mod.F90:404-438).  In: trajectory (which=0) and perturbation
 *                  (which=1) at t; out: both advanced to t+dt.
 *  fv3lm_step_nl : nonlinear sweep that also stores the stage checkpoints (FV_DYNAMICS_FWD role, :507).
 *  fv3lm_step_ad : FV_DYNAMICS_BWD + compute_fv3_pressures_bwd (:615-638); call after fv3lm_step_nl on the
 *                  same trajectory.  In: adjoint of the state at t+dt (which=1); out: adjoint at t.  The forward
 *                  sweep's checkpoints stay valid: further fv3lm_step_ad calls on the same trajectory need no new
 *                  fv3lm_step_nl (the role of cp_iter, utils/tapenade/tapenade_iter.F90). */
int fv3lm_step_tl(fv3lm_handle* h);
int fv3lm_step_nl(fv3lm_handle* h);
int fv3lm_step_ad(fv3lm_handle* h);
/* The host's boundary copies on the device -- traj_to_fv3, pert_to_fv3, fv3_to_pert (DYN/fv3jedi_lm_dynamics_mod.F90:717-807, :846-889,
 * :893-933).  Arrays are the host's own traj% / pert% fields: COMPACT, (isc:iec, jsc:jec, npz) per resident tile in Fortran order
 * (i fastest, tile slowest), no halo; q[n] one array per tracer (qv ql qi o3 ...), w / delz only when hydrostatic = 0, phis
 * (isc:iec, jsc:jec) or NULL to keep the one given at create.
 *   fv3lm_traj_to_fv3: halos zeroed, interiors copied, the D-grid edge rows u(:, jec+1) / v(iec+1, :) filled from the neighbour faces
 *                      (mpp_get_boundary :781-793), halo of phis updated (:798), pe / peln / pk / pkz computed (:803-805).
 *   fv3lm_pert_to_fv3: perturbation (step_tl) or adjoint forcing (step_ad) in, halos zeroed.
 *   fv3lm_fv3_to_pert: compute-domain perturbation / adjoint out; the device copy is cleared as the reference clears FV_AtmP. */
int fv3lm_traj_to_fv3(fv3lm_handle* h, const double* u, const double* v, const double* t, const double* delp, const double* const* q,
                      const double* w, const double* delz, const double* phis);
int fv3lm_pert_to_fv3(fv3lm_handle* h, const double* u, const double* v, const double* t, const double* delp, const double* const* q,
                      const double* w, const double* delz);
int fv3lm_fv3_to_pert(fv3lm_handle* h, double* u, double* v, double* t, double* delp, double* const* q, double* w, double* delz);
/* Sub-operators, same mode argument (tests and kernel-level benchmarks): */
int fv3lm_pressures(fv3lm_handle* h, int mode);                 /* compute_fv3_pressures{,_tlm,_bwd} TLM/fv_pressure.F90 */
int fv3lm_tracer_2d(fv3lm_handle* h, int mode);                 /* TRACER_2D_TLM fv_tracer2d_tlm.F90:757 / _FWD+_BWD */
int fv3lm_traj_slots(fv3lm_handle* h);                          /* acoustic steps (of n_split*k_split) whose intermediates the forward sweep keeps
                                                                   in HBM so that the backward sweep need not recompute them; chosen at create from free memory,
                                                                   FV3LM_TRAJ_SLOTS caps it */
int fv3lm_tracer_nsplt(fv3lm_handle* h);                        /* largest tracer sub-step count (nsplt, fv_tracer2d_tlm.F90:1317) used so far */
int fv3lm_remap(fv3lm_handle* h, int mode, int last_step);      /* LAGRANGIAN_TO_EULERIAN_TLM fv_mapz_tlm.F90:69 / _FWD+_BWD */
int fv3lm_fv_dynamics(fv3lm_handle* h, int mode);               /* FV_DYNAMICS_TLM fv_dynamics_tlm.F90:87 / _FWD+_BWD */
/* Per-kernel HIP-event profile of everything launched between begin and end, on the library's stream:
 * lines "kernel count total_ms algorithmic_bytes".  Returns the buffer length needed. */
int fv3lm_profile_begin(fv3lm_handle* h);
int fv3lm_profile_end(fv3lm_handle* h, char* buf, int buflen);
int fv3lm_set_device(int dev);                 /* one process per GPU: select the device before fv3lm_create */
int fv3lm_state_save(fv3lm_handle* h);         /* device-side snapshot of u v pt delp q* (traj + pert) */
int fv3lm_state_restore(fv3lm_handle* h);
int fv3lm_zero_work_adjoint(fv3lm_handle* h);
int fv3lm_sync(fv3lm_handle* h);
long fv3lm_launch_count(fv3lm_handle* h);
int fv3lm_level_params(fv3lm_handle* h, int k, int* iparams10, double* rparams6);

#ifdef __cplusplus
}
#endif
#endif
