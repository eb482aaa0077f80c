"""Configuration containers mirroring fv_flags_type / fv_flags_pert_type.

Reference: NLM/fv_arrays_nlm.F90:236-506 (fv_core_nml), TLM/fv_arrays_tlmadm.F90:37-92
(fv_core_pert_nml), utils/fv3jedi_lm_const_mod.F90:17-41 (JEDI constants).  The struct layouts
must match include/fv3lm.h exactly.
"""
import ctypes as C


class Options(C.Structure):
    _fields_ = (
        [(n, C.c_int) for n in (
            "hord_mt hord_vt hord_tm hord_dp hord_tr nord do_vort_damp n_sponge "
            "hord_mt_pert hord_vt_pert hord_tm_pert hord_dp_pert hord_tr_pert "
            "nord_pert do_vort_damp_pert n_sponge_pert hord_ks_traj hord_ks_pert "
            "hord_mt_ks_traj hord_vt_ks_traj hord_tm_ks_traj hord_dp_ks_traj hord_tr_ks_traj "
            "hord_mt_ks_pert hord_vt_ks_pert hord_tm_ks_pert hord_dp_ks_pert hord_tr_ks_pert "
            "kord_tm kord_mt kord_wz kord_tr kord_tm_pert kord_mt_pert kord_wz_pert kord_tr_pert hydrostatic split_damp").split()]
        + [(n, C.c_double) for n in (
            "dddmp d2_bg d4_bg vtdm4 d2_bg_k1 d2_bg_k2 d_con ke_bg "
            "dddmp_pert d2_bg_pert d4_bg_pert vtdm4_pert d2_bg_k1_pert d2_bg_k2_pert d2_bg_ks_pert "
            "akap cp zvir grav_jedi cp_air rdgas rvgas grav radius omega hlv ptop a_imp p_fac scale_z").split()]
    )

    def int_list(self):
        return [getattr(self, n) for n, t in self._fields_ if t is C.c_int]

    def real_list(self):
        return [getattr(self, n) for n, t in self._fields_ if t is C.c_double and n not in ("a_imp", "p_fac", "scale_z")]


class Dims(C.Structure):
    _fields_ = ([(n, C.c_int) for n in "nx ny npz ntile nq n_split k_split face".split()] + [("dt", C.c_double)]
                + [("nface", C.c_int), ("pad_", C.c_int), ("tile_ij0", C.POINTER(C.c_int))])


def default_options(**kw):
    """fv_core_pert_nml defaults (TLM/fv_arrays_tlmadm.F90:40-90) with split_hord = split_kord =
    split_damp = .false. so that run_setup_pert copies them onto the trajectory
    (TLM/fv_control_tlmadm.F90:219-252); BASELINE.md §4 scheme flags."""
    o = Options()
    for n in "hord_mt hord_vt hord_tm hord_dp hord_tr".split():
        setattr(o, n, 2); setattr(o, n + "_pert", 2)
        setattr(o, n + "_ks_traj", 1); setattr(o, n + "_ks_pert", 1)
    o.nord = o.nord_pert = 1
    o.do_vort_damp = o.do_vort_damp_pert = 1
    o.n_sponge = 1            # fv_flags_type default, NLM/fv_arrays_nlm.F90:327
    o.n_sponge_pert = 9
    o.hord_ks_traj = o.hord_ks_pert = 1
    o.kord_tm, o.kord_mt, o.kord_wz, o.kord_tr = -17, 17, 17, 17
    o.kord_tm_pert, o.kord_mt_pert, o.kord_wz_pert, o.kord_tr_pert = -17, 17, 17, 17
    o.hydrostatic = 1
    o.dddmp = o.dddmp_pert = 0.2
    o.d2_bg = o.d2_bg_pert = 0.015
    o.d4_bg = o.d4_bg_pert = 0.15
    o.vtdm4 = o.vtdm4_pert = 0.0005
    # The reference defaults d2_bg_k1/k2/ks = 4/2/2 are absolute del-2 coefficients far outside the
    # stable range (they are meant to be set by namelist); use typical operational values.
    o.d2_bg_k1 = o.d2_bg_k1_pert = 0.20
    o.d2_bg_k2 = o.d2_bg_k2_pert = 0.12
    o.d2_bg_ks_pert = 0.06
    o.d_con = 0.0
    o.ke_bg = 0.0
    # JEDI constants (utils/fv3jedi_lm_const_mod.F90:17-41)
    runiv, airmw, h2omw, kappa = 8314.47, 28.965, 18.015, 2.0 / 7.0
    rgas = runiv / airmw
    o.akap = kappa
    o.cp = rgas / kappa
    o.zvir = (runiv / h2omw) / rgas - 1.0
    o.grav_jedi = 9.80665
    # FMS constants_mod, GFDL variant
    o.rdgas, o.rvgas, o.grav, o.radius, o.omega, o.hlv = 287.04, 461.50, 9.80, 6371.0e3, 7.292e-5, 2.500e6
    o.cp_air = o.rdgas / kappa
    o.ptop = 1.0
    o.a_imp, o.p_fac, o.scale_z = 0.75, 0.05, 0.0    # non-hydrostatic solver (fv_flags_type defaults; the reference default a_imp = 0.75)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o
