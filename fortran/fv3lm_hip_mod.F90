!> fv3lm_hip_mod — ISO_C_BINDING shim between the Fortran host (FV3-JEDI / fv3jedi_lm_dynamics_mod)
!! and the MI355X-native TL/AD dynamical core behind the C-ABI of include/fv3lm.h.
!!
!! The host keeps everything it does today (trajectory management, traj/pert -> FV_Atm copies, the
!! mpp_get_boundary edge fill, fv3_to_pert); only the calls that fv3jedi_lm_dynamics_mod makes into
!! the Tapenade code are replaced (INTEGRATION.md shows the three edited call sites):
!!   create  : call fv3lm_hip_create(...)  after fv_init / fv_init_pert  (fv3jedi_lm_dynamics_mod.F90:155)
!!   step_tl : call fv3lm_hip_step_tl(...) in place of compute_fv3_pressures_tlm + fv_dynamics_tlm (:404-438)
!!   step_ad : call fv3lm_hip_step_ad(...) in place of fv_dynamics_fwd / fv_dynamics_bwd / compute_fv3_pressures_bwd (:507-638)
!!   delete  : call fv3lm_hip_destroy(...) (:693-713)
!! A nonzero status from the library becomes a fatal error, as the reference does with
!! mpp_error(FATAL)/exit(1) (src/fv3jedi_lm_mod.F90:93).
module fv3lm_hip_mod
  use iso_c_binding
  implicit none
  private
  public :: fv3lm_options, fv3lm_dims, fv3lm_hip_type
  public :: fv3lm_hip_create, fv3lm_hip_destroy, fv3lm_hip_put, fv3lm_hip_get
  public :: fv3lm_hip_step_tl, fv3lm_hip_step_ad

  integer, parameter :: ng = 3   ! halo width, tools/fv_mp_nlm_mod.F90:67

  !> mirrors `struct fv3lm_options` (include/fv3lm.h) field for field
  type, bind(C) :: fv3lm_options
    integer(c_int) :: hord_mt, hord_vt, hord_tm, hord_dp, hord_tr
    integer(c_int) :: nord, do_vort_damp, n_sponge
    integer(c_int) :: hord_mt_pert, hord_vt_pert, hord_tm_pert, hord_dp_pert, hord_tr_pert
    integer(c_int) :: nord_pert, do_vort_damp_pert, n_sponge_pert, hord_ks_traj, hord_ks_pert
    integer(c_int) :: hord_mt_ks_traj, hord_vt_ks_traj, hord_tm_ks_traj, hord_dp_ks_traj, hord_tr_ks_traj
    integer(c_int) :: hord_mt_ks_pert, hord_vt_ks_pert, hord_tm_ks_pert, hord_dp_ks_pert, hord_tr_ks_pert
    integer(c_int) :: kord_tm, kord_mt, kord_wz, kord_tr
    integer(c_int) :: hydrostatic, pad_
    real(c_double) :: dddmp, d2_bg, d4_bg, vtdm4, d2_bg_k1, d2_bg_k2, d_con, ke_bg
    real(c_double) :: dddmp_pert, d2_bg_pert, d4_bg_pert, vtdm4_pert, d2_bg_k1_pert, d2_bg_k2_pert, d2_bg_ks_pert
    real(c_double) :: akap, cp, zvir, grav_jedi
    real(c_double) :: cp_air, rdgas, rvgas, grav, radius, omega, hlv
    real(c_double) :: ptop
  end type fv3lm_options

  type, bind(C) :: fv3lm_dims
    integer(c_int) :: nx, ny, npz, ntile, nq, n_split, k_split, pad_
    real(c_double) :: dt
  end type fv3lm_dims

  type :: fv3lm_hip_type
    type(c_ptr) :: handle = c_null_ptr
    integer :: nx = 0, ny = 0, npz = 0
  end type fv3lm_hip_type

  interface
    function c_create(h, dims, opt, metrics, da_min, da_min_c, phis, ak, bk) bind(C, name="fv3lm_create") result(rc)
      import :: c_ptr, c_int, c_double, fv3lm_dims, fv3lm_options
      type(c_ptr), intent(out) :: h
      type(fv3lm_dims), intent(in) :: dims
      type(fv3lm_options), intent(in) :: opt
      type(c_ptr), intent(in) :: metrics(*)
      real(c_double), value :: da_min, da_min_c
      real(c_double), intent(in) :: phis(*), ak(*), bk(*)
      integer(c_int) :: rc
    end function
    function c_destroy(h) bind(C, name="fv3lm_destroy") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int) :: rc
    end function
    function c_put(h, name, which, host) bind(C, name="fv3lm_field_put") result(rc)
      import :: c_ptr, c_int, c_char, c_double
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: which
      real(c_double), intent(in) :: host(*)
      integer(c_int) :: rc
    end function
    function c_get(h, name, which, host) bind(C, name="fv3lm_field_get") result(rc)
      import :: c_ptr, c_int, c_char, c_double
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), value :: which
      real(c_double), intent(out) :: host(*)
      integer(c_int) :: rc
    end function
    function c_step_tl(h) bind(C, name="fv3lm_step_tl") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int) :: rc
    end function
    function c_step_nl(h) bind(C, name="fv3lm_step_nl") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int) :: rc
    end function
    function c_step_ad(h) bind(C, name="fv3lm_step_ad") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int) :: rc
    end function
    function c_last_error() bind(C, name="fv3lm_last_error") result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
  end interface

contains

  subroutine check(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    integer :: n
    if (rc == 0) return
    call c_f_pointer(c_last_error(), msg, [512])
    n = 1
    do while (n < 512 .and. msg(n) /= c_null_char)
      n = n + 1
    end do
    write(*, '(4a)') 'FATAL fv3lm_hip ', where, ': ', transfer(msg(1:n-1), repeat(' ', n-1))
    call exit(1)      ! the reference's own failure mode, src/fv3jedi_lm_mod.F90:93
  end subroutine check

  !> metrics(:) holds c_loc of the 50 metric planes in the order of fv3lm_metric_names(), each
  !! (isd:ied+1, jsd:jed+1) — the host copies gridstruct%... into padded planes once at create.
  subroutine fv3lm_hip_create(self, dims, opt, metrics, da_min, da_min_c, phis, ak, bk)
    type(fv3lm_hip_type), intent(inout) :: self
    type(fv3lm_dims), intent(in) :: dims
    type(fv3lm_options), intent(in) :: opt
    type(c_ptr), intent(in) :: metrics(:)
    real(c_double), intent(in) :: da_min, da_min_c
    real(c_double), intent(in) :: phis(:, :), ak(:), bk(:)
    call check(c_create(self%handle, dims, opt, metrics, da_min, da_min_c, phis, ak, bk), 'create')
    self%nx = dims%nx; self%ny = dims%ny; self%npz = dims%npz
  end subroutine fv3lm_hip_create

  subroutine fv3lm_hip_destroy(self)
    type(fv3lm_hip_type), intent(inout) :: self
    if (c_associated(self%handle)) call check(c_destroy(self%handle), 'destroy')
    self%handle = c_null_ptr
  end subroutine fv3lm_hip_destroy

  !> Upload one FV_Atm / FV_AtmP array.  `a` has the reference's own bounds
  !! (ilo:ihi, jlo:jhi, nk), e.g. u(isd:ied, jsd:jed+1, npz) or traj%u(isc:iec, jsc:jec, npz);
  !! it is repacked into the padded plane (isd:ied+1, jsd:jed+1) the device uses.
  subroutine fv3lm_hip_put(self, name, which, a, ilo, jlo)
    type(fv3lm_hip_type), intent(in) :: self
    character(len=*), intent(in) :: name
    integer, intent(in) :: which, ilo, jlo
    real(c_double), intent(in) :: a(ilo:, jlo:, :)
    real(c_double), allocatable :: pad(:, :, :)
    allocate(pad(1-ng:self%nx+ng+1, 1-ng:self%ny+ng+1, size(a, 3)))
    pad = 0.0_c_double
    pad(ilo:ubound(a, 1), jlo:ubound(a, 2), :) = a
    call check(c_put(self%handle, trim(name)//c_null_char, int(which, c_int), pad), 'put '//name)
  end subroutine fv3lm_hip_put

  subroutine fv3lm_hip_get(self, name, which, a, ilo, jlo)
    type(fv3lm_hip_type), intent(in) :: self
    character(len=*), intent(in) :: name
    integer, intent(in) :: which, ilo, jlo
    real(c_double), intent(inout) :: a(ilo:, jlo:, :)
    real(c_double), allocatable :: pad(:, :, :)
    allocate(pad(1-ng:self%nx+ng+1, 1-ng:self%ny+ng+1, size(a, 3)))
    call check(c_get(self%handle, trim(name)//c_null_char, int(which, c_int), pad), 'get '//name)
    a = pad(ilo:ubound(a, 1), jlo:ubound(a, 2), :)
  end subroutine fv3lm_hip_get

  !> Replaces compute_fv3_pressures_tlm + fv_dynamics_tlm (fv3jedi_lm_dynamics_mod.F90:404-438).
  subroutine fv3lm_hip_step_tl(self)
    type(fv3lm_hip_type), intent(in) :: self
    call check(c_step_tl(self%handle), 'step_tl')
  end subroutine fv3lm_hip_step_tl

  !> Replaces fv_dynamics_fwd + fv_dynamics_bwd + compute_fv3_pressures_bwd (:507-638).
  subroutine fv3lm_hip_step_ad(self)
    type(fv3lm_hip_type), intent(in) :: self
    call check(c_step_nl(self%handle), 'step_ad (forward sweep)')
    call check(c_step_ad(self%handle), 'step_ad (backward sweep)')
  end subroutine fv3lm_hip_step_ad

end module fv3lm_hip_mod
