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
  public :: fv3lm_hip_traj_to_fv3, fv3lm_hip_pert_to_fv3, fv3lm_hip_fv3_to_pert

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
    integer(c_int) :: kord_tm_pert, kord_mt_pert, kord_wz_pert, kord_tr_pert
    integer(c_int) :: hydrostatic, split_damp
    real(c_double) :: dddmp, d2_bg, d4_bg, vtdm4, d2_bg_k1, d2_bg_k2, d_con, ke_bg
    real(c_double) :: dddmp_pert, d2_bg_pert, d4_bg_pert, vtdm4_pert, d2_bg_k1_pert, d2_bg_k2_pert, d2_bg_ks_pert
    real(c_double) :: akap, cp, zvir, grav_jedi
    real(c_double) :: cp_air, rdgas, rvgas, grav, radius, omega, hlv
    real(c_double) :: ptop
    real(c_double) :: a_imp, p_fac, scale_z
  end type fv3lm_options

  type, bind(C) :: fv3lm_dims
    integer(c_int) :: nx, ny, npz, ntile, nq, n_split, k_split
    integer(c_int) :: face   ! 1: every resident tile is a whole cube face (edge/corner branches on); 0: one edge-free periodic tile
    real(c_double) :: dt
    ! sub-face tiles (layout > 1 x 1): cells per edge of a whole face (0: the tile is the face) and, per resident tile, (is, js) in the global
    ! indices of its face (c_null_ptr: (1, 1)); nx, ny are then the cells of a tile
    integer(c_int) :: nface = 0, pad_ = 0
    type(c_ptr) :: tile_ij0 = c_null_ptr
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
    function c_set_face_data(h, edge, ecorner) bind(C, name="fv3lm_set_face_data") result(rc)
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: edge(*), ecorner(*)
      integer(c_int) :: rc
    end function
    function c_set_exchange(h, kind, rows, nrows) bind(C, name="fv3lm_set_exchange") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: kind, nrows
      integer(c_int), intent(in) :: rows(*)
      integer(c_int) :: rc
    end function
    function c_set_exchange_remote(h, kind, npeers, peers, nsend, send_rows, nrecv, recv_rows) &
        bind(C, name="fv3lm_set_exchange_remote") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: kind, npeers
      integer(c_int), intent(in) :: peers(*), nsend(*), send_rows(*), nrecv(*), recv_rows(*)
      integer(c_int) :: rc
    end function
    function c_comm_init(rccl_path, id128, nranks, rank) bind(C, name="fv3lm_comm_init") result(rc)
      import :: c_char, c_int
      character(kind=c_char), intent(in) :: rccl_path(*), id128(*)
      integer(c_int), value :: nranks, rank
      integer(c_int) :: rc
    end function
    function c_comm_unique_id(rccl_path, id128) bind(C, name="fv3lm_comm_unique_id") result(rc)
      import :: c_char, c_int
      character(kind=c_char), intent(in) :: rccl_path(*)
      character(kind=c_char), intent(out) :: id128(*)
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
    function c_traj_to_fv3(h, u, v, t, delp, q, w, delz, phis) bind(C, name="fv3lm_traj_to_fv3") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h, u, v, t, delp, w, delz, phis
      type(c_ptr), intent(in) :: q(*)
      integer(c_int) :: rc
    end function
    function c_pert_to_fv3(h, u, v, t, delp, q, w, delz) bind(C, name="fv3lm_pert_to_fv3") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h, u, v, t, delp, w, delz
      type(c_ptr), intent(in) :: q(*)
      integer(c_int) :: rc
    end function
    function c_fv3_to_pert(h, u, v, t, delp, q, w, delz) bind(C, name="fv3lm_fv3_to_pert") result(rc)
      import :: c_ptr, c_int
      type(c_ptr), value :: h, u, v, t, delp, w, delz
      type(c_ptr), intent(in) :: q(*)
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

  !> Face mode: a2b_ord4 edge weights gridstruct%edge_w/e/s/n as (pj, 4, ntile) and the extrap_corner factors (3, 4, ntile).
  subroutine fv3lm_hip_set_face_data(self, edge, ecorner)
    type(fv3lm_hip_type), intent(inout) :: self
    real(c_double), intent(in) :: edge(:, :, :), ecorner(:, :, :)
    call check(c_set_face_data(self%handle, edge, ecorner), 'set_face_data')
  end subroutine fv3lm_hip_set_face_data

  !> One halo-exchange table (kind 0..4, include/fv3lm.h): rows(7, n) for the faces resident on this GPU, and the
  !! per-peer send/receive lists for faces held by other ranks (replaces mpp_update_domains / mpp_get_boundary).
  subroutine fv3lm_hip_set_exchange(self, kind, rows)
    type(fv3lm_hip_type), intent(inout) :: self
    integer(c_int), intent(in) :: kind, rows(:, :)
    call check(c_set_exchange(self%handle, kind, rows, int(size(rows, 2), c_int)), 'set_exchange')
  end subroutine fv3lm_hip_set_exchange

  subroutine fv3lm_hip_set_exchange_remote(self, kind, peers, nsend, send_rows, nrecv, recv_rows)
    type(fv3lm_hip_type), intent(inout) :: self
    integer(c_int), intent(in) :: kind, peers(:), nsend(:), send_rows(:, :), nrecv(:), recv_rows(:, :)
    call check(c_set_exchange_remote(self%handle, kind, int(size(peers), c_int), peers, nsend, send_rows, nrecv, recv_rows), &
               'set_exchange_remote')
  end subroutine fv3lm_hip_set_exchange_remote

  !> RCCL communicator for the face exchange: rank 0 calls fv3lm_hip_comm_unique_id, the host broadcasts the 128 bytes
  !! (mpp_broadcast / MPI_Bcast), every rank calls fv3lm_hip_comm_init.
  subroutine fv3lm_hip_comm_unique_id(rccl_path, id128)
    character(len=*), intent(in) :: rccl_path
    character(kind=c_char), intent(out) :: id128(128)
    call check(c_comm_unique_id(trim(rccl_path)//c_null_char, id128), 'comm_unique_id')
  end subroutine fv3lm_hip_comm_unique_id

  subroutine fv3lm_hip_comm_init(rccl_path, id128, nranks, rank)
    character(len=*), intent(in) :: rccl_path
    character(kind=c_char), intent(in) :: id128(128)
    integer, intent(in) :: nranks, rank
    call check(c_comm_init(trim(rccl_path)//c_null_char, id128, int(nranks, c_int), int(rank, c_int)), 'comm_init')
  end subroutine fv3lm_hip_comm_init

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

  !> traj_to_fv3 / pert_to_fv3 / fv3_to_pert on the device (fv3jedi_lm_dynamics_mod.F90:717-933): the host's own traj% / pert% arrays,
  !! (isc:iec, jsc:jec, npz), no halo -- halos, D-grid edge rows, phis halo and pressures are the library's business.  q(:,:,:,n) in
  !! the reference's tracer order; w, delz only when hydrostatic = .false. (pass any array otherwise: not read).
  subroutine fv3lm_hip_traj_to_fv3(self, u, v, t, delp, q, phis, w, delz)
    type(fv3lm_hip_type), intent(in) :: self
    real(c_double), intent(in), target, contiguous :: u(:, :, :), v(:, :, :), t(:, :, :), delp(:, :, :), q(:, :, :, :), phis(:, :)
    real(c_double), intent(in), target, contiguous, optional :: w(:, :, :), delz(:, :, :)
    type(c_ptr) :: qp(max(1, size(q, 4))), wp, zp
    integer :: n
    do n = 1, size(q, 4)
      qp(n) = c_loc(q(1, 1, 1, n))
    end do
    wp = c_null_ptr; zp = c_null_ptr
    if (present(w)) wp = c_loc(w)
    if (present(delz)) zp = c_loc(delz)
    call check(c_traj_to_fv3(self%handle, c_loc(u), c_loc(v), c_loc(t), c_loc(delp), qp, wp, zp, c_loc(phis)), 'traj_to_fv3')
  end subroutine fv3lm_hip_traj_to_fv3

  subroutine fv3lm_hip_pert_to_fv3(self, u, v, t, delp, q, w, delz)
    type(fv3lm_hip_type), intent(in) :: self
    real(c_double), intent(in), target, contiguous :: u(:, :, :), v(:, :, :), t(:, :, :), delp(:, :, :), q(:, :, :, :)
    real(c_double), intent(in), target, contiguous, optional :: w(:, :, :), delz(:, :, :)
    type(c_ptr) :: qp(max(1, size(q, 4))), wp, zp
    integer :: n
    do n = 1, size(q, 4)
      qp(n) = c_loc(q(1, 1, 1, n))
    end do
    wp = c_null_ptr; zp = c_null_ptr
    if (present(w)) wp = c_loc(w)
    if (present(delz)) zp = c_loc(delz)
    call check(c_pert_to_fv3(self%handle, c_loc(u), c_loc(v), c_loc(t), c_loc(delp), qp, wp, zp), 'pert_to_fv3')
  end subroutine fv3lm_hip_pert_to_fv3

  subroutine fv3lm_hip_fv3_to_pert(self, u, v, t, delp, q, w, delz)
    type(fv3lm_hip_type), intent(in) :: self
    real(c_double), intent(inout), target, contiguous :: u(:, :, :), v(:, :, :), t(:, :, :), delp(:, :, :), q(:, :, :, :)
    real(c_double), intent(inout), target, contiguous, optional :: w(:, :, :), delz(:, :, :)
    type(c_ptr) :: qp(max(1, size(q, 4))), wp, zp
    integer :: n
    do n = 1, size(q, 4)
      qp(n) = c_loc(q(1, 1, 1, n))
    end do
    wp = c_null_ptr; zp = c_null_ptr
    if (present(w)) wp = c_loc(w)
    if (present(delz)) zp = c_loc(delz)
    call check(c_fv3_to_pert(self%handle, c_loc(u), c_loc(v), c_loc(t), c_loc(delp), qp, wp, zp), 'fv3_to_pert')
  end subroutine fv3lm_hip_fv3_to_pert

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
