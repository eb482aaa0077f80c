!> shim_driver — a Fortran host in miniature: links fv3lm_hip_mod (the ISO_C_BINDING shim) against libfv3lm_hip.so and drives
!! create / put / step_tl / get / step_ad / get / destroy through it, the way the edited fv3jedi_lm_dynamics_mod would
!! (INTEGRATION.md).  Inputs and outputs travel through two stream files so that tests/test_gpu_fortran_shim.py can hold the
!! results against the same calls made through ctypes (bit for bit): that checks the bind(C) type layouts, the repacking of the
!! reference-shaped arrays (u(isd:ied, jsd:jed+1, npz), v(isd:ied+1, jsd:jed, npz), ...) into padded planes, and the error path.
!! usage: shim_driver <input file> <output file>
program shim_driver
  use iso_c_binding
  use fv3lm_hip_mod
  implicit none
  integer, parameter :: ng = 3
  character(len=512) :: fin, fout
  type(fv3lm_dims) :: dims
  type(fv3lm_options) :: opt
  type(fv3lm_hip_type) :: dyn
  integer(c_int8_t), allocatable :: raw(:)
  integer :: nx, ny, npz, nq, n, m, isd, ied, jsd, jed, nraw, bad
  real(c_double) :: da_min, da_min_c
  real(c_double), allocatable, target :: metrics(:, :, :)
  type(c_ptr) :: mptr(50)
  real(c_double), allocatable :: phis(:, :), ak(:), bk(:)
  real(c_double), allocatable :: u(:, :, :), v(:, :, :), pt(:, :, :), delp(:, :, :), q(:, :, :, :)
  real(c_double), allocatable :: up(:, :, :), vp(:, :, :), ptp(:, :, :), delpp(:, :, :), qp(:, :, :, :)
  character(len=8) :: qn

  call get_command_argument(1, fin); call get_command_argument(2, fout)
  open(11, file=trim(fin), access='stream', form='unformatted', status='old')
  read(11) nraw; allocate(raw(nraw)); read(11) raw; dims = transfer(raw, dims); deallocate(raw)
  read(11) nraw; allocate(raw(nraw)); read(11) raw; opt = transfer(raw, opt); deallocate(raw)
  nx = dims%nx; ny = dims%ny; npz = dims%npz; nq = dims%nq
  isd = 1 - ng; ied = nx + ng; jsd = 1 - ng; jed = ny + ng
  allocate(metrics(isd:ied+1, jsd:jed+1, 50), phis(isd:ied+1, jsd:jed+1), ak(npz+1), bk(npz+1))
  read(11) da_min, da_min_c; read(11) metrics; read(11) phis; read(11) ak; read(11) bk
  ! the reference's own array shapes (FV_Atm%u, %v, %pt, %delp, %q)
  allocate(u(isd:ied, jsd:jed+1, npz), v(isd:ied+1, jsd:jed, npz), pt(isd:ied, jsd:jed, npz), delp(isd:ied, jsd:jed, npz), q(isd:ied, jsd:jed, npz, nq))
  allocate(up(isd:ied, jsd:jed+1, npz), vp(isd:ied+1, jsd:jed, npz), ptp(isd:ied, jsd:jed, npz), delpp(isd:ied, jsd:jed, npz), qp(isd:ied, jsd:jed, npz, nq))
  read(11) u, v, pt, delp, q; read(11) up, vp, ptp, delpp, qp
  read(11) bad
  close(11)
  do m = 1, 50
    mptr(m) = c_loc(metrics(isd, jsd, m))
  end do
  if (bad == 1) opt%nord = 7        ! an option the library refuses: exercises status -> fv3lm_last_error -> fatal exit
  call fv3lm_hip_create(dyn, dims, opt, mptr, da_min, da_min_c, phis, ak, bk)
  open(12, file=trim(fout), access='stream', form='unformatted', status='replace')
  ! ---- step_tl: trajectory and perturbation in, both advanced
  call put_all(0, u, v, pt, delp, q); call put_all(1, up, vp, ptp, delpp, qp)
  call fv3lm_hip_step_tl(dyn)
  call get_all(0); call get_all(1)
  ! ---- step_ad: same trajectory, the perturbation arrays as the adjoint forcing
  call put_all(0, u, v, pt, delp, q); call put_all(1, up, vp, ptp, delpp, qp)
  call fv3lm_hip_step_ad(dyn)
  call get_all(1)
  ! ---- the host's own boundary: compact traj% / pert% arrays (isc:iec, jsc:jec, npz), no halo (traj_to_fv3, pert_to_fv3, fv3_to_pert)
  call boundary_run()
  close(12)
  call fv3lm_hip_destroy(dyn)
  write(*, '(a)') 'shim_driver OK'
contains
  subroutine put_all(which, au, av, apt, adelp, aq)
    integer, intent(in) :: which
    real(c_double), intent(in) :: au(isd:, jsd:, :), av(isd:, jsd:, :), apt(isd:, jsd:, :), adelp(isd:, jsd:, :), aq(isd:, jsd:, :, :)
    call fv3lm_hip_put(dyn, 'u', which, au, isd, jsd); call fv3lm_hip_put(dyn, 'v', which, av, isd, jsd)
    call fv3lm_hip_put(dyn, 'pt', which, apt, isd, jsd); call fv3lm_hip_put(dyn, 'delp', which, adelp, isd, jsd)
    do n = 1, nq
      write(qn, '(a,i0)') 'q', n
      call fv3lm_hip_put(dyn, trim(qn), which, aq(:, :, :, n), isd, jsd)
    end do
  end subroutine put_all
  subroutine boundary_run()
    real(c_double), allocatable :: cu(:, :, :), cv(:, :, :), ct(:, :, :), cd(:, :, :), cq(:, :, :, :), cph(:, :)
    real(c_double), allocatable :: pu(:, :, :), pv(:, :, :), pt_(:, :, :), pd(:, :, :), pq(:, :, :, :)
    allocate(cu(nx, ny, npz), cv(nx, ny, npz), ct(nx, ny, npz), cd(nx, ny, npz), cq(nx, ny, npz, nq), cph(nx, ny))
    allocate(pu(nx, ny, npz), pv(nx, ny, npz), pt_(nx, ny, npz), pd(nx, ny, npz), pq(nx, ny, npz, nq))
    cu = u(1:nx, 1:ny, :); cv = v(1:nx, 1:ny, :); ct = pt(1:nx, 1:ny, :); cd = delp(1:nx, 1:ny, :); cq = q(1:nx, 1:ny, :, :)
    cph = phis(1:nx, 1:ny)
    pu = up(1:nx, 1:ny, :); pv = vp(1:nx, 1:ny, :); pt_ = ptp(1:nx, 1:ny, :); pd = delpp(1:nx, 1:ny, :); pq = qp(1:nx, 1:ny, :, :)
    call fv3lm_hip_traj_to_fv3(dyn, cu, cv, ct, cd, cq, cph)
    call fv3lm_hip_pert_to_fv3(dyn, pu, pv, pt_, pd, pq)
    call fv3lm_hip_step_tl(dyn)
    call fv3lm_hip_fv3_to_pert(dyn, pu, pv, pt_, pd, pq)
    write(12) pu, pv, pt_, pd, pq
  end subroutine boundary_run
  subroutine get_all(which)
    integer, intent(in) :: which
    real(c_double), allocatable :: gu(:, :, :), gv(:, :, :), ga(:, :, :)
    allocate(gu(isd:ied, jsd:jed+1, npz), gv(isd:ied+1, jsd:jed, npz), ga(isd:ied, jsd:jed, npz))
    call fv3lm_hip_get(dyn, 'u', which, gu, isd, jsd); write(12) gu
    call fv3lm_hip_get(dyn, 'v', which, gv, isd, jsd); write(12) gv
    call fv3lm_hip_get(dyn, 'pt', which, ga, isd, jsd); write(12) ga
    call fv3lm_hip_get(dyn, 'delp', which, ga, isd, jsd); write(12) ga
    do n = 1, nq
      write(qn, '(a,i0)') 'q', n
      call fv3lm_hip_get(dyn, trim(qn), which, ga, isd, jsd); write(12) ga
    end do
  end subroutine get_all
end program shim_driver
