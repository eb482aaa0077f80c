! TEST INFRASTRUCTURE — bind(C) entry points (ours) onto the reference's fv_pressure_mod, so that
! tests can call compute_fv3_pressures_tlm / _bwd of the reference itself through ctypes.
module fvp_wrap
  use iso_c_binding
  use fv_pressure_mod, only: compute_fv3_pressures_tlm, compute_fv3_pressures_bwd
  implicit none
contains
  subroutine ref_pressures_tlm(is, ie, js, je, isd, ied, jsd, jed, npz, kappa, ptop, delp, delp_tl, &
                               pe, pe_tl, pk, pk_tl, pkz, pkz_tl, peln, peln_tl) bind(C, name="ref_pressures_tlm")
    integer(c_int), value :: is, ie, js, je, isd, ied, jsd, jed, npz
    real(c_double), value :: kappa, ptop
    real(c_double) :: delp(isd:ied, jsd:jed, npz), delp_tl(isd:ied, jsd:jed, npz)
    real(c_double) :: pe(is-1:ie+1, npz+1, js-1:je+1), pe_tl(is-1:ie+1, npz+1, js-1:je+1)
    real(c_double) :: pk(is:ie, js:je, npz+1), pk_tl(is:ie, js:je, npz+1)
    real(c_double) :: peln(is:ie, npz+1, js:je), peln_tl(is:ie, npz+1, js:je)
    real(c_double) :: pkz(is:ie, js:je, npz), pkz_tl(is:ie, js:je, npz)
    call compute_fv3_pressures_tlm(is, ie, js, je, isd, ied, jsd, jed, npz, kappa, ptop, delp, delp_tl, &
                                   pe, pe_tl, pk, pk_tl, pkz, pkz_tl, peln, peln_tl)
  end subroutine
  subroutine ref_pressures_bwd(is, ie, js, je, isd, ied, jsd, jed, npz, kappa, ptop, delp, delp_ad, &
                               pe, pe_ad, pk, pk_ad, pkz, pkz_ad, peln, peln_ad) bind(C, name="ref_pressures_bwd")
    integer(c_int), value :: is, ie, js, je, isd, ied, jsd, jed, npz
    real(c_double), value :: kappa, ptop
    real(c_double) :: delp(isd:ied, jsd:jed, npz), delp_ad(isd:ied, jsd:jed, npz)
    real(c_double) :: pe(is-1:ie+1, npz+1, js-1:je+1), pe_ad(is-1:ie+1, npz+1, js-1:je+1)
    real(c_double) :: pk(is:ie, js:je, npz+1), pk_ad(is:ie, js:je, npz+1)
    real(c_double) :: peln(is:ie, npz+1, js:je), peln_ad(is:ie, npz+1, js:je)
    real(c_double) :: pkz(is:ie, js:je, npz), pkz_ad(is:ie, js:je, npz)
    call compute_fv3_pressures_bwd(is, ie, js, je, isd, ied, jsd, jed, npz, kappa, ptop, delp, delp_ad, &
                                   pe, pe_ad, pk, pk_ad, pkz, pkz_ad, peln, peln_ad)
  end subroutine
end module fvp_wrap
