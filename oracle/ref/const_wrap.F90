! TEST INFRASTRUCTURE -- bind(C) read-out (ours) of the reference's own constants modules, compiled where they lie:
! fv3jedi_lm_const_mod (the JEDI set handed to FV_DYNAMICS_TLM, DYN/fv3jedi_lm_dynamics_mod.F90:423-424) and MAPL_ConstantsMod.
module const_wrap
  use iso_c_binding
  use fv3jedi_lm_const_mod, only: grav, radius, omega, airmw, h2omw, runiv, rdry, cpdry, rvap, kappa, rgas, cp, zvir, pi
  implicit none
contains
  subroutine ref_jedi_constants(v) bind(C, name="ref_jedi_constants")
    real(c_double) :: v(14)
    v(1) = kappa; v(2) = cp; v(3) = zvir; v(4) = grav; v(5) = rgas; v(6) = rdry; v(7) = cpdry; v(8) = rvap
    v(9) = runiv; v(10) = airmw; v(11) = h2omw; v(12) = radius; v(13) = omega; v(14) = pi
  end subroutine
end module const_wrap
