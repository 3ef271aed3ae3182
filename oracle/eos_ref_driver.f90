! eos_ref_driver.f90 -- C-callable entry points into the REFERENCE's eos module (test infrastructure).
! Compiled by ref_build.sh together with /root/reference/src/eos.f90 (where it lies, unmodified)
! into oracle/_ref/libeos_ref.so.  Used only to pin oracle/cdf_oracle.c's EOS restatement and to
! generate tests/golden/eos_*.npz (tests/make_golden_eos.py).
SUBROUTINE ref_eos_init(kteos10) BIND(C, name='ref_eos_init')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kteos10
  CALL eos_init( kteos10 /= 0 )
END SUBROUTINE ref_eos_init

SUBROUTINE ref_sigma0(ptem, psal, kpi, kpj, dout) BIND(C, name='ref_sigma0')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  dout(:,:) = sigma0(ptem, psal, kpi, kpj)
END SUBROUTINE ref_sigma0

SUBROUTINE ref_sigmai(ptem, psal, pref, kpi, kpj, dout) BIND(C, name='ref_sigmai')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj
  REAL(C_FLOAT),  VALUE :: pref
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  dout(:,:) = sigmai(ptem, psal, pref, kpi, kpj)
END SUBROUTINE ref_sigmai

SUBROUTINE ref_sigmai2d(ptem, psal, pref, kpi, kpj, dout) BIND(C, name='ref_sigmai2d')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj), pref(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  dout(:,:) = sigmai(ptem, psal, pref, kpi, kpj)
END SUBROUTINE ref_sigmai2d

SUBROUTINE ref_sigmai_jmd94(ptem, psal, pref, kpi, kpj, dout) BIND(C, name='ref_sigmai_jmd94')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj
  REAL(C_FLOAT),  VALUE :: pref
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  dout(:,:) = sigmai(ptem, psal, pref, kpi, kpj, .TRUE.)
END SUBROUTINE ref_sigmai_jmd94

SUBROUTINE ref_sigmai2d_jmd94(ptem, psal, pref, kpi, kpj, dout) BIND(C, name='ref_sigmai2d_jmd94')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj), pref(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  dout(:,:) = sigmai(ptem, psal, pref, kpi, kpj, .TRUE.)
END SUBROUTINE ref_sigmai2d_jmd94

SUBROUTINE ref_sigmantr(ptem, psal, kpi, kpj, dout) BIND(C, name='ref_sigmantr')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  dout(:,:) = sigmantr(ptem, psal, kpi, kpj)
END SUBROUTINE ref_sigmantr

! ---- the rest of the module's public list: spice, alpha, beta, albet, eosbn2 (each generic member)
SUBROUTINE ref_spice(ptem, psal, kpi, kpj, kflament, dout) BIND(C, name='ref_spice')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kpi, kpj, kflament
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  IF ( kflament /= 0 ) THEN
     dout(:,:) = spice(ptem, psal, kpi, kpj)                ! spice_80
  ELSE
     dout(:,:) = spice(ptem, psal, kpi, kpj, .TRUE.)        ! spice_pol
  ENDIF
END SUBROUTINE ref_spice

! kwhich: 0 alpha_2d, 1 beta_2d, 2 albet_2d, 3 beta_2d_80
SUBROUTINE ref_coef2d(kwhich, ptem, psal, pdep, kpi, kpj, dout) BIND(C, name='ref_coef2d')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kwhich, kpi, kpj
  REAL(C_FLOAT),  VALUE :: pdep
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj), psal(kpi,kpj)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpi,kpj)
  SELECT CASE ( kwhich )
  CASE ( 0 ) ; dout(:,:) = alpha(ptem, psal, pdep, kpi, kpj)
  CASE ( 1 ) ; dout(:,:) = beta (ptem, psal, pdep, kpi, kpj)
  CASE ( 2 ) ; dout(:,:) = albet(ptem, psal, pdep, kpi, kpj)
  CASE ( 3 ) ; dout(:,:) = beta (ptem, psal, pdep, kpi, kpj, .TRUE.)
  END SELECT
END SUBROUTINE ref_coef2d

! kwhich: 0 alpha_z, 1 beta_z, 2 albet_z, 3 beta_z_80
SUBROUTINE ref_coefz(kwhich, ptem, psal, pdep, kpk, dout) BIND(C, name='ref_coefz')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kwhich, kpk
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpk), psal(kpk), pdep(kpk)
  REAL(C_DOUBLE), INTENT(out) :: dout(kpk)
  SELECT CASE ( kwhich )
  CASE ( 0 ) ; dout(:) = alpha(ptem, psal, pdep, kpk)
  CASE ( 1 ) ; dout(:) = beta (ptem, psal, pdep, kpk)
  CASE ( 2 ) ; dout(:) = albet(ptem, psal, pdep, kpk)
  CASE ( 3 ) ; dout(:) = beta (ptem, psal, pdep, kpk, .TRUE.)
  END SELECT
END SUBROUTINE ref_coefz

! kwhich: 0 alpha_0d, 1 beta_0d
SUBROUTINE ref_coef0d(kwhich, dtem, dsal, ddep, dout) BIND(C, name='ref_coef0d')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kwhich
  REAL(C_DOUBLE), VALUE :: dtem, dsal, ddep
  REAL(C_DOUBLE), INTENT(out) :: dout
  IF ( kwhich == 0 ) THEN
     dout = alpha(dtem, dsal, ddep)
  ELSE
     dout = beta (dtem, dsal, ddep)
  ENDIF
END SUBROUTINE ref_coef0d

! ptem, psal: (kpi,kpj,2) with the upper level in plane kup; kform 0 eosbn2_pol, 1 eosbn2_80
SUBROUTINE ref_bn2(kform, ptem, psal, pdep, pe3w, kpi, kpj, kup, kdown, pout) BIND(C, name='ref_bn2')
  USE ISO_C_BINDING
  USE eos
  INTEGER(C_INT), VALUE :: kform, kpi, kpj, kup, kdown
  REAL(C_FLOAT),  VALUE :: pdep
  REAL(C_FLOAT),  INTENT(in)  :: ptem(kpi,kpj,2), psal(kpi,kpj,2), pe3w(kpi,kpj)
  REAL(C_FLOAT),  INTENT(out) :: pout(kpi,kpj)
  REAL(KIND=4) :: zdep
  zdep = pdep
  IF ( kform == 0 ) THEN
     pout(:,:) = eosbn2(ptem, psal, zdep, pe3w, kpi, kpj, kup, kdown)
  ELSE
     pout(:,:) = eosbn2(ptem, psal, zdep, pe3w, kpi, kpj, kup, kdown, .TRUE.)
  ENDIF
END SUBROUTINE ref_bn2
