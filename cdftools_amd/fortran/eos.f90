MODULE eos
  !!======================================================================
  !! eos : equation of state -- the reference's module surface (src/eos.f90:24-56): eos_init, sigma0, the
  !! generic sigmai with its four specifics, sigmantr, and the generics spice, eosbn2, albet, alpha, beta with
  !! every specific.  The arithmetic runs on the GPU (cdft_eos_* of libcdftools_hip.so, bit-identical to the
  !! reference's compiled loops: tests/test_eos.py, tests/test_eos_more.py); this module only keeps the interface.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdftools_hip
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: eos_init, sigma0, sigmai, spice, sigmantr, eosbn2, albet, alpha, beta

  INTEGER(C_INT) :: neos = CDFT_EOS_80          ! what eos_init selected

  INTERFACE sigmai
     MODULE PROCEDURE sigmai_dep, sigmai_dep2d, sigmai_dep_80, sigmai_dep2d_80
  END INTERFACE

  INTERFACE spice
     MODULE PROCEDURE spice_pol, spice_80
  END INTERFACE
  INTERFACE eosbn2
     MODULE PROCEDURE eosbn2_pol, eosbn2_80
  END INTERFACE
  INTERFACE albet
     MODULE PROCEDURE albet_2d, albet_z
  END INTERFACE
  INTERFACE alpha
     MODULE PROCEDURE alpha_0d, alpha_2d, alpha_z
  END INTERFACE
  INTERFACE beta
     MODULE PROCEDURE beta_0d, beta_2d, beta_z, beta_2d_80, beta_z_80
  END INTERFACE

CONTAINS

  SUBROUTINE eos_init(ld_teos10)
    !! src/eos.f90:207-598: selects the TEOS-10 or the EOS-80 coefficient set of the polynomial EOS
    LOGICAL, INTENT(in) :: ld_teos10
    IF ( ld_teos10 ) THEN
       PRINT *, ' Equation of state for sea-water : TEOS-10 '
       neos = CDFT_EOS_TEOS10
    ELSE
       PRINT *, ' Equation of state for sea-water : EOS-80 '
       neos = CDFT_EOS_80
    ENDIF
  END SUBROUTINE eos_init

  FUNCTION sigma0(ptem, psal, kpi, kpj)
    !! src/eos.f90:601-632
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=8), DIMENSION(kpi,kpj) :: sigma0
    sigma0(:,:) = sigmai_dep(ptem, psal, 0.0, kpi, kpj)
  END FUNCTION sigma0

  FUNCTION sigmai_dep(ptem, psal, pref, kpi, kpj)
    !! src/eos.f90:802-887
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=4),                     INTENT(in) :: pref
    REAL(KIND=8), DIMENSION(kpi,kpj) :: sigmai_dep
    CALL hip_check(cdft_eos_sigmai(ptem, psal, INT(kpi, C_INT64_T) * kpj, pref, neos, 0_C_INT, sigmai_dep, &
         &         CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'sigmai')
  END FUNCTION sigmai_dep

  FUNCTION sigmai_dep2d(ptem, psal, pref, kpi, kpj)
    !! src/eos.f90:1002-1087
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal, pref
    REAL(KIND=8), DIMENSION(kpi,kpj) :: sigmai_dep2d
    CALL hip_check(cdft_eos_sigmai2d(ptem, psal, pref, INT(kpi, C_INT64_T) * kpj, neos, 0_C_INT, sigmai_dep2d, &
         &         CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'sigmai (2d depth)')
  END FUNCTION sigmai_dep2d

  FUNCTION sigmai_dep_80(ptem, psal, pref, kpi, kpj, ld_eos80)
    !! src/eos.f90:889-1000 (Jackett and McDougall 1994); ld_eos80 only selects this specific
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=4),                     INTENT(in) :: pref
    LOGICAL,                          INTENT(in) :: ld_eos80
    REAL(KIND=8), DIMENSION(kpi,kpj) :: sigmai_dep_80
    CALL hip_check(cdft_eos_sigmai(ptem, psal, INT(kpi, C_INT64_T) * kpj, pref, CDFT_EOS_JMD94, 0_C_INT, sigmai_dep_80, &
         &         CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'sigmai (JMD94)')
  END FUNCTION sigmai_dep_80

  FUNCTION sigmai_dep2d_80(ptem, psal, pref, kpi, kpj, ld_eos80)
    !! src/eos.f90:1089-1201
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal, pref
    LOGICAL,                          INTENT(in) :: ld_eos80
    REAL(KIND=8), DIMENSION(kpi,kpj) :: sigmai_dep2d_80
    CALL hip_check(cdft_eos_sigmai2d(ptem, psal, pref, INT(kpi, C_INT64_T) * kpj, CDFT_EOS_JMD94, 0_C_INT,           &
         &         sigmai_dep2d_80, CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'sigmai (JMD94, 2d depth)')
  END FUNCTION sigmai_dep2d_80

  FUNCTION sigmantr(ptem, psal, kpi, kpj)
    !! src/eos.f90:634-687: neutral density of McDougall and Jackett (2005) - 1000, not masked
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=8), DIMENSION(kpi,kpj) :: sigmantr
    CALL hip_check(cdft_eos_sigmai(ptem, psal, INT(kpi, C_INT64_T) * kpj, 0.0, CDFT_EOS_NTR, 0_C_INT, sigmantr, &
         &         CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'sigmantr')
  END FUNCTION sigmantr

  ! ---- spiciness --------------------------------------------------------------------------------------------
  FUNCTION spice_80(ptem, psal, kpi, kpj)
    !! src/eos.f90:689-753 (Flament 2002)
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=8), DIMENSION(kpi,kpj) :: spice_80
    CALL field(CDFT_EOSQ_SPICE_F02, ptem, psal, 0.0, kpi * kpj, spice_80, 'spice (Flament 2002)')
  END FUNCTION spice_80

  FUNCTION spice_pol(ptem, psal, kpi, kpj, ld_teos10)
    !! src/eos.f90:755-800 (spiciness-0 of the GSW toolbox); ld_teos10 only selects this specific
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    LOGICAL,                          INTENT(in) :: ld_teos10
    REAL(KIND=8), DIMENSION(kpi,kpj) :: spice_pol
    CALL field(CDFT_EOSQ_SPICE_POL, ptem, psal, 0.0, kpi * kpj, spice_pol, 'spice (GSW)')
  END FUNCTION spice_pol

  ! ---- Brunt-Vaisala frequency ------------------------------------------------------------------------------
  FUNCTION eosbn2_pol(ptem, psal, pdep, pe3w, kpi, kpj, kup, kdown)
    !! src/eos.f90:1203-1244
    INTEGER(KIND=4),                    INTENT(in) :: kpi, kpj, kup, kdown
    REAL(KIND=4), DIMENSION(kpi,kpj,2), INTENT(in) :: ptem, psal
    REAL(KIND=4)                                   :: pdep
    REAL(KIND=4), DIMENSION(kpi,kpj),   INTENT(in) :: pe3w
    REAL(KIND=4), DIMENSION(kpi,kpj) :: eosbn2_pol
    CALL hip_check(cdft_eos_bn2(ptem(:,:,kup), ptem(:,:,kdown), psal(:,:,kup), psal(:,:,kdown), pe3w, pdep,           &
         &         INT(kpi, C_INT64_T) * kpj, neos, eosbn2_pol, CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'eosbn2')
  END FUNCTION eosbn2_pol

  FUNCTION eosbn2_80(ptem, psal, pdep, pe3w, kpi, kpj, kup, kdown, ld_eos80)
    !! src/eos.f90:1246-1312 (McDougall 1987); ld_eos80 only selects this specific
    INTEGER(KIND=4),                    INTENT(in) :: kpi, kpj, kup, kdown
    REAL(KIND=4), DIMENSION(kpi,kpj,2), INTENT(in) :: ptem, psal
    REAL(KIND=4)                                   :: pdep
    REAL(KIND=4), DIMENSION(kpi,kpj),   INTENT(in) :: pe3w
    LOGICAL,                            INTENT(in) :: ld_eos80
    REAL(KIND=4), DIMENSION(kpi,kpj) :: eosbn2_80
    CALL hip_check(cdft_eos_bn2(ptem(:,:,kup), ptem(:,:,kdown), psal(:,:,kup), psal(:,:,kdown), pe3w, pdep,           &
         &         INT(kpi, C_INT64_T) * kpj, CDFT_EOS_MCD87, eosbn2_80, CDFT_MEM_HOST, CDFT_MEM_HOST, C_NULL_PTR), 'eosbn2 (McDougall 1987)')
  END FUNCTION eosbn2_80

  ! ---- alpha / beta / their ratio ---------------------------------------------------------------------------
  SUBROUTINE field(kq, ptem, psal, pdep, kn, dout, cdwhat, pdepz)
    ! one call of cdft_eos_field on kn points; pdepz: a depth per point
    INTEGER(C_INT),   INTENT(in)  :: kq
    INTEGER(KIND=4),  INTENT(in)  :: kn
    REAL(KIND=4),     INTENT(in)  :: ptem(kn), psal(kn), pdep
    REAL(KIND=8),     INTENT(out) :: dout(kn)
    CHARACTER(LEN=*), INTENT(in)  :: cdwhat
    REAL(KIND=4), OPTIONAL, TARGET, INTENT(in) :: pdepz(kn)
    TYPE(C_PTR) :: cp
    cp = C_NULL_PTR
    IF ( PRESENT(pdepz) ) cp = C_LOC(pdepz)
    CALL hip_check(cdft_eos_field(kq, ptem, psal, cp, pdep, INT(kn, C_INT64_T), neos, dout, CDFT_MEM_HOST, CDFT_MEM_HOST, &
         &         C_NULL_PTR), cdwhat)
  END SUBROUTINE field

  FUNCTION albet_2d(ptem, psal, pdep, kpi, kpj)
    !! src/eos.f90:1315-1356
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=4),                     INTENT(in) :: pdep
    REAL(KIND=8), DIMENSION(kpi,kpj) :: albet_2d
    CALL field(CDFT_EOSQ_ALBET_MCD87, ptem, psal, pdep, kpi * kpj, albet_2d, 'albet')
  END FUNCTION albet_2d

  FUNCTION albet_z(ptem, psal, pdep, kpk)
    !! src/eos.f90:1359-1398
    INTEGER(KIND=4),              INTENT(in) :: kpk
    REAL(KIND=4), DIMENSION(kpk), INTENT(in) :: ptem, psal, pdep
    REAL(KIND=8), DIMENSION(kpk) :: albet_z
    CALL field(CDFT_EOSQ_ALBET_MCD87, ptem, psal, 0.0, kpk, albet_z, 'albet (profile)', pdepz=pdep)
  END FUNCTION albet_z

  FUNCTION beta_2d(ptem, psal, pdep, kpi, kpj)
    !! src/eos.f90:1400-1450
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=4),                     INTENT(in) :: pdep
    REAL(KIND=8), DIMENSION(kpi,kpj) :: beta_2d
    CALL field(CDFT_EOSQ_BETA, ptem, psal, pdep, kpi * kpj, beta_2d, 'beta')
  END FUNCTION beta_2d

  FUNCTION alpha_2d(ptem, psal, pdep, kpi, kpj)
    !! src/eos.f90:1452-1503
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=4),                     INTENT(in) :: pdep
    REAL(KIND=8), DIMENSION(kpi,kpj) :: alpha_2d
    CALL field(CDFT_EOSQ_ALPHA, ptem, psal, pdep, kpi * kpj, alpha_2d, 'alpha')
  END FUNCTION alpha_2d

  FUNCTION alpha_0d(ptem, psal, pdep)
    !! src/eos.f90:1505-1543
    REAL(KIND=8), INTENT(in) :: ptem, psal, pdep
    REAL(KIND=8) :: alpha_0d
    CALL hip_check(cdft_eos_point(CDFT_EOSQ_ALPHA, ptem, psal, pdep, neos, alpha_0d), 'alpha (one point)')
  END FUNCTION alpha_0d

  FUNCTION alpha_z(ptem, psal, pdep, kpk)
    !! src/eos.f90:1545-1594
    INTEGER(KIND=4),              INTENT(in) :: kpk
    REAL(KIND=4), DIMENSION(kpk), INTENT(in) :: ptem, psal, pdep
    REAL(KIND=8), DIMENSION(kpk) :: alpha_z
    CALL field(CDFT_EOSQ_ALPHA, ptem, psal, 0.0, kpk, alpha_z, 'alpha (profile)', pdepz=pdep)
  END FUNCTION alpha_z

  FUNCTION beta_0d(ptem, psal, pdep)
    !! src/eos.f90:1596-1641
    REAL(KIND=8), INTENT(in) :: ptem, psal, pdep
    REAL(KIND=8) :: beta_0d
    CALL hip_check(cdft_eos_point(CDFT_EOSQ_BETA, ptem, psal, pdep, neos, beta_0d), 'beta (one point)')
  END FUNCTION beta_0d

  FUNCTION beta_z(ptem, psal, pdep, kpk)
    !! src/eos.f90:1643-1691
    INTEGER(KIND=4),              INTENT(in) :: kpk
    REAL(KIND=4), DIMENSION(kpk), INTENT(in) :: ptem, psal, pdep
    REAL(KIND=8), DIMENSION(kpk) :: beta_z
    CALL field(CDFT_EOSQ_BETA, ptem, psal, 0.0, kpk, beta_z, 'beta (profile)', pdepz=pdep)
  END FUNCTION beta_z

  FUNCTION beta_2d_80(ptem, psal, pdep, kpi, kpj, ld_eos80)
    !! src/eos.f90:1693-1734 (McDougall 1987); ld_eos80 only selects this specific
    INTEGER(KIND=4),                  INTENT(in) :: kpi, kpj
    REAL(KIND=4), DIMENSION(kpi,kpj), INTENT(in) :: ptem, psal
    REAL(KIND=4),                     INTENT(in) :: pdep
    LOGICAL,                          INTENT(in) :: ld_eos80
    REAL(KIND=8), DIMENSION(kpi,kpj) :: beta_2d_80
    CALL field(CDFT_EOSQ_BETA_MCD87, ptem, psal, pdep, kpi * kpj, beta_2d_80, 'beta (McDougall 1987)')
  END FUNCTION beta_2d_80

  FUNCTION beta_z_80(ptem, psal, pdep, kpk, ld_eos80)
    !! src/eos.f90:1736-1775
    INTEGER(KIND=4),              INTENT(in) :: kpk
    REAL(KIND=4), DIMENSION(kpk), INTENT(in) :: ptem, psal, pdep
    LOGICAL,                      INTENT(in) :: ld_eos80
    REAL(KIND=8), DIMENSION(kpk) :: beta_z_80
    CALL field(CDFT_EOSQ_BETA_MCD87, ptem, psal, 0.0, kpk, beta_z_80, 'beta (McDougall 1987, profile)', pdepz=pdep)
  END FUNCTION beta_z_80

END MODULE eos
