MODULE eos
  !!======================================================================
  !! eos : equation of state -- the part of the reference's module surface the hot path uses
  !! (src/eos.f90:24-36): eos_init, sigma0 and the generic sigmai with its four specifics.  The
  !! arithmetic runs on the GPU (cdft_eos_* of libcdftools_hip.so, bit-identical to the reference's
  !! loops: tests/test_eos.py); this module only keeps the interface.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdftools_hip
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: eos_init, sigma0, sigmai

  INTEGER(C_INT) :: neos = CDFT_EOS_80          ! what eos_init selected

  INTERFACE sigmai
     MODULE PROCEDURE sigmai_dep, sigmai_dep2d, sigmai_dep_80, sigmai_dep2d_80
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

END MODULE eos
