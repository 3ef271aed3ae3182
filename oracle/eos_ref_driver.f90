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
