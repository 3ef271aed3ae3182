PROGRAM cdfsigi
  !!======================================================================
  !!                     ***  PROGRAM  cdfsigi  ***
  !! Potential density referred to a given depth, from potential temperature and salinity.
  !!
  !! Drop-in for the reference tool (src/cdfsigi.f90): same options, same sigi.nc (vosigmai float
  !! (t,z,y,x), kg/m3, :150-181).  The level loop :127-141 -- sigmai(T,S,ref) * (S /= spval) stored in r4 --
  !! runs on the GPU for a whole record at once (cdft_eos_sig_tool, bit-identical to the reference's
  !! arithmetic).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE eos
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, npiglo, npjglo, npk, npkk, npt, ncout
  INTEGER(KIND=4) :: ipk(1), id_varout(1)
  REAL(KIND=4), ALLOCATABLE :: ztemp(:,:,:), zsal(:,:,:), zsigi(:,:,:)
  REAL(KIND=4) :: zsps, ref_dep = -10.
  REAL(KIND=8), ALLOCATABLE :: dtim(:)
  CHARACTER(LEN=256) :: cf_tfil = 'none', cf_sfil = 'none', cf_out = 'sigi.nc', cldum, cv_sal, cv_tem, cl_ref = ' '
  TYPE(variable) :: stypvar(1)
  LOGICAL :: lnc4 = .FALSE., ll_teos10 = .FALSE., lfeed
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfsigi.inc"
     STOP
  ENDIF

  cv_sal = cn_vosaline ; cv_tem = cn_votemper
  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-t'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_tfil) ; ijarg = ijarg + 1
     CASE ( '-s'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_sfil) ; ijarg = ijarg + 1
     CASE ( '-o'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out ) ; ijarg = ijarg + 1
     CASE ( '-r'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cl_ref ) ; ijarg = ijarg + 1 ; READ(cl_ref, *) ref_dep ; cldum = cl_ref
     CASE ( '-nc4'    ) ; lnc4 = .TRUE.
     CASE ( '-teos10' ) ; ll_teos10 = .TRUE.
     CASE DEFAULT       ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  CALL eos_init( ll_teos10 )
  IF ( cf_sfil == 'none' ) cf_sfil = cf_tfil
  IF ( chkfile(cf_tfil) .OR. chkfile(cf_sfil) ) STOP 99

  zsps   = getatt(cf_sfil, cn_vosaline, cn_missing_value)   ! src/cdfsigi.f90:125
  npiglo = getdim(cf_tfil, cn_x)
  npjglo = getdim(cf_tfil, cn_y)
  npk    = getdim(cf_tfil, cn_z)
  npt    = getdim(cf_tfil, cn_t)
  npkk = npk ; IF ( npk == 0 ) npkk = 1

  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt

  ! large fields: a slab of levels at a time through the page-locked ring (module cdffeed), nothing record-sized on the host
  lfeed = npk > 0 .AND. feed_wanted(4_8 * npiglo * npjglo * npkk)
  ALLOCATE( dtim(npt) )
  IF ( .NOT. lfeed ) ALLOCATE( ztemp(npiglo,npjglo,npkk), zsal(npiglo,npjglo,npkk), zsigi(npiglo,npjglo,npkk) )
  CALL CreateOutput

  DO jt = 1, npt
     PRINT *, 'time: ', jt
     IF ( lfeed ) THEN
        CALL feed_sig_record(cf_tfil, cv_tem, cf_sfil, cv_sal, npiglo, npjglo, npkk, jt, ref_dep, zsps,  &
             &               MERGE(CDFT_EOS_TEOS10, CDFT_EOS_80, ll_teos10), ncout, id_varout(1))
        CYCLE
     ENDIF
     CALL readvar3d(cf_tfil, cv_tem, ztemp, ktime=jt)
     CALL readvar3d(cf_sfil, cv_sal, zsal, ktime=jt)
     CALL hip_check(cdft_eos_sig_tool(ztemp, zsal, INT(npiglo, C_INT64_T) * npjglo * npkk, ref_dep, zsps,           &
          &         MERGE(CDFT_EOS_TEOS10, CDFT_EOS_80, ll_teos10), 0_C_INT, zsigi, CDFT_MEM_HOST, CDFT_MEM_HOST, &
          &         C_NULL_PTR), 'cdft_eos_sig_tool')
     DO jk = 1, npkk
        ierr = putvar(ncout, id_varout(1), zsigi(:,:,jk), jk, npiglo, npjglo, ktime=jt)
     END DO
  END DO
  ierr = closeout(ncout)
  IF ( lfeed ) CALL feed_close()

CONTAINS

  SUBROUTINE CreateOutput
    ipk(:)                       = npkk
    stypvar(1)%cname             = cn_vosigmai
    stypvar(1)%cunits            = 'kg/m3'
    stypvar(1)%rmissing_value    = 0.
    stypvar(1)%valid_min         = 0.001
    stypvar(1)%valid_max         = 45.
    ! the reference names the depth by its parser's scratch string (src/cdfsigi.f90:98, :166): the value of -r when nothing
    ! but value-less options follow it on the command line, else the last option the parser looked at -- kept, file for file
    stypvar(1)%clong_name        = 'Potential_density:refered to '//TRIM(cldum)//' m'
    stypvar(1)%cshort_name       = cn_vosigmai
    stypvar(1)%conline_operation = 'N/A'
    stypvar(1)%caxis             = 'TZYX'
    stypvar(1)%ichunk            = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    ncout = create      (cf_out, cf_tfil, npiglo, npjglo, npk, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, 1, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_tfil, npiglo, npjglo, npk)
    dtim  = getvar1d(cf_tfil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfsigi
