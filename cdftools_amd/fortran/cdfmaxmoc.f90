PROGRAM cdfmaxmoc
  !!======================================================================
  !!                     ***  PROGRAM  cdfmaxmoc  ***
  !! Maximum and minimum of the overturning of a moc.nc file in a latitude / depth window, and where they
  !! are found.
  !!
  !! Drop-in for the reference tool (src/cdfmaxmoc.f90): a consumer of the file cdfmoc writes (one
  !! (y,depthw) slice of the first record): no grid-sized work, so nothing here runs on the GPU; it is part of
  !! the tool set because it checks the moc.nc produced by this build end to end.
  !!======================================================================
  USE cdfio
  USE modcdfnames
  IMPLICIT NONE

  INTEGER(KIND=4) :: jk, jv, npjglo, npk, narg, ijarg, nx = 1, ny = 1, nz = 1, nvarout = 6, ncout, ierr
  INTEGER(KIND=4) :: iwin(4)                   ! the window: first row, last row, first level, last level
  INTEGER(KIND=4) :: iat(2,2)                  ! (row, level) of the maximum, of the minimum
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  REAL(KIND=4) :: zext(2), zout(6), rlatmin = 0., rlatmax = 0., rdepmin = 0., rdepmax = 0.
  REAL(KIND=4), ALLOCATABLE :: gdepw(:), rdumlon(:,:), rdumlat(:,:), rlat(:,:), rmoc(:,:,:)
  REAL(KIND=8) :: dtim(1)
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  CHARACTER(LEN=256) :: cf_moc = 'none', cf_ncout = 'maxmoc.nc', cldum, cbasin = ' ', cv_in
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfmaxmoc.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_moc) ; ijarg = ijarg + 1
     CASE ( '-b' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cbasin) ; ijarg = ijarg + 1
     CASE ( '-w' ) ; CALL getr(rlatmin) ; CALL getr(rlatmax) ; CALL getr(rdepmin) ; CALL getr(rdepmax)
     CASE ( '-o' ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ncout) ; ijarg = ijarg + 1
     CASE DEFAULT  ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  IF ( chkfile(cf_moc) ) STOP 99

  npjglo = getdim(cf_moc, cn_y)
  npk    = getdim(cf_moc, cn_z)
  ALLOCATE( rmoc(1,npjglo,npk), gdepw(npk), rlat(1,npjglo) )
  gdepw(:)  = -getvar1d(cf_moc, cn_vdepthw, npk)
  rlat(:,:) =  getvar  (cf_moc, cn_vlat2d, 1, 1, npjglo)
  SELECT CASE ( cbasin )
  CASE ( 'atl' ) ; cv_in = cn_zomsfatl
  CASE ( 'glo' ) ; cv_in = cn_zomsfglo
  CASE ( 'pac' ) ; cv_in = cn_zomsfpac
  CASE ( 'inp' ) ; cv_in = cn_zomsfinp
  CASE ( 'ind' ) ; cv_in = cn_zomsfind
  CASE DEFAULT   ; PRINT *, ' basin not found !' ; STOP 99
  END SELECT
  ALLOCATE( stypvar(nvarout), ipk(nvarout), id_varout(nvarout), rdumlon(1,1), rdumlat(1,1) )
  CALL CreateOutput

  DO jk = 1, npk                                            ! one (y) row of the first record per level
     rmoc(:,:,jk) = getvar(cf_moc, cv_in, jk, 1, npjglo)
  END DO
  ! The window in index space: the LAST row / level whose latitude / depth is at or below each limit, 1 when none is
  ! (what the reference's four IF-in-a-loop statements leave behind, src/cdfmaxmoc.f90:162-173).
  iwin(:) = (/ last_at_or_below(rlat(1,:), rlatmin), last_at_or_below(rlat(1,:), rlatmax), &
       &       last_at_or_below(gdepw,     rdepmin), last_at_or_below(gdepw,     rdepmax) /)
  ! The extrema and where they are: the first one met going through the window row by row, level after level -- the order
  ! in which MAXLOC / MINLOC of the (1, y, z) section look (:175-183).
  CALL first_extreme( 1., zext(1), iat(1,1), iat(2,1))
  CALL first_extreme(-1., zext(2), iat(1,2), iat(2,2))
  PRINT *, ' Maximum ', zext(1), ' Sv latitude ', rlat(1,iat(1,1)), ' depth = ', gdepw(iat(2,1))
  PRINT *, ' Minimum ', zext(2), ' Sv latitude ', rlat(1,iat(1,2)), ' depth = ', gdepw(iat(2,2))
  zout(:) = (/ zext(1), zext(2), rlat(1,iat(1,1)), rlat(1,iat(1,2)), gdepw(iat(2,1)), gdepw(iat(2,2)) /)
  DO jv = 1, nvarout                                        ! maxmoc, minmoc, latmaxmoc, latminmoc, depthmaxmoc, depthminmoc
     ierr = putvar0d(ncout, id_varout(jv), zout(jv))
  END DO
  ierr = closeout(ncout)

CONTAINS

  SUBROUTINE getr(pval)
    REAL(KIND=4), INTENT(out) :: pval
    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) pval
  END SUBROUTINE getr

  INTEGER(KIND=4) FUNCTION last_at_or_below(pval, plim)
    REAL(KIND=4), INTENT(in) :: pval(:), plim
    INTEGER(KIND=4) :: ji
    last_at_or_below = 1
    DO ji = 1, SIZE(pval)
       IF ( pval(ji) <= plim ) last_at_or_below = ji
    END DO
  END FUNCTION last_at_or_below

  SUBROUTINE first_extreme(psign, pext, kj, kk)
    !! psign = 1.: the maximum of rmoc over the window, -1.: the minimum; (kj, kk): its row and level in the file
    REAL(KIND=4),    INTENT(in)  :: psign
    REAL(KIND=4),    INTENT(out) :: pext
    INTEGER(KIND=4), INTENT(out) :: kj, kk
    INTEGER(KIND=4) :: ij, ik
    kj = iwin(1) ; kk = iwin(3) ; pext = rmoc(1,kj,kk)
    DO ik = iwin(3), iwin(4)
       DO ij = iwin(1), iwin(2)
          IF ( psign * rmoc(1,ij,ik) > psign * pext ) THEN
             pext = rmoc(1,ij,ik) ; kj = ij ; kk = ik
          ENDIF
       END DO
    END DO
  END SUBROUTINE first_extreme

  SUBROUTINE CreateOutput
    !! src/cdfmaxmoc.f90:196-241
    rdumlon(:,:) = 0. ; rdumlat(:,:) = 0.
    ipk(:) = 1
    stypvar%rmissing_value    = 99999.
    stypvar%scale_factor      = 1.
    stypvar%add_offset        = 0.
    stypvar%savelog10         = 0.
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'T'
    stypvar(1)%cname = 'maxmoc'      ; stypvar(1)%clong_name = 'Maximum_Overturing'
    stypvar(2)%cname = 'minmoc'      ; stypvar(2)%clong_name = 'Minimum_Overtuning'
    stypvar(3)%cname = 'latmaxmoc'   ; stypvar(3)%clong_name = 'Latitude_of_Maximum_Overturing'
    stypvar(4)%cname = 'latminmoc'   ; stypvar(4)%clong_name = 'Latitude_of_Minimum_Overtuning'
    stypvar(5)%cname = 'depthmaxmoc' ; stypvar(5)%clong_name = 'Depth_of_Maximum_Overturing'
    stypvar(6)%cname = 'depthminmoc' ; stypvar(6)%clong_name = 'Depth_of_Minimum_Overtuning'
    DO jv = 1, nvarout
       stypvar(jv)%cshort_name = stypvar(jv)%cname
    END DO
    stypvar(1:2)%cunits = 'Sverdrup' ; stypvar(1:2)%valid_min = -1000.  ; stypvar(1:2)%valid_max = 1000.
    stypvar(3:4)%cunits = 'Degrees'  ; stypvar(3:4)%valid_min = -90.    ; stypvar(3:4)%valid_max = 90.
    stypvar(5:6)%cunits = 'Meters'   ; stypvar(5:6)%valid_min = -10000. ; stypvar(5:6)%valid_max = 0.
    ncout = create      (cf_ncout, 'none', nx, ny, nz, cdep=cn_vdepthw)
    ierr  = createvar   (ncout, stypvar, nvarout, ipk, id_varout)
    ierr  = putheadervar(ncout, cf_moc, nx, ny, nz, pnavlon=rdumlon, pnavlat=rdumlat, pdep=gdepw)
    dtim  = getvar1d(cf_moc, cn_vtimec, 1)
    ierr  = putvar1d(ncout, dtim, 1, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfmaxmoc
