MODULE modutils
  !!======================================================================
  !! modutils : the helpers of the reference's module (src/modutils.f90) that the hot-path tools call:
  !!   SetGlobalAtt (:45-83)   command line -> 'history' attribute text
  !!   SetFileName  (:85-114)  <CONF>_<TAG>_grid<X>.nc or <CONF>_<TAG>_grid_<X>.nc, STOP 97 when neither exists
  !!   GetList      (:116-169) '1-3,6,10-12' -> integer list
  !! and heading (:934-973), the rhumb-line bearing the section tools use.  The pool-filling and Shapiro routines of
  !! the reference's module (:171-932) belong to tools outside the hot path and are not provided.
  !!======================================================================
  USE cdfio, ONLY : chkfile
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: SetGlobalAtt, SetFileName, GetList, heading

CONTAINS

  SUBROUTINE SetGlobalAtt(cdglobal, cd_append)
    CHARACTER(LEN=*),           INTENT(inout) :: cdglobal
    CHARACTER(LEN=1), OPTIONAL, INTENT(in   ) :: cd_append       ! present: append to cdglobal after ' ; '
    CHARACTER(LEN=100) :: cl_arg
    INTEGER(KIND=4) :: jarg
    CALL GET_COMMAND_ARGUMENT(0, cl_arg)
    IF ( PRESENT(cd_append) ) THEN ; cdglobal = TRIM(cdglobal)//' ; '//TRIM(cl_arg)
    ELSE                           ; cdglobal = TRIM(cl_arg)
    ENDIF
    DO jarg = 1, COMMAND_ARGUMENT_COUNT()
       CALL GET_COMMAND_ARGUMENT(jarg, cl_arg)
       cdglobal = TRIM(cdglobal)//' '//TRIM(cl_arg)
    END DO
  END SUBROUTINE SetGlobalAtt

  CHARACTER(LEN=256) FUNCTION SetFileName(cdconf, cdtag, cdgrid, ld_stop)
    CHARACTER(LEN=*),  INTENT(in) :: cdconf, cdtag, cdgrid
    LOGICAL, OPTIONAL, INTENT(in) :: ld_stop
    LOGICAL :: ll_stop
    ll_stop = .TRUE. ; IF ( PRESENT(ld_stop) ) ll_stop = ld_stop
    SetFileName = TRIM(cdconf)//'_'//TRIM(cdtag)//'_grid'//TRIM(cdgrid)//'.nc'
    IF ( .NOT. chkfile(SetFileName, ld_verbose=.FALSE.) ) RETURN
    SetFileName = TRIM(cdconf)//'_'//TRIM(cdtag)//'_grid_'//TRIM(cdgrid)//'.nc'
    IF ( chkfile(SetFileName, ld_verbose=.FALSE.) .AND. ll_stop ) THEN
       PRINT *, ' ERROR : missing grid', TRIM(cdgrid), 'or even grid_', TRIM(cdgrid), ' file '
       STOP 97
    ENDIF
  END FUNCTION SetFileName

  SUBROUTINE GetList(cd_list, klist, ksiz)
    CHARACTER(LEN=*),                           INTENT(in ) :: cd_list
    INTEGER(KIND=4), DIMENSION(:), ALLOCATABLE, INTENT(out) :: klist
    INTEGER(KIND=4),                            INTENT(out) :: ksiz
    INTEGER(KIND=4), PARAMETER :: jp_maxlist = 500
    INTEGER(KIND=4) :: itmp(jp_maxlist), ia, ib, ic, id, i1, i2, jk
    ksiz = 0 ; ia = 1
    DO WHILE ( ia <= LEN_TRIM(cd_list) )
       ib = INDEX(cd_list(ia:), ',')
       IF ( ib == 0 ) THEN ; ic = LEN_TRIM(cd_list) ; ELSE ; ic = ia + ib - 2 ; ENDIF
       id = INDEX(cd_list(ia:ic), '-')
       IF ( id == 0 ) THEN
          READ(cd_list(ia:ic), *) i1 ; i2 = i1
       ELSE
          READ(cd_list(ia:ia+id-2), *) i1 ; READ(cd_list(ia+id:ic), *) i2
       ENDIF
       DO jk = i1, i2
          ksiz = ksiz + 1
          IF ( ksiz > jp_maxlist ) THEN ; PRINT *, 'jp_maxlist too small in getlist ' ; STOP 97 ; ENDIF
          itmp(ksiz) = jk
       END DO
       ia = ic + 2
    END DO
    ALLOCATE( klist(ksiz) )
    klist(:) = itmp(1:ksiz)
  END SUBROUTINE GetList

  REAL(KIND=8) FUNCTION heading(dplona, dplonb, dplata, dplatb)
    !! true heading (degrees, 0..360) from A to B along the rhumb line: the bearing of the segment on a Mercator map
    !! (src/modutils.f90:934-973).  The longitude difference is folded into (-pi, pi) so that a segment crossing the
    !! date line is measured the short way.
    REAL(KIND=8), INTENT(in) :: dplona, dplonb, dplata, dplatb
    REAL(KIND=8) :: dlpi, dlrad, dldx, dlya, dlyb
    dlpi  = ACOS(-1.d0)
    dlrad = dlpi/180.d0
    dlya  = -LOG(TAN(dlpi/4. - dlrad*dplata/2.d0))          ! Mercator ordinates
    dlyb  = -LOG(TAN(dlpi/4. - dlrad*dplatb/2.d0))
    dldx  = MOD(dplonb*dlrad - dplona*dlrad, 2*dlpi)
    IF ( dldx >=  dlpi ) dldx = dldx - 2*dlpi
    IF ( dldx <= -dlpi ) dldx = dldx + 2*dlpi
    heading = ATAN2(dldx, dlyb - dlya)*180.d0/dlpi
    IF ( heading < 0 ) heading = heading + 360.d0
  END FUNCTION heading

END MODULE modutils
