PROGRAM cdfpsi
  !!======================================================================
  !!                     ***  PROGRAM  cdfpsi  ***
  !! Barotropic stream function from the vertically integrated transports.
  !!
  !! Drop-in for the reference tool (src/cdfpsi.f90): same options, same option report on stdout
  !! (:208-224), same psi.nc (sobarstf float (t,y,x), m3/s, history = command line, :444-505).  The
  !! vertical integration (:271-298), the integration along j or i (:377-402), -mean, the reference point
  !! and the SNGL/mask at output (:412-438) run on the GPU (cdft_psi_*, bit-identical: every sum keeps the
  !! reference's sequential order).
  !! Not carried: -open, --ssh-file and -lev stop with an explicit message.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE modutils
  USE cdftools_hip
  IMPLICIT NONE

  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, it, npiglo, npjglo, npk, npt, ncout, iiref = -1, ijref = -1
  INTEGER(KIND=4) :: ipk(1), id_varout(1)
  REAL(KIND=4), ALLOCATABLE, TARGET :: zu(:,:,:), zv(:,:,:), e3u(:,:,:), e3v(:,:,:), e1v(:,:), e2u(:,:), zmask(:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE :: glamf(:,:), gphif(:,:), zpsi(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:)
  CHARACTER(LEN=256) :: cf_ufil = 'none', cf_vfil = 'none', cf_out = 'psi.nc', cv_out = 'sobarstf', cglobal, cldum
  TYPE(variable) :: stypvar(1)
  TYPE(C_PTR) :: plan, pzu, pe2u, pe3u, pzv, pe1v, pe3v, pe31d, pmask
  LOGICAL :: lchk = .FALSE., lfull = .FALSE., lmask = .FALSE., lmean = .FALSE., lopen = .FALSE., lssh = .FALSE.
  LOGICAL :: llev = .FALSE., lnc4 = .FALSE., ll_u = .TRUE., ll_v = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
     PRINT *, ' usage : cdfpsi -u U-file -v V-file [-V] [-full ] [-mask ] [-mean] [-nc4] ...'
     PRINT *, '          ... [--ssh-file SSH-file] [-open ] [-ref iref jref ] [-o OUT-file] [-vvl] [-lev]'
     PRINT *, '      '
     PRINT *, '     PURPOSE :'
     PRINT *, '       Compute the barotropic stream function (at F points) by integrating the zonal transport'
     PRINT *, '       from south to north (default) or the meridional transport from east to west (-V).'
     PRINT *, '       MI355X build: integration and scans run on the GPU.'
     PRINT *, '      '
     PRINT *, '     ARGUMENTS :'
     PRINT *, '       -u U-file  : netcdf file with the zonal velocity.'
     PRINT *, '       -v V-file  : netcdf file with the meridional velocity.'
     PRINT *, '      '
     PRINT *, '     OPTIONS :'
     PRINT *, '       [-V ]      : use the meridional component (default: zonal component)'
     PRINT *, '       [-full ]   : full step'
     PRINT *, '       [-mask ]   : mask the output with fmask'
     PRINT *, '       [-mean ]   : average of the U and V estimates'
     PRINT *, '       [-ref iref jref] : point where psi is set to 0 (default: upper right corner)'
     PRINT *, '       [-vvl ]    : time-varying vertical metrics'
     PRINT *, '       [-nc4 ]    : accepted; the bundled NetCDF backend writes classic files'
     PRINT *, '       [-o OUT-file] : output file instead of ', TRIM(cf_out)
     PRINT *, '       [-open ] [--ssh-file SSH-file] [-lev] : accepted for compatibility, not available in this build'
     PRINT *, '      '
     PRINT *, '     REQUIRED FILES :'
     PRINT *, '       ', TRIM(cn_fhgr), ' and ', TRIM(cn_fzgr), ' (', TRIM(cn_fmsk), ' with -mask)'
     PRINT *, '      '
     PRINT *, '     OUTPUT : '
     PRINT *, '       netcdf file : ', TRIM(cf_out)
     PRINT *, '         variables : ', TRIM(cv_out), ' (m3/s )'
     STOP
  ENDIF

  CALL SetGlobalAtt(cglobal)
  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-u'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ufil) ; ijarg = ijarg + 1
     CASE ( '-v'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil) ; ijarg = ijarg + 1
     CASE ( '-V'         ) ; ll_v  = .TRUE. ; ll_u = .FALSE.
     CASE ( '-full'      ) ; lfull = .TRUE.
     CASE ( '-mask'      ) ; lmask = .TRUE.
     CASE ( '-mean'      ) ; lmean = .TRUE. ; ll_v = .TRUE. ; ll_u = .TRUE.
     CASE ( '--ssh-file' ) ; lssh  = .TRUE. ; ijarg = ijarg + 1
     CASE ( '-open'      ) ; lopen = .TRUE.
     CASE ( '-o'         ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE ( '-nc4'       ) ; lnc4  = .TRUE.
     CASE ( '-ref'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) iiref
        ;                    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) ijref
     CASE ( '-vvl'       ) ; lg_vvl = .TRUE.
     CASE ( '-lev'       ) ; llev  = .TRUE.
     CASE DEFAULT          ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  lchk = lchk .OR. chkfile(cn_fhgr)
  lchk = lchk .OR. chkfile(cn_fzgr)
  IF ( lmask ) lchk = lchk .OR. chkfile(cn_fmsk)
  lchk = lchk .OR. chkfile(cf_ufil)
  lchk = lchk .OR. chkfile(cf_vfil)
  IF ( lchk ) STOP 99
  IF ( lopen .OR. lssh .OR. llev ) THEN
     PRINT *, ' ERROR : options -open, --ssh-file and -lev are not available in the MI355X build of cdfpsi.'
     STOP 99
  ENDIF
  IF ( lg_vvl ) THEN
     cn_fe3u = cf_ufil ; cn_fe3v = cf_vfil
     cn_ve3u = cn_ve3uvvl ; cn_ve3v = cn_ve3vvvl
  ENDIF

  npiglo = getdim(cf_ufil, cn_x)
  npjglo = getdim(cf_ufil, cn_y)
  npk    = getdim(cf_ufil, cn_z)
  npt    = getdim(cf_ufil, cn_t)
  IF ( iiref == -1 .OR. ijref == -1 ) THEN
     iiref = npiglo
     ijref = npjglo
  ENDIF

  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt
  PRINT *, ' Option is use :'
  PRINT *, '    -full : ', lfull
  PRINT *, '    -mask : ', lmask
  PRINT *, '    -mean : ', lmean
  PRINT *, '    -ssh  : ', lssh
  PRINT *, '    -open : ', lopen
  PRINT *, '    -ref  : ', iiref, ijref
  PRINT *, '  U-comp  : ', ll_u
  PRINT *, '  V-comp  : ', ll_v
  PRINT *, '      -o  : ', TRIM(cf_out)
  PRINT *, '    -lnc4 : ', lnc4
  PRINT *, '    -vvl  : ', lg_vvl

  ALLOCATE( glamf(npiglo,npjglo), gphif(npiglo,npjglo), zpsi(npiglo,npjglo), dtim(npt) )
  ALLOCATE( e1v(npiglo,npjglo), e2u(npiglo,npjglo) )
  IF ( ll_u ) ALLOCATE( zu(npiglo,npjglo,npk) )
  IF ( ll_v ) ALLOCATE( zv(npiglo,npjglo,npk) )
  IF ( lfull ) THEN ; ALLOCATE( e31d(npk) )
  ELSE
     IF ( ll_u ) ALLOCATE( e3u(npiglo,npjglo,npk) )
     IF ( ll_v ) ALLOCATE( e3v(npiglo,npjglo,npk) )
  ENDIF

  glamf(:,:) = getvar(cn_fhgr, cn_glamf, 1, npiglo, npjglo)
  gphif(:,:) = getvar(cn_fhgr, cn_gphif, 1, npiglo, npjglo)
  CALL CreateOutput
  e1v(:,:) = getvar(cn_fhgr, cn_ve1v, 1, npiglo, npjglo)
  e2u(:,:) = getvar(cn_fhgr, cn_ve2u, 1, npiglo, npjglo)
  pmask = C_NULL_PTR
  IF ( lmask ) THEN
     ALLOCATE( zmask(npiglo,npjglo) )
     zmask(:,:) = getvar(cn_fmsk, cn_fmask, 1, npiglo, npjglo)
     WHERE ( zmask >= 2 ) zmask = 1                       ! get rid of the free-slip / no-slip coding
     pmask = C_LOC(zmask)
  ENDIF
  pe31d = C_NULL_PTR
  IF ( lfull ) THEN
     e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
     pe31d = C_LOC(e31d)
  ENDIF

  CALL hip_check(cdft_psi_create(plan, npiglo, npjglo, npk), 'cdft_psi_create')
  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     pzu = C_NULL_PTR ; pe2u = C_NULL_PTR ; pe3u = C_NULL_PTR ; pzv = C_NULL_PTR ; pe1v = C_NULL_PTR ; pe3v = C_NULL_PTR
     IF ( ll_v ) THEN
        zv = getvar3d(cf_vfil, cn_vomecrty, npiglo, npjglo, npk, ktime=jt)
        pzv = C_LOC(zv) ; pe1v = C_LOC(e1v)
        IF ( .NOT. lfull ) THEN
           IF ( jt == 1 .OR. lg_vvl ) THEN
              DO jk = 1, npk
                 e3v(:,:,jk) = getvar(cn_fe3v, cn_ve3v, jk, npiglo, npjglo, ktime=it, ldiom=.NOT. lg_vvl)
              END DO
           ENDIF
           pe3v = C_LOC(e3v)
        ENDIF
     ENDIF
     IF ( ll_u ) THEN
        zu = getvar3d(cf_ufil, cn_vozocrtx, npiglo, npjglo, npk, ktime=jt)
        pzu = C_LOC(zu) ; pe2u = C_LOC(e2u)
        IF ( .NOT. lfull ) THEN
           IF ( jt == 1 .OR. lg_vvl ) THEN
              DO jk = 1, npk
                 e3u(:,:,jk) = getvar(cn_fe3u, cn_ve3u, jk, npiglo, npjglo, ktime=it, ldiom=.NOT. lg_vvl)
              END DO
           ENDIF
           pe3u = C_LOC(e3u)
        ENDIF
     ENDIF
     CALL hip_check(cdft_psi_compute(plan, pzu, pe2u, pe3u, pzv, pe1v, pe3v, pe31d, INT(iiref, C_INT), INT(ijref, C_INT), &
          &         pmask, CDFT_MEM_HOST, zpsi, C_NULL_PTR, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_psi_compute')
     IF ( lmask ) THEN ; PRINT *, ' Write masked BSF'
     ELSE              ; PRINT *, ' Write BSF'
     ENDIF
     ierr = putvar(ncout, id_varout(1), zpsi, 1, npiglo, npjglo, ktime=jt)
  END DO
  ierr = cdft_psi_destroy(plan)
  ierr = closeout(ncout)

CONTAINS

  SUBROUTINE CreateOutput
    !! src/cdfpsi.f90:444-505 (single variable: no --ssh-file, no -lev)
    ipk(:) = 1
    stypvar(1)%cunits            = 'm3/s'
    stypvar(1)%valid_min         = -300.e6
    stypvar(1)%valid_max         = 300.e6
    stypvar(1)%conline_operation = 'N/A'
    stypvar(1)%caxis             = 'TYX'
    stypvar(1)%ichunk            = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    stypvar(1)%cname             = cv_out
    stypvar(1)%rmissing_value    = 0.
    stypvar(1)%clong_name        = 'Barotropic_Stream_Function'
    stypvar(1)%cshort_name       = cv_out
    ncout = create      (cf_out, cf_ufil, npiglo, npjglo, 1, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, 1, ipk, id_varout, cdglobal=TRIM(cglobal), ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_ufil, npiglo, npjglo, 1, glamf, gphif)
    dtim  = getvar1d(cf_ufil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfpsi
