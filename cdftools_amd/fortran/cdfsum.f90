PROGRAM cdfsum
  !!======================================================================
  !!                     ***  PROGRAM  cdfsum  ***
  !! Sum of a field over the ocean or a 3-D window of it: per level SUM(zv*zmask*e1*e2), and its total
  !! over the levels, per record and cumulated in time.
  !!
  !! Drop-in for the reference tool (src/cdfsum.f90): same options, same lines on stdout, same
  !! cdfsum.nc (sum_<var>, sum_3D<var>, sum_3D_t<var>, :359-415) and cdfsum.txt for a file without depth.
  !! The three sums of every level (:303-327) run on the GPU for all levels of a record at once
  !! (cdft_mean_* with CDFT_MEAN_FORM_SUM: r4 products as written, r8 sums).
  !! As in the reference, -full is announced by the usage text but not known to the option parser.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: jk, jt, it, ik, narg, ijarg, ierr, ncout, numout = 10
  INTEGER(KIND=4) :: iimin = 0, iimax = 0, ijmin = 0, ijmax = 0, ikmin = 0, ikmax = 0
  INTEGER(KIND=4) :: npiglo, npjglo, npk, npkk, npt, nvpk, nlev
  INTEGER(KIND=4) :: ipk(3), id_varout(3)
  INTEGER(KIND=2), ALLOCATABLE, TARGET :: ibmask(:,:,:)
  REAL(KIND=4) :: zspval, rdummy(1,1)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1(:,:), e2(:,:)
  REAL(KIND=4), ALLOCATABLE :: gdep(:), zv(:,:,:), zmask(:,:,:)
  REAL(KIND=4), POINTER :: zv_d(:)                                 ! the record in HBM (host feed), never touched by the host
  REAL(KIND=4), ALLOCATABLE, TARGET :: e3(:,:,:)
  REAL(KIND=8) :: dvol, dsum, dsumt
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dsums(:,:,:)               ! dsums(4, 1, nlev)
  CHARACTER(LEN=256) :: cldum, cf_in = 'none', cf_e3, cf_out = 'cdfsum.nc', cv_dep, cdep, cv_in = 'none'
  CHARACTER(LEN=256) :: clunits = ' ', cllong_name = ' ', clshort_name = ' '
  CHARACTER(LEN=20)  :: cv_e1, cv_e2, cv_e3, cv_msk, cl_vmsk, cvartype = 'none'
  TYPE(variable) :: stypvar(3)
  TYPE(C_PTR) :: plan, pd_zv = C_NULL_PTR
  LOGICAL :: lfeed
  LOGICAL :: lforcing, lchk = .FALSE., lnc4 = .FALSE., lerror = .FALSE., lfmsk = .FALSE., lnwgh = .FALSE., lnmsk = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfsum.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_in   ) ; ijarg = ijarg + 1
     CASE ( '-v'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cv_in   ) ; ijarg = ijarg + 1
     CASE ( '-p'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cvartype) ; ijarg = ijarg + 1
     CASE ( '-o'      ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out  ) ; ijarg = ijarg + 1
     CASE ( '-vvl'    ) ; lg_vvl = .TRUE.
     CASE ( '-nc4'    ) ; lnc4   = .TRUE.
     CASE ( '-nowght' ) ; lnwgh  = .TRUE.
     CASE ( '-nomsk'  ) ; lnmsk  = .TRUE.
     CASE ( '-w'      ) ; CALL geti(iimin) ; CALL geti(iimax) ; CALL geti(ijmin) ; CALL geti(ijmax) ; CALL geti(ikmin) ; CALL geti(ikmax)
     CASE ( '-M'      ) ; lfmsk = .TRUE.
        ;                 CALL GET_COMMAND_ARGUMENT(ijarg, cn_fmsk) ; ijarg = ijarg + 1
        ;                 CALL GET_COMMAND_ARGUMENT(ijarg, cl_vmsk) ; ijarg = ijarg + 1
     CASE DEFAULT       ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO

  IF ( cf_in == 'none' ) THEN
     PRINT *, ' You must specify an input file with -f option' ; lerror = .TRUE.
  ENDIF
  IF ( cv_in == 'none' ) THEN
     PRINT *, ' You must specify an input variable with -v option' ; lerror = .TRUE.
  ENDIF
  IF ( cvartype == 'none' ) THEN
     PRINT *, ' You must specify a point type with -p option' ; lerror = .TRUE.
  ENDIF
  IF ( lerror ) STOP 99

  IF ( .NOT. lnwgh ) lchk = chkfile(cn_fhgr)
  IF ( .NOT. lnwgh ) lchk = chkfile(cn_fzgr) .OR. lchk
  IF ( .NOT. lnmsk ) lchk = chkfile(cn_fmsk) .OR. lchk
  lchk = chkfile(cf_in) .OR. lchk
  IF ( lchk ) STOP 99

  npiglo = getdim(cf_in, cn_x)
  npjglo = getdim(cf_in, cn_y)
  npk    = getdim(cf_in, cn_z)
  nvpk   = getvdim(cf_in, cv_in)
  npt    = getdim(cf_in, cn_t)
  npkk   = npk
  IF ( iimin /= 0 ) THEN ; npiglo = iimax - iimin + 1 ; ELSE ; iimin = 1
  ENDIF
  IF ( ijmin /= 0 ) THEN ; npjglo = ijmax - ijmin + 1 ; ELSE ; ijmin = 1
  ENDIF
  IF ( ikmin /= 0 ) THEN ; npkk   = ikmax - ikmin + 1 ; ELSE ; ikmin = 1 ; ikmax = npk
  ENDIF
  IF ( nvpk == 2 ) nvpk = 1
  IF ( nvpk == 3 ) nvpk = npkk

  PRINT *, 'Size of the extracted area :'
  PRINT *, '  npiglo = ', npiglo
  PRINT *, '  npjglo = ', npjglo
  PRINT *, '  npk    = ', npk
  PRINT *, '  nvpk   = ', nvpk
  PRINT *, '  npt    = ', npt

  lforcing = .FALSE.
  IF ( npk == 0 ) THEN
     lforcing = .TRUE.
     npk      = 1
     PRINT *, 'W A R N I N G : you used a forcing field'
  ENDIF
  IF ( lforcing ) OPEN(UNIT=numout, FILE='cdfsum.txt', FORM='formatted', STATUS='new', IOSTAT=ierr)

  nlev = MAX(nvpk, 1)
  ! large fields with the file's mask and weights: mask, e3 and every record from the file to HBM through the ring (cdffeed)
  lfeed = nvpk >= 1 .AND. .NOT. ( lnmsk .OR. lnwgh ) .AND. feed_wanted(4_8 * npiglo * npjglo * nlev)
  ALLOCATE( ibmask(1,npiglo,npjglo) )
  IF ( .NOT. lfeed ) ALLOCATE( zmask(npiglo,npjglo,nlev), zv(npiglo,npjglo,nlev), e3(npiglo,npjglo,nlev) )
  ALLOCATE( e1(npiglo,npjglo), e2(npiglo,npjglo), gdep(npk), dtim(npt), dsums(4,1,nlev) )

  SELECT CASE ( TRIM(cvartype) )                               ! metrics, mask and depth of each C-grid point (:209-258)
  CASE ( 'T' ) ; CALL setgrid(cn_ve1t, cn_ve2t, cn_fe3t, cn_ve3t, cn_ve3tvvl, cn_tmask, cn_gdept, cn_vdeptht)
  CASE ( 'U' ) ; CALL setgrid(cn_ve1u, cn_ve2u, cn_fe3u, cn_ve3u, cn_ve3uvvl, cn_umask, cn_gdept, cn_vdepthu)
  CASE ( 'V' ) ; CALL setgrid(cn_ve1v, cn_ve2v, cn_fe3v, cn_ve3v, cn_ve3vvvl, cn_vmask, cn_gdept, cn_vdepthv)
  CASE ( 'F' ) ; CALL setgrid(cn_ve1f, cn_ve2f, cn_fe3t, cn_ve3t, cn_ve3tvvl, cn_fmask, cn_gdept, cn_vdeptht)
  CASE ( 'W' ) ; CALL setgrid(cn_ve1t, cn_ve2t, cn_fe3w, cn_ve3w, cn_ve3wvvl, cn_tmask, cn_gdepw, cn_vdepthw)
  CASE DEFAULT
     PRINT *, 'this type of variable is not known :', TRIM(cvartype)
     STOP 99
  END SELECT
  IF ( lg_vvl ) cf_e3 = cf_in
  IF ( lfmsk ) cv_msk = cl_vmsk

  IF ( lnwgh ) THEN
     e1(:,:) = 1 ; e2(:,:) = 1
     gdep(:) = (/ (jk, jk = 1, npk) /)
  ELSE
     CALL getvar_to(cn_fhgr, cv_e1, 1, e1(:,:), kimin=iimin, kjmin=ijmin)
     CALL getvar_to(cn_fhgr, cv_e2, 1, e2(:,:), kimin=iimin, kjmin=ijmin)
     gdep(:) = getvare3(cn_fzgr, cv_dep, npk)
  ENDIF

  CALL CreateOutput

  ! the mask of the working levels does not change with the record
  IF ( lnmsk ) THEN ; zmask(:,:,:) = 1
  ELSE IF ( nvpk >= 1 .AND. .NOT. lfeed ) THEN ; CALL readvar3d(cn_fmsk, cv_msk, zmask, kimin=iimin, kjmin=ijmin, kkmin=ikmin)      ! (every level in one read)
  ENDIF
  ibmask(:,:,:) = 1
  IF ( nvpk >= 1 ) THEN
     CALL hip_check(cdft_mean_create(plan, npiglo, npjglo, nvpk, 1), 'cdft_mean_create')
     IF ( lforcing ) THEN ; CALL hip_check(cdft_mean_set_form(plan, CDFT_MEAN_FORM_SUM_FORCING), 'cdft_mean_set_form')
     ELSE                 ; CALL hip_check(cdft_mean_set_form(plan, CDFT_MEAN_FORM_SUM), 'cdft_mean_set_form')
     ENDIF
  ENDIF

  dsumt = 0.d0
  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     dvol = 0.d0 ; dsum = 0.d0
     IF ( lfeed ) THEN
        IF ( jt == 1 .OR. lg_vvl ) CALL feed_mean_mesh(plan, C_LOC(e1), C_LOC(e2), C_NULL_PTR, cf_e3, cv_e3, cn_fmsk, cv_msk, C_LOC(ibmask), &
             &                                         npiglo, npjglo, nvpk, 1, iimin, ijmin, ikmin, it, .NOT. lg_vvl)
        IF ( jt == 1 ) THEN
           pd_zv = feed_device(4_8 * npiglo * npjglo * nvpk)
           CALL C_F_POINTER(pd_zv, zv_d, (/ INT(npiglo, 8) * npjglo * nvpk /))
        ENDIF
        CALL feed_field(cf_in, cv_in, pd_zv, npiglo, npjglo, nvpk, ktime=jt, kimin=iimin, kjmin=ijmin, kkmin=ikmin)
        CALL hip_check(cdft_mean_compute(plan, zv_d, CDFT_MEM_DEVICE, dsums, CDFT_MEM_HOST, feed_stream), 'cdft_mean_compute')
     ENDIF
     IF ( nvpk >= 1 .AND. ( jt == 1 .OR. lg_vvl ) .AND. .NOT. lfeed ) THEN
        IF ( lnwgh ) THEN ; e3(:,:,:) = 1.
        ELSE ; CALL readvar3d(cf_e3, cv_e3, e3, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=it, ldiom=.NOT. lg_vvl)
        ENDIF
        CALL hip_check(cdft_mean_set_mesh(plan, e1, e2, C_LOC(e3), C_NULL_PTR, zmask, ibmask, CDFT_MEM_HOST, C_NULL_PTR), &
             &         'cdft_mean_set_mesh')
     ENDIF
     IF ( nvpk >= 1 .AND. .NOT. lfeed ) CALL readvar3d(cf_in, cv_in, zv, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=jt)
     IF ( nvpk >= 1 .AND. .NOT. lfeed ) CALL hip_check(cdft_mean_compute(plan, zv, CDFT_MEM_HOST, dsums, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_mean_compute')

     ! The kernel's sums per level (dsums: surface, volume, weighted sum) become the record's text and output values: this
     ! build's routine level_out below, to the output contract of src/cdfsum.f90:318-349 (lines on stdout, the forcing-file
     ! text dump, the three variables).
     DO jk = 1, nvpk
        dsum = dsum + dsums(3,1,jk)
        IF ( .NOT. lforcing ) dvol = dvol + dsums(2,1,jk)
        CALL level_out(jk, dsums(1,1,jk), dsums(2,1,jk), dsums(3,1,jk))
     END DO
     dsumt = dsumt + dsum
     IF ( .NOT. lforcing ) THEN
        PRINT *, ' Sum value over the ocean: (cumulated in time)', dsumt
        PRINT *, ' Sum value over the ocean: (function of  time)', dsum
     ENDIF
     CALL put_scalar(2, 1, REAL(dsumt))
     CALL put_scalar(3, 1, REAL(dsum))
  END DO

  PRINT *, ' mean Sum over time ', dsumt/npt
  PRINT *, ' Sum over time ', dsumt

  CLOSE(numout)
  IF ( nvpk >= 1 ) ierr = cdft_mean_destroy(plan)
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_zv) ; CALL feed_close()
  ENDIF

CONTAINS

  SUBROUTINE level_out(kk, psurf, pvol, pwsum)
    !! one working level: the text of the level and the value of the level variable (99999. for a level without water)
    INTEGER(KIND=4), INTENT(in) :: kk
    REAL(KIND=8),    INTENT(in) :: psurf, pvol, pwsum
    INTEGER(KIND=4) :: ilev
    ilev = kk + ikmin - 1
    IF ( lforcing ) THEN
       PRINT *, ' Sum value at time ', jt, ' = ', pwsum
       PRINT *, '          Surface  = ', psurf/1.d6, ' km^2'
       PRINT *, '       mean value  = ', pwsum/psurf
       WRITE(numout, '(i4," ",1e13.6)') jt, pwsum
       CALL put_scalar(1, kk, REAL(pwsum))
    ELSE IF ( pvol /= 0 ) THEN
       PRINT *, ' Sum value at level ', ilev, '(', gdep(ilev), ' m) ', pwsum
       CALL put_scalar(1, kk, REAL(pwsum))
    ELSE
       PRINT *, ' No points in the water at level ', ilev, '(', gdep(ilev), ' m) '
       CALL put_scalar(1, kk, 99999.)
    ENDIF
  END SUBROUTINE level_out

  SUBROUTINE put_scalar(kvar, kk, pval)
    INTEGER(KIND=4), INTENT(in) :: kvar, kk
    REAL(KIND=4),    INTENT(in) :: pval
    rdummy(1,1) = pval
    ierr = putvar(ncout, id_varout(kvar), rdummy, kk, 1, 1, ktime=jt)
  END SUBROUTINE put_scalar

  SUBROUTINE geti(kval)
    INTEGER(KIND=4), INTENT(out) :: kval
    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) kval
  END SUBROUTINE geti

  SUBROUTINE setgrid(cd_e1, cd_e2, cd_fe3, cd_e3, cd_e3vvl, cd_msk, cd_dep, cd_depname)
    CHARACTER(LEN=*), INTENT(in) :: cd_e1, cd_e2, cd_fe3, cd_e3, cd_e3vvl, cd_msk, cd_dep, cd_depname
    cv_e1 = cd_e1 ; cv_e2 = cd_e2 ; cf_e3 = cd_fe3 ; cv_e3 = cd_e3 ; cv_msk = cd_msk ; cv_dep = cd_dep ; cdep = cd_depname
    IF ( lg_vvl ) cv_e3 = cd_e3vvl
  END SUBROUTINE setgrid

  SUBROUTINE CreateOutput
    !! src/cdfsum.f90:359-415
    REAL(KIND=4) :: zdumlon(1,1), zdumlat(1,1)
    INTEGER(KIND=4) :: jv
    zdumlon = 0. ; zdumlat = 0.
    ierr = getvaratt(cf_in, cv_in, clunits, zspval, cllong_name, clshort_name)
    stypvar%rmissing_value    = 99999.
    stypvar%valid_min         = -1000.
    stypvar%valid_max         = 1000.
    stypvar%scale_factor      = 1.
    stypvar%add_offset        = 0.
    stypvar%savelog10         = 0.
    stypvar%conline_operation = 'N/A'
    DO jv = 1, 3
       stypvar(jv)%ichunk = (/ npiglo, MAX(1, npjglo/30), 1, 1 /)
    END DO
    ipk(1) = nvpk
    CALL defvar(1, 'sum_', 'ZT')
    IF ( lnwgh ) THEN ; stypvar(1)%cunits = TRIM(clunits)
    ELSE              ; stypvar(1)%cunits = TRIM(clunits)//'.m2'
    ENDIF
    ipk(2) = 1
    CALL defvar(2, 'sum_3D', 'T')
    stypvar(2)%cunits = TRIM(clunits)//'.m3'
    ipk(3) = 1
    CALL defvar(3, 'sum_3D_t', 'T')
    stypvar(3)%cunits = TRIM(clunits)//'.m3'

    ncout = create      (cf_out, 'none', 1, 1, nvpk, cdep=cdep, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, 3, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_in, 1, 1, npkk, pnavlon=zdumlon, pnavlat=zdumlat, pdep=gdep(ikmin:ikmax), cdep=cdep)
    dtim  = getvar1d(cf_in, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

  SUBROUTINE defvar(kv, cd_pre, cd_axis)
    INTEGER(KIND=4),  INTENT(in) :: kv
    CHARACTER(LEN=*), INTENT(in) :: cd_pre, cd_axis
    stypvar(kv)%cname       = cd_pre//TRIM(cv_in)
    IF ( kv == 1 ) THEN                                       ! (:381-382: no separator after 'sum')
       stypvar(kv)%clong_name  = 'sum'//TRIM(cllong_name)
       stypvar(kv)%cshort_name = 'sum'//TRIM(clshort_name)
    ELSE
       stypvar(kv)%clong_name  = cd_pre//TRIM(cllong_name)
       stypvar(kv)%cshort_name = cd_pre//TRIM(clshort_name)
    ENDIF
    stypvar(kv)%caxis       = cd_axis
  END SUBROUTINE defvar

END PROGRAM cdfsum
