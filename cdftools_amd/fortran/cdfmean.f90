PROGRAM cdfmean
  !!======================================================================
  !!                     ***  PROGRAM  cdfmean  ***
  !! Weighted mean (and variance, sum) of a field, per level and over the volume, in a window and / or
  !! per sub-basin.
  !!
  !! Drop-in for the reference tool (src/cdfmean.f90): same options, same text on stdout and in
  !! cdfmean.txt [cdfvar.txt, cdfsum.txt], same cdfmean.nc (mean_<var>, mean_3D<var> [var_, var_3D, sum_,
  !! sum_3D] [_<basin>], :556-625).  The four weighted sums of every level and basin (:405-424) run on the
  !! GPU for all levels of a record at once (cdft_mean_*: r4 products as written, r8 sums).
  !! -zeromean (:490-527) is a second, pointwise pass over the whole field that is bound by reading and writing the
  !! file: it stays on the host, as in the reference.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: jk, jt, it, ik, jbasin, ivarb, narg, ijarg, ierr, idep, idep_max
  INTEGER(KIND=4) :: iimin = 0, iimax = 0, ijmin = 0, ijmax = 0, ikmin = 0, ikmax = 0
  INTEGER(KIND=4) :: npiglo, npjglo, npk, npt, npiglo_fi, npjglo_fi, npk_fi, nvpk, nbasin = 1
  INTEGER(KIND=4) :: numout = 10, numvar = 11, numsum = 12, ikx = 1, iky = 1, nvars, nvars_bas, ncout
  INTEGER(KIND=4), ALLOCATABLE :: n_sum(:), n_sum3d(:), n_mean(:), n_mean3d(:), n_var(:), n_var3d(:), ipk(:), id_varout(:)
  INTEGER(KIND=2), ALLOCATABLE, TARGET :: ibmask(:,:,:)            ! (nbasin, npiglo, npjglo)
  REAL(KIND=4) :: zspval, rdummy(1,1)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e3(:,:,:), e31d(:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1(:,:), e2(:,:)
  REAL(KIND=4), ALLOCATABLE :: gdep(:), zdep(:), zv(:,:,:), zmask(:,:,:), zwk(:,:)
  REAL(KIND=4), POINTER :: zv_d(:)                                 ! the record in HBM (host feed), never touched by the host
  REAL(KIND=4), ALLOCATABLE :: rdumlon(:,:), rdumlat(:,:), rdummymean(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), dvmeanout(:), dvariance(:), dvol(:), dsum(:), dvar(:)
  REAL(KIND=8), ALLOCATABLE :: dvmeanout3d(:,:), dvariance3d(:,:), dsums(:,:,:)      ! dsums(4, nbasin, nvpk)
  CHARACTER(LEN=256) :: cv_nam, cv_dep, cf_e3, cf_in = 'none', cf_out = 'cdfmean.txt', cf_var = 'cdfvar.txt'
  CHARACTER(LEN=256) :: cf_sum = 'cdfsum.txt', cf_ncout = 'cdfmean.nc', cf_zerom = 'zeromean.nc', clbas = ''
  CHARACTER(LEN=256) :: cl_prefix = '', ctype = ' ', clunits = ' ', cllong_name = ' ', clshort_name = ' ', cglobal, cldum
  CHARACTER(LEN=20)  :: cv_e1, cv_e2, cv_e3, cv_e31d, cv_msk = ''
  CHARACTER(LEN=256), ALLOCATABLE :: clv_dep(:), cbasins(:)
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: plan, pe3, pe31d, pd_zv = C_NULL_PTR
  LOGICAL :: lfeed
  LOGICAL :: lbas = .FALSE., lfull = .FALSE., lvar = .FALSE., lsum = .FALSE., lzeromean = .FALSE., lnodep = .FALSE., lchk
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfmean.inc"
     STOP
  ENDIF

  cglobal = 'Partial step computation'
  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-f'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_in ) ; ijarg = ijarg + 1
     CASE ( '-v'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cv_nam) ; ijarg = ijarg + 1
     CASE ( '-p'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, ctype ) ; ijarg = ijarg + 1
     CASE ( '-full'     ) ; lfull     = .TRUE. ; cglobal = 'full step computation'
     CASE ( '-var'      ) ; lvar      = .TRUE.
     CASE ( '-zeromean' ) ; lzeromean = .TRUE.
     CASE ( '-vvl'      ) ; lg_vvl    = .TRUE.
     CASE ( '-w'        ) ; CALL geti(iimin) ; CALL geti(iimax) ; CALL geti(ijmin) ; CALL geti(ijmax) ; CALL geti(ikmin) ; CALL geti(ikmax)
     CASE ( '-o'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ncout) ; ijarg = ijarg + 1
     CASE ( '-oz'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_zerom) ; ijarg = ijarg + 1
     CASE ( '-ov'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_var  ) ; ijarg = ijarg + 1
     CASE ( '-os'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_sum  ) ; ijarg = ijarg + 1
     CASE ( '-ot'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out  ) ; ijarg = ijarg + 1
     CASE ( '-M'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cn_fmsk ) ; ijarg = ijarg + 1
        ;                   CALL GET_COMMAND_ARGUMENT(ijarg, cv_msk  ) ; ijarg = ijarg + 1
     CASE ( '-S'        ) ; lsum = .TRUE.
     CASE ( '-B'        ) ; lbas = .TRUE.
        ;                   CALL GET_COMMAND_ARGUMENT(ijarg, cn_fbasins) ; ijarg = ijarg + 1
        ;                   CALL GetLstMask
     CASE ( '-P'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cl_prefix) ; ijarg = ijarg + 1
     CASE DEFAULT         ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  cf_ncout = TRIM(cl_prefix)//TRIM(cf_ncout)
  cf_out   = TRIM(cl_prefix)//TRIM(cf_out  )
  cf_var   = TRIM(cl_prefix)//TRIM(cf_var  )
  cf_sum   = TRIM(cl_prefix)//TRIM(cf_sum  )
  cf_zerom = TRIM(cl_prefix)//TRIM(cf_zerom)

  lchk = chkfile(cn_fhgr)
  lchk = chkfile(cn_fzgr) .OR. lchk
  lchk = chkfile(cn_fmsk) .OR. lchk
  lchk = chkfile(cf_in  ) .OR. lchk
  IF ( lbas ) lchk = chkfile(cn_fbasins) .OR. lchk
  IF ( lchk ) STOP 99
  IF ( .NOT. ALLOCATED(cbasins) ) ALLOCATE( cbasins(1) )

  cv_dep = 'none'
  npiglo = getdim(cf_in, cn_x)
  npjglo = getdim(cf_in, cn_y)
  idep_max = 8
  ALLOCATE( clv_dep(idep_max) )
  clv_dep(1) = cn_z ; clv_dep(2) = 'z' ; clv_dep(3) = 'sigma' ; clv_dep(4) = 'nav_lev' ; clv_dep(5) = 'levels'
  clv_dep(6) = 'ncatice' ; clv_dep(7) = 'icbcla' ; clv_dep(8) = 'icbsect'
  idep = 1 ; ierr = 1000
  DO WHILE ( ierr /= 0 .AND. idep <= idep_max )
     npk  = getdim(cf_in, clv_dep(idep), cdtrue=cv_dep, kstatus=ierr)
     idep = idep + 1
  END DO
  IF ( ierr /= 0 ) THEN
     PRINT *, ' assume file with no depth'
     npk = 0
  ENDIF
  npt  = getdim(cf_in, cn_t)
  nvpk = getvdim(cf_in, cv_nam)
  npiglo_fi = npiglo ; npjglo_fi = npjglo ; npk_fi = npk
  IF ( npk == 0 ) THEN ; lnodep = .TRUE. ; npk = 1 ; npk_fi = 1
  ENDIF
  IF ( iimin /= 0 ) THEN ; npiglo = iimax - iimin + 1 ; ELSE ; iimin = 1
  ENDIF
  IF ( ijmin /= 0 ) THEN ; npjglo = ijmax - ijmin + 1 ; ELSE ; ijmin = 1
  ENDIF
  IF ( ikmin /= 0 ) THEN ; npk    = ikmax - ikmin + 1 ; ELSE ; ikmin = 1
  ENDIF
  IF ( nvpk == 2 ) nvpk = 1
  IF ( nvpk == 3 ) nvpk = npk

  WRITE(6, *) 'npiglo = ', npiglo
  WRITE(6, *) 'npjglo = ', npjglo
  WRITE(6, *) 'npk    = ', npk
  WRITE(6, *) 'npt    = ', npt
  WRITE(6, *) 'nvpk   = ', nvpk
  WRITE(6, *) 'depth dim name is ', TRIM(cv_dep)

  ! large fields: the mask, e3 and every record go from the file to HBM through the page-locked ring (module cdffeed)
  lfeed = feed_wanted(4_8 * npiglo * npjglo * nvpk)
  ALLOCATE( ibmask(nbasin,npiglo,npjglo), zwk(npiglo,npjglo) )
  IF ( .NOT. lfeed ) ALLOCATE( zmask(npiglo,npjglo,nvpk), zv(npiglo,npjglo,nvpk) )
  ALLOCATE( e1(npiglo,npjglo), e2(npiglo,npjglo), dvariance3d(nbasin,npt), dvmeanout3d(nbasin,npt) )
  ALLOCATE( gdep(npk), e31d(npk), dtim(npt), dvol(nbasin), dsum(nbasin), dvar(nbasin), zdep(npk_fi) )
  ALLOCATE( n_mean(nbasin), n_mean3d(nbasin), n_var(nbasin), n_var3d(nbasin), n_sum(nbasin), n_sum3d(nbasin) )
  ALLOCATE( dsums(4, nbasin, nvpk) )

  SELECT CASE ( TRIM(ctype) )                                ! metrics, mask and depth of each C-grid point (:296-355)
  CASE ( 'T' ) ; CALL setgrid(cn_ve1t, cn_ve2t, cn_fe3t, cn_ve3t, cn_ve3t1d, cn_ve3tvvl, cn_tmask, cn_gdept)
  CASE ( 'U' ) ; CALL setgrid(cn_ve1u, cn_ve2u, cn_fe3u, cn_ve3u, cn_ve3t1d, cn_ve3uvvl, cn_umask, cn_gdept)
  CASE ( 'V' ) ; CALL setgrid(cn_ve1v, cn_ve2v, cn_fe3v, cn_ve3v, cn_ve3t1d, cn_ve3vvvl, cn_vmask, cn_gdept)
  CASE ( 'F' ) ; CALL setgrid(cn_ve1f, cn_ve2f, cn_fe3t, cn_ve3t, cn_ve3t1d, cn_ve3tvvl, cn_fmask, cn_gdept)
  CASE ( 'W' ) ; CALL setgrid(cn_ve1t, cn_ve2t, cn_fe3t, cn_ve3w, cn_ve3w1d, cn_ve3wvvl, cn_tmask, cn_gdepw)
  CASE DEFAULT
     PRINT *, 'this type of variable is not known :', TRIM(ctype)
     STOP 99
  END SELECT

  CALL getvar_to(cn_fhgr, cv_e1, 1, e1(:,:), kimin=iimin, kjmin=ijmin)
  CALL getvar_to(cn_fhgr, cv_e2, 1, e2(:,:), kimin=iimin, kjmin=ijmin)
  IF ( lfull ) e31d(:) = getvare3(cn_fzgr, cv_e31d, npk)
  zdep(:) = getvare3(cn_fzgr, cv_dep, npk_fi)
  gdep(:) = zdep(ikmin:npk - ikmin + 1)

  IF ( lvar ) THEN ; nvars_bas = 4
  ELSE             ; nvars_bas = 2
  ENDIF
  IF ( lsum ) nvars_bas = nvars_bas + 2
  nvars = nvars_bas * nbasin
  ALLOCATE( stypvar(nvars), ipk(nvars), id_varout(nvars) )
  ALLOCATE( rdumlon(ikx,iky), rdumlat(ikx,iky), rdummymean(ikx,iky), dvmeanout(npk) )
  IF ( lvar ) ALLOCATE( dvariance(npk) )
  CALL CreateOutput

  IF ( lbas ) THEN
     DO jbasin = 1, nbasin
        CALL getvar_to(cn_fbasins, cbasins(jbasin), 1, zwk(:,:), kimin=iimin, kjmin=ijmin)
        ibmask(jbasin,:,:) = INT(zwk(:,:), 2)
     END DO
  ELSE
     ibmask(1,:,:) = 1
  ENDIF

  OPEN(numout, FILE=cf_out)
  IF ( lvar ) OPEN(numvar, FILE=cf_var)
  IF ( lsum ) OPEN(numsum, FILE=cf_sum)
  IF ( lg_vvl ) cf_e3 = cf_in

  ! the mask of the working levels is the same for every record
  IF ( .NOT. lfeed ) CALL readvar3d(cn_fmsk, cv_msk, zmask, kimin=iimin, kjmin=ijmin, kkmin=ikmin)      ! (every level in one read)
  IF ( .NOT. ( lfull .OR. lfeed ) ) ALLOCATE( e3(npiglo,npjglo,nvpk) )
  CALL hip_check(cdft_mean_create(plan, npiglo, npjglo, nvpk, nbasin), 'cdft_mean_create')
  ! without -var nothing reads dvar2d: the kernel then leaves its four products per cell out (the slot comes back 0)
  IF ( .NOT. lvar ) CALL hip_check(cdft_mean_set_form(plan, CDFT_MEAN_FORM_MEAN_NOVAR), 'cdft_mean_set_form')

  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     IF ( lfeed ) THEN
        IF ( jt == 1 .OR. lg_vvl ) THEN
           pe31d = C_NULL_PTR ; IF ( lfull ) pe31d = C_LOC(e31d)
           CALL feed_mean_mesh(plan, C_LOC(e1), C_LOC(e2), pe31d, cf_e3, cv_e3, cn_fmsk, cv_msk, C_LOC(ibmask), npiglo, npjglo, nvpk, &
                &              nbasin, iimin, ijmin, ikmin, it, .NOT. lg_vvl)
        ENDIF
        IF ( jt == 1 ) THEN
           pd_zv = feed_device(4_8 * npiglo * npjglo * nvpk)
           CALL C_F_POINTER(pd_zv, zv_d, (/ INT(npiglo, 8) * npjglo * nvpk /))
        ENDIF
        CALL feed_field(cf_in, cv_nam, pd_zv, npiglo, npjglo, nvpk, ktime=jt, kimin=iimin, kjmin=ijmin, kkmin=ikmin)
        CALL hip_check(cdft_mean_compute(plan, zv_d, CDFT_MEM_DEVICE, dsums, CDFT_MEM_HOST, feed_stream), 'cdft_mean_compute')
     ELSE
        IF ( jt == 1 .OR. lg_vvl ) THEN
           pe3 = C_NULL_PTR ; pe31d = C_NULL_PTR
           IF ( lfull ) THEN
              pe31d = C_LOC(e31d)                               ! e3(:,:) = e31d(jk), with jk the working-level index (:412)
           ELSE
              CALL readvar3d(cf_e3, cv_e3, e3, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=it, ldiom=.NOT. lg_vvl)
              pe3 = C_LOC(e3)
           ENDIF
           CALL hip_check(cdft_mean_set_mesh(plan, e1, e2, pe3, pe31d, zmask, ibmask, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_mean_set_mesh')
        ENDIF
        CALL readvar3d(cf_in, cv_nam, zv, kimin=iimin, kjmin=ijmin, kkmin=ikmin, ktime=jt)
        CALL hip_check(cdft_mean_compute(plan, zv, CDFT_MEM_HOST, dsums, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_mean_compute')
     ENDIF

     ! What is left for the host: the kernel's four sums per level and basin (dsums) become the record's statistics, text
     ! and output values.  Routines of this build (level_out, ocean_out below); what they must reproduce is the output
     ! contract of src/cdfmean.f90:426-480 -- the lines on stdout and in the text files, in the order level (outer), basin
     ! (inner), then the basins' volume values, and the value of every variable -- with the divisions in the order the goldens
     ! hold (tests/golden/lift_cdfmean*.npz): mean*mean for a level, the chained dsum/dvol*dsum/dvol for the volume.
     DO jbasin = 1, nbasin                               ! the volume totals: the levels added in order
        dvol(jbasin) = 0.d0 ; dsum(jbasin) = 0.d0 ; dvar(jbasin) = 0.d0
        DO jk = 1, nvpk
           dvol(jbasin) = dvol(jbasin) + dsums(2,jbasin,jk)
           dsum(jbasin) = dsum(jbasin) + dsums(3,jbasin,jk)
           dvar(jbasin) = dvar(jbasin) + dsums(4,jbasin,jk)
        END DO
     END DO
     DO jk = 1, nvpk
        DO jbasin = 1, nbasin
           CALL level_out(jbasin, jk, dsums(:,jbasin,jk))
        END DO
     END DO
     DO jbasin = 1, nbasin
        CALL ocean_out(jbasin)
     END DO
  END DO

  CLOSE(numout)
  IF ( lvar ) CLOSE(numvar)
  ierr = cdft_mean_destroy(plan)
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_zv) ; CALL feed_close()
  ENDIF

  ! -zeromean (:490-527): the field of the WHOLE domain minus the 3-D mean found above (possibly that of a window),
  ! times the masks, level by level into zeromean.nc -- a read-modify-write stream at the speed of the files
  IF ( lzeromean ) CALL ZeroMean

CONTAINS

  SUBROUTINE level_out(kb, kk, psum)
    !! one working level of one basin: psum = (surface, volume, weighted sum, weighted sum of squares)
    INTEGER(KIND=4), INTENT(in) :: kb, kk
    REAL(KIND=8),    INTENT(in) :: psum(4)
    INTEGER(KIND=4) :: ilev
    LOGICAL :: llwet
    ilev  = kk + ikmin - 1
    llwet = psum(2) /= 0
    IF ( lbas ) clbas = ' for basin '//cbasins(kb)
    dvmeanout(kk) = 99999.d0
    IF ( lvar ) dvariance(kk) = 99999.d0
    IF ( llwet ) THEN
       dvmeanout(kk) = psum(3) / psum(2)
       IF ( lvar ) dvariance(kk) = psum(4) / psum(2) - dvmeanout(kk) * dvmeanout(kk)
       WRITE(6,*) ' Mean value', TRIM(clbas), ' at level ', ilev, '(', gdep(kk), ' m) ', dvmeanout(kk), 'surface = ', psum(1)/1.e6, ' km^2'
       WRITE(numout, 9004) gdep(kk), ilev, dvmeanout(kk)
       IF ( lvar ) THEN
          WRITE(6,*) ' Variance value', TRIM(clbas), ' at level ', ilev, '(', gdep(kk), ' m) ', dvariance(kk), 'surface = ', psum(1)/1.e6, ' km^2'
          WRITE(numvar, 9004) gdep(kk), ilev, dvariance(kk)
       ENDIF
       IF ( lsum ) WRITE(numsum, 9004) gdep(kk), ilev, psum(2)
    ELSE
       WRITE(6,*) ' No points in the water at level ', ilev, '(', gdep(kk), ' m) '
    ENDIF
    CALL put_level(n_mean(kb), kk, dvmeanout(kk))
    IF ( lvar ) CALL put_level(n_var(kb), kk, dvariance(kk))
    IF ( lsum ) CALL put_level(n_sum(kb), kk, psum(3))
9004 FORMAT(f9.2,' ',i2,' ',f9.2)
  END SUBROUTINE level_out

  SUBROUTINE put_level(kvar, kk, pval)
    INTEGER(KIND=4), INTENT(in) :: kvar, kk
    REAL(KIND=8),    INTENT(in) :: pval
    rdummymean(1,1) = pval
    ierr = putvar(ncout, id_varout(kvar), rdummymean, kk, ikx, iky, ktime=jt)
  END SUBROUTINE put_level

  SUBROUTINE put_ocean(kvar, pval)
    INTEGER(KIND=4), INTENT(in) :: kvar
    REAL(KIND=8),    INTENT(in) :: pval
    rdummy(:,:) = pval
    ierr = putvar0d(ncout, id_varout(kvar), rdummy, ktime=jt)
  END SUBROUTINE put_ocean

  SUBROUTINE ocean_out(kb)
    !! the volume values of one basin from the totals dvol, dsum, dvar of the record
    INTEGER(KIND=4), INTENT(in) :: kb
    dvmeanout3d(kb,jt) = dsum(kb) / dvol(kb)
    WRITE(6,*) ' Mean value over the ocean: ', dvmeanout3d(kb,jt), jt
    WRITE(numout,*) ' Mean value over the ocean ', TRIM(clbas), ': ', dvmeanout3d(kb,jt), jt
    CALL put_ocean(n_mean3d(kb), dvmeanout3d(kb,jt))
    IF ( lvar ) THEN
       dvariance3d(kb,jt) = dvar(kb)/dvol(kb) - dsum(kb) / dvol(kb) * dsum(kb) / dvol(kb)
       WRITE(6,*) ' Variance over the ocean: ', dvariance3d(kb,jt), jt
       WRITE(numvar,*) ' Variance over the ocean ', TRIM(clbas), ': ', dvariance3d(kb,jt), jt
       CALL put_ocean(n_var3d(kb), dvariance3d(kb,jt))
    ENDIF
    IF ( lsum ) THEN
       WRITE(6,*) ' Sum over the ocean: ', dsum, jt
       WRITE(numsum,*) ' Sum over the ocean:  ', TRIM(clbas), '', dsum, jt
       CALL put_ocean(n_sum3d(kb), dsum(kb))
    ENDIF
  END SUBROUTINE ocean_out

  SUBROUTINE geti(kval)
    INTEGER(KIND=4), INTENT(out) :: kval
    CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) kval
  END SUBROUTINE geti

  SUBROUTINE setgrid(cd_e1, cd_e2, cd_fe3, cd_e3, cd_e31d, cd_e3vvl, cd_msk, cd_dep)
    CHARACTER(LEN=*), INTENT(in) :: cd_e1, cd_e2, cd_fe3, cd_e3, cd_e31d, cd_e3vvl, cd_msk, cd_dep
    cv_e1 = cd_e1 ; cv_e2 = cd_e2 ; cf_e3 = cd_fe3 ; cv_e3 = cd_e3 ; cv_e31d = cd_e31d
    IF ( lg_vvl ) cv_e3 = cd_e3vvl
    IF ( cv_msk == '' ) cv_msk = cd_msk
    cv_dep = cd_dep
  END SUBROUTINE setgrid

  SUBROUTINE CreateOutput
    !! src/cdfmean.f90:530-632
    INTEGER(KIND=4) :: ivar
    CHARACTER(LEN=80) :: cl_suffix
    rdumlon(:,:) = 0. ; rdumlat(:,:) = 0.
    ivar = 0
    ierr = getvaratt(cf_in, cv_nam, clunits, zspval, cllong_name, clshort_name)
    stypvar%cunits            = TRIM(clunits)
    stypvar%rmissing_value    = 99999.
    stypvar%valid_min         = -1000.
    stypvar%valid_max         = 1000.
    stypvar%scale_factor      = 1.
    stypvar%add_offset        = 0.
    stypvar%savelog10         = 0.
    stypvar%conline_operation = 'N/A'
    DO jbasin = 1, nbasin
       IF ( lbas ) THEN ; cl_suffix = '_'//TRIM(cbasins(jbasin))
       ELSE             ; cl_suffix = ''
       ENDIF
       ivar = ivar + 1 ; n_mean(jbasin) = ivar
       CALL defvar(ivar, nvpk, 'mean_', TRIM(clunits), 'ZT', cl_suffix)
       ivar = ivar + 1 ; n_mean3d(jbasin) = ivar
       CALL defvar(ivar, 1, 'mean_3D', TRIM(clunits), 'T', cl_suffix)
       IF ( lvar ) THEN
          ivar = ivar + 1 ; n_var(jbasin) = ivar
          CALL defvar(ivar, nvpk, 'var_', TRIM(clunits)//'^2', 'ZT', cl_suffix)
          ivar = ivar + 1 ; n_var3d(jbasin) = ivar
          CALL defvar(ivar, 1, 'var_3D', TRIM(clunits)//'^2', 'T', cl_suffix)
       ENDIF
       IF ( lsum ) THEN
          ivar = ivar + 1 ; n_sum(jbasin) = ivar
          CALL defvar(ivar, nvpk, 'sum_', TRIM(clunits)//'.m^3', 'ZT', cl_suffix)
          ivar = ivar + 1 ; n_sum3d(jbasin) = ivar
          CALL defvar(ivar, 1, 'sum_3D', TRIM(clunits)//'.m^3', 'T', cl_suffix)
       ENDIF
    END DO
    ncout = create      (cf_ncout, 'none', ikx, iky, nvpk, cdep=cv_dep)
    ierr  = createvar   (ncout, stypvar, nvars, ipk, id_varout, cdglobal=TRIM(cglobal))
    ierr  = putheadervar(ncout, cf_in, ikx, iky, npk, pnavlon=rdumlon, pnavlat=rdumlat, pdep=gdep(1:nvpk), cdep=cv_dep)
    dtim  = getvar1d(cf_in, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

  SUBROUTINE ZeroMean
    INTEGER(KIND=4) :: ivin, ivar0, jv
    INTEGER(KIND=4), ALLOCATABLE :: ipkz(:), id_varz(:)
    INTEGER(KIND=2), ALLOCATABLE :: ibfull(:,:,:)
    REAL(KIND=4),    ALLOCATABLE :: zvf(:,:), zmf(:,:), zvzm(:,:)
    CHARACTER(LEN=256), ALLOCATABLE :: cv_names(:)
    CHARACTER(LEN=80) :: cl_suffix
    TYPE(variable), ALLOCATABLE :: stypvarin(:), stypvarzero(:)
    npiglo = npiglo_fi ; npjglo = npjglo_fi
    ALLOCATE( zvf(npiglo,npjglo), zmf(npiglo,npjglo), zvzm(npiglo,npjglo), ibfull(nbasin,npiglo,npjglo) )
    ivin = getnvar(cf_in)
    ALLOCATE( stypvarin(ivin), cv_names(ivin), id_varz(nbasin), ipkz(nbasin), stypvarzero(nbasin) )
    cv_names(:) = getvarname(cf_in, ivin, stypvarin)
    ivar0 = ivin + 1
    DO jv = 1, ivin
       IF ( TRIM(cv_names(jv)) == TRIM(cv_nam) ) THEN
          ivar0 = jv ; EXIT
       ENDIF
    END DO
    IF ( ivar0 > ivin ) RETURN
    DO jbasin = 1, nbasin
       cl_suffix = '' ; IF ( lbas ) cl_suffix = '_'//TRIM(cbasins(jbasin))
       ipkz(jbasin)                          = nvpk
       ! The reference appends to the UNtrimmed 256-character strings (src/cdfmean.f90:647-653): what it appends falls off the end
       ! on assignment.  Its long_name is therefore the input's, unchanged -- kept, the reference program's own file says so
       ! (tests/golden/prog_cdfmean_zeromean.npz).  Its variable NAMES lose their basin suffix the same way, so that with -B every
       ! basin defines the same <IN-var> again and one variable is left in the file: not kept -- the names carry the suffix here.
       stypvarzero(jbasin)%cname             = TRIM(cv_nam)//TRIM(cl_suffix)
       stypvarzero(jbasin)%cunits            = stypvarin(ivar0)%cunits
       stypvarzero(jbasin)%rmissing_value    = stypvarin(ivar0)%rmissing_value
       stypvarzero(jbasin)%valid_min         = stypvarin(ivar0)%valid_min - MAXVAL(dvmeanout3d)
       stypvarzero(jbasin)%valid_max         = stypvarin(ivar0)%valid_max - MINVAL(dvmeanout3d)
       stypvarzero(jbasin)%clong_name        = stypvarin(ivar0)%clong_name
       stypvarzero(jbasin)%cshort_name       = TRIM(cv_nam)//TRIM(cl_suffix)
       stypvarzero(jbasin)%conline_operation = 'N/A'
       stypvarzero(jbasin)%caxis             = stypvarin(ivar0)%caxis
       stypvarzero(jbasin)%scale_factor      = 1.
       stypvarzero(jbasin)%add_offset        = 0.
       stypvarzero(jbasin)%savelog10         = 0.
    END DO
    ik = nvpk ; IF ( lnodep ) ik = 0                         ! no depth in the input file: none in the output
    ncout = create      (cf_zerom, cf_in, npiglo, npjglo, ik)
    ierr  = createvar   (ncout, stypvarzero, nbasin, ipkz, id_varz)
    ierr  = putheadervar(ncout, cf_in, npiglo, npjglo, ik, pdep=zdep)
    dtim  = getvar1d(cf_in, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
    IF ( lbas ) THEN
       DO jbasin = 1, nbasin
          CALL getvar_to(cn_fbasins, cbasins(jbasin), 1, zvf(:,:))
          ibfull(jbasin,:,:) = INT(zvf(:,:), 2)
       END DO
    ELSE
       ibfull(1,:,:) = 1
    ENDIF
    DO jt = 1, npt
       DO jk = 1, nvpk
          ik = jk + ikmin - 1
          CALL getvar_to(cf_in, cv_nam, ik, zvf(:,:), ktime=jt)
          CALL getvar_to(cn_fmsk, cv_msk, ik, zmf(:,:))
          DO jbasin = 1, nbasin
             zvzm(:,:) = (zvf(:,:) - dvmeanout3d(jbasin,jt)) * zmf(:,:) * ibfull(jbasin,:,:)
             ierr = putvar(ncout, id_varz(jbasin), zvzm, ik, npiglo, npjglo, ktime=jt)
          END DO
       END DO
    END DO
    ierr = closeout(ncout)
  END SUBROUTINE ZeroMean

  SUBROUTINE defvar(kv, kpk, cd_pre, cd_units, cd_axis, cd_suffix)
    INTEGER(KIND=4),  INTENT(in) :: kv, kpk
    CHARACTER(LEN=*), INTENT(in) :: cd_pre, cd_units, cd_axis, cd_suffix
    ipk(kv)                 = kpk
    stypvar(kv)%cunits      = cd_units
    stypvar(kv)%cname       = cd_pre//TRIM(cv_nam)//TRIM(cd_suffix)
    stypvar(kv)%clong_name  = cd_pre//TRIM(cllong_name)//TRIM(cd_suffix)
    stypvar(kv)%cshort_name = cd_pre//TRIM(clshort_name)//TRIM(cd_suffix)
    stypvar(kv)%caxis       = cd_axis
  END SUBROUTINE defvar

  SUBROUTINE GetLstMask
    !! -B BASIN-file name1 name2 ... : the names up to the next option (src/cdfmean.f90:670-699)
    INTEGER(KIND=4) :: iiarg
    LOGICAL :: lchkv = .FALSE.
    iiarg = ijarg
    cldum = 'xxxx'
    DO WHILE ( cldum(1:1) /= '-' .AND. iiarg <= narg )
       CALL GET_COMMAND_ARGUMENT(iiarg, cldum) ; iiarg = iiarg + 1
    END DO
    nbasin = iiarg - ijarg - 1                             ! as the reference: a list that runs to the end of the
    !                                                        command line loses its last name (put -B before another option)
    ALLOCATE( cbasins(nbasin) )
    DO jbasin = 1, nbasin
       CALL GET_COMMAND_ARGUMENT(ijarg, cbasins(jbasin)) ; ijarg = ijarg + 1
    END DO
    PRINT *, 'Basin File : ', TRIM(cn_fbasins)
    PRINT *, '  nbasin : ', nbasin
    DO jbasin = 1, nbasin
       PRINT *, '  basin ', jbasin, ' : ', TRIM(cbasins(jbasin))
    END DO
    DO jbasin = 1, nbasin
       lchkv = lchkv .OR. chkvar(cn_fbasins, cbasins(jbasin))
    END DO
    IF ( lchkv ) STOP 99
  END SUBROUTINE GetLstMask

END PROGRAM cdfmean
