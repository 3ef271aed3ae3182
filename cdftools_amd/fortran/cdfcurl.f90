PROGRAM cdfcurl
  !!======================================================================
  !!                     ***  PROGRAM  cdfcurl  ***
  !! Relative vorticity (curl of the horizontal velocity) at F points, for a list of levels.
  !!
  !! Drop-in for the reference tool (src/cdfcurl.f90): same options (-u file var, -v file var, -l levels,
  !! -T, -8, -surf, -overf, -nometric, -nc4, -o), same curl.nc (socurl | socurlt | socurloverf, float or
  !! double, :395-438).  The level loop :242-304 runs on the GPU for all selected levels of a record at
  !! once (cdft_curl_*, bit-identical to the reference's r4 expression).
  !! Level 0 (forcing fields on the A grid, :250-258): the 2-point averages to U / V points are host work,
  !! the curl itself is the same kernel.
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: narg, ijarg, ierr, jt, jk, ji, jj, npiglo, npjglo, npk, npt, nlev = 0, ncout
  INTEGER(KIND=4) :: ipk(1), id_varout(1)
  INTEGER(KIND=4), ALLOCATABLE :: nilev(:)
  REAL(KIND=4), ALLOCATABLE :: e1u(:,:), e1f(:,:), e2v(:,:), e2f(:,:), zlon(:,:), zlat(:,:)
  REAL(KIND=4), ALLOCATABLE :: un(:,:,:), vn(:,:,:), gdep(:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: zrot4(:,:,:)
  REAL(KIND=8), ALLOCATABLE, TARGET :: drotn(:,:,:), dl_ff(:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:), zdep(:)
  REAL(KIND=8) :: dl_pi, dl_omega
  CHARACTER(LEN=256) :: cf_ufil = 'none', cf_vfil = 'none', cf_out = 'curl.nc', cv_u = 'none', cv_v = 'none', cldum
  TYPE(variable) :: stypvar(1)
  TYPE(C_PTR) :: plan, pff
  TYPE(C_PTR) :: pd_u = C_NULL_PTR, pd_v = C_NULL_PTR, ph_rot(2) = C_NULL_PTR      ! the host feed's level in HBM, two page-locked result levels
  REAL(KIND=4), POINTER :: un_d(:), vn_d(:)
  INTEGER(KIND=8) :: nij8
  LOGICAL :: lfeed
  LOGICAL :: lchk = .FALSE., lperio = .FALSE., ldblpr = .FALSE., lnc4 = .FALSE., ltpoint = .FALSE., lsurf = .FALSE.
  LOGICAL :: loverf = .FALSE., l_metric = .TRUE., lforcing = .FALSE.
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfcurl.inc"
     STOP
  ENDIF

  ijarg = 1
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-u'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_ufil) ; ijarg = ijarg + 1
        ;                   CALL GET_COMMAND_ARGUMENT(ijarg, cv_u   ) ; ijarg = ijarg + 1
     CASE ( '-v'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil) ; ijarg = ijarg + 1
        ;                   CALL GET_COMMAND_ARGUMENT(ijarg, cv_v   ) ; ijarg = ijarg + 1
     CASE ( '-l'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum  ) ; ijarg = ijarg + 1
        ;                   CALL ParseLevel(cldum)
     CASE ( '-nc4'      ) ; lnc4    = .TRUE.
     CASE ( '-T'        ) ; ltpoint = .TRUE.
     CASE ( '-8'        ) ; ldblpr  = .TRUE.
     CASE ( '-surf'     ) ; lsurf   = .TRUE.
     CASE ( '-overf'    ) ; loverf  = .TRUE.
     CASE ( '-nometric' ) ; l_metric = .FALSE. ; cf_out = 'curl_grid.nc'
     CASE ( '-o'        ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_out) ; ijarg = ijarg + 1
     CASE DEFAULT         ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  IF ( ltpoint .AND. loverf ) THEN
     PRINT *, ' ERROR : You might choose only one of -T or -overf option !'
     STOP 99
  ENDIF

  IF ( l_metric ) lchk = chkfile(cn_fhgr) .OR. lchk
  lchk = chkfile(cf_ufil) .OR. lchk
  lchk = chkfile(cf_vfil) .OR. lchk
  IF ( lchk ) STOP 99

  npiglo = getdim(cf_ufil, cn_x)
  npjglo = getdim(cf_ufil, cn_y)
  npk    = getdim(cf_ufil, cn_z)
  npt    = getdim(cf_ufil, cn_t)

  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt
  PRINT *, 'nlev   = ', nlev

  IF ( npk == 0 .AND. nlev > 0 .AND. .NOT. lsurf ) THEN
     PRINT *, 'Problem : npk = 0 and lev > 0 STOP 99'
     PRINT *, '  Use -surf option is dealing with single level file on C grid '
     STOP 99
  ENDIF
  IF ( lsurf ) THEN
     nlev = 1
     IF ( ALLOCATED(nilev) ) DEALLOCATE(nilev)
     ALLOCATE( nilev(1) )
     npk = 1 ; nilev(1) = 1
  ENDIF
  IF ( .NOT. ALLOCATED(nilev) ) THEN
     PRINT *, ' ERROR : option -l levlist is mandatory.' ; STOP 99
  ENDIF
  IF ( nilev(1) == 0 ) THEN                                  ! forcing fields (A grid, no depth), :170-174
     lforcing = .TRUE.
     npk = 1 ; nilev(1) = 1
     PRINT *, 'npk =0, assume 1'
  ENDIF
  IF ( npt == 0 ) THEN
     PRINT *, 'npt=0, assume 1'
     npt = 1
  ENDIF
  DO jk = 1, nlev                                            ! the list stops at the first level >= npk
     IF ( nilev(jk) >= npk ) THEN
        nlev = jk
        EXIT
     ENDIF
  END DO
  PRINT *, 'NLEV', nlev

  CALL cdf_phase('start -> arguments, dimensions')
  ALLOCATE( e1u(npiglo,npjglo), e1f(npiglo,npjglo), e2v(npiglo,npjglo), e2f(npiglo,npjglo) )
  ! large fields: a level at a time through the page-locked ring (module cdffeed), nothing of nlev levels on the host
  nij8 = INT(npiglo, 8) * npjglo
  lfeed = feed_wanted(4 * nij8 * nlev) .AND. .NOT. ( lforcing .OR. lsurf )
  ALLOCATE( zlon(npiglo,npjglo), zlat(npiglo,npjglo), dtim(npt), gdep(nlev), zdep(MAX(npk, 1)) )
  IF ( .NOT. lfeed ) THEN
     ALLOCATE( un(npiglo,npjglo,nlev), vn(npiglo,npjglo,nlev) )
     IF ( ldblpr ) THEN ; ALLOCATE( drotn(npiglo,npjglo,nlev) ) ; ELSE ; ALLOCATE( zrot4(npiglo,npjglo,nlev) ) ; ENDIF
  ENDIF

  IF ( l_metric ) THEN
     CALL getvar_to(cn_fhgr, cn_ve1u, 1, e1u)
     CALL getvar_to(cn_fhgr, cn_ve1f, 1, e1f)
     CALL getvar_to(cn_fhgr, cn_ve2v, 1, e2v)
     CALL getvar_to(cn_fhgr, cn_ve2f, 1, e2f)
     CALL getvar_to(cn_fhgr, cn_glamf, 1, zlon)
     CALL getvar_to(cn_fhgr, cn_gphif, 1, zlat)
  ELSE
     e1u = 1. ; e1f = 1. ; e2v = 1. ; e2f = 1. ; zlon = 10. ; zlat = 10.
  ENDIF
  CALL cdf_phase('allocations, metric and coordinates read')
  pff = C_NULL_PTR
  IF ( loverf ) THEN                                         ! src/cdfcurl.f90:224-227
     ALLOCATE( dl_ff(npiglo,npjglo) )
     dl_pi = ACOS(-1.d0)
     dl_omega = 2 * dl_pi / 86400.d0
     dl_ff = 2 * dl_omega * SIN( zlat * dl_pi / 180.d0 )
     pff = C_LOC(dl_ff)
  ENDIF
  IF ( lforcing .OR. lsurf ) THEN
     gdep(1) = 0.
  ELSE
     zdep(:) = getvar1d(cf_ufil, cn_vdepthu, npk)
     DO jk = 1, nlev
        gdep(jk) = REAL(zdep(nilev(jk)), 4)
     END DO
  ENDIF
  IF ( zlon(1,1) == zlon(npiglo-1,1) ) lperio = .TRUE.       ! E-W periodicity, :238

  CALL cdf_phase('depths, periodicity')
  CALL CreateOutput
  CALL cdf_phase('CreateOutput')
  CALL hip_check(cdft_curl_create(plan, npiglo, npjglo), 'cdft_curl_create')
  CALL hip_check(cdft_curl_set_mesh(plan, e1u, e2v, e1f, e2f, pff, MERGE(1_C_INT, 0_C_INT, lperio), CDFT_MEM_HOST, &
       &         C_NULL_PTR), 'cdft_curl_set_mesh')

  CALL cdf_phase('plan, mesh to the device')
  IF ( lfeed ) THEN
     pd_u = feed_device(4 * nij8) ; pd_v = feed_device(4 * nij8)
     CALL C_F_POINTER(pd_u, un_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_v, vn_d, (/ nij8 /))
     CALL hip_check(cdft_host_alloc(ph_rot(1), INT(MERGE(8, 4, ldblpr) * nij8, C_SIZE_T)), 'cdft_host_alloc')
     CALL hip_check(cdft_host_alloc(ph_rot(2), INT(MERGE(8, 4, ldblpr) * nij8, C_SIZE_T)), 'cdft_host_alloc')
  ENDIF

  CALL cdf_phase('device and page-locked arrays')
  DO jt = 1, npt
     IF ( MOD(jt, 100) == 0 ) PRINT *, jt, '/', npt
     IF ( lfeed ) THEN      ! level jk+1 is read and computed while level jk is on its way to the host and to the file
        CALL load_level(1) ; CALL compute_level(1)
        DO jk = 1, nlev
           IF ( jk < nlev ) CALL load_level(jk + 1)       ! (level jk-1 is being written by the output file's thread meanwhile)
           CALL hip_check(cdft_stream_sync(feed_stream), 'cdft_stream_sync')
           CALL wait_writes(ncout)                        ! level jk-1 is in the file: its array takes level jk+1
           IF ( jk < nlev ) CALL compute_level(jk + 1)
           ierr = putvar_behind(ncout, id_varout(1), ph_rot(1 + MOD(jk - 1, 2)), ldblpr, jk, npiglo, npjglo, ktime=jt)
        END DO
        CALL wait_writes(ncout)
        CYCLE
     ENDIF
     DO jk = 1, nlev
        PRINT *, TRIM(cf_ufil), ' ', TRIM(cv_u), ' ', jk, nilev(jk), npiglo, npjglo, jt
        CALL getvar_to(cf_ufil, cv_u, nilev(jk), un(:,:,jk), ktime=jt)
        CALL getvar_to(cf_vfil, cv_v, nilev(jk), vn(:,:,jk), ktime=jt)
        IF ( lforcing ) THEN                                 ! A grid -> U and V points (:250-258); the last
           DO ji = 1, npiglo - 1                             ! column of un / row of vn is not used by the stencil
              un(ji,:,jk) = 0.5*( un(ji,:,jk) + un(ji+1,:,jk) )
           END DO
           DO jj = 1, npjglo - 1
              vn(:,jj,jk) = 0.5*( vn(:,jj,jk) + vn(:,jj+1,jk) )
           END DO
        ENDIF
     END DO
     IF ( ldblpr ) THEN
        CALL hip_check(cdft_curl_compute(plan, un, vn, nlev, CDFT_MEM_HOST, MERGE(1_C_INT, 0_C_INT, ltpoint),        &
             &         C_LOC(drotn), 1_C_INT, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_curl_compute')
     ELSE
        CALL hip_check(cdft_curl_compute(plan, un, vn, nlev, CDFT_MEM_HOST, MERGE(1_C_INT, 0_C_INT, ltpoint),        &
             &         C_LOC(zrot4), 0_C_INT, CDFT_MEM_HOST, C_NULL_PTR), 'cdft_curl_compute')
     ENDIF
     DO jk = 1, nlev
        IF ( ldblpr ) THEN ; ierr = putvar(ncout, id_varout(1), drotn(:,:,jk), jk, npiglo, npjglo, ktime=jt)
        ELSE               ; ierr = putvar(ncout, id_varout(1), zrot4(:,:,jk), jk, npiglo, npjglo, ktime=jt)
        ENDIF
     END DO
  END DO
  CALL cdf_phase('records')
  ierr = cdft_curl_destroy(plan)
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_u) ; CALL feed_release(pd_v)
     ierr = cdft_host_free(ph_rot(1)) ; ierr = cdft_host_free(ph_rot(2))
     CALL feed_close()
  ENDIF
  CALL cdf_phase('destroy, close, release')

CONTAINS

  SUBROUTINE load_level(kk)
    !! u and v of level nilev(kk) of record jt from the file to HBM through the ring (the line the reference prints, :246)
    INTEGER(KIND=4), INTENT(in) :: kk
    PRINT *, TRIM(cf_ufil), ' ', TRIM(cv_u), ' ', kk, nilev(kk), npiglo, npjglo, jt
    CALL feed_field(cf_ufil, cv_u, pd_u, npiglo, npjglo, 1, ktime=jt, kkmin=nilev(kk))
    CALL feed_field(cf_vfil, cv_v, pd_v, npiglo, npjglo, 1, ktime=jt, kkmin=nilev(kk))
  END SUBROUTINE load_level

  SUBROUTINE compute_level(kk)
    !! queued behind the copies of load_level(kk); the result on its way to the page-locked array of kk's parity
    INTEGER(KIND=4), INTENT(in) :: kk
    CALL hip_check(cdft_curl_compute(plan, un_d, vn_d, 1, CDFT_MEM_DEVICE, MERGE(1_C_INT, 0_C_INT, ltpoint),            &
         &         ph_rot(1 + MOD(kk - 1, 2)), MERGE(1_C_INT, 0_C_INT, ldblpr), CDFT_MEM_PINNED, feed_stream), 'cdft_curl_compute')
  END SUBROUTINE compute_level

  SUBROUTINE ParseLevel(cdum)
    !! '5', '1,3,5', '1-20', '1-3,10-12', '10-' (open ended, capped at 300) -> nilev(1:nlev)
    CHARACTER(LEN=*), INTENT(in) :: cdum
    INTEGER(KIND=4) :: ilev(350), ia, ib, ic, id, i1, i2, jl
    nlev = 0 ; ia = 1
    DO WHILE ( ia <= LEN_TRIM(cdum) )
       ib = INDEX(cdum(ia:), ',')
       IF ( ib == 0 ) THEN ; ic = LEN_TRIM(cdum) ; ELSE ; ic = ia + ib - 2 ; ENDIF
       id = INDEX(cdum(ia:ic), '-')
       IF ( id == 0 ) THEN
          READ(cdum(ia:ic), *) i1 ; i2 = i1
       ELSE IF ( ia + id - 1 == ic ) THEN
          READ(cdum(ia:ic-1), *) i1 ; i2 = 300
       ELSE
          READ(cdum(ia:ia+id-2), *) i1 ; READ(cdum(ia+id:ic), *) i2
       ENDIF
       DO jl = i1, i2
          IF ( nlev < 350 ) THEN ; nlev = nlev + 1 ; ilev(nlev) = jl ; ENDIF
       END DO
       ia = ic + 2
    END DO
    IF ( ALLOCATED(nilev) ) DEALLOCATE(nilev)
    ALLOCATE( nilev(nlev) )
    nilev(:) = ilev(1:nlev)
  END SUBROUTINE ParseLevel

  SUBROUTINE CreateOutput
    !! src/cdfcurl.f90:395-438
    ipk(1) = nlev
    stypvar(1)%ichunk = (/ npiglo, MAX(1, npjglo / 30), 1, 1 /)
    IF      ( ltpoint ) THEN ; stypvar(1)%cname = 'socurlt'     ; stypvar(1)%cunits = 's-1'
    ELSE IF ( loverf  ) THEN ; stypvar(1)%cname = 'socurloverf' ; stypvar(1)%cunits = '-'
    ELSE                     ; stypvar(1)%cname = 'socurl'      ; stypvar(1)%cunits = 's-1'
    ENDIF
    stypvar(1)%cprecision = 'r4' ; IF ( ldblpr ) stypvar(1)%cprecision = 'r8'
    stypvar(1)%rmissing_value    = 0.
    stypvar(1)%valid_min         = -1000.
    stypvar(1)%valid_max         =  1000.
    stypvar(1)%clong_name        = 'Relative_Vorticity (curl)'
    stypvar(1)%cshort_name       = 'socurl'
    stypvar(1)%conline_operation = 'N/A'
    stypvar(1)%caxis             = 'TYX'
    ncout = create      (cf_out, cf_ufil, npiglo, npjglo, nlev, ld_nc4=lnc4)
    ierr  = createvar   (ncout, stypvar, 1, ipk, id_varout, ld_nc4=lnc4)
    ierr  = putheadervar(ncout, cf_ufil, npiglo, npjglo, nlev, pnavlon=zlon, pnavlat=zlat, pdep=gdep)
    dtim  = getvar1d(cf_ufil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutput

END PROGRAM cdfcurl
