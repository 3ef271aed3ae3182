PROGRAM cdfmocsig
  !!======================================================================
  !!                     ***  PROGRAM  cdfmocsig  ***
  !! Meridional overturning stream function in density classes (sigma coordinates).
  !!
  !! Drop-in for the reference tool (src/cdfmocsig.f90): same options, same mocsig.nc (x=1, y, sigma, time;
  !! zomsfglo [atl inp ind pac] [zoisoglo ...] float, Sverdrup / m, :490-586).  The level loop, the density
  !! binning, the zonal sums and the cumulation over the density classes (:368-470) run on the GPU for a whole
  !! record at once (cdft_mocsig_compute); -ntr bins the neutral density (CDFT_EOS_NTR) the same way.
  !! Host feed (module cdffeed): when the fields are large (feed_wanted) nothing 3-D is held in host memory -- V, T, S [, the
  !! eddy-induced velocity, e3v] go from the file to HBM slab by slab through the page-locked ring, the transfer of a slab
  !! under the read of the next, so a record costs its read, not its read plus its transfer (CDFT_FEED=0: the old way).
  !!======================================================================
  USE ISO_C_BINDING
  USE cdfio
  USE modcdfnames
  USE modutils
  USE eos
  USE cdftools_hip
  USE cdffeed
  IMPLICIT NONE

  INTEGER(KIND=4) :: ji, jj, jk, jt, jbasin, jbin, it, npiglo, npjglo, npk, npt, nbins, narg, ijarg, ii, ncout, nbasins, nvaro, ierr
  INTEGER(KIND=4) :: npglo = 1, npatl = 2, npinp = 3, npind = 4, nppac = 5, iloc(2)
  INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
  INTEGER(KIND=2), ALLOCATABLE, TARGET :: ibmask(:,:,:)
  REAL(KIND=4) :: pref = 0., sigmin, sigstp, zsps, zspt, zspv, rp_spval = 99999.
  REAL(KIND=4), ALLOCATABLE, TARGET :: e3v(:,:,:), e31d(:), gdep(:), zveiv(:,:,:)
  REAL(KIND=4), ALLOCATABLE, TARGET :: e1v(:,:)
  REAL(KIND=4), ALLOCATABLE :: gphiv(:,:), zv(:,:,:), zt(:,:,:), zs(:,:,:), rdumlon(:,:), rdumlat(:,:), sigma(:), zwk(:,:), zbas(:,:,:)
  REAL(KIND=8), ALLOCATABLE :: dtim(:)
  REAL(KIND=8), ALLOCATABLE, TARGET :: dmoc(:,:,:), depi(:,:,:)            ! (nbasins, nbins, npjglo)
  CHARACTER(LEN=256) :: cf_vfil, cf_tfil, cf_sfil, cf_moc = 'mocsig.nc', cldum, cglobal
  TYPE(variable), ALLOCATABLE :: stypvar(:)
  TYPE(C_PTR) :: pe3, pe31d, peiv, pgdep, pdepi
  TYPE(C_PTR) :: pd_zv = C_NULL_PTR, pd_t = C_NULL_PTR, pd_s = C_NULL_PTR, pd_eiv = C_NULL_PTR, pd_e3v = C_NULL_PTR
  TYPE(C_PTR) :: pd_e1v = C_NULL_PTR, pd_ib = C_NULL_PTR, pd_e31 = C_NULL_PTR
  REAL(KIND=4), POINTER :: zv_d(:), zt_d(:), zs_d(:), e1v_d(:)          ! device arrays seen as arrays (never touched by the host)
  INTEGER(KIND=2), POINTER :: ib_d(:)
  INTEGER(KIND=8) :: nij8, n38
  LOGICAL :: lfeed
  LOGICAL :: lfull = .FALSE., lchk = .FALSE., lbas, lbin(3), lisodep = .FALSE., lprint = .FALSE., leiv = .FALSE.
  LOGICAL :: lntr = .FALSE., ll_teos10 = .FALSE.
  INTEGER(C_INT) :: ieos
  !!----------------------------------------------------------------------
  CALL ReadCdfNames()
  narg = COMMAND_ARGUMENT_COUNT()
  IF ( narg == 0 ) THEN
#include "usage/cdfmocsig.inc"
     STOP
  ENDIF

  cglobal = 'Partial step computation'
  lbin = (/ .TRUE., .TRUE., .TRUE. /)
  ijarg = 1 ; ii = 0
  cf_sfil = 'none'
  DO WHILE ( ijarg <= narg )
     CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1
     SELECT CASE ( cldum )
     CASE ( '-v'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_vfil) ; ijarg = ijarg + 1 ; ii = ii + 1
     CASE ( '-t'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_tfil) ; ijarg = ijarg + 1 ; ii = ii + 1
     CASE ( '-r'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum  ) ; ijarg = ijarg + 1 ; ii = ii + 1 ; READ(cldum, *) pref
     CASE ( '-ntr'     ) ; lntr = .TRUE. ; ii = ii + 1
     CASE ( '-s'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_sfil) ; ijarg = ijarg + 1
     CASE ( '-full'    ) ; lfull   = .TRUE. ; cglobal = 'Full step computation'
     CASE ( '-eiv'     ) ; leiv    = .TRUE.
     CASE ( '-sigmin'  ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) sigmin ; lbin(1) = .FALSE.
     CASE ( '-nbins'   ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) nbins  ; lbin(2) = .FALSE.
     CASE ( '-sigstp'  ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cldum) ; ijarg = ijarg + 1 ; READ(cldum, *) sigstp ; lbin(3) = .FALSE.
     CASE ( '-o'       ) ; CALL GET_COMMAND_ARGUMENT(ijarg, cf_moc) ; ijarg = ijarg + 1
     CASE ( '-vvl'     ) ; lg_vvl  = .TRUE.
     CASE ( '-teos10'  ) ; ll_teos10 = .TRUE.
     CASE ( '-isodep'  ) ; lisodep = .TRUE.
     CASE ( '-verbose' ) ; lprint  = .TRUE.
     CASE DEFAULT        ; PRINT *, ' ERROR : ', TRIM(cldum), ' : unknown option.' ; STOP 99
     END SELECT
  END DO
  CALL eos_init( ll_teos10 )
  IF ( ii /= 3 ) THEN ; PRINT *, ' ERROR : mandatory arguments missing, see usage please !' ; STOP 99
  ENDIF
  IF ( cf_sfil == 'none' ) cf_sfil = cf_tfil
  ieos = MERGE(CDFT_EOS_TEOS10, CDFT_EOS_80, ll_teos10)
  IF ( lntr ) ieos = CDFT_EOS_NTR
  lchk = lchk .OR. chkfile(cn_fhgr)
  lchk = lchk .OR. chkfile(cn_fzgr)
  lchk = lchk .OR. chkfile(cn_fmsk)
  lchk = lchk .OR. chkfile(cf_vfil)
  lchk = lchk .OR. chkfile(cf_tfil)
  lchk = lchk .OR. chkfile(cf_sfil)
  IF ( lchk ) STOP 99
  zsps = getspval(cf_sfil, cn_vosaline)
  zspt = getspval(cf_tfil, cn_votemper)
  zspv = getspval(cf_vfil, cn_vomecrty)
  IF ( lg_vvl ) THEN
     cn_fe3v = cf_vfil
     cn_ve3v = cn_ve3vvvl
  ENDIF
  lchk = lbin(1) .OR. lbin(2) .OR. lbin(3)                  ! TRUE if no particular binning specified

  npiglo = getdim(cf_vfil, cn_x)
  npjglo = getdim(cf_vfil, cn_y)
  npk    = getdim(cf_vfil, cn_z)
  npt    = getdim(cf_vfil, cn_t)
  PRINT *, 'npiglo = ', npiglo
  PRINT *, 'npjglo = ', npjglo
  PRINT *, 'npk    = ', npk
  PRINT *, 'npt    = ', npt
  CALL SetGlobalAtt(cglobal, 'A')
  lbas = .NOT. chkfile(cn_fbasins)
  IF ( lbas ) THEN ; nbasins = 5
  ELSE             ; nbasins = 1
  ENDIF
  IF ( lisodep ) THEN ; nvaro = 2 * nbasins
  ELSE                ; nvaro =     nbasins
  ENDIF
  ALLOCATE( stypvar(nvaro), ipk(nvaro), id_varout(nvaro) )
  IF ( lchk .AND. lntr ) THEN                               ! (:267-270, "to be confirmed": sigmantr is rho - 1000)
     nbins = 52 ; sigmin = 1023. ; sigstp = 0.1
  ELSE IF ( lchk ) THEN                                     ! default classes of the reference depth (:282-318)
     SELECT CASE ( INT(pref) )
     CASE ( 0 )    ; nbins = 52  ; sigmin = 23. ; sigstp = 0.1
     CASE ( 1000 ) ; nbins = 88  ; sigmin = 24. ; sigstp = 0.1
     CASE ( 2000 ) ; nbins = 158 ; sigmin = 30. ; sigstp = 0.05
     CASE DEFAULT
        PRINT *, ' This value of depth_ref (', pref, ') is not implemented as standard'
        PRINT *, ' You must use the -sigmin, -sigstp and -nbins options to precise'
        PRINT *, ' the density bining you want to use.'
        STOP 99
     END SELECT
  ENDIF
  IF ( lntr ) THEN ; PRINT '(a)',         '  For Neutral density MOC'
  ELSE             ; PRINT '(a,f6.1,a)', '  For reference depth ', pref, ' m, '
  ENDIF
  PRINT '(a,f5.2,a,f5.2,a,i3)', '  You are using -sigmin ', sigmin, ' -sigstp ', sigstp, ' -nbins ', nbins
  ALLOCATE( sigma(nbins) )
  DO ji = 1, nbins
     sigma(ji) = sigmin + (ji-0.5)*sigstp
  END DO
  IF ( lprint ) PRINT *, ' min density:', sigma(1), ' max density:', sigma(nbins)

  nij8 = INT(npiglo, 8) * npjglo ; n38 = nij8 * npk
  lfeed = feed_wanted(4 * n38)
  ALLOCATE( ibmask(nbasins,npiglo,npjglo) )
  IF ( .NOT. lfeed ) ALLOCATE( zv(npiglo,npjglo,npk), zt(npiglo,npjglo,npk), zs(npiglo,npjglo,npk) )
  ALLOCATE( e1v(npiglo,npjglo), gphiv(npiglo,npjglo), zwk(npiglo,npjglo), dmoc(nbasins,nbins,npjglo) )
  ALLOCATE( rdumlon(1,npjglo), rdumlat(1,npjglo), dtim(npt), e31d(npk), gdep(npk) )
  IF ( lisodep ) ALLOCATE( depi(nbasins,nbins,npjglo) )
  IF ( leiv .AND. .NOT. lfeed ) ALLOCATE( zveiv(npiglo,npjglo,npk) )
  IF ( .NOT. ( lfull .OR. lfeed ) ) ALLOCATE( e3v(npiglo,npjglo,npk) )
  CALL getvar_to(cn_fhgr, cn_ve1v, 1, e1v(:,:))
  IF ( lfull   ) e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
  IF ( lisodep ) gdep(:) = -getvare3(cn_fzgr, cn_gdept, npk)
  IF ( npjglo > 1 ) THEN
     CALL getvar_to(cn_fhgr, cn_gphiv, 1, gphiv(:,:))
     iloc         = MAXLOC(gphiv)
     rdumlat(1,:) = gphiv(iloc(1),:)
  ELSE
     rdumlat(1,:) = 0.
  ENDIF
  rdumlon(:,:) = 0.
  CALL CreateOutputFile

  ! (the same values as the reference's five strided assignments and its WHERE, in one pass with the basin index innermost: cdfmoc.f90)
  CALL getvar_to(cn_fmsk, cn_vmask, 1, zwk(:,:))
  IF ( .NOT. lbas ) ibmask(npglo,:,:) = INT(zwk, 2)
  IF ( lbas ) THEN
     ALLOCATE( zbas(npiglo, npjglo, 3) )
     CALL getvar_to(cn_fbasins, cn_tmaskatl, 1, zbas(:,:,1))
     CALL getvar_to(cn_fbasins, cn_tmaskind, 1, zbas(:,:,2))
     CALL getvar_to(cn_fbasins, cn_tmaskpac, 1, zbas(:,:,3))
     DO jj = 1, npjglo
        DO ji = 1, npiglo
           ibmask(npglo,ji,jj) = INT(zwk(ji,jj), 2)
           ibmask(npatl,ji,jj) = INT(zbas(ji,jj,1), 2)
           ibmask(npind,ji,jj) = INT(zbas(ji,jj,2), 2)
           ibmask(nppac,ji,jj) = INT(zbas(ji,jj,3), 2)
           ibmask(npinp,ji,jj) = ibmask(nppac,ji,jj) + ibmask(npind,ji,jj)
           IF ( ibmask(npinp,ji,jj) > 0 ) ibmask(npinp,ji,jj) = 1
        END DO
     END DO
     DEALLOCATE( zbas )
     ibmask(1,1,     :) = 0                                 ! global periodic condition (only with the basin file, :357-358)
     ibmask(1,npiglo,:) = 0
  ENDIF

  DO jt = 1, npt
     it = 1 ; IF ( lg_vvl ) it = jt
     IF ( lprint ) THEN                                     ! -verbose: the reference's two lines per level of its loop (src/cdfmocsig.f90:372, :412)
        DO jk = 1, npk - 1
           PRINT *, ' working at depth ', jk
           PRINT *, ' Entering main J loop '
        END DO
     ENDIF
     IF ( lfeed ) THEN                                      ! file -> HBM through the ring, the compute call ordered behind it
        pe3 = C_NULL_PTR ; pe31d = C_NULL_PTR ; peiv = C_NULL_PTR ; pgdep = C_NULL_PTR ; pdepi = C_NULL_PTR
        IF ( jt == 1 ) THEN
           pd_e1v = feed_device(4 * nij8) ; pd_ib = feed_device(2 * nbasins * nij8)
           pd_zv  = feed_device(4 * n38)  ; pd_t  = feed_device(4 * n38) ; pd_s = feed_device(4 * n38)
           CALL C_F_POINTER(pd_e1v, e1v_d, (/ nij8 /)) ; CALL C_F_POINTER(pd_ib, ib_d, (/ nbasins * nij8 /))
           CALL C_F_POINTER(pd_zv, zv_d, (/ n38 /)) ; CALL C_F_POINTER(pd_t, zt_d, (/ n38 /)) ; CALL C_F_POINTER(pd_s, zs_d, (/ n38 /))
           CALL feed_host(C_LOC(e1v), 4 * nij8, pd_e1v)
           CALL feed_host(C_LOC(ibmask), 2 * nbasins * nij8, pd_ib)
           IF ( leiv ) pd_eiv = feed_device(4 * n38)
           IF ( lfull ) THEN
              pd_e31 = feed_device(4_8 * npk) ; CALL feed_host(C_LOC(e31d), 4_8 * npk, pd_e31)
           ELSE
              pd_e3v = feed_device(4 * n38)
           ENDIF
        ENDIF
        IF ( .NOT. lfull .AND. ( jt == 1 .OR. lg_vvl ) )  &
             &  CALL feed_field(cn_fe3v, cn_ve3v, pd_e3v, npiglo, npjglo, npk, ktime=it, ldiom=.NOT. lg_vvl)
        CALL feed_field(cf_vfil, cn_vomecrty, pd_zv, npiglo, npjglo, npk, ktime=jt)
        CALL feed_field(cf_tfil, cn_votemper, pd_t,  npiglo, npjglo, npk, ktime=jt)
        CALL feed_field(cf_sfil, cn_vosaline, pd_s,  npiglo, npjglo, npk, ktime=jt)
        IF ( leiv ) CALL feed_field(cf_vfil, cn_vomeeivv, pd_eiv, npiglo, npjglo, npk, ktime=jt)
        IF ( lisodep ) THEN ; pgdep = C_LOC(gdep) ; pdepi = C_LOC(depi) ; ENDIF         ! (gdep: npk host values in every case)
        CALL hip_check(cdft_mocsig_compute(zv_d, pd_eiv, zt_d, zs_d, e1v_d, pd_e3v, pd_e31, ib_d, pgdep, npiglo, npjglo, npk, nbasins, &
             &         zspv, zspt, zsps, pref, ieos, nbins, sigmin, sigstp, CDFT_MEM_DEVICE, dmoc, pdepi, CDFT_MEM_HOST,               &
             &         feed_stream), 'cdft_mocsig_compute')
     ELSE
        CALL readvar3d(cf_vfil, cn_vomecrty, zv, ktime=jt)
        CALL readvar3d(cf_tfil, cn_votemper, zt, ktime=jt)
        CALL readvar3d(cf_sfil, cn_vosaline, zs, ktime=jt)
        pe3 = C_NULL_PTR ; pe31d = C_NULL_PTR ; peiv = C_NULL_PTR ; pgdep = C_NULL_PTR ; pdepi = C_NULL_PTR
        IF ( leiv ) THEN
           CALL readvar3d(cf_vfil, cn_vomeeivv, zveiv, ktime=jt)
           peiv = C_LOC(zveiv)
        ENDIF
        IF ( lfull ) THEN
           pe31d = C_LOC(e31d)
        ELSE
           IF ( jt == 1 .OR. lg_vvl ) THEN
              DO ji = 1, npk
                 CALL getvar_to(cn_fe3v, cn_ve3v, ji, e3v(:,:,ji), ktime=it, ldiom=.NOT. lg_vvl)
              END DO
           ENDIF
           pe3 = C_LOC(e3v)
        ENDIF
        IF ( lisodep ) THEN ; pgdep = C_LOC(gdep) ; pdepi = C_LOC(depi) ; ENDIF
        CALL hip_check(cdft_mocsig_compute(zv, peiv, zt, zs, e1v, pe3, pe31d, ibmask, pgdep, npiglo, npjglo, npk, nbasins, zspv, zspt, &
             &         zsps, pref, ieos, nbins, sigmin, sigstp, CDFT_MEM_HOST, dmoc, pdepi,  &
             &         CDFT_MEM_HOST, C_NULL_PTR), 'cdft_mocsig_compute')
     ENDIF
     DO jbasin = 1, nbasins
        DO jbin = 1, nbins
           ierr = putvar(ncout, id_varout(jbasin), REAL(dmoc(jbasin,jbin,:)), jbin, 1, npjglo, ktime=jt)
           IF ( lisodep ) ierr = putvar(ncout, id_varout(jbasin+nbasins), REAL(depi(jbasin,jbin,:)), jbin, 1, npjglo, ktime=jt)
        END DO
     END DO
  END DO
  ierr = closeout(ncout)
  IF ( lfeed ) THEN
     CALL feed_release(pd_zv) ; CALL feed_release(pd_t) ; CALL feed_release(pd_s) ; CALL feed_release(pd_eiv)
     CALL feed_release(pd_e3v) ; CALL feed_release(pd_e31) ; CALL feed_release(pd_e1v) ; CALL feed_release(pd_ib)
     CALL feed_close()
  ENDIF

CONTAINS

  SUBROUTINE CreateOutputFile
    !! src/cdfmocsig.f90:490-586
    CHARACTER(LEN=256) :: cl_moc(5), cl_iso(5), cl_tag(5)
    INTEGER(KIND=4) :: jb
    stypvar%cunits            = 'Sverdrup'
    stypvar%rmissing_value    = rp_spval
    stypvar%valid_min         = -1000.
    stypvar%valid_max         =  1000.
    stypvar%scale_factor      = 1.
    stypvar%add_offset        = 0.
    stypvar%savelog10         = 0.
    stypvar%conline_operation = 'N/A'
    stypvar%caxis             = 'TZY'
    ipk(:) = nbins
    cl_moc = (/ cn_zomsfglo, cn_zomsfatl, cn_zomsfinp, cn_zomsfind, cn_zomsfpac /)
    cl_iso = (/ cn_zoisoglo, cn_zoisoatl, cn_zoisoinp, cn_zoisoind, cn_zoisopac /)
    cl_tag(1) = 'Global' ; cl_tag(2) = 'Atlantic' ; cl_tag(3) = 'IndoPacif' ; cl_tag(4) = 'Indian' ; cl_tag(5) = 'pacif'
    DO jb = 1, nbasins
       stypvar(jb)%cname       = cl_moc(jb)
       stypvar(jb)%clong_name  = 'Meridional_Overt.Cell_'//TRIM(cl_tag(jb))
       stypvar(jb)%cshort_name = cl_moc(jb)
       IF ( lisodep ) THEN
          stypvar(jb+nbasins)%cunits      = 'm'
          stypvar(jb+nbasins)%cname       = cl_iso(jb)
          stypvar(jb+nbasins)%clong_name  = 'Zonal_mean_isopycnal_depth_'//TRIM(cl_tag(jb))
          stypvar(jb+nbasins)%cshort_name = cl_iso(jb)
          stypvar(jb+nbasins)%valid_min   = 0.
          stypvar(jb+nbasins)%valid_max   = 8000.
       ENDIF
    END DO
    ncout = create      (cf_moc, 'none', 1, npjglo, nbins, cdep='sigma')
    ierr  = createvar   (ncout, stypvar, nvaro, ipk, id_varout, cdglobal=cglobal)
    ierr  = putheadervar(ncout, cf_vfil, 1, npjglo, nbins, pnavlon=rdumlon, pnavlat=rdumlat, pdep=sigma)
    dtim  = getvar1d(cf_vfil, cn_vtimec, npt)
    ierr  = putvar1d(ncout, dtim, npt, 'T')
  END SUBROUTINE CreateOutputFile

END PROGRAM cdfmocsig
