MODULE cdfmoc_rapid
  !!======================================================================
  !!                     ***  MODULE  cdfmoc_rapid  ***
  !! cdfmoc -rapid : the AMOC at 26.5 N decomposed as the RAPID-MOCHA array does it (src/cdfmoc.f90:598-1004,
  !! output variables :1190-1407).
  !!
  !! One (i,k) section of the model (a few thousand cells) and 33 numbers per record: there is nothing
  !! data-parallel to offload, so this option is host code, as in the reference.  Every sum below runs in
  !! the reference's loop order with its r4 / r8 mixing, so the results are the reference's bit for bit
  !! given the same compiler intrinsics (the Coriolis factor uses an r4 SIN).
  !!
  !! The section is the row ijrapid between the F-points nearest to 80.1 W and 12.7 E; its first npigs
  !! columns (to 77.4 W) are the Florida Strait, the rest the mid-ocean.  All terms are v*e1v*e3v based:
  !!   tsum_*(fld)    = sum over a box of columns and levels of  fld * e1 * e3            (r8 from e1 on)
  !!   wmean(fld)     = the same weighted by the mask, with the masked area
  !!======================================================================
  USE cdfio
  USE modcdfnames
  USE cdftools
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: rapid_amoc

CONTAINS

  SUBROUTINE rapid_amoc(cf_vfil, cf_tfil, cf_sfil, cf_ufil, cd_moc, lfull)
    CHARACTER(*), INTENT(in) :: cf_vfil, cf_tfil, cf_sfil, cf_ufil, cd_moc      ! cd_moc: the -o name (moc.nc)
    LOGICAL,      INTENT(in) :: lfull
    REAL(KIND=4), PARAMETER :: pp_lat = 26.5, pp_lonw = -80.1, pp_lone = 12.7, pp_longs = -77.4
    INTEGER(KIND=4), PARAMETER :: jp_class = 5
    REAL(KIND=4), PARAMETER :: pp_zlim(jp_class+1) = (/ 0., 800., 1100., 3000., 5000., 10000. /)
    INTEGER(KIND=4) :: ijrapid, iiw, iie, iigs, idum, npis, npigs, ji, jk, jt, jclass, jv, jk100, idvar, it
    INTEGER(KIND=4) :: npk, npt, ncout, ierr, nvarout
    INTEGER(KIND=4), ALLOCATABLE :: ipk(:), id_varout(:)
    REAL(KIND=4), ALLOCATABLE :: rdumlon(:,:), rdumlat(:,:), gdepw(:), e31d(:)
    REAL(KIND=8), ALLOCATABLE :: dtim(:)
    TYPE(variable), ALLOCATABLE :: stypvar(:)
    CHARACTER(LEN=256) :: cf_moc
    INTEGER(KIND=4), ALLOCATABLE :: itm(:,:)
    REAL(KIND=4), ALLOCATABLE :: zv2(:,:), zt2(:,:), zs2(:,:), ze3(:,:), zrow(:,:), zdept(:)
    REAL(KIND=4), POINTER :: zfld(:,:)
    REAL(KIND=8), ALLOCATABLE :: de1(:), dtaux(:,:), damoc(:), dmv(:), dmt(:)
    REAL(KIND=8) :: dtrp, dsurf, dl0, dtrek, dmv0, dmt0, dms0
    REAL(KIND=4) :: zmin, zmax, zbot, zalpha, zrho, zpi, zf
    CHARACTER(LEN=1) :: cl_name(3)
    INTEGER(KIND=4) :: jvar
    TARGET :: zv2, zt2, zs2

    npk = getdim(cf_vfil, cn_z)
    PRINT *, 'cf_vfil = ', TRIM(cf_vfil)
    PRINT *, 'cf_tfil = ', TRIM(cf_tfil)
    PRINT *, 'cf_sfil = ', TRIM(cf_sfil)
    PRINT *, 'cf_ufil = ', TRIM(cf_ufil)
    npt = getdim(cf_vfil, cn_t)
    ! 1) the section
    CALL cdf_findij(pp_lonw, pp_lone,  pp_lat, pp_lat, iiw,  iie,  ijrapid, idum, cd_coord=cn_fhgr, cd_point='F')
    CALL cdf_findij(pp_lonw, pp_longs, pp_lat, pp_lat, idum, iigs, idum,    idum, cd_coord=cn_fhgr, cd_point='F')
    IF ( lg_vvl ) THEN
       cn_fe3v = cf_vfil
       cn_ve3v = cn_ve3vvvl
    ENDIF
    npis  = iie  - iiw + 1
    npigs = iigs - iiw + 1
    IF ( iiw < 1 .OR. ijrapid < 1 .OR. npis < 1 ) THEN
       PRINT *, ' ERROR : the RAPID section (26.5 N, 80.1 W - 12.7 E) was not found in ', TRIM(cn_fhgr)
       STOP 99
    ENDIF
    ALLOCATE( zv2(npis,npk), zt2(npis,npk), zs2(npis,npk), ze3(npis,npk), itm(npis,npk), zrow(npis,1), zdept(npk) )
    ALLOCATE( de1(npis), dtaux(npis,1), damoc(npk), dmv(MAX(npis,npk)), dmt(MAX(npis,npk)), gdepw(npk), e31d(npk) )
    ALLOCATE( rdumlon(1,1), rdumlat(1,1), dtim(npt) )

    zrow(:,:) = getvar(cn_fhgr, cn_gphiv, 1, npis, 1, kimin=iiw, kjmin=ijrapid)
    rdumlon(:,:) = 0.0
    rdumlat(:,:) = zrow(1,1)
    IF ( lfull ) e31d(:) = getvare3(cn_fzgr, cn_ve3t1d, npk)
    zrow(:,:) = getvar(cn_fhgr, cn_ve1v, 1, npis, 1, kimin=iiw, kjmin=ijrapid)
    de1(:)    = zrow(:,1)
    gdepw(:)  = getvare3(cn_fzgr, cn_gdepw, npk)
    DO jk = 1, npk
       zrow(:,:) = getvar(cn_fmsk, cn_tmask, jk, npis, 1, kimin=iiw, kjmin=ijrapid)
       itm(:,jk) = zrow(:,1)
    END DO
    zdept(:) = getvare3(cn_fzgr, cn_gdept, npk)

    CALL CreateOutputRapid

    zrho = 1020
    zpi  = 4.*ATAN(1.)
    zf   = 2.*2.*zpi/86400.*SIN(pp_lat*zpi/180.)
    jk100 = 0
    DO jk = npk, 1, -1
       IF ( zdept(jk) >= 100. ) jk100 = jk                      ! first level at or below 100 m
    END DO

    DO jt = 1, npt
       it = 1 ; IF ( lg_vvl ) it = jt
       IF ( it == jt ) THEN                                     ! every record only with -vvl
          DO jk = 1, npk
             IF ( lfull ) THEN
                ze3(:,jk) = e31d(jk)
             ELSE
                zrow(:,:) = getvar(cn_fe3v, cn_ve3v, jk, npis, 1, kimin=iiw, kjmin=ijrapid, ktime=it, ldiom=.NOT. lg_vvl)
                WHERE ( zrow(:,1) == 0.d0 ) zrow(:,1) = 0.1     ! no division by 0 below
                ze3(:,jk) = zrow(:,1)
             ENDIF
          END DO
       ENDIF
       CALL slab(cf_vfil, cn_vomecrty, zv2)
       dtaux(:,:) = getvar(cf_ufil, cn_sozotaux, 1, npis, 1, kimin=iiw, kjmin=ijrapid, ktime=jt)
       CALL slab(cf_tfil, cn_votemper, zt2)
       CALL slab(cf_sfil, cn_vosaline, zs2)
       zv2(:,:)   = zv2(:,:) * itm(:,:)                         ! tmask stands for vmask here
       dtaux(:,1) = dtaux(:,1) * itm(:,1)
       zt2(:,:)   = zt2(:,:) * itm(:,:)
       zs2(:,:)   = zs2(:,:) * itm(:,:)
       PRINT *, 'max vrapid  ', MAXVAL(zv2)
       PRINT *, 'max trapid  ', MAXVAL(zt2)
       PRINT *, 'max srapid  ', MAXVAL(zs2)
       PRINT *, 'max dtaux   ', MAXVAL(ABS(dtaux))

       ! 2) overturning from the surface down, level by level
       damoc(1) = 0.d0
       CALL put1(1, 0.d0, 1)
       DO jk = 2, npk
          damoc(jk) = damoc(jk-1)
          DO ji = 1, npis
             damoc(jk) = damoc(jk) + zv2(ji,jk-1) * de1(ji) * ze3(ji,jk-1)*1.d0
          END DO
          CALL put1(1, damoc(jk)/1.d6, jk)
       END DO
       dtrp = MAXVAL(damoc(:))                                  ! Q1
       CALL put1(8, dtrp/1.d6, 1)
       PRINT *, 'JT = ', jt, ' Total maximum AMOC = ', REAL(dtrp/1.d6)

       ! 3) Gulf Stream = the western npigs columns, all levels
       dtrp = tsum_v(1, npigs)
       CALL put1(2, dtrp/1.d6, 1)
       PRINT *, 'JT = ', jt, ' GS = ', dtrp/1.d6, ' Sv'

       ! 4) mid-ocean transport by depth class; a level contributes the fraction of its thickness inside the class
       DO jclass = 1, jp_class
          zmin = pp_zlim(jclass) ; zmax = pp_zlim(jclass+1)
          dtrp = 0.d0
          DO ji = npigs + 1, npis
             DO jk = 1, npk
                zbot = gdepw(jk) + ze3(ji,jk)
                IF ( gdepw(jk) >= zmin .AND. zbot <= zmax ) zalpha = 1.0
                IF ( gdepw(jk) >= zmax .OR.  zbot <= zmin ) zalpha = 0.0
                IF ( gdepw(jk) <= zmin .AND. zbot >= zmin ) zalpha = ( zbot - zmin ) / ze3(ji,jk)
                IF ( gdepw(jk) <= zmax .AND. zbot >= zmax ) zalpha = ( zmax - gdepw(jk) ) / ze3(ji,jk)
                dtrp = dtrp + zv2(ji,jk) * de1(ji) * ze3(ji,jk)*1.d0 * zalpha
             END DO
          END DO
          CALL put1(jclass + 2, dtrp/1.d6, 1)
          PRINT *, 'JT = ', jt, ' trp_class:', zmin, zmax, dtrp/1.d6, ' Sv'
       END DO

       ! 5) Ekman transport from the zonal wind stress (Q3)
       dtrp = 0.d0
       DO ji = 1, npis
          dtrp = dtrp - dtaux(ji,1) * de1(ji) / (zrho*zf) *1.d0
       END DO
       dtrek = dtrp/1.d6
       CALL put1(9, dtrp/1.d6, 1)
       PRINT *, 'JT = ', jt, ' Total Ekman transport = ', dtrp/1.d6, ' Sv'

       ! 6) Florida to Africa (Q4)
       dtrp = tsum_v(1, npis)
       CALL put1(10, dtrp/1.d6, 1)
       PRINT *, 'JT = ', jt, ' Total Transport = ', dtrp/1.d6, ' Sv'

       ! 7) area means of V, T, S over the section, the Florida Strait and the mid-ocean (Q5)
       cl_name = (/ 'V', 'T', 'S' /)
       DO jv = 1, 3
          CALL pick(jv)
          idvar = 11 + (jv-1)*3
          CALL wmean(1, npis, dtrp, dsurf)
          CALL put1(idvar, dtrp/dsurf*1.d0, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Mean '//cl_name(jv)//' Total section = ', dtrp/dsurf, ' u.S.I'
          idvar = idvar + 1
          CALL wmean(1, npigs, dtrp, dsurf)
          CALL put1(idvar, dtrp/dsurf*1.d0, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Mean '//cl_name(jv)//' Florida section = ', dtrp/dsurf, ' u.S.I'
          idvar = idvar + 1
          CALL wmean(npigs + 1, npis, dtrp, dsurf)
          CALL put1(idvar, dtrp/dsurf*1.d0, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Mean '//cl_name(jv)//' MidOcean section = ', dtrp/dsurf, ' u.S.I'
       END DO

       ! 8) mid-ocean: zonal-mean v(z) times zonal-mean T(z) or S(z), integrated (Q6)
       cl_name = (/ 'T', 'S', ' ' /)
       DO jv = 1, 2
          CALL pick(jv + 1)
          dmv(:) = 0.d0 ; dmt(:) = 0.d0
          DO jk = 1, npk
             dsurf = 0.
             DO ji = npigs + 1, npis
                dmv(jk) = dmv(jk) + zv2 (ji,jk) * de1(ji) *1.d0
                dmt(jk) = dmt(jk) + zfld(ji,jk) * de1(ji) *1.d0
                dsurf   = dsurf + de1(ji) * itm(ji,jk) *1.d0
             END DO
             IF ( dsurf /= 0. ) dmv(jk) = dmv(jk) / dsurf
             IF ( dsurf /= 0. ) dmt(jk) = dmt(jk) / dsurf
          END DO
          dtrp = 0.d0
          DO jk = 1, npk
             DO ji = npigs + 1, npis
                dtrp = dtrp + dmv(jk) * dmt(jk) * de1(ji) * ze3(ji,jk)*1.d0
             END DO
          END DO
          CALL put1(19 + jv, dtrp/1.d6, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Total  Transport of mean V x mean '//cl_name(jv)//'= ', dtrp/1.d6, ' Sv.uSI'
       END DO

       ! 8) transports of T and S (Q7), and the Ekman ones from the surface and the top 100 m values (Q8)
       DO jv = 1, 2
          CALL pick(jv + 1)
          idvar = 22 + (jv-1)*5
          dtrp = tsum_vx(1, npis)
          CALL put1(idvar, dtrp/1.d6, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Total '//cl_name(jv)//' Transport = ', dtrp/1.d6, ' u.S.I'
          idvar = idvar + 1
          dtrp = tsum_vx(1, npigs)
          CALL put1(idvar, dtrp/1.d6, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Florida '//cl_name(jv)//' Transport = ', dtrp/1.d6, ' u.S.I'
          idvar = idvar + 1
          dtrp = tsum_vx(npigs + 1, npis)
          CALL put1(idvar, dtrp/1.d6, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' MidOcean '//cl_name(jv)//' Transport = ', dtrp/1.d6, ' u.S.I'
          idvar = idvar + 1
          CALL tsum_x(1, dtrp, dsurf)
          CALL put1(idvar, dtrek*dtrp/dsurf, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Ekman '//cl_name(jv)//' Transport based on SST/S = ', dtrek*dtrp/dsurf, ' u.S.I'
          idvar = idvar + 1
          IF ( jk100 >= 1 ) PRINT *, zdept(jk100), ' ~= 100 m for index jk = ', jk100
          CALL tsum_x(jk100, dtrp, dsurf)
          CALL put1(idvar, dtrek*dtrp/dsurf, 1)
          PRINT *, TRIM(stypvar(idvar)%cname), ' :'
          PRINT *, 'JT = ', jt, ' Total Ekman '//cl_name(jv)//' Transport based on T/S100= ', dtrek*dtrp/dsurf, ' u.S.I'
       END DO

       ! 9) throughflow: section-mean V times section-mean T or S of the top 100 m (Q9)
       dsurf = 0.d0 ; dmv0 = 0.d0 ; dmt0 = 0.d0 ; dms0 = 0.d0
       DO ji = 1, npis
          DO jk = 1, jk100
             dl0   = de1(ji) * ze3(ji,jk)*itm(ji,jk)*1.d0
             dsurf = dsurf + dl0
             dmv0  = dmv0 + zv2(ji,jk)*dl0*1.d0
             dmt0  = dmt0 + zt2(ji,jk)*dl0*1.d0
             dms0  = dms0 + zs2(ji,jk)*dl0*1.d0
          END DO
       END DO
       dtrp = dmv0 * dmt0 / dsurf *1.d0
       CALL put1(32, dtrp/1.d6, 1)
       PRINT *, TRIM(stypvar(32)%cname), ' :'
       PRINT *, 'JT = ', jt, ' Total throughflow temperature transport = ', REAL(dtrp/1.d6), ' Sv.deg C.'
       dtrp = dmv0 * dms0 / dsurf *1.d0
       CALL put1(33, dtrp/1.d6, 1)
       PRINT *, TRIM(stypvar(33)%cname), ' :'
       PRINT *, 'JT = ', jt, ' Total throughflow salinity transport = ', REAL(dtrp/1.d6), ' Sv.PSU'
    END DO

    dtim = getvar1d(cf_vfil, cn_vtimec, npt)
    ierr = putvar1d(ncout, dtim, npt, 'T')
    ierr = closeout(ncout)

  CONTAINS

    SUBROUTINE slab(cd_file, cd_var, pslab)
      !! the (i,k) section of one record of a 3-D variable
      CHARACTER(*), INTENT(in)  :: cd_file, cd_var
      REAL(KIND=4), INTENT(out) :: pslab(:,:)
      INTEGER(KIND=4) :: ik
      DO ik = 1, npk
         zrow(:,:) = getvar(cd_file, cd_var, ik, npis, 1, kimin=iiw, kjmin=ijrapid, ktime=jt)
         pslab(:,ik) = zrow(:,1)
      END DO
    END SUBROUTINE slab

    SUBROUTINE pick(kv)
      INTEGER(KIND=4), INTENT(in) :: kv
      SELECT CASE ( kv )
      CASE ( 1 ) ; zfld => zv2
      CASE ( 2 ) ; zfld => zt2
      CASE ( 3 ) ; zfld => zs2
      END SELECT
    END SUBROUTINE pick

    SUBROUTINE put1(kvar, ddval, klev)
      !! one r4 number of record jt
      INTEGER(KIND=4), INTENT(in) :: kvar, klev
      REAL(KIND=8),    INTENT(in) :: ddval
      REAL(KIND=4) :: zval(1,1)
      zval(1,1) = REAL(ddval)
      ierr = putvar(ncout, id_varout(kvar), zval, klev, 1, 1, ktime=jt)
    END SUBROUTINE put1

    REAL(KIND=8) FUNCTION tsum_v(ki1, ki2)
      !! volume transport through columns ki1..ki2, all levels
      INTEGER(KIND=4), INTENT(in) :: ki1, ki2
      INTEGER(KIND=4) :: ii, ik
      tsum_v = 0.d0
      DO ii = ki1, ki2
         DO ik = 1, npk
            tsum_v = tsum_v + zv2(ii,ik) * de1(ii) * ze3(ii,ik)*1.d0
         END DO
      END DO
    END FUNCTION tsum_v

    REAL(KIND=8) FUNCTION tsum_vx(ki1, ki2)
      !! transport of the picked tracer (v*tracer formed in r4 at the T points, fields already masked)
      INTEGER(KIND=4), INTENT(in) :: ki1, ki2
      INTEGER(KIND=4) :: ii, ik
      tsum_vx = 0.d0
      DO ii = ki1, ki2
         DO ik = 1, npk
            tsum_vx = tsum_vx + zv2(ii,ik) * zfld(ii,ik) * de1(ii) * ze3(ii,ik)*1.d0
         END DO
      END DO
    END FUNCTION tsum_vx

    SUBROUTINE tsum_x(kk2, ddsum, ddarea)
      !! the picked tracer times the cell face, and the masked area, over the whole section down to level kk2
      INTEGER(KIND=4), INTENT(in)  :: kk2
      REAL(KIND=8),    INTENT(out) :: ddsum, ddarea
      INTEGER(KIND=4) :: ii, ik
      ddsum = 0.d0 ; ddarea = 0.d0
      DO ii = 1, npis
         DO ik = 1, kk2
            ddsum  = ddsum  + zfld(ii,ik) * de1(ii) * ze3(ii,ik)*1.d0
            ddarea = ddarea + de1(ii) * ze3(ii,ik)*itm(ii,ik)*1.d0
         END DO
      END DO
    END SUBROUTINE tsum_x

    SUBROUTINE wmean(ki1, ki2, ddsum, ddarea)
      !! masked-area weighted sum of the picked field and the area, columns ki1..ki2, all levels
      INTEGER(KIND=4), INTENT(in)  :: ki1, ki2
      REAL(KIND=8),    INTENT(out) :: ddsum, ddarea
      INTEGER(KIND=4) :: ii, ik
      REAL(KIND=8) :: dlw
      ddsum = 0.d0 ; ddarea = 0.
      DO ii = ki1, ki2
         DO ik = 1, npk
            dlw    = de1(ii) * ze3(ii,ik)*itm(ii,ik)*1.d0
            ddarea = ddarea + dlw
            ddsum  = ddsum + zfld(ii,ik) * dlw *1.d0
         END DO
      END DO
    END SUBROUTINE wmean

    SUBROUTINE CreateOutputRapid
      !! rapid_<moc.nc>: 33 variables, one value per record except amoc_rapid (npk levels); src/cdfmoc.f90:1190-1407
      cf_moc  = 'rapid_'//TRIM(cd_moc)
      nvarout = 33
      ALLOCATE( stypvar(nvarout), ipk(nvarout), id_varout(nvarout) )
      stypvar%cunits            = 'Sverdrup'
      stypvar%rmissing_value    = 99999.
      stypvar%valid_min         = -1000.
      stypvar%valid_max         =  1000.
      stypvar%scale_factor      = 1.
      stypvar%add_offset        = 0.
      stypvar%savelog10         = 0.
      stypvar%conline_operation = 'N/A'
      stypvar%caxis             = 'T'
      ipk(:) = 1
      ipk(1) = npk                                                   ! only amoc_rapid is a profile
      jvar = 0
      CALL def('Sverdrup', 'amoc_rapid',           'Rapid Overturning ',                         'amoc_rapid')
      CALL def('Sverdrup', 'tr_GS',                'Gulf Stream Contribution (Q2)',              'tr_GS')
      CALL def('Sverdrup', 'tr_THERM',             'Overturning contrib of Thermocline waters',  'tr_THERM')
      CALL def('Sverdrup', 'tr_AIW',               'Overturning contrib of intermediate waters', 'tr_AIW')
      CALL def('Sverdrup', 'tr_UNADW',             'Overturning contrib of Upper NADW ',         'tr_UNADW')
      CALL def('Sverdrup', 'tr_LNADW',             'Overturning contrib of Lower NADW ',         'tr_LNADW')
      CALL def('Sverdrup', 'tr_BW',                'Overturning contrib of Bottom Waters',       'tr_BW')
      CALL def('Sverdrup', 'Total_max_amoc_rapid', 'Total Max Rapid Overturning (Q1)',           'Total_max_amoc_rapid')
      CALL def('Sverdrup', 'tr_EKMAN',             'Total Ekman transport (Q3)',                 'tr_EKMAN')
      CALL def('Sverdrup', 'tr_TOTAL',             'Total transport at 26.5N (Q4)',              'tr_TOTAL')
      CALL def('m.s**-1', 'mean_v_total_section',    'Total section mean meridional velocity at 26.5N (Q5)',    'mean_V_total_section')
      CALL def('m.s**-1', 'mean_v_Florida_section',  'Florida section mean meridional velocity at 26.5N (Q5)',  'mean_V_Florida_section')
      CALL def('m.s**-1', 'mean_v_MidOcean_section', 'MidOcean section mean meridional velocity at 26.5N (Q5)', 'mean_V_MidOcean_section')
      CALL def('deg. Celsius', 'mean_T_total_section',    'Total section mean Temperature at 26.5N (Q5)',    'mean_T_total_section')
      CALL def('deg. Celsius', 'mean_T_Florida_section',  'Florida section mean Temperature at 26.5N (Q5)',  'mean_T_Florida_section')
      CALL def('deg. Celsius', 'mean_T_MidOcean_section', 'MidOcean section mean Temperature at 26.5N (Q5)', 'mean_T_MidOcean_section')
      CALL def('PSU', 'mean_S_total_section',    'Total section mean Salinity at 26.5N (Q5)',      'mean_S_total_section')
      CALL def('PSU', 'mean_S_Florida_section',  'Florida section mean Salinity at 26.5N (Q5)',    'mean_S_Florida_section')
      CALL def('PSU', 'mean_S_MidOcean_section', 'MidOcean  section mean Salinity at 26.5N (Q5)',  'mean_S_MidOcean_section')
      CALL def('Sv*deg. Celsius', 'MO_meanVtimesmeanT', 'Mid Ocean mean V x mean T at 26.5N (Q6)', 'MO_meanVtimesmeanT')
      CALL def('Sv*deg. Celsius', 'MO_meanVtimesmeanS', 'Mid Ocean mean V x mean S at 26.5N (Q6)', 'MO_meanVtimesmeanS')
      CALL def('Sv*deg. Celsius', 'Total_temp_transport',    'Total temperature transport V x T at 26.5N (Q7)',            'Total_temp_transport')
      CALL def('Sv*deg. Celsius', 'Florida_temp_transport',  'Florida section temperature transport V x T at 26.5N (Q7)',  'Florida_temp_transport')
      CALL def('Sv*deg. Celsius', 'MidOcean_temp_transport', 'MidOcean section temperature transport V x T at 26.5N (Q7)', 'MidOcean_temp_transport')
      CALL def('Sv*deg. Celsius', 'Ekman_temp_transport_SST',  'Ekman temperature transport based on SST at 26.5N (Q8.1)',   'Total_temp_transport_SST')
      CALL def('Sv*deg. Celsius', 'Ekman_temp_transport_T100', 'Ekman temperature transport based on T100m at 26.5N (Q8.2)', 'Total_temp_transport_T100')
      CALL def('Sv*PSU', 'Total_salt_transport',    'Total salinity transport V x S at 26.5N (Q7)',            'Total_salt_transport')
      CALL def('Sv*PSU', 'Florida_salt_transport',  'Florida section salinity transport V x S at 26.5N (Q7)',  'Florida_salt_transport')
      CALL def('Sv*PSU', 'MidOcean_salt_transport', 'MidOcean section salinity transport V x S at 26.5N (Q7)', 'MidOcean_salt_transport')
      CALL def('Sv*PSU', 'Ekman_salt_transport_SSS',  'Ekman salinity transport based on SSS at 26.5N (Q8.1)',   'Total_salt_transport_SSS')
      CALL def('Sv*PSU', 'Ekman_salt_transport_S100', 'Ekman salinity transport based on S100m at 26.5N (Q8.2)', 'Total_salt_transport_S100')
      CALL def('Sv*deg. Celsius', 'Total_meanVtimesmeanT', 'Throughflow temperature transport = mean V x mean T 26.5N (Q9)', 'Total_meanVtimesmeanT')
      CALL def('Sv*PSU',          'Total_meanVtimesmeanS', 'Throughflow salinity transport = mean V x mean S 26.5N (Q9)',    'Total_meanVtimesmeanS')
      PRINT *, 'idvar = ', jvar

      ncout = create      (cf_moc, 'none', 1, 1, npk, cdep=cn_vdepthw)
      ierr  = createvar   (ncout, stypvar, nvarout, ipk, id_varout)
      ierr  = putheadervar(ncout, cf_vfil, 1, 1, npk, pnavlon=rdumlon, pnavlat=rdumlat, pdep=gdepw)
    END SUBROUTINE CreateOutputRapid

    SUBROUTINE def(cd_units, cd_name, cd_long, cd_short)
      CHARACTER(*), INTENT(in) :: cd_units, cd_name, cd_long, cd_short
      jvar = jvar + 1
      stypvar(jvar)%cunits      = cd_units
      stypvar(jvar)%cname       = cd_name
      stypvar(jvar)%clong_name  = cd_long
      stypvar(jvar)%cshort_name = cd_short
    END SUBROUTINE def

  END SUBROUTINE rapid_amoc

END MODULE cdfmoc_rapid
