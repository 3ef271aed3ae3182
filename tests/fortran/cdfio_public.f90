PROGRAM cdfio_public
  ! Test driver (tests/test_cdfio_public.py): calls the routines of cdfio the tools of this build rely on, with the
  ! reference's argument order and keywords (src/cdfio.F90), on the files the Python side wrote into the directory given
  ! as argument; prints "key values" lines the test compares with NumPy.
  USE cdfio
  USE modcdfnames
  IMPLICIT NONE
  INTEGER(KIND=4), PARAMETER :: npi = 6, npj = 5, npk = 4, npt = 3
  CHARACTER(LEN=256) :: cdir, cf_t, cf_m, cf_o, cltrue, clvar, clunits, cllong, clshort, clmiss
  CHARACTER(LEN=256), ALLOCATABLE :: clnames(:)
  INTEGER(KIND=4) :: ierr, ncout, ipk(4), id_varout(4), nvars, istat
  INTEGER(KIND=4), ALLOCATABLE :: ipkin(:)
  REAL(KIND=4) :: v3(npi-1,npj-2,npk-1), w3(npi,npj,npk), vw(4,npj), zspv, z2(npi,npj), zline(npj)
  REAL(KIND=8) :: d2(npi,npj), d1d(2)
  INTEGER(KIND=1) :: m3(npi,npj,npk)
  INTEGER(KIND=2) :: i2(npi,npj)
  TYPE(variable) :: stypvar(4)
  TYPE(variable), ALLOCATABLE :: stypin(:)

  CALL getarg(1, cdir)
  cf_t = TRIM(cdir)//'/T.nc'
  cf_m = TRIM(cdir)//'/mesh.nc'
  cf_o = TRIM(cdir)//'/out.nc'

  ! ---- inquiries
  PRINT '(a,4i6)', 'getdim', getdim(cf_t, cn_x), getdim(cf_t, cn_y), getdim(cf_t, 'dep', cdtrue=cltrue), getdim(cf_t, cn_t)
  PRINT '(a,1x,a)', 'getdim_true', TRIM(cltrue)
  ierr = getdim(cf_t, 'nosuchdim', cdtrue=cltrue, kstatus=istat)
  PRINT '(a,2i4,1x,a)', 'getdim_absent', ierr, istat, TRIM(cltrue)
  clvar = 'votemper'
  PRINT '(a,2i4)', 'getvdim', getvdim(cf_t, clvar), MERGE(1, 0, chkvar(cf_t, 'nosuchvar', .FALSE.))
  zspv = getspval(cf_t, 'sossheig', clmiss)
  PRINT '(a,es16.8,1x,a)', 'getspval', zspv, TRIM(clmiss)
  PRINT '(a,es16.8,i6,1x,a)', 'getatt', getatt(cf_t, 'votemper', 'valid_max'), getatt(cf_t, 'votemper', 'iweight', 1), &
       &  TRIM(getatt(cf_t, 'votemper', 'units', 'c'))
  clunits = 'keep' ; cllong = 'keep' ; clshort = 'keep'
  ierr = getvaratt(cf_t, 'votemper', clunits, zspv, cllong, clshort)
  PRINT '(a,es16.8,7a)', 'getvaratt', zspv, ' |', TRIM(clunits), '|', TRIM(cllong), '|', TRIM(clshort), '|'
  nvars = getnvar(cf_t)
  ALLOCATE( clnames(nvars), stypin(nvars), ipkin(nvars) )
  clnames = getvarname(cf_t, nvars, stypin)
  ipkin = getipk(cf_t, nvars)
  PRINT '(a,i4)', 'getnvar', nvars
  PRINT '(a,99(1x,a))', 'getvarname', (TRIM(clnames(ierr)), ierr = 1, nvars)
  PRINT '(a,99i4)', 'getipk', ipkin
  PRINT '(a,es16.8,3a)', 'stypin_votemper', stypin(5)%valid_max, ' |', TRIM(stypin(5)%cunits), '|'

  ! ---- slab and block reads
  z2 = getvar(cf_t, 'packed', 2, npi, npj, ktime=2)
  PRINT '(a,999es16.8)', 'getvar_packed', z2
  vw = getvar(cf_t, cn_votemper, 3, 4, npj, kimin=4, ktime=1)            ! columns 4 5, then 2 3 (the wrap skips column 1 = 6)
  PRINT '(a,999es16.8)', 'getvar_wrap', vw
  vw = getvar(cf_t, cn_votemper, 3, 4, npj, kimin=3, ktime=1)            ! ends on the last column: no wrap
  PRINT '(a,999es16.8)', 'getvar_edge', vw
  v3 = getvar3d(cf_t, cn_votemper, npi-1, npj-2, npk-1, kimin=2, kjmin=3, kkmin=2, ktime=2)
  PRINT '(a,999es16.8)', 'getvar3d', v3
  CALL readvar3d(cf_t, 'packed', w3, ktime=3)
  PRINT '(a,999es16.8)', 'readvar3d_packed', w3
  CALL readvar3d(cf_t, cn_sossheig, w3, ktime=2)                           ! (x,y,t): the slab on every level
  PRINT '(a,999es16.8)', 'readvar3d_xyt', w3
  CALL readvar3d(cf_m, 'e3static', w3)                                     ! (x,y,z) without a time axis
  PRINT '(a,999es16.8)', 'readvar3d_xyz', w3
  CALL readvar3d(cf_m, 'e3static', v3, kimin=2, kjmin=3, kkmin=2)
  PRINT '(a,999es16.8)', 'readvar3d_xyz_window', v3
  CALL readvar3d(cf_m, 'area', w3)                                         ! (x,y)
  PRINT '(a,999es16.8)', 'readvar3d_xy', w3
  CALL readvar3d(cf_t, cn_votemper, w3(1:4,:,:), ktime=2, kimin=4)         ! over the eastern edge
  PRINT '(a,999es16.8)', 'readvar3d_wrap', w3(1:4,:,:)
  CALL readvar3d_i1(cf_m, 'tmaskstatic', m3)
  PRINT '(a,999i3)', 'readvar3d_i1_xyz', m3
  CALL readvar3d_i1(cf_t, 'tmask4d', m3, ktime=2)
  PRINT '(a,999i3)', 'readvar3d_i1_xyzt', m3
  d1d = getvar1d(cf_t, cn_vtimec, 2)
  PRINT '(a,2es24.16)', 'getvar1d', d1d
  d1d = getvar1d(cf_t, 'nosuchvar', 2, istat)
  PRINT '(a,es24.16,i4)', 'getvar1d_absent', d1d(1), MERGE(1, 0, istat /= 0)

  ! ---- a new file: every kind of putvar
  ipk = (/ npk, 1, 1, -1 /)
  stypvar(:)%cunits = 'm' ; stypvar(:)%rmissing_value = -1. ; stypvar(:)%valid_min = 0. ; stypvar(:)%valid_max = 9.
  stypvar(:)%clong_name = 'long' ; stypvar(:)%cshort_name = 'short' ; stypvar(:)%conline_operation = 'N/A' ; stypvar(:)%caxis = 'TZYX'
  stypvar(1)%cname = 'cube'  ; stypvar(1)%cprecision = 'r8'
  stypvar(2)%cname = 'plane' ; stypvar(2)%scale_factor = 0.5                ! becomes a short
  stypvar(3)%cname = 'count' ; stypvar(3)%cprecision = 'i2'
  stypvar(4)%cname = 'scalar'
  ncout = create(cf_o, cf_t, npi, npj, npk, cdep='deptht')
  ierr  = createvar(ncout, stypvar, 4, ipk, id_varout, cdglobal='made by the test')
  ierr  = putheadervar(ncout, cf_t, npi, npj, npk)
  d2 = RESHAPE((/ (DBLE(ierr) * 0.25d0, ierr = 1, npi*npj) /), (/ npi, npj /))
  DO ierr = 1, npk
     istat = putvar(ncout, id_varout(1), d2 * ierr, ierr, npi, npj, ktime=2)
  END DO
  z2 = REAL(d2, 4)
  istat = putvar(ncout, id_varout(2), z2, 1, npi, npj, ktime=1, kwght=5)
  i2 = INT(4.d0 * d2, 2)
  istat = putvar(ncout, id_varout(3), i2, 1, npi, npj, ktime=2)
  istat = putvar0d(ncout, id_varout(4), 5.5, ktime=1)
  istat = putvar0d(ncout, id_varout(4), RESHAPE((/ 6.5 /), (/ 1, 1 /)), ktime=2)
  d1d = (/ 86400.d0, 172800.d0 /)
  istat = putvar1d(ncout, d1d, 2, 'T')
  ierr = closeout(ncout)
  ! a zonal line (1-D r4) into a (1,npj) file, with the coordinates given
  cf_o = TRIM(cdir)//'/zonal.nc'
  stypvar(1)%cname = 'line' ; stypvar(1)%cprecision = 'r4'
  ncout = create(cf_o, 'none', 1, npj, npk, cdep='depthw')
  ierr  = createvar(ncout, stypvar, 1, ipk, id_varout)
  z2 = 0.
  ierr  = putheadervar(ncout, 'none', 1, npj, npk, pnavlon=z2(1:1,:), pnavlat=z2(1:1,:) + 2., pdep=(/ 1., 2., 3., 4. /))
  zline = (/ 1., 2., 3., 4., 5. /)
  istat = putvar(ncout, id_varout(1), zline, 3, 1, npj, ktime=1)
  ierr = closeout(ncout)
  CALL the_rest_of_the_public_list
  PRINT '(a)', 'done'

CONTAINS

  SUBROUTINE the_rest_of_the_public_list
    ! the names of src/cdfio.F90:173-180 that none of the 18 programs calls: one call each, the reference's argument order
    CHARACTER(LEN=256) :: cl_cal, cl_tit, cl_lon, cl_axi, cl_uni, cl_ori, cl_dep, cf_e, cl256a, cl256b, cl256c, cl256d
    CHARACTER(LEN=20)  :: cl_tvar
    INTEGER(KIND=4) :: ids(8), iout, ivarid(1), ipk1(1)
    REAL(KIND=4) :: zdep(npk), xz(npi-1,npk-1), yz(npj,npk), xyt(npi,npj,2), x4(3,2,2,2), zspv2
    TYPE(variable) :: styp1(1)
    ids = getvarid(cf_t, 8)
    PRINT '(a,99i4)', 'getvarid', ids
    ierr = gettimeatt(cf_t, cn_vtimec, cl_cal, cl_tit, cl_lon, cl_axi, cl_uni, cl_ori)
    PRINT '(a,1x,13a)', 'gettimeatt', '|', TRIM(cl_cal), '|', TRIM(cl_tit), '|', TRIM(cl_lon), '|', TRIM(cl_axi), '|', TRIM(cl_uni), '|', TRIM(cl_ori), '|'
    zdep = getdimvar(cf_t, npk, cl_dep)
    PRINT '(a,99es16.8)', 'getdimvar', zdep
    PRINT '(a,1x,a)', 'getdimvar_name', TRIM(cl_dep)
    xz = getvarxz(cf_t, 'packed', 3, npi-1, npk-1, kimin=2, kkmin=2, ktime=2)
    PRINT '(a,999es16.8)', 'getvarxz', xz
    yz = getvaryz(cf_t, cn_votemper, 4, npj, npk, ktime=3)
    PRINT '(a,999es16.8)', 'getvaryz', yz
    xyt = getvar3dt(cf_t, cn_votemper, 2, npi, npj, 2, ktmin=2)
    PRINT '(a,999es16.8)', 'getvar3dt', xyt
    xyt = getvar3dt(cf_t, cn_sossheig, 1, npi, npj, 2, kimin=1, kjmin=1, ktmin=1)
    PRINT '(a,999es16.8)', 'getvar3dt_xyt', xyt
    x4 = getvar4d(cf_t, 'packed', 3, 2, 2, 2, kimin=2, kjmin=3, kkmin=2, ktmin=2)
    PRINT '(a,999es16.8)', 'getvar4d', x4
    PRINT '(a)', 'gettimeseries_begin'
    CALL gettimeseries(cf_t, 'packed', 4, 3, klev=2)
    PRINT '(a)', 'gettimeseries_end'
    ! a file of its own to edit: create + copyatt (no reference file) + puttimeatt, then the editors on the closed file
    cf_e = TRIM(cdir)//'/edit.nc'
    styp1(1)%cname = 'field' ; styp1(1)%cunits = 'old' ; styp1(1)%rmissing_value = -9. ; styp1(1)%clong_name = 'old long'
    styp1(1)%cshort_name = 'old short' ; styp1(1)%conline_operation = 'N/A' ; styp1(1)%caxis = 'TYX'
    ipk1 = 1
    iout = create(cf_e, cf_t, npi, npj, 0)
    ierr = createvar(iout, styp1, 1, ipk1, ivarid)
    ierr = putheadervar(iout, cf_t, npi, npj, 0)
    cl_tvar = cn_vtimec
    cl_cal = 'gregorian' ; cl_tit = 'Time!' ; cl_lon = 'the axis of time' ; cl_axi = 'T' ; cl_uni = 'days since then' ; cl_ori = 'then'
    ierr = puttimeatt(iout, cl_tvar, cl_cal, cl_tit, cl_lon, cl_axi, cl_uni, cl_ori)
    ierr = closeout(iout)
    iout = ncopen(cf_e)
    PRINT '(a,i3)', 'ncopen', MERGE(1, 0, iout >= 0)
    ierr = closeout(iout)
    cl256a = cf_e ; cl256b = 'field' ; cl256c = 'new units' ; cl256d = 'new long'
    ierr = cvaratt(cl256a, cl256b, cl256c, 77., cl256d, cl256b)
    ierr = atted(cf_e, 'field', 'comment', 'edited')
    ierr = atted(cf_e, 'field', 'factor', 2.5)
    ierr = atted(cf_e, 'field', 'comment', 'edited twice')
    zspv2 = getspval(cf_e, 'field')
    PRINT '(a,es16.8,1x,a,1x,a)', 'edited', zspv2, TRIM(getatt(cf_e, 'field', 'units', 'c')), TRIM(getatt(cf_e, 'field', 'comment', 'c'))
    ! the specifics of putvar / putvar1d that take a file NAME (reputvarr4, reputvar1d4 / 8) and the profile one (putvare3)
    z2 = 7.
    ierr = putvar(cf_e, 'field', 1, 2, 2, kimin=3, kjmin=2, ktime=1, ptab=z2(1:2,1:2), kwght=9)
    ierr = putvar1d(cf_e, cn_vtimec, (/ 5.d0 /), 1)
    z2 = getvar(cf_e, 'field', 1, npi, npj)
    PRINT '(a,999es16.8)', 'reputvar', z2
    PRINT '(a,i4,es16.8)', 'reputvar_att', getatt(cf_e, 'field', 'iweight', 1), REAL(getvar1d(cf_e, cn_vtimec, 1), 4)
    ! copyatt between two open files, and from no file at all
    cf_o = TRIM(cdir)//'/copy.nc'
    iout = create(cf_o, cf_t, npi, npj, 0)
    ierr = createvar(iout, styp1, 1, ipk1, ivarid)
    ierr = closeout(iout)
  END SUBROUTINE the_rest_of_the_public_list
END PROGRAM cdfio_public
