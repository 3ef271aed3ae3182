PROGRAM cdfio_public
  ! Test driver (tests/test_cdfio_public.py): calls every routine of cdfio's public list that no tool of the hot path
  ! uses, the way a reference program would (argument order and keywords of src/cdfio.F90), on the files the Python
  ! side wrote into the directory given as argument; prints "key values" lines the test compares with NumPy.
  USE cdfio
  USE modcdfnames
  IMPLICIT NONE
  INTEGER(KIND=4), PARAMETER :: npi = 6, npj = 5, npk = 4, npt = 3
  CHARACTER(LEN=256) :: cdir, cf_t, cf_e, cf_xz, cf_o, cldep, cl1, cl2, cl3, cl4, cl5, cl6, clvar, clunits, cllong, clshort
  CHARACTER(LEN=20)  :: cltime
  INTEGER(KIND=4) :: ierr, ncout, ipk(2), id_varout(2), ivid(5)
  REAL(KIND=4) :: v3(npi-1,npj-2,npk-1), v3t(npi,npj,npt), v4(npi-2,npj-1,npk,npt-1), vxz(npi-1,npk-1), vyz(npj-1,npk), vx2(npi,npk)
  REAL(KIND=4) :: zdep(npk), zwin(3,2), zprof(npk), z1d(npk), zlon(npi,npj), zlat(npi,npj)
  REAL(KIND=8) :: d1d(npk)
  TYPE(variable) :: stypvar(2)

  CALL getarg(1, cdir)
  cf_t  = TRIM(cdir)//'/T.nc'
  cf_e  = TRIM(cdir)//'/edit.nc'
  cf_xz = TRIM(cdir)//'/xz.nc'
  cf_o  = TRIM(cdir)//'/out.nc'

  ivid = getvarid(cf_t, 5)
  PRINT '(a,5i4)', 'getvarid', ivid
  zdep = getdimvar(cf_t, npk, cd_depnam=cldep)
  PRINT '(a,1x,a)', 'getdimvar_name', TRIM(cldep)
  PRINT '(a,4es16.8)', 'getdimvar', zdep
  ierr = gettimeatt(cf_t, cn_vtimec, cl1, cl2, cl3, cl4, cl5, cl6)
  PRINT '(13a)', 'gettimeatt |', TRIM(cl1), '|', TRIM(cl2), '|', TRIM(cl3), '|', TRIM(cl4), '|', TRIM(cl5), '|', TRIM(cl6), '|'

  v3 = getvar3d(cf_t, cn_votemper, npi-1, npj-2, npk-1, kimin=2, kjmin=3, kkmin=2, ktime=2)
  PRINT '(a,999es16.8)', 'getvar3d', v3
  v3t = getvar3dt(cf_t, cn_votemper, 3, npi, npj, npt)
  PRINT '(a,999es16.8)', 'getvar3dt', v3t
  v3t(:,:,1:2) = getvar3dt(cf_t, cn_sossheig, 1, npi, npj, 2, ktmin=2)
  PRINT '(a,999es16.8)', 'getvar3dt_2d', v3t(:,:,1:2)
  v4 = getvar4d(cf_t, 'packed', npi-2, npj-1, npk, npt-1, kimin=3, kjmin=2, ktmin=2)
  PRINT '(a,999es16.8)', 'getvar4d', v4
  vxz = getvarxz(cf_t, cn_votemper, 4, npi-1, npk-1, kimin=2, kkmin=2, ktime=3)
  PRINT '(a,999es16.8)', 'getvarxz', vxz
  vyz = getvaryz(cf_t, 'packed', 5, npj-1, npk, kjmin=2, ktime=2)
  PRINT '(a,999es16.8)', 'getvaryz', vyz
  vx2 = getvarxz(cf_xz, 'sec', 1, npi, npk, ktime=2)
  PRINT '(a,999es16.8)', 'getvarxz_noy', vx2
  PRINT '(a)', 'gettimeseries_begin'
  CALL gettimeseries(cf_t, cn_votemper, 2, 3, klev=4)
  CALL gettimeseries(cf_t, cn_sossheig, 6, 1)
  PRINT '(a)', 'gettimeseries_end'

  ! ---- editing an existing file
  ierr = atted(cf_e, 'votemper', 'units', 'degC')                  ! overwrite, longer than before
  ierr = atted(cf_e, 'votemper', 'a_new_text_attribute', 'made by atted_char, long enough to move the data section')
  ierr = atted(cf_e, 'votemper', 'valid_max', 41.5)               ! overwrite r4
  ierr = atted(cf_e, 'sossheig', 'new_r4', -2.25)
  clvar = 'sossheig' ; clunits = 'metres' ; cllong = 'sea surface height' ; clshort = 'ssh'
  ierr = cvaratt(cf_e, clvar, clunits, -9999., cllong, clshort)
  PRINT '(a,i4)', 'cvaratt', ierr
  zwin = RESHAPE((/ 1., 2., 3., 4., 5., 6. /), (/ 3, 2 /))
  ierr = putvar(cf_e, 'votemper', 2, 3, 2, kimin=2, kjmin=4, ktime=3, ptab=zwin, kwght=7)
  PRINT '(a,i4)', 'reputvarr4', ierr
  ierr = putvar(cf_e, 'sossheig', 1, 3, 2, kimin=4, kjmin=1, ktime=2, ptab=-zwin)
  z1d = (/ 10., 20., 30., 40. /) ; d1d = (/ 1.5d0, 2.5d0, 3.5d0, 4.5d0 /)
  ierr = putvar1d(cf_e, 'deptht', z1d, npk)
  ierr = putvar1d(cf_e, 'dprof', d1d, npk)

  ! ---- a new file: puttimeatt, putvare3
  ipk = (/ npk, 1 /)
  stypvar(1)%cname = 'prof' ; stypvar(1)%cunits = 'm' ; stypvar(1)%clong_name = 'a profile' ; stypvar(1)%cshort_name = 'prof'
  stypvar(2)%cname = 'flat' ; stypvar(2)%cunits = 'm' ; stypvar(2)%clong_name = 'a map'     ; stypvar(2)%cshort_name = 'flat'
  ncout = create(cf_o, cf_t, 1, 1, npk)
  ierr  = createvar(ncout, stypvar, 2, ipk, id_varout)
  zlon = 0. ; zlat = 0.
  ierr  = putheadervar(ncout, cf_t, 1, 1, npk, pnavlon=zlon(1:1,1:1), pnavlat=zlat(1:1,1:1))
  zprof = (/ 4., 3., 2., 1. /)
  ierr = putvar(ncout, id_varout(1), zprof, npk, 'vert', ktime=1)
  ierr = putvar(ncout, id_varout(1), 2. * zprof, npk, 'vert', ktime=2, kwght=3)
  ierr = putvar0d(ncout, id_varout(2), 5.5, ktime=1)
  ierr = putvar0d(ncout, id_varout(2), 6.5, ktime=2)
  d1d(1:2) = (/ 86400.d0, 172800.d0 /)
  ierr = putvar1d(ncout, d1d(1:2), 2, 'T')
  cltime = cn_vtimec
  cl1 = 'gregorian' ; cl2 = 'Time' ; cl3 = 'Time axis' ; cl4 = 'T' ; cl5 = 'seconds since 1958-01-01 00:00:00' ; cl6 = '1958-JAN-01 00:00:00'
  ierr = puttimeatt(ncout, cltime, cl1, cl2, cl3, cl4, cl5, cl6)
  PRINT '(a,i4)', 'puttimeatt', ierr
  ierr = closeout(ncout)
  PRINT '(a)', 'done'
END PROGRAM cdfio_public
