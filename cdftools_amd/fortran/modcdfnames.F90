MODULE modcdfnames
  !!======================================================================
  !! modcdfnames : every dimension / variable / file name the tools look for, as module variables.
  !!
  !! This is the configuration surface of CDFTOOLS (reference: src/modcdfnames.F90:17-187 for the names
  !! and their defaults, :202-257 for the namelist groups, :268-326 ReadCdfNames, :362-388 chkenv) and
  !! is part of the drop-in boundary: same identifiers, same defaults, same namelist groups, same
  !! lookup order (./nam_cdf_names or $NAM_CDF_NAMES, then $HOME/CDFTOOLS_cfg/), same CDFT_* overrides.
  !! Only the implementation is new.
  !!======================================================================
  !!
  !! Compiled with -Dkey_CMIP6 the twenty names whose CMIP6 spelling differs take the defaults of the reference's
  !! src/modcdfnames_CMIP6.h90 (thetao, so, uo, vo, wo, zos, thkcello ...) and the seven names only that header has
  !! (cn_votemper2 ... cn_somxldep) exist; every other name keeps its default and stays declared, so a program written
  !! against either name set compiles against this module.
  !!======================================================================
#if defined key_CMIP6
#  define CMIP6_(std, cmip) cmip
#else
#  define CMIP6_(std, cmip) std
#endif
  IMPLICIT NONE
  PUBLIC
  INTEGER, PARAMETER, PRIVATE :: lc = 256

  CHARACTER(LEN=lc) :: cn_output = CMIP6_('default', 'CMIP6')

  ! --- dimensions and their coordinate variables -------------------------------------------
  CHARACTER(LEN=lc) :: cn_x = 'x', cn_y = 'y', cn_z = 'depth', cn_t = 'time_counter'
  CHARACTER(LEN=lc) :: cn_vlon2d = 'nav_lon', cn_vlat2d = 'nav_lat', cn_vlon1d = 'lon', cn_vlat1d = 'lat'
  CHARACTER(LEN=lc) :: cn_vdeptht = 'deptht', cn_vdepthu = 'depthu', cn_vdepthv = 'depthv', cn_vdepthw = 'depthw'
  CHARACTER(LEN=lc) :: cn_vtimec = 'time_counter'
  CHARACTER(LEN=lc) :: cn_missing_value = '_FillValue'

#if defined key_CMIP6
  ! --- the names only the CMIP6 set has (src/modcdfnames_CMIP6.h90:74-85) -----------------------
  CHARACTER(LEN=lc) :: cn_votemper2 = 'thetao_sqd', cn_vosaline2 = 'so_sqd', cn_vozocrtx2 = 'u2o', cn_vomecrty2 = 'v2o'
  CHARACTER(LEN=lc) :: cn_vovecrtz2 = 'w2o', cn_sossheig2 = 'zossq', cn_somxldep = 'mlotst'
#endif

  ! --- metrics ---------------------------------------------------------------------------------
  CHARACTER(LEN=lc) :: cn_ve1t = 'e1t', cn_ve2t = 'e2t', cn_ve1u = 'e1u', cn_ve2u = 'e2u'
  CHARACTER(LEN=lc) :: cn_ve1v = 'e1v', cn_ve2v = 'e2v', cn_ve1f = 'e1f', cn_ve2f = 'e2f'
  CHARACTER(LEN=lc) :: cn_ve3t1d = 'e3t', cn_ve3t = 'e3t', cn_ve3w = 'e3w', cn_ve3u = 'e3u', cn_ve3v = 'e3v'
  CHARACTER(LEN=255) :: cn_ve3w1d = 'e3w'
  CHARACTER(LEN=lc) :: cn_ve3tvvl = CMIP6_('e3t', 'thkcello'), cn_ve3wvvl = CMIP6_('e3w', 'thkcello'), cn_ve3uvvl = CMIP6_('e3u', 'thkcello'), cn_ve3vvvl = CMIP6_('e3v', 'thkcello')
  CHARACTER(LEN=lc) :: cn_ve3t0 = 'e3t_0', cn_ve3w0 = 'e3w_0', cn_ve3u0 = 'e3u_0', cn_ve3v0 = 'e3v_0'
  CHARACTER(LEN=lc) :: cn_vff = 'ff'
  CHARACTER(LEN=lc) :: cn_gdept = 'gdept', cn_gdepw = 'gdepw', cn_hdept = 'hdept', cn_hdepw = 'hdepw'
  CHARACTER(LEN=lc) :: cn_dept3d = 'gdept_0', cn_depu3d = 'depu3d', cn_depw3d = 'depw3d'
  CHARACTER(LEN=lc) :: cn_glamt = 'glamt', cn_gphit = 'gphit', cn_glamu = 'glamu', cn_gphiu = 'gphiu'
  CHARACTER(LEN=lc) :: cn_glamv = 'glamv', cn_gphiv = 'gphiv', cn_glamf = 'glamf', cn_gphif = 'gphif'

  ! --- masks -----------------------------------------------------------------------------------
  CHARACTER(LEN=lc) :: cn_tmask = 'tmask', cn_umask = 'umask', cn_vmask = 'vmask', cn_fmask = 'fmask'
  CHARACTER(LEN=lc) :: cn_tmaskutil = 'tmaskutil', cn_polymask = 'polymask'
  CHARACTER(LEN=lc) :: cn_tmaskatl = 'tmaskatl', cn_tmaskpac = 'tmaskpac', cn_tmaskind = 'tmaskind'
  CHARACTER(LEN=lc) :: cn_tmaskant = 'tmaskant', cn_tmaskmed = 'tmaskmed'

  ! --- mesh / mask files ----------------------------------------------------------------------
  CHARACTER(LEN=lc) :: cn_fzgr = 'mesh_zgr.nc', cn_fhgr = 'mesh_hgr.nc', cn_fmsk = 'mask.nc'
  CHARACTER(LEN=lc) :: cn_fe3t = 'mesh_zgr.nc', cn_fe3u = 'mesh_zgr.nc', cn_fe3v = 'mesh_zgr.nc', cn_fe3w = 'mesh_zgr.nc'
  CHARACTER(LEN=lc) :: cn_fcoo = 'coordinates.nc', cn_fbasins = 'new_maskglo.nc'

  ! --- model variables ---------------------------------------------------------------------
  CHARACTER(LEN=lc) :: cn_votemper = CMIP6_('votemper', 'thetao'), cn_vosaline = CMIP6_('vosaline', 'so'), cn_vozocrtx = CMIP6_('vozocrtx', 'uo')
  CHARACTER(LEN=lc) :: cn_vomecrty = CMIP6_('vomecrty', 'vo'), cn_vomeeivv = 'vomeeivv', cn_vovecrtz = CMIP6_('vovecrtz', 'wo')
  CHARACTER(LEN=lc) :: cn_sossheig = CMIP6_('sossheig', 'zos'), cn_somxl010 = 'somxl010', cn_somxlt02 = 'somxlt02'
  CHARACTER(LEN=lc) :: cn_sozotaux = CMIP6_('sozotaux', 'tauuo')
  CHARACTER(LEN=lc) :: cn_sohefldo = CMIP6_('sohefldo', 'hfds'), cn_solhflup = 'solhflup', cn_sosbhfup = 'sosbhfup'
  CHARACTER(LEN=lc) :: cn_solwfldo = 'solwfldo', cn_soshfldo = 'soshfldo'
  CHARACTER(LEN=lc) :: cn_sowaflup = 'sowaflup', cn_sowaflcd = 'sowaflcd', cn_sowafldp = 'sowafldp'
  CHARACTER(LEN=lc) :: cn_sorunoff = 'sorunoff', cn_iowaflup = 'iowaflup', cn_soicecov = 'soicecov'
  CHARACTER(LEN=lc) :: cn_iicbmass = 'berg_mass', cn_iicbmelt = 'berg_melt'
  ! overturning
  CHARACTER(LEN=lc) :: cn_zomsfatl = 'zomsfatl', cn_zomsfglo = 'zomsfglo', cn_zomsfpac = 'zomsfpac'
  CHARACTER(LEN=lc) :: cn_zomsfinp = 'zomsfinp', cn_zomsfind = 'zomsfind', cn_zomsfinp0 = 'zomsfinp0'
  CHARACTER(LEN=lc) :: cn_zoisoatl = 'zoisoatl', cn_zoisoglo = 'zoisoglo', cn_zoisopac = 'zoisopac'
  CHARACTER(LEN=lc) :: cn_zoisoinp = 'zoisoinp', cn_zoisoind = 'zoisoind', cn_zoisoinp0 = 'zoisoinp0'
  ! transports
  CHARACTER(LEN=lc) :: cn_vozout = CMIP6_('vozout', 'uto'), cn_vomevt = CMIP6_('vomevt', 'vto'), cn_vozous = CMIP6_('vozous', 'uso'), cn_vomevs = CMIP6_('vomevs', 'vso')
  CHARACTER(LEN=lc) :: cn_sozout = 'sozout', cn_somevt = 'somevt', cn_sozous = 'sozous', cn_somevs = 'somevs'
  CHARACTER(LEN=lc) :: cn_sozoutrp = CMIP6_('sozoutrp', 'umo_vint'), cn_somevtrp = 'somevtrp', cn_vouv = 'vouv'
  ! density
  CHARACTER(LEN=lc) :: cn_vosigma0 = 'vosigma0', cn_vosigmai = 'vosigmai', cn_vosigntr = 'vosigntr'
  CHARACTER(LEN=lc) :: cn_vodepiso = 'vodepiso', cn_isothick = 'isothick'
  ! passive tracers, ice
  CHARACTER(LEN=lc) :: cn_invcfc = 'INVCFC', cn_cfc11 = 'CFC11', cn_pendep = 'pendep'
  CHARACTER(LEN=lc) :: cn_iicethic = 'iicethic', cn_ileadfra = 'ileadfra', cn_iicevelu = 'iicevelu', cn_iicevelv = 'iicevelv'
  CHARACTER(LEN=lc) :: cn_iicethic3 = CMIP6_('sivolu', 'iicethic'), cn_ileadfra3 = CMIP6_('siconc', 'iiceconc'), cn_iicevelu3 = 'sivelu', cn_iicevelv3 = 'sivelv'
  ! bathymetry
  CHARACTER(LEN=lc) :: cn_fbathymet = 'bathy_meter.nc', cn_fbathylev = 'bathy_level.nc'
  CHARACTER(LEN=lc) :: cn_bathymet = 'Bathymetry', cn_bathylev = 'bathy_level', cn_mbathy = 'mbathy'

  ! --- lists used by cdfmoy (kept so that existing namelist files still parse) ---------------
  INTEGER(KIND=4), PARAMETER :: jp_sqdvarmax = 10, jp_cubvarmax = 10
  INTEGER(KIND=4) :: nn_sqdvar = 4, nn_cubvar = 2
  CHARACTER(LEN=15), DIMENSION(jp_sqdvarmax) :: cn_sqdvar = (/ 'vozocrtx       ', 'vomecrty       ', 'vovecrtz       ', &
       & 'sossheig       ', '               ', '               ', '               ', '               ',               &
       & '               ', '               ' /)
  CHARACTER(LEN=15), DIMENSION(jp_cubvarmax) :: cn_cubvar = (/ 'sossheig       ', 'votemper       ', '               ', &
       & '               ', '               ', '               ', '               ', '               ',               &
       & '               ', '               ' /)

  NAMELIST /namdim/     cn_x, cn_y, cn_z, cn_t
  NAMELIST /namdimvar/  cn_vlon2d, cn_vlat2d, cn_vdeptht, cn_vdepthu, cn_vdepthv, cn_vdepthw, cn_vtimec, cn_missing_value
  NAMELIST /nammetrics/ cn_ve1t, cn_ve1u, cn_ve1v, cn_ve1f, cn_ve2t, cn_ve2u, cn_ve2v, cn_ve2f, cn_ve3t1d, cn_ve3w1d, &
       &                cn_ve3t, cn_ve3w, cn_ve3u, cn_ve3v, cn_ve3tvvl, cn_ve3wvvl, cn_ve3uvvl, cn_ve3vvvl, cn_vff,  &
       &                cn_glamt, cn_glamu, cn_glamv, cn_glamf, cn_gphit, cn_gphiu, cn_gphiv, cn_gphif,             &
       &                cn_gdept, cn_gdepw, cn_hdept, cn_hdepw
  NAMELIST /nambroken/  cn_dept3d, cn_depu3d, cn_depw3d
  NAMELIST /namvars/    cn_votemper, cn_vosaline, cn_vozocrtx, cn_vomecrty, cn_vomeeivv, cn_vovecrtz, cn_sossheig, &
       &                cn_somxl010, cn_somxlt02, cn_sohefldo, cn_solhflup, cn_sosbhfup, cn_solwfldo, cn_soshfldo, &
       &                cn_sowaflup, cn_sowaflcd, cn_sowafldp, cn_iowaflup, cn_zomsfatl, cn_zomsfglo, cn_zomsfpac, &
       &                cn_zomsfinp, cn_zomsfind, cn_zoisoatl, cn_zoisoglo, cn_zoisopac, cn_zoisoinp, cn_zoisoind, &
       &                cn_vozout, cn_vomevt, cn_vozous, cn_vomevs, cn_sozout, cn_somevt, cn_sozous, cn_somevs,    &
       &                cn_sozoutrp, cn_somevtrp, cn_soicecov, cn_vosigma0, cn_vosigmai, cn_vosigntr, cn_vodepiso, &
       &                cn_isothick, cn_iicethic, cn_ileadfra, cn_iicevelu, cn_iicevelv, cn_invcfc, cn_cfc11, cn_pendep
  NAMELIST /namsqdvar/  nn_sqdvar, cn_sqdvar
  NAMELIST /namcubvar/  nn_cubvar, cn_cubvar
  NAMELIST /nammeshmask/ cn_fzgr, cn_fe3t, cn_fe3u, cn_fe3v, cn_fe3w, cn_fhgr, cn_fmsk, cn_fcoo, cn_fbasins
  NAMELIST /nammask/    cn_tmask, cn_umask, cn_vmask, cn_fmask, cn_polymask, cn_tmaskutil, cn_tmaskatl, cn_tmaskpac, &
       &                cn_tmaskind, cn_tmaskant, cn_tmaskmed
  NAMELIST /nambathy/   cn_fbathymet, cn_fbathylev, cn_bathymet, cn_bathylev, cn_mbathy

  PRIVATE :: env, override

CONTAINS

  FUNCTION env(cdname) RESULT(cdval)
    CHARACTER(LEN=*), INTENT(in) :: cdname
    CHARACTER(LEN=lc) :: cdval
    INTEGER :: istat
    cdval = ' '
    CALL GET_ENVIRONMENT_VARIABLE(cdname, cdval, STATUS=istat)
    IF ( istat /= 0 ) cdval = ' '
  END FUNCTION env

  SUBROUTINE override(cdname, cdtarget)
    ! a non-empty environment variable replaces a file name
    CHARACTER(LEN=*), INTENT(in)    :: cdname
    CHARACTER(LEN=*), INTENT(inout) :: cdtarget
    CHARACTER(LEN=lc) :: cl
    cl = env(cdname)
    IF ( LEN_TRIM(cl) > 0 ) cdtarget = cl
  END SUBROUTINE override

  SUBROUTINE ReadCdfNames()
    !! Same lookup as the reference (src/modcdfnames.F90:283-322): $NAM_CDF_NAMES or 'nam_cdf_names' in
    !! the working directory, then in $HOME/CDFTOOLS_cfg/; the seven groups are read in the reference's
    !! order; no namelist file is not an error.  CDFT_* overrides are applied last in every case.
    CHARACTER(LEN=lc) :: cl_nam, cl_try
    LOGICAL :: ll_found
    INTEGER :: inum
    cl_nam = env('NAM_CDF_NAMES')
    IF ( LEN_TRIM(cl_nam) == 0 ) cl_nam = 'nam_cdf_names'
    INQUIRE(FILE=cl_nam, EXIST=ll_found)
    IF ( .NOT. ll_found ) THEN
       cl_try = TRIM(env('HOME'))//'/CDFTOOLS_cfg/'//TRIM(cl_nam)
       INQUIRE(FILE=cl_try, EXIST=ll_found)
       cl_nam = cl_try
    ENDIF
    IF ( ll_found ) THEN
       PRINT *, ' CAUTION : dim names and variable names are now set according to '
       PRINT *, ' =======   the following namelist : ', TRIM(cl_nam)
       OPEN(NEWUNIT=inum, FILE=cl_nam, STATUS='OLD', ACTION='READ')
       READ(inum, namdim) ; READ(inum, namdimvar) ; READ(inum, nammetrics) ; READ(inum, namvars)
       READ(inum, nambathy) ; READ(inum, namsqdvar) ; READ(inum, nammeshmask)
       CLOSE(inum)
    ENDIF
    CALL chkenv
  END SUBROUTINE ReadCdfNames

  SUBROUTINE PrintCdfNames()
    !! writes the current names as a namelist file (src/modcdfnames.F90:328-360)
    INTEGER :: inum
    CHARACTER(LEN=8) :: cl_d ; CHARACTER(LEN=10) :: cl_t
    CALL DATE_AND_TIME(DATE=cl_d, TIME=cl_t)
    OPEN(NEWUNIT=inum, FILE='PrintCdfNames.namlist', STATUS='REPLACE', ACTION='WRITE')
    WRITE(inum, '(4a)') ' ! ', cl_d, ' ', cl_t
    WRITE(inum, '(a)') ' ! Namelist automatically generated by PrintCdfNames '
    WRITE(inum, '(a)') ' ! Do not edit without changing its name ... '
    WRITE(inum, namdim) ; WRITE(inum, namdimvar) ; WRITE(inum, nammetrics) ; WRITE(inum, namvars)
    WRITE(inum, nambathy) ; WRITE(inum, namsqdvar) ; WRITE(inum, nammeshmask)
    CLOSE(inum)
  END SUBROUTINE PrintCdfNames

  SUBROUTINE chkenv()
    !! CDFT_MESH_HGR / CDFT_MESH_ZGR / CDFT_MASK / CDFT_BASINS / CDFT_COORD replace the mesh-mask file
    !! names; the e3 files follow cn_fzgr (src/modcdfnames.F90:376-385)
    CALL override('CDFT_MESH_HGR', cn_fhgr)
    CALL override('CDFT_MESH_ZGR', cn_fzgr)
    CALL override('CDFT_MASK',     cn_fmsk)
    CALL override('CDFT_BASINS',   cn_fbasins)
    CALL override('CDFT_COORD',    cn_fcoo)
    cn_fe3t = cn_fzgr ; cn_fe3u = cn_fzgr ; cn_fe3v = cn_fzgr ; cn_fe3w = cn_fzgr
  END SUBROUTINE chkenv

END MODULE modcdfnames
