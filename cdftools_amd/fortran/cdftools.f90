MODULE cdftools
  !!======================================================================
  !!                     ***  MODULE  cdftools  ***
  !! The one routine of the reference's module of the same name that the path needs: cdf_findij, the model
  !! (i,j) indices of a geographical window (src/cdftools.f90:41-264), used by cdfmoc -rapid to locate the
  !! 26.5N section.  Host code: a search over the 2-D coordinate arrays, nothing to offload.
  !!
  !! Behaviour kept from the reference:
  !!   - the coordinates are read once and kept between calls; only that first call brings the window's
  !!     longitudes into the grid's 0-360 convention (the great-circle distance does not care);
  !!   - the search walks 21 x 21 blocks from the middle of the domain towards the smallest distance
  !!     (in r4, strict decrease moves the point) and gives up on the domain's inner border;
  !!   - a candidate further than one cell diagonal, MAX(e1,e2)*SQRT(2), is refused and the search restarts from
  !!     up to 15 other points; after that the indices are -1000.
  !!======================================================================
  USE cdfio
  USE modcdfnames
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: cdf_findij, dist

  INTEGER(KIND=4), PARAMETER :: jp_halfblock = 10, jp_restarts = 15
  INTEGER(KIND=4), SAVE :: nxg = 0, nyg = 0
  REAL(KIND=8), SAVE, ALLOCATABLE :: dlam(:,:), dphi(:,:), dsc1(:,:), dsc2(:,:)

CONTAINS

  SUBROUTINE cdf_findij(pxmin, pxmax, pymin, pymax, kimin, kimax, kjmin, kjmax, cd_coord, cd_point, cd_verbose, &
       &                plonmin, plonmax, platmin, platmax)
    REAL(KIND=4),           INTENT(in)  :: pxmin, pxmax, pymin, pymax
    INTEGER(KIND=4),        INTENT(out) :: kimin, kimax, kjmin, kjmax
    CHARACTER(*), OPTIONAL, INTENT(in)  :: cd_coord, cd_point, cd_verbose
    REAL(KIND=4), OPTIONAL, INTENT(out) :: plonmin, plonmax, platmin, platmax

    INTEGER(KIND=4) :: ii, ij, imin, imax, jmin, jmax
    REAL(KIND=8)    :: dxmin, dxmax, dymin, dymax, dlam0
    REAL(KIND=4)    :: zlonmin, zlonmax, zlatmin, zlatmax
    CHARACTER(LEN=256) :: clcoo, cltype
    LOGICAL :: lbnd, lverbose

    CALL ReadCdfNames()
    imin = 0 ; jmin = 0 ; imax = 0 ; jmax = 0
    dxmin = pxmin ; dxmax = pxmax ; dymin = pymin ; dymax = pymax
    clcoo = cn_fhgr ; cltype = 'F' ; lverbose = .FALSE.
    IF ( PRESENT(cd_coord) ) clcoo  = cd_coord
    IF ( PRESENT(cd_point) ) cltype = cd_point
    IF ( PRESENT(cd_verbose) ) lverbose = ( cd_verbose(1:1) == 'Y' .OR. cd_verbose(1:1) == 'y' )

    IF ( .NOT. ALLOCATED(dlam) ) THEN
       IF ( chkfile(clcoo) ) STOP 96
       nxg = getdim(clcoo, cn_x)
       nyg = getdim(clcoo, cn_y)
       ALLOCATE( dlam(nxg,nyg), dphi(nxg,nyg), dsc1(nxg,nyg), dsc2(nxg,nyg) )
       SELECT CASE ( cltype(1:1) )
       CASE ( 'T', 't' ) ; CALL load(cn_glamt, cn_gphit, cn_ve1t, cn_ve2t)
       CASE ( 'U', 'u' ) ; CALL load(cn_glamu, cn_gphiu, cn_ve1u, cn_ve2u)
       CASE ( 'V', 'v' ) ; CALL load(cn_glamv, cn_gphiv, cn_ve1v, cn_ve2v)
       CASE ( 'F', 'f' ) ; CALL load(cn_glamf, cn_gphif, cn_ve1f, cn_ve2f)
       CASE DEFAULT
          PRINT *, ' ERROR : type of point not known: ', TRIM(cltype)
       END SELECT
       ! longitudes counted eastward from the grid's first column (ORCA grids start near 70 E)
       WHERE ( dlam < 0 ) dlam = dlam + 360.d0
       dlam0 = dlam(1, nyg/2)
       WHERE ( dlam < dlam0 ) dlam = dlam + 360.d0
       IF ( dxmin < 0.   ) dxmin = dxmin + 360.d0
       IF ( dxmax < 0.   ) dxmax = dxmax + 360.d0
       IF ( dxmin < dlam0 ) dxmin = dxmin + 360.d0
       IF ( dxmax < dlam0 ) dxmax = dxmax + 360.d0
    ENDIF

    CALL locate(dxmin, dymin, ii, ij, lbnd)
    IF ( lbnd ) THEN
       IF ( lverbose ) WRITE(*,*) 'Point  Out of domain or on boundary'
    ELSE
       imin = ii ; jmin = ij
    ENDIF
    IF ( pxmin == pxmax .AND. pymin == pymax ) THEN
       imax = imin ; jmax = jmin
    ELSE
       CALL locate(dxmax, dymax, ii, ij, lbnd)
       IF ( lbnd ) THEN
          IF ( lverbose ) WRITE(*,*) 'Point  Out of domain or on boundary'
       ELSE
          imax = ii ; jmax = ij
       ENDIF
    ENDIF
    IF ( lverbose ) PRINT '(4i10)', imin, imax, jmin, jmax

    kimin = imin ; kimax = imax ; kjmin = jmin ; kjmax = jmax
    IF ( lbnd .OR. imin < 0 ) THEN
       zlonmin = -9999. ; zlonmax = -9999. ; zlatmin = -9999. ; zlatmax = -9999.
    ELSE
       zlonmin = dlam(imin,jmin) ; zlonmax = dlam(imax,jmax)
       zlatmin = dphi(imin,jmin) ; zlatmax = dphi(imax,jmax)
    ENDIF
    IF ( zlonmin > 180 ) zlonmin = zlonmin - 360.
    IF ( zlonmax > 180 ) zlonmax = zlonmax - 360.
    IF ( lverbose ) PRINT '(4f10.4)', zlonmin, zlonmax, zlatmin, zlatmax
    IF ( PRESENT(plonmin) ) plonmin = zlonmin
    IF ( PRESENT(plonmax) ) plonmax = zlonmax
    IF ( PRESENT(platmin) ) platmin = zlatmin
    IF ( PRESENT(platmax) ) platmax = zlatmax

  CONTAINS

    SUBROUTINE load(cd_lam, cd_phi, cd_e1, cd_e2)
      CHARACTER(*), INTENT(in) :: cd_lam, cd_phi, cd_e1, cd_e2
      dlam(:,:) = getvar(clcoo, cd_lam, 1, nxg, nyg)
      dphi(:,:) = getvar(clcoo, cd_phi, 1, nxg, nyg)
      dsc1(:,:) = getvar(clcoo, cd_e1,  1, nxg, nyg)
      dsc2(:,:) = getvar(clcoo, cd_e2,  1, nxg, nyg)
    END SUBROUTINE load

    SUBROUTINE locate(ddx, ddy, ki, kj, ld_bnd)
      !! nearest grid point of (ddx,ddy), accepted when closer than a cell diagonal; other start points otherwise
      REAL(KIND=8),    INTENT(in)  :: ddx, ddy
      INTEGER(KIND=4), INTENT(out) :: ki, kj
      LOGICAL,         INTENT(out) :: ld_bnd
      INTEGER(KIND=4) :: itry
      REAL(KIND=8)    :: dldis, dldiag
      itry = 1
      ki = nxg/2 ; kj = nyg/2
      DO
         CALL walk(ddx, ddy, ki, kj, ld_bnd)
         dldis  = dist(ddx, dlam(ki,kj), ddy, dphi(ki,kj))                      ! km
         dldiag = MAX(dsc1(ki,kj), dsc2(ki,kj))/1000.*SQRT(2.)                    ! km
         IF ( dldis <= dldiag ) THEN
            IF ( lverbose ) PRINT '("#  dl_dis= ",f8.3," km")', dldis
            EXIT
         ENDIF
         IF ( itry >= jp_restarts ) THEN
            ki = -1000 ; kj = -1000
            EXIT
         ENDIF
         itry = itry + 1
         kj = (itry - 1)*nyg/itry
         ki = (itry - 1)*nxg/jp_restarts
      END DO
    END SUBROUTINE locate

  END SUBROUTINE cdf_findij

  SUBROUTINE walk(ddlon, ddlat, ki, kj, ld_bnd)
    !! src/cdftools.f90:420-495: block walk towards the smallest r4 distance; ld_bnd when it ends on the inner border
    REAL(KIND=8),    INTENT(in)    :: ddlon, ddlat
    INTEGER(KIND=4), INTENT(inout) :: ki, kj
    LOGICAL,         INTENT(out)   :: ld_bnd
    INTEGER(KIND=4) :: ji, jj, i0, i1, j0, j1
    REAL(KIND=4)    :: zd, zbest, zprev
    LOGICAL :: ledge
    zbest = 1000000. ; zprev = 1000000.
    ledge = .TRUE. ; ld_bnd = .FALSE.
    DO WHILE ( ledge .AND. .NOT. ld_bnd )
       i0 = ki - jp_halfblock ; i1 = ki + jp_halfblock
       j0 = kj - jp_halfblock ; j1 = kj + jp_halfblock
       IF ( i0 <= 0  ) i0 = 2
       IF ( i1 > nxg ) i1 = nxg - 1
       IF ( j0 <= 0  ) j0 = 2
       IF ( j1 > nyg ) j1 = nyg - 1
       DO jj = j0, j1
          DO ji = i0, i1
             zd    = dist(ddlon, dlam(ji,jj), ddlat, dphi(ji,jj))
             zbest = MIN(zbest, zd)
             IF ( zbest /= zprev ) THEN
                ki = ji ; kj = jj
             ENDIF
             zprev = zbest
          END DO
       END DO
       ledge = ( ki == i0 .OR. ki == i1 .OR. kj == j0 .OR. kj == j1 )               ! on the block's edge: next block
       ld_bnd = ( ki == 2 .OR. ki == nxg - 1 .OR. kj == 2 .OR. kj == nyg - 1 )      ! on the domain's inner border: lost
    END DO
  END SUBROUTINE walk

  REAL(KIND=8) FUNCTION dist(ddlona, ddlonb, ddlata, ddlatb)
    !! great-circle distance in km between A and B (degrees), src/cdftools.f90:498-556
    REAL(KIND=8), INTENT(in) :: ddlona, ddlonb, ddlata, ddlatb
    REAL(KIND=8) :: dlpi, dlconv, dlr, dlux, dluy, dluz, dlvx, dlvy, dlvz, dlpds
    dlpi   = ACOS(-1.d0)
    dlconv = dlpi/180.d0
    dlr    = (6378.137d0 + 6356.7523d0)/2.0d0
    dlux = COS(ddlona*dlconv)*COS(ddlata*dlconv)
    dluy = SIN(ddlona*dlconv)*COS(ddlata*dlconv)
    dluz = SIN(ddlata*dlconv)
    dlvx = COS(ddlonb*dlconv)*COS(ddlatb*dlconv)
    dlvy = SIN(ddlonb*dlconv)*COS(ddlatb*dlconv)
    dlvz = SIN(ddlatb*dlconv)
    dlpds = dlux*dlvx + dluy*dlvy + dluz*dlvz
    IF ( dlpds >= 1. ) THEN
       dist = 0.
    ELSE
       dist = dlr*ACOS(dlpds)
    ENDIF
  END FUNCTION dist

END MODULE cdftools
