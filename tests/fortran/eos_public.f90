PROGRAM eos_public
  ! Test driver (tests/test_eos_more.py, GPU): calls every member of the eos module's generics the way a reference
  ! program does (src/eos.f90:24-56) on the arrays of <dir>/in.bin and writes the results, in call order, as r8 to
  ! <dir>/out.bin.  Argument 2: 0 EOS-80, 1 TEOS-10.
  USE eos
  IMPLICIT NONE
  CHARACTER(LEN=512) :: cdir, carg
  INTEGER(KIND=4) :: npi, npj, iteos, iu, io, jp
  REAL(KIND=4), ALLOCATABLE :: zt(:,:), zs(:,:), zd(:,:), zt2(:,:), zs2(:,:), ze(:,:), zt3(:,:,:), zs3(:,:,:)
  REAL(KIND=4) :: zdep
  REAL(KIND=8) :: dpt(3,4)

  CALL getarg(1, cdir) ; CALL getarg(2, carg) ; READ(carg, *) iteos
  OPEN(NEWUNIT=iu, FILE=TRIM(cdir)//'/in.bin', ACCESS='stream', FORM='unformatted', STATUS='old')
  READ(iu) npi, npj
  ALLOCATE( zt(npi,npj), zs(npi,npj), zd(npi,npj), zt2(npi,npj), zs2(npi,npj), ze(npi,npj), zt3(npi,npj,2), zs3(npi,npj,2) )
  READ(iu) zt, zs, zd, zt2, zs2, ze, dpt
  CLOSE(iu)
  OPEN(NEWUNIT=io, FILE=TRIM(cdir)//'/out.bin', ACCESS='stream', FORM='unformatted', STATUS='replace')
  CALL eos_init( iteos == 1 )
  WRITE(io) sigmantr(zt, zs, npi, npj)
  WRITE(io) spice(zt, zs, npi, npj, .TRUE.)                       ! spice_pol
  WRITE(io) spice(zt, zs, npi, npj)                               ! spice_80
  zdep = 1234.5
  WRITE(io) alpha(zt, zs, zdep, npi, npj)                         ! alpha_2d
  WRITE(io) beta (zt, zs, zdep, npi, npj)                         ! beta_2d
  WRITE(io) albet(zt, zs, zdep, npi, npj)                         ! albet_2d
  WRITE(io) beta (zt, zs, zdep, npi, npj, .TRUE.)                 ! beta_2d_80
  WRITE(io) alpha(zt(:,1), zs(:,1), zd(:,1), npi)                 ! alpha_z
  WRITE(io) beta (zt(:,1), zs(:,1), zd(:,1), npi)                 ! beta_z
  WRITE(io) albet(zt(:,1), zs(:,1), zd(:,1), npi)                 ! albet_z
  WRITE(io) beta (zt(:,1), zs(:,1), zd(:,1), npi, .TRUE.)         ! beta_z_80
  DO jp = 1, 4
     WRITE(io) alpha(dpt(1,jp), dpt(2,jp), dpt(3,jp)), beta(dpt(1,jp), dpt(2,jp), dpt(3,jp))   ! alpha_0d, beta_0d
  END DO
  zdep = 777.
  zt3(:,:,1) = zt ; zt3(:,:,2) = zt2 ; zs3(:,:,1) = zs ; zs3(:,:,2) = zs2
  WRITE(io) REAL(eosbn2(zt3, zs3, zdep, ze, npi, npj, 1, 2), 8)           ! eosbn2_pol, upper level in plane 1
  WRITE(io) REAL(eosbn2(zt3, zs3, zdep, ze, npi, npj, 1, 2, .TRUE.), 8)   ! eosbn2_80
  zt3(:,:,2) = zt ; zt3(:,:,1) = zt2 ; zs3(:,:,2) = zs ; zs3(:,:,1) = zs2
  WRITE(io) REAL(eosbn2(zt3, zs3, zdep, ze, npi, npj, 2, 1), 8)           ! the planes swapped, as cdfbn2 does from level to level
  CLOSE(io)
  PRINT '(a)', 'done'
END PROGRAM eos_public
