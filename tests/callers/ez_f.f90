! A Fortran caller of the same library through the reference's Fortran entry points (hidden string lengths, every argument by reference,
! LOGICAL = 4 bytes: src/PUBLIC_INCLUDES/rmn/rpnmacros.h:21,32-49, src/interp/ezqkdef.c:39-55, src/interpv/Interp1D_Linear.F90:22-99):
!     ezqkdef x 2, ezsetopt, ezdefset, ezsint, ezuvint, then Interp1D_FindPos + Interp1D_Linear on columns built from the interpolated field.
! usage: ez_f in.bin out.bin   in.bin = int32 ni nj no mo, real z(ni*nj) u(ni*nj) v(ni*nj)
! out.bin = real zout(no*mo) uout(no*mo) vout(no*mo), real prof(ncol, nd)
program ez_f
  implicit none
  integer, external :: ezqkdef, ezsetopt, ezdefset, ezsint, ezuvint
  external :: interp1d_findpos, interp1d_linear
  integer :: ni, nj, no, mo, gdin, gdout, ier, i, k, ncol
  integer, parameter :: ns = 6, nd = 4
  real, allocatable :: z(:), u(:), v(:), zo(:), uo(:), vo(:)
  real, allocatable :: lev_s(:,:), st_s(:,:), der_s(:,:), lev_d(:,:), st_d(:,:), der_d(:,:)
  integer, allocatable :: posn(:,:)
  logical :: exdown, exup
  real :: gdown, gup
  character(len=256) :: fin, fout
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(10, file=trim(fin), access='stream', form='unformatted', status='old')
  read(10) ni, nj, no, mo
  allocate(z(ni*nj), u(ni*nj), v(ni*nj), zo(no*mo), uo(no*mo), vo(no*mo))
  read(10) z, u, v
  close(10)
  gdin = ezqkdef(ni, nj, 'G', 0, 0, 0, 0, 0)
  gdout = ezqkdef(no, mo, 'L', 200, 200, 0, 0, 0)
  if (gdin < 0 .or. gdout < 0) stop 1
  ier = ezsetopt('INTERP_DEGREE', 'CUBIC')
  if (ier /= 0) stop 1
  ier = ezdefset(gdout, gdin)
  if (ier < 0) stop 1
  ier = ezsint(zo, z)
  if (ier < 0) stop 1
  ier = ezuvint(uo, vo, u, v)
  if (ier < 0) stop 1
  ! vertical step on the interpolated field: ncol columns, ns source levels (ascending), nd destination levels; dimensioned wider than used
  ncol = min(no, 64)
  allocate(lev_s(ncol+3, ns), st_s(ncol+3, ns), der_s(ncol+3, ns), lev_d(ncol+2, nd), st_d(ncol+2, nd), der_d(ncol+2, nd), posn(ncol+2, nd))
  lev_s = 0.; st_s = 0.; der_s = 0.; lev_d = 0.; st_d = -1.; der_d = 0.; posn = 0
  do k = 1, ns
    do i = 1, ncol
      lev_s(i, k) = 100. * k + 0.25 * i
      st_s(i, k) = zo(i + (k - 1) * no)
    end do
  end do
  do k = 1, nd
    do i = 1, ncol
      lev_d(i, k) = 130. + 110. * k + 0.125 * i
    end do
  end do
  exdown = .false.; exup = .true.; gdown = 0.; gup = 0.
  call interp1d_findpos(ncol, ns, nd, ncol+3, ncol+2, lev_s, posn, lev_d)
  call interp1d_linear(ncol, ns, nd, ncol+3, ncol+2, lev_s, st_s, der_s, posn, lev_d, st_d, der_d, exdown, exup, gdown, gup)
  open(11, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(11) zo, uo, vo
  write(11) st_d(1:ncol, 1:nd)
  close(11)
  print '(a,i0,a,i0,a,f12.6,a,f12.6)', 'ez_f: gdin ', gdin, ' gdout ', gdout, ' z(1) ', zo(1), ' prof(1,1) ', st_d(1, 1)
end program ez_f
