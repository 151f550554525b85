!=======================================================================
! Test driver for the bind(C) shim modules (rf_inv_amd/fortran): follows the
! reference main program's init order (src/rf_inv.f90:69-88) with the
! reference's own params / mt19937 / model modules and OUR forward /
! likelihood modules, then dumps every chain's model, sigma, logL and trace so
! that tests/test_fortran_shim.py can check them against the CPU oracle.
! Also exercises the fwd_flag = .false. branch and calc_rf.
!=======================================================================
program drive_shim
  use params
  use mt19937
  use model
  use forward
  use likelihood
  implicit none
  character(clen_max) :: param_file
  integer :: ichain, i, itrc, nlay, u
  real(8) :: alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max)
  real(8), allocatable :: prop_rft(:,:), sig2(:), rf2(:,:)
  real(8) :: ll2
  logical :: is_valid

  param_file = "params.in"
  if (command_argument_count() > 0) call get_command_argument(1, param_file)
  call get_params(.false., param_file)
  call read_obs(.false.)
  call sgrnd(iseed)
  call init_forward(.false.)
  call read_ref_model(.false.)
  call init_model(.false.)
  call init_likelihood(.false.)

  allocate(prop_rft(nfft, ntrc), sig2(ntrc), rf2(nfft, ntrc))
  u = 77
  open(u, file = "shim_dump.txt", status = "unknown")
  write(u, *) nchains, ntrc, nfft, nsmp, k_max
  write(u, '(es25.17)') delta
  write(u, *) merge(1, 0, is_ray_common)
  do ichain = 1, nchains
     call format_model(k(ichain), z(:, ichain), dvp(:, ichain), dvs(:, ichain), &
          & nlay, alpha, beta, rho, h, is_valid)
     write(u, *) nlay
     do i = 1, nlay
        write(u, '(4es25.17)') alpha(i), beta(i), rho(i), h(i)
     end do
     write(u, '(es25.17)') (sig(itrc, ichain), itrc = 1, ntrc)
     write(u, '(es25.17)') log_likelihood(ichain)
     do itrc = 1, ntrc
        write(u, '(es25.17)') (rft(i, itrc, ichain), i = 1, nfft)
     end do
     ! sigma-only branch (src/likelihood.f90:81) on the stored trace
     sig2 = 2.d0 * sig(:, ichain)
     call calc_likelihood(ichain, .false., k(ichain), z(:, ichain), dvp(:, ichain), &
          & dvs(:, ichain), sig2, ll2, prop_rft)
     write(u, '(es25.17)') ll2
     if (any(prop_rft /= rft(:, :, ichain))) then
        write(u, *) 0
     else
        write(u, *) 1
     end if
     ! plain calc_rf on the same stack
     call calc_rf(ichain, nlay, nfft, ntrc, rayps, alpha, beta, rho, h, rf2)
     if (any(rf2 /= rft(:, :, ichain))) then
        write(u, *) 0
     else
        write(u, *) 1
     end if
  end do
  write(u, '(es25.17)') (flt(i, 1), i = 1, nfft / 2 + 1)
  close(u)
  write(*,*) "drive_shim: ok"
end program drive_shim
