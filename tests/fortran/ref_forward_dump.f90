!=======================================================================
! The reference's OWN forward module (src/forward.f90, compiled unmodified) as the consumer of the drop-in
! `module fftw` (rf_inv_amd/fortran/fftw.f90): calc_rf executes the module's c2r plan itself --
! `call dfftw_execute(ifft)` on cx -> rx, src/forward.f90:172,200 -- which the drop-in runs on the GPU
! (rf_fft_c2r: the transform's definition).  Everything else in the traces this program writes -- filter, E^-1,
! solid and liquid layer matrices, the propagator chain, land / ocean boundary conditions, P and S spectra,
! water-level deconvolution, direct arrival, shift maps, normalisation -- is the reference's own code and arithmetic.
!   usage: ref_forward_dump params.in stacks.txt out.bin [reps]
!   reps > 0: after the dump, `reps` timed passes of calc_rf over all stacks (one core): prints the seconds -- the
!   reference's own forward cost per evaluation on this host (its c2r round trip to the GPU included, a few per cent)
!   stacks.txt: n; then per stack: nlay, and nlay lines "alpha beta rho h"
!   out.bin (stream): int32 nfft, ntrc, nh, n; real64 flt(nh, ntrc); then per stack real64 rft(nfft, ntrc)
!   optional 5th argument extras.bin (stream), per stack: real64 tp(ntrc), int32 npre(ntrc) -- tp from the reference's
!   own public direct_arrival under calc_rf's branch rule (src/forward.f90:148-163; 0 with deconvolution; common rays
!   reuse trace 1's), npre by the expressions of :177 / :186 (calc_rf keeps both local: they are re-evaluated here)
! Built twice: (1) oracle/Makefile.ref -- with the reference's OWN src/fftw.f90 on MKL's FFTW3 interface, CPU only, no
! product object linked: the generator of tests/golden/ref/ (oracle/gen_golden.py); (2) oracle/Makefile.dropin -- on the
! drop-in module fftw as described above (the drop-in's own test, needs a GPU).
! Test infrastructure (tests/test_reference_fixtures.py, tests/test_reference_forward.py).
!=======================================================================
program ref_forward_dump
  use params
  use fftw
  use forward
  implicit none
  character(clen_max) :: param_file, stack_file, out_file
  integer :: n, i, j, nlay, u, v, reps, irep, x, itrc
  integer(4), allocatable :: npre(:)
  real(8), allocatable :: tp(:)
  character(clen_max) :: extra_file
  integer(8) :: c0, c1, crate
  character(32) :: arg
  real(8), allocatable :: alpha(:), beta(:), rho(:), h(:), rft(:,:)
  real(8), allocatable :: sa(:,:), sb(:,:), sr(:,:), sh(:,:)
  integer, allocatable :: snl(:)

  call get_command_argument(1, param_file)
  call get_command_argument(2, stack_file)
  call get_command_argument(3, out_file)
  call get_params(.false., param_file)
  call read_obs(.false.)
  call init_fftw()
  call init_forward(.false.)
  allocate(rft(nfft, ntrc), tp(ntrc), npre(ntrc))
  x = 0
  if (command_argument_count() > 4) then
     call get_command_argument(5, extra_file)
     x = 73
     open(x, file = trim(extra_file), status = "replace", access = "stream", form = "unformatted")
  end if
  u = 71
  v = 72
  open(u, file = trim(stack_file), status = "old")
  open(v, file = trim(out_file), status = "replace", access = "stream", form = "unformatted")
  read(u, *) n
  allocate(snl(n), sa(nlay_max, n), sb(nlay_max, n), sr(nlay_max, n), sh(nlay_max, n))
  write(v) int(nfft, 4), int(ntrc, 4), int(nfft / 2 + 1, 4), int(n, 4)
  write(v) flt
  do i = 1, n
     read(u, *) nlay
     allocate(alpha(nlay), beta(nlay), rho(nlay), h(nlay))
     do j = 1, nlay
        read(u, *) alpha(j), beta(j), rho(j), h(j)
     end do
     call calc_rf(1, nlay, nfft, ntrc, rayps, alpha, beta, rho, h, rft)
     write(v) rft
     if (x > 0) then
        do itrc = 1, ntrc
           if (itrc == 1 .or. .not. is_ray_common) then
              if (deconv_mode == 1) then
                 tp(itrc) = 0.d0
              else if (ipha(itrc) == 1) then
                 call direct_arrival(nlay, h(1:nlay), alpha(1:nlay), rayps(itrc), tp(itrc))
              else
                 call direct_arrival(nlay, h(1:nlay), beta(1:nlay), rayps(itrc), tp(itrc))
              end if
           else
              tp(itrc) = tp(1)
           end if
           if (ipha(itrc) == 1) then
              npre(itrc) = nint((-t_start - tp(itrc)) / delta)
           else
              npre(itrc) = nint((-t_start + tp(itrc)) / delta)
           end if
        end do
        write(x) tp
        write(x) npre
     end if
     snl(i) = nlay
     sa(1:nlay, i) = alpha;  sb(1:nlay, i) = beta;  sr(1:nlay, i) = rho;  sh(1:nlay, i) = h
     deallocate(alpha, beta, rho, h)
  end do
  close(u)
  close(v)
  if (x > 0) close(x)
  write(*,*) "ref_forward_dump: ok", n, is_ray_common
  reps = 0
  if (command_argument_count() > 3) then
     call get_command_argument(4, arg)
     read(arg, *) reps
  end if
  if (reps > 0) then
     call system_clock(c0, crate)
     do irep = 1, reps
        do i = 1, n
           nlay = snl(i)
           call calc_rf(1, nlay, nfft, ntrc, rayps, sa(1:nlay, i), sb(1:nlay, i), sr(1:nlay, i), sh(1:nlay, i), rft)
        end do
     end do
     call system_clock(c1)
     write(*,'(A,F12.6,A,I0)') " ref_forward_dump: seconds ", dble(c1 - c0) / dble(crate), " evaluations ", reps * n
  end if
end program ref_forward_dump
