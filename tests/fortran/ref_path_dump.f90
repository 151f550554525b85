!=======================================================================
! The reference's OWN forward AND likelihood modules (src/forward.f90, src/likelihood.f90, compiled unmodified) on the
! drop-in `module fftw` (rf_inv_amd/fortran/fftw.f90: calc_rf's `call dfftw_execute(ifft)` runs the c2r on the GPU as
! the transform's definition) with LAPACK's dgesvd from the Intel MKL the image ships under /opt/conda/lib.
! calc_likelihood (src/likelihood.f90:56-101) is then the reference's own code end to end -- format_model, calc_rf,
! calc_seis, the misfit, matmul(misfits, r_inv), the log-likelihood -- except for the inverse transform.
! Follows the init order of src/rf_inv.f90:69-88.
!   usage: ref_path_dump params.in models.txt out.bin [reps]
!   reps > 0: after the dump, `reps` timed passes of calc_likelihood(fwd_flag = .true.) over all models on one core:
!   prints the seconds -- the reference's own cost per forward + likelihood evaluation on this host (its c2r round
!   trips to the GPU included)
!   models.txt: n; then per model: k; z(1:k_max-1); dvp(1:k_max); dvs(1:k_max); sig(1:ntrc)   (list-directed)
!   out.bin (stream): int32 nfft, ntrc, nsmp, n, m;  real64 r_inv(nsmp, nsmp, ntrc) as built below;
!                     per model: real64 logL, rft(nfft, ntrc)          [calc_likelihood, fwd_flag = .true.]
!                     per probe j = 1 .. m: real64 logL, trace(nfft, ntrc), sig(ntrc)   [fwd_flag = .false. on a trace
!                                           this program stored in likelihood's public rft(:, :, 1)]
!   optional 5th argument extras.bin (stream), per model: int32 nlay, valid; real64 alpha, beta, rho, h (nlay_max each)
!   from the reference's own public format_model (src/model.f90:175-290); real64 tp(ntrc), int32 npre(ntrc) -- tp from
!   its public direct_arrival under calc_rf's branch rule (src/forward.f90:148-163), npre by the expressions of
!   :177 / :186 (calc_rf keeps both local: they are re-evaluated here)
! module likelihood keeps r_inv private: the matrix written to out.bin is formed here by the steps of init_r_inv
! (src/likelihood.f90:183-222: R(j, i) = r ** ((i - j) ** 2), dgesvd, reciprocals of the singular values above 1.0d-3,
! transpose(vt) . diag . transpose(u)) with the same LAPACK and the same intrinsic matmul; the probes check that it is
! the module's (their log-likelihoods are reproduced from it to rounding, tests/test_reference_forward.py).
! Built twice: (1) oracle/Makefile.ref -- with the reference's OWN src/fftw.f90 on MKL's FFTW3 interface, CPU only, no
! product object linked: the generator of tests/golden/ref/ (oracle/gen_golden.py) and bench.py's cpu_baseline;
! (2) oracle/Makefile.dropin -- on the drop-in module fftw as described above (the drop-in's own test, needs a GPU).
! Test infrastructure.
!=======================================================================
program ref_path_dump
  use params
  use mt19937
  use fftw
  use model
  use forward
  use likelihood
  implicit none
  integer, parameter :: m = 6
  character(clen_max) :: param_file, model_file, out_file
  integer :: n, i, j, it, jt, u, v, pk, info, lw, reps, irep, x, nl
  integer(4), allocatable :: npre(:)
  real(8), allocatable :: tp(:), fa(:), fb(:), fr(:), fh(:)
  logical :: ok
  character(clen_max) :: extra_file
  integer(8) :: c0, c1, crate
  character(32) :: arg
  integer, allocatable :: sk(:)
  real(8), allocatable :: sz(:,:), sdvp(:,:), sdvs(:,:), ssig(:,:)
  real(8), allocatable :: pz(:), pdvp(:), pdvs(:), psig(:), prft(:,:)
  real(8), allocatable :: cov(:,:), sv(:), uu(:,:), vt(:,:), work(:), dg(:,:), pinv(:,:,:)
  real(8) :: ll, r, wq(1)

  call get_command_argument(1, param_file)
  call get_command_argument(2, model_file)
  call get_command_argument(3, out_file)
  call get_params(.false., param_file)
  call read_obs(.false.)
  call sgrnd(iseed)
  call init_fftw()
  call init_forward(.false.)
  call read_ref_model(.false.)
  call init_model(.false.)
  call init_likelihood(.false.)

  ! the pseudo-inverse as init_r_inv forms it (src/likelihood.f90:183-222)
  allocate(cov(nsmp, nsmp), sv(nsmp), uu(nsmp, nsmp), vt(nsmp, nsmp), dg(nsmp, nsmp), pinv(nsmp, nsmp, ntrc))
  do jt = 1, ntrc
     r = exp(-a_gus(jt)**2 * delta**2)
     do i = 1, nsmp
        do j = 1, nsmp
           cov(j, i) = r ** ((i - j) ** 2)
        end do
     end do
     call dgesvd('A', 'A', nsmp, nsmp, cov, nsmp, sv, uu, nsmp, vt, nsmp, wq, -1, info)
     lw = nint(wq(1))
     if (.not. allocated(work)) allocate(work(lw))
     call dgesvd('A', 'A', nsmp, nsmp, cov, nsmp, sv, uu, nsmp, vt, nsmp, work, lw, info)
     if (info /= 0) stop 3
     dg = 0.d0
     do i = 1, nsmp
        if (sv(i) > 1.0d-3) dg(i, i) = 1.d0 / sv(i)
     end do
     pinv(:, :, jt) = matmul(matmul(transpose(vt), dg), transpose(uu))
  end do

  allocate(pz(k_max - 1), pdvp(k_max), pdvs(k_max), psig(ntrc), prft(nfft, ntrc))
  allocate(tp(ntrc), npre(ntrc), fa(nlay_max), fb(nlay_max), fr(nlay_max), fh(nlay_max))
  x = 0
  if (command_argument_count() > 4) then
     call get_command_argument(5, extra_file)
     x = 73
     open(x, file = trim(extra_file), status = "replace", access = "stream", form = "unformatted")
  end if
  u = 71
  v = 72
  open(u, file = trim(model_file), status = "old")
  open(v, file = trim(out_file), status = "replace", access = "stream", form = "unformatted")
  read(u, *) n
  allocate(sk(n), sz(k_max - 1, n), sdvp(k_max, n), sdvs(k_max, n), ssig(ntrc, n))
  write(v) int(nfft, 4), int(ntrc, 4), int(nsmp, 4), int(n, 4), int(m, 4)
  write(v) pinv
  do i = 1, n
     read(u, *) pk
     read(u, *) pz
     read(u, *) pdvp
     read(u, *) pdvs
     read(u, *) psig
     call calc_likelihood(1, .true., pk, pz, pdvp, pdvs, psig, ll, prft)
     write(v) ll
     write(v) prft
     if (x > 0) then
        call format_model(pk, pz, pdvp, pdvs, nl, fa, fb, fr, fh, ok)
        do jt = 1, ntrc
           if (jt == 1 .or. .not. is_ray_common) then
              if (deconv_mode == 1) then
                 tp(jt) = 0.d0
              else if (ipha(jt) == 1) then
                 call direct_arrival(nl, fh(1:nl), fa(1:nl), rayps(jt), tp(jt))
              else
                 call direct_arrival(nl, fh(1:nl), fb(1:nl), rayps(jt), tp(jt))
              end if
           else
              tp(jt) = tp(1)
           end if
           if (ipha(jt) == 1) then
              npre(jt) = nint((-t_start - tp(jt)) / delta)
           else
              npre(jt) = nint((-t_start + tp(jt)) / delta)
           end if
        end do
        write(x) int(nl, 4), int(merge(1, 0, ok), 4)
        write(x) fa, fb, fr, fh
        write(x) tp
        write(x) npre
     end if
     sk(i) = pk;  sz(:, i) = pz;  sdvp(:, i) = pdvp;  sdvs(:, i) = pdvs;  ssig(:, i) = psig
  end do
  close(u)
  reps = 0
  if (command_argument_count() > 3) then
     call get_command_argument(4, arg)
     read(arg, *) reps
  end if
  if (reps > 0) then
     call system_clock(c0, crate)
     do irep = 1, reps
        do i = 1, n
           call calc_likelihood(1, .true., sk(i), sz(:, i), sdvp(:, i), sdvs(:, i), ssig(:, i), ll, prft)
        end do
     end do
     call system_clock(c1)
     write(*,'(A,F12.6,A,I0)') " ref_path_dump: seconds ", dble(c1 - c0) / dble(crate), " evaluations ", reps * n
  end if
  ! sigma-only branch on traces stored by the host (src/likelihood.f90:81): obs + a deterministic wiggle
  do j = 1, m
     do jt = 1, ntrc
        do it = 1, nfft
           rft(it, jt, 1) = 0.05d0 * sin(0.37d0 * dble(it) * dble(j) + dble(jt))
           if (it <= nsmp) rft(it, jt, 1) = rft(it, jt, 1) + obs(it, jt)
        end do
        psig(jt) = 0.01d0 * dble(j) + 0.002d0 * dble(jt)
     end do
     call calc_likelihood(1, .false., k(1), z(:, 1), dvp(:, 1), dvs(:, 1), psig, ll, prft)
     write(v) ll
     write(v) prft
     write(v) psig
  end do
  close(v)
  if (x > 0) close(x)
  write(*,*) "ref_path_dump: ok", n, m
end program ref_path_dump
