!=======================================================================
! module likelihood -- drop-in replacement of RF_INV's src/likelihood.f90.
!
! Same public interface as the reference module (src/likelihood.f90:28-37):
!     real(8), allocatable, public :: sig(:,:), rft(:,:,:), log_likelihood(:)
!     subroutine init_likelihood(verb)
!     subroutine calc_likelihood(chain_id, fwd_flag, prop_k, prop_z, prop_dvp,
!                                prop_dvs, sig, prop_log_likelihood, prop_rft)
! The forward model + misfit run on the GPU through librfgpu (include/rfgpu.h).
! format_model stays the host's (module model), exactly where the reference
! calls it (src/likelihood.f90:75-76).  Written from scratch.
!
! Build with -DRFGPU_USE_LAPACK to build the noise-covariance pseudo-inverse
! with the host's own LAPACK dgesvd (as the reference does, bit-for-bit with
! that LAPACK); without it librfgpu's own SVD is used.
!=======================================================================
module likelihood
  use iso_c_binding
  use rfgpu_c
  implicit none
  real(8), allocatable, public :: sig(:,:)
  real(8), allocatable, public :: rft(:,:,:)
  real(8), allocatable, public :: log_likelihood(:)

  public init_likelihood, calc_likelihood
  private init_sig, init_rft
#ifdef RFGPU_USE_LAPACK
  private init_r_inv_lapack
#endif

contains

  !---------------------------------------------------------------------
  subroutine init_likelihood(verb)
    logical, intent(in) :: verb
    call init_sig(verb)
#ifdef RFGPU_USE_LAPACK
    call init_r_inv_lapack(verb)
#endif
    call init_rft()
  end subroutine init_likelihood

  !---------------------------------------------------------------------
  subroutine calc_likelihood(chain_id, fwd_flag, prop_k, prop_z, &
       & prop_dvp, prop_dvs, sig, prop_log_likelihood, prop_rft)
    use params, only: k_max, ntrc, nfft, nlay_max
    use forward, only: rf_ctx
    use model, only: format_model
    integer, intent(in) :: prop_k, chain_id
    logical, intent(in) :: fwd_flag
    real(8), intent(in) :: prop_z(k_max-1), prop_dvp(k_max)
    real(8), intent(in) :: prop_dvs(k_max), sig(ntrc)
    real(8), intent(out) :: prop_log_likelihood
    real(8), intent(out) :: prop_rft(nfft, ntrc)
    integer :: nlay
    real(8) :: alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max)
    logical :: is_valid

    if (fwd_flag) then
       call format_model(prop_k, prop_z, prop_dvp, prop_dvs, &
            & nlay, alpha, beta, rho, h, is_valid)
       call rfgpu_check(rf_calc_likelihood(rf_ctx, int(chain_id - 1, c_int32_t), 1_c_int32_t, &
            & int(nlay, c_int32_t), alpha, beta, rho, h, sig, prop_log_likelihood, prop_rft), &
            & "rf_calc_likelihood")
    else
       ! sigma-only proposal: the host owns the stored trace of the chain
       prop_rft(1:nfft, 1:ntrc) = rft(1:nfft, 1:ntrc, chain_id)
       call rfgpu_check(rf_calc_likelihood_of_trace(rf_ctx, prop_rft, sig, prop_log_likelihood), &
            & "rf_calc_likelihood_of_trace")
    end if
  end subroutine calc_likelihood

  !---------------------------------------------------------------------
  ! noise sigma of every chain: fixed, or uniform in [sig_min, sig_max]
  subroutine init_sig(verb)
    use params, only: sig_min, sig_max, nchains, ntrc, sig_mode
    use mt19937, only: grnd
    logical, intent(in) :: verb
    integer :: ichain, itrc

    allocate(sig(ntrc, nchains))
    do ichain = 1, nchains
       do itrc = 1, ntrc
          sig(itrc, ichain) = sig_min(itrc)
          if (sig_mode(itrc) == 1) then
             sig(itrc, ichain) = sig_min(itrc) + grnd() * (sig_max(itrc) - sig_min(itrc))
          end if
       end do
    end do
    if (verb) then
       write(*,*)
       write(*,*) "--- Initialize noise sigma ---"
       do ichain = 1, nchains
          write(*,*) ichain, sig(1, ichain)
       end do
    end if
  end subroutine init_sig

  !---------------------------------------------------------------------
  ! first evaluation of every chain
  subroutine init_rft()
    use params, only: nfft, ntrc, nchains
    use model, only: k, z, dvp, dvs
    integer :: ichain

    allocate(rft(nfft, ntrc, nchains), log_likelihood(nchains))
    do ichain = 1, nchains
       call calc_likelihood(ichain, .true., k(ichain), z(:, ichain), dvp(:, ichain), &
            & dvs(:, ichain), sig(:, ichain), log_likelihood(ichain), rft(:, :, ichain))
    end do
  end subroutine init_rft

#ifdef RFGPU_USE_LAPACK
  !---------------------------------------------------------------------
  ! Gaussian-correlated noise matrix -> SVD -> pseudo-inverse (s > 1e-3),
  ! through the host's LAPACK, then handed to the engine.
  subroutine init_r_inv_lapack(verb)
    use params, only: nsmp, ntrc, a_gus, delta
    use forward, only: rf_ctx
    logical, intent(in) :: verb
    real(8), allocatable :: rm(:,:), s(:), u(:,:), vt(:,:), work(:), pinv(:,:,:), vd(:,:)
    real(8) :: r, wq(1)
    integer :: itrc, i, j, info, lwork

    allocate(rm(nsmp, nsmp), s(nsmp), u(nsmp, nsmp), vt(nsmp, nsmp), vd(nsmp, nsmp))
    allocate(pinv(nsmp, nsmp, ntrc))
    do itrc = 1, ntrc
       r = exp(-a_gus(itrc)**2 * delta**2)
       do i = 1, nsmp
          do j = 1, nsmp
             rm(j, i) = r ** ((i - j) ** 2)
          end do
       end do
       call dgesvd('A', 'A', nsmp, nsmp, rm, nsmp, s, u, nsmp, vt, nsmp, wq, -1, info)
       lwork = nint(wq(1))
       allocate(work(lwork))
       call dgesvd('A', 'A', nsmp, nsmp, rm, nsmp, s, u, nsmp, vt, nsmp, work, lwork, info)
       deallocate(work)
       if (info /= 0) call rfgpu_check(int(info, c_int), "dgesvd")
       do i = 1, nsmp
          if (s(i) > 1.0d-3) then
             vd(:, i) = vt(i, :) / s(i)
          else
             vd(:, i) = 0.d0
          end if
       end do
       pinv(:, :, itrc) = matmul(vd, transpose(u))
    end do
    call rfgpu_check(rf_set_r_inv(rf_ctx, pinv), "rf_set_r_inv")
    if (verb) write(*,*) "R inverse built with host LAPACK"
  end subroutine init_r_inv_lapack
#endif

end module likelihood
