!=======================================================================
! module likelihood -- GPU-backed stand-in for RF_INV's likelihood module.
!
! Public names and argument lists are those the RF_INV host expects
! (sig, rft, log_likelihood, init_likelihood, calc_likelihood); the work is
! done by librfgpu (include/rfgpu.h) through module rfgpu_c.  The layer stack
! is still produced by the host's own format_model, at the point where the
! original module called it.  Written from scratch for rf_inv_amd.
!
! -DRFGPU_USE_LAPACK : build the noise-covariance pseudo-inverse with the
!                      host's LAPACK (dgesvd) and hand it to the engine;
!                      default is the engine's own SVD.
!=======================================================================
module likelihood
  use iso_c_binding, only: c_int, c_int32_t, c_double
  use rfgpu_c
  implicit none
  private

  real(8), allocatable, public :: sig(:,:)            ! (ntrc, nchains) noise level of each chain
  real(8), allocatable, public :: rft(:,:,:)          ! (nfft, ntrc, nchains) current trace of each chain
  real(8), allocatable, public :: log_likelihood(:)   ! (nchains)
  public :: init_likelihood, calc_likelihood

contains

  !---------------------------------------------------------------------
  ! State of every chain: noise levels, then one forward evaluation each.
  subroutine init_likelihood(verb)
    use params, only: nchains, ntrc, nfft, sig_mode, sig_min, sig_max
    use mt19937, only: grnd
    use model, only: k, z, dvp, dvs
    logical, intent(in) :: verb
    integer :: jc, jt

    allocate(sig(ntrc, nchains), rft(nfft, ntrc, nchains), log_likelihood(nchains))

    ! noise level: fixed at sig_min, or uniform in [sig_min, sig_max] when it is solved for
    ! (one uniform draw per solved trace, chain-major order)
    do jc = 1, nchains
       do jt = 1, ntrc
          if (sig_mode(jt) == 1) then
             sig(jt, jc) = sig_min(jt) + grnd() * (sig_max(jt) - sig_min(jt))
          else
             sig(jt, jc) = sig_min(jt)
          end if
       end do
    end do
    if (verb) then
       write(*,*)
       write(*,*) "--- Initialize noise sigma ---"
       write(*,'(i8,es16.6)') (jc, sig(1, jc), jc = 1, nchains)
    end if

#ifdef RFGPU_USE_LAPACK
    call pseudo_inverse_from_lapack(verb)
#endif

    do jc = 1, nchains
       call calc_likelihood(jc, .true., k(jc), z(:, jc), dvp(:, jc), dvs(:, jc), &
            sig(:, jc), log_likelihood(jc), rft(:, :, jc))
    end do
  end subroutine init_likelihood

  !---------------------------------------------------------------------
  ! log-likelihood (and trace) of a proposed model of chain `chain_id`;
  ! fwd_flag = .false. means "same model, new noise level": the chain's stored
  ! trace is returned and only the misfit term is re-evaluated.
  subroutine calc_likelihood(chain_id, fwd_flag, prop_k, prop_z, prop_dvp, prop_dvs, sig, &
       prop_log_likelihood, prop_rft)
    use params, only: k_max, ntrc, nfft, nlay_max
    use forward, only: rf_ctx
    use model, only: format_model
    integer, intent(in)  :: chain_id, prop_k
    logical, intent(in)  :: fwd_flag
    real(8), intent(in)  :: prop_z(k_max-1), prop_dvp(k_max), prop_dvs(k_max), sig(ntrc)
    real(8), intent(out) :: prop_log_likelihood, prop_rft(nfft, ntrc)
    real(8) :: vp(nlay_max), vs(nlay_max), dens(nlay_max), thick(nlay_max)
    integer :: nl
    logical :: usable

    if (.not. fwd_flag) then
       prop_rft = rft(:, :, chain_id)
       call rfgpu_check(rf_calc_likelihood_of_trace(rf_ctx, prop_rft, sig, prop_log_likelihood), &
            "rf_calc_likelihood_of_trace")
       return
    end if
    call format_model(prop_k, prop_z, prop_dvp, prop_dvs, nl, vp, vs, dens, thick, usable)
    call rfgpu_check(rf_calc_likelihood(rf_ctx, int(chain_id - 1, c_int32_t), 1_c_int32_t, &
         int(nl, c_int32_t), vp, vs, dens, thick, sig, prop_log_likelihood, prop_rft), "rf_calc_likelihood")
  end subroutine calc_likelihood

#ifdef RFGPU_USE_LAPACK
  !---------------------------------------------------------------------
  ! Truncated pseudo-inverse of the Gaussian noise-correlation matrix
  ! R(i,j) = r**((i-j)**2), r = exp(-(a*delta)**2), singular values <= 1e-3 dropped,
  ! computed with the host's dgesvd and uploaded to the engine.
  subroutine pseudo_inverse_from_lapack(verb)
    use params, only: nsmp, ntrc, a_gus, delta
    use forward, only: rf_ctx
    logical, intent(in) :: verb
    real(8), allocatable :: cov(:,:), sv(:), u(:,:), vt(:,:), work(:), pinv(:,:,:), scaled_v(:,:)
    real(8) :: r, wsize(1)
    integer :: jt, i, j, info

    allocate(cov(nsmp, nsmp), sv(nsmp), u(nsmp, nsmp), vt(nsmp, nsmp), scaled_v(nsmp, nsmp))
    allocate(pinv(nsmp, nsmp, ntrc))
    do jt = 1, ntrc
       r = exp(-a_gus(jt)**2 * delta**2)
       forall (i = 1:nsmp, j = 1:nsmp) cov(j, i) = r ** ((i - j) ** 2)
       call dgesvd('A', 'A', nsmp, nsmp, cov, nsmp, sv, u, nsmp, vt, nsmp, wsize, -1, info)
       allocate(work(nint(wsize(1))))
       call dgesvd('A', 'A', nsmp, nsmp, cov, nsmp, sv, u, nsmp, vt, nsmp, work, size(work), info)
       deallocate(work)
       if (info /= 0) call rfgpu_check(int(info, c_int), "dgesvd")
       ! matmul(transpose(vt), diag) of src/likelihood.f90:221 with diag(i,i) = 1.d0 / s(i) or 0 (:212-219):
       ! column i of the product is transpose(vt)(:, i) TIMES the reciprocal (not divided by s(i): that
       ! rounds differently); the zero off-diagonal terms of the reference's sum add exact zeros
       do i = 1, nsmp
          if (sv(i) > 1.0d-3) then
             scaled_v(:, i) = vt(i, :) * (1.d0 / sv(i))
          else
             scaled_v(:, i) = 0.d0
          end if
       end do
       pinv(:, :, jt) = matmul(scaled_v, transpose(u))
    end do
    call rfgpu_check(rf_set_r_inv(rf_ctx, pinv), "rf_set_r_inv")
    if (verb) write(*,*) "R inverse built with host LAPACK"
  end subroutine pseudo_inverse_from_lapack
#endif

end module likelihood
