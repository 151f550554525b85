!=======================================================================
! module pt_mcmc_batched -- throughput form of the reference sampler loop.
!
!   call pt_control_batched(verb)     ! instead of  call pt_control(verb)
!
! One iteration = propose every chain -> ONE batched forward+likelihood call
! on the GPU (rf_eval_batch) -> accept/reject every chain -> rf_commit ->
! temperature swap.  It shares all state with the reference's modules
! (model: k, z, dvp, dvs; likelihood: sig, log_likelihood; params: temps;
! pt_mcmc: every counter and histogram), so init_pt_mcmc before it and
! output_results after it work unchanged.
!
! The random stream is consumed in exactly the order of the reference's
! sequential loop (src/pt_mcmc.f90:488-571): per chain the proposal draws and,
! iff the proposal is not null, the acceptance uniform of judge_mcmc -- which
! does not depend on the likelihood and can therefore be drawn before the
! batched evaluation.  The trajectory is identical to pt_control's.
!
! Written from scratch for rf_inv_amd (it restates what one step of the
! reference's private `mcmc` does; the reference subroutine itself evaluates
! one chain per call and cannot be batched).  The traces stay on the device;
! the host copy likelihood::rft is NOT kept up to date by this loop.
!=======================================================================
module pt_mcmc_batched
  use iso_c_binding
  use rfgpu_c
  implicit none
  public pt_control_batched
  private

contains

  subroutine pt_control_batched(verb)
    use params
    use mt19937, only: grnd
    use math, only: gauss
    use prior, only: laplace, log_prior_ratio
    use model
    use likelihood, only: sig, log_likelihood
    use forward, only: rf_ctx
    use pt_mcmc
    include "mpif.h"
    logical, intent(in) :: verb
    integer :: nproc, rank, ierr, it, n_tot_iter, n_all, ichain, nb, ib, i
    integer :: itype, nlay, nlay_pad, cand_k
    logical :: live, yn
    real(8) :: alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max)
    real(8) :: cand_z(k_max), cand_dvp(k_max), cand_dvs(k_max), cand_sig(ntrc)
    real(8) :: lpr, r, del_s, t_cold
    ! per-chain proposals of the current iteration
    integer, allocatable :: p_k(:), p_type(:)
    logical, allocatable :: p_live(:), p_acc(:)
    real(8), allocatable :: p_z(:,:), p_dvp(:,:), p_dvs(:,:), p_sig(:,:), p_lp(:), p_logr(:)
    ! the batch handed to the engine
    integer(c_int32_t), allocatable :: b_id(:), b_fwd(:), b_nlay(:), b_acc(:)
    real(c_double), allocatable :: b_layers(:,:,:), b_sig(:,:), b_logl(:)
    ! traces of the chains recorded in this iteration (one gather per recording iteration)
    integer(c_int32_t), allocatable :: r_id(:)
    real(c_double), allocatable :: r_trace(:,:,:)
    integer :: nrec, irec
    ! temperature swap
    integer :: ipack(4), rank1, rank2, ichain1, ichain2, itarget1, itarget2
    integer :: status(MPI_STATUS_SIZE)
    real(8) :: temp1, temp2, e1, e2, rpack(2)

    call mpi_comm_size(MPI_COMM_WORLD, nproc, ierr)
    call mpi_comm_rank(MPI_COMM_WORLD, rank, ierr)
    n_all = nproc * nchains
    n_tot_iter = nburn + niter
    nlay_pad = k_max + 2
    t_cold = 1.d0 + 1.0e-6

    allocate(p_k(nchains), p_type(nchains), p_live(nchains), p_acc(nchains))
    allocate(p_z(k_max, nchains), p_dvp(k_max, nchains), p_dvs(k_max, nchains))
    allocate(p_sig(ntrc, nchains), p_lp(nchains), p_logr(nchains))
    allocate(b_id(nchains), b_fwd(nchains), b_nlay(nchains), b_acc(nchains))
    allocate(b_layers(nlay_pad, 4, nchains), b_sig(ntrc, nchains), b_logl(nchains))
    allocate(r_id(nchains), r_trace(nsmp, ntrc, nchains))

    ! the first evaluation of every chain (init_likelihood) becomes its current trace
    do ichain = 1, nchains
       b_id(ichain) = ichain - 1
       b_acc(ichain) = 1
    end do
    call rfgpu_check(rf_commit(rf_ctx, int(nchains, c_int32_t), b_id, b_acc), "rf_commit")

    do it = 1, n_tot_iter
       if (verb .and. mod(it, ncorr) == 0) write(*,*) "Iteration #:", it, "/", n_tot_iter

       !----------------------------------------------------------------
       ! 1. proposals, chain by chain, in the reference's draw order
       !----------------------------------------------------------------
       nb = 0
       do ichain = 1, nchains
          call draw_candidate(ichain)
          p_type(ichain) = itype
          p_live(ichain) = live
          p_acc(ichain) = .false.
          if (.not. live) cycle
          ! the acceptance uniform of the Metropolis-Hastings test, drawn at its place in the
          ! reference's stream (it does not depend on the likelihood)
          do
             r = grnd()
             if (r >= epsilon(1.d0)) exit
          end do
          p_logr(ichain) = log(r)
          p_lp(ichain) = lpr
          p_k(ichain) = cand_k
          p_z(1:k_max-1, ichain) = cand_z(1:k_max-1)
          p_dvp(:, ichain) = cand_dvp
          p_dvs(:, ichain) = cand_dvs
          p_sig(:, ichain) = cand_sig
          nb = nb + 1
          b_id(nb) = ichain - 1
          b_sig(:, nb) = cand_sig
          b_layers(:, :, nb) = 1.d0
          if (itype == itype_sig) then
             b_fwd(nb) = 0          ! noise-level move: the chain's stored trace is re-used
             b_nlay(nb) = 2
          else
             b_fwd(nb) = 1
             b_nlay(nb) = nlay
             b_layers(1:nlay, 1, nb) = alpha(1:nlay)
             b_layers(1:nlay, 2, nb) = beta(1:nlay)
             b_layers(1:nlay, 3, nb) = rho(1:nlay)
             b_layers(1:nlay, 4, nb) = h(1:nlay)
          end if
       end do

       !----------------------------------------------------------------
       ! 2. one batched forward + likelihood evaluation on the GPU
       !----------------------------------------------------------------
       if (nb > 0) then
          call rfgpu_check(rf_eval_batch(rf_ctx, int(nb, c_int32_t), b_id, b_fwd, b_nlay, &
               & int(nlay_pad, c_int32_t), b_layers, b_sig, b_logl), "rf_eval_batch")
       end if

       !----------------------------------------------------------------
       ! 3. Metropolis-Hastings decisions and state update
       !----------------------------------------------------------------
       do ib = 1, nb
          ichain = b_id(ib) + 1
          del_s = (b_logl(ib) - log_likelihood(ichain)) / temps(ichain) + p_lp(ichain)
          yn = (p_logr(ichain) <= del_s)
          b_acc(ib) = 0
          if (yn) then
             b_acc(ib) = 1
             log_likelihood(ichain) = b_logl(ib)
             k(ichain) = p_k(ichain)
             dvp(1:k_max, ichain) = p_dvp(1:k_max, ichain)
             dvs(1:k_max, ichain) = p_dvs(1:k_max, ichain)
             z(1:k_max-1, ichain) = p_z(1:k_max-1, ichain)
             sig(1:ntrc, ichain) = p_sig(1:ntrc, ichain)
          end if
          p_acc(ichain) = yn
       end do
       if (nb > 0) call rfgpu_check(rf_commit(rf_ctx, int(nb, c_int32_t), b_id, b_acc), "rf_commit")

       !----------------------------------------------------------------
       ! 4. counters and posterior records of the non-tempered chains
       !----------------------------------------------------------------
       nrec = 0
       do ichain = 1, nchains
          if (temps(ichain) <= t_cold) then
             nprop(p_type(ichain)) = nprop(p_type(ichain)) + 1
             if (p_acc(ichain)) naccept(p_type(ichain)) = naccept(p_type(ichain)) + 1
             likelihood_hist(it) = likelihood_hist(it) + log_likelihood(ichain)
             if (it > nburn .and. mod(it, ncorr) == 0) then
                nrec = nrec + 1
                r_id(nrec) = ichain - 1
             end if
          end if
       end do
       if (nrec > 0) then
          ! the recorded chains' current traces live on the device: one gather for all
          call rfgpu_check(rf_get_rft_batch(rf_ctx, int(nrec, c_int32_t), r_id, 0_c_int32_t, &
               & int(nsmp, c_int32_t), r_trace), "rf_get_rft_batch")
          do irec = 1, nrec
             call record_sample(r_id(irec) + 1, r_trace(:, :, irec))
          end do
       end if

       !----------------------------------------------------------------
       ! 5. one temperature-swap proposal for the whole ensemble
       !----------------------------------------------------------------
       if (nchains < 2) cycle
       if (rank == 0) then
          itarget1 = int(grnd() * n_all)
          do
             itarget2 = int(grnd() * n_all)
             if (itarget2 /= itarget1) exit
          end do
          ipack(1) = itarget1 / nchains
          ipack(2) = itarget2 / nchains
          ipack(3) = mod(itarget1, nchains) + 1
          ipack(4) = mod(itarget2, nchains) + 1
       end if
       call mpi_bcast(ipack, 4, MPI_INTEGER4, 0, MPI_COMM_WORLD, ierr)
       rank1 = ipack(1)
       rank2 = ipack(2)
       ichain1 = ipack(3)
       ichain2 = ipack(4)
       if (rank1 == rank .and. rank2 == rank) then
          temp1 = temps(ichain1)
          temp2 = temps(ichain2)
          if (swap_ok(temp1, temp2, log_likelihood(ichain1), log_likelihood(ichain2))) then
             temps(ichain2) = temp1
             temps(ichain1) = temp2
          end if
       else if (rank1 == rank) then
          call mpi_recv(rpack, 2, MPI_REAL8, rank2, 2018, MPI_COMM_WORLD, status, ierr)
          temp1 = temps(ichain1)
          temp2 = rpack(1)
          e1 = log_likelihood(ichain1)
          e2 = rpack(2)
          if (swap_ok(temp1, temp2, e1, e2)) then
             temps(ichain1) = temp2
             rpack(1) = temp1
          end if
          call mpi_send(rpack, 1, MPI_REAL8, rank2, 1988, MPI_COMM_WORLD, ierr)
       else if (rank2 == rank) then
          rpack(1) = temps(ichain2)
          rpack(2) = log_likelihood(ichain2)
          call mpi_send(rpack, 2, MPI_REAL8, rank1, 2018, MPI_COMM_WORLD, ierr)
          call mpi_recv(rpack, 1, MPI_REAL8, rank1, 1988, MPI_COMM_WORLD, status, ierr)
          temps(ichain2) = rpack(1)
       end if
    end do

  contains

    ! One chain's trans-dimensional proposal.  Sets (host-associated) itype, cand_*, lpr,
    ! live and -- for live candidates -- the formatted layer stack nlay/alpha/beta/rho/h.
    ! Every grnd()/gauss()/laplace() call sits where the reference's step routine makes it,
    ! so the stream position after this call is the reference's.
    subroutine draw_candidate(jc)
      integer, intent(in) :: jc
      integer :: pick
      logical :: ok

      cand_k = k(jc)
      cand_dvp(:) = dvp(1:k_max, jc)
      cand_dvs(:) = dvs(1:k_max, jc)
      cand_z(1:k_max-1) = z(1:k_max-1, jc)
      cand_sig(:) = sig(1:ntrc, jc)
      lpr = 0.d0
      live = .true.

      itype = int(grnd() * ntype) + 1
      if (itype == itype_birth) then
         ! add an interface: perturbations first (dVp, dVs), depth last
         cand_k = cand_k + 1
         live = cand_k < k_max
         if (live) then
            select case (prior_mode)
            case (1)
               cand_dvp(cand_k) = laplace() * dvp_prior
               cand_dvs(cand_k) = laplace() * dvs_prior
            case (2)
               cand_dvp(cand_k) = gauss() * dvp_prior
               cand_dvs(cand_k) = gauss() * dvs_prior
            end select
            cand_z(cand_k) = z_min + grnd() * (z_max - z_min)
         end if
      else if (itype == itype_death) then
         ! remove interface `pick`: close the gap, clear the vacated slot
         cand_k = cand_k - 1
         live = cand_k >= k_min
         if (live) then
            pick = int(grnd() * (cand_k + 1)) + 1
            if (pick <= cand_k) then
               cand_dvp(pick:cand_k) = dvp(pick+1:cand_k+1, jc)
               cand_dvs(pick:cand_k) = dvs(pick+1:cand_k+1, jc)
               cand_z(pick:cand_k) = z(pick+1:cand_k+1, jc)
            end if
            cand_dvp(cand_k + 1) = 0.d0
            cand_dvs(cand_k + 1) = 0.d0
            cand_z(cand_k + 1) = 0.d0
         end if
      else if (itype == itype_z) then
         pick = int(grnd() * cand_k) + 1
         cand_z(pick) = cand_z(pick) + gauss() * dev_z
         live = .not. (cand_z(pick) < z_min .or. cand_z(pick) > z_max)
      else if (itype == itype_dvs .or. itype == itype_dvp) then
         ! velocity perturbation of one layer (the last index addresses the half-space slot)
         pick = int(grnd() * (cand_k + 1)) + 1
         if (pick == cand_k + 1) pick = k_max
         if (itype == itype_dvs) then
            cand_dvs(pick) = cand_dvs(pick) + gauss() * dev_dvs
            lpr = log_prior_ratio(cand_dvs(pick), dvs(pick, jc), dvs_prior, prior_mode)
         else
            cand_dvp(pick) = cand_dvp(pick) + gauss() * dev_dvp
            lpr = log_prior_ratio(cand_dvp(pick), dvp(pick, jc), dvp_prior, prior_mode)
         end if
      else if (itype == itype_sig) then
         pick = isig_trc(int(grnd() * nsig_trc) + 1)
         cand_sig(pick) = cand_sig(pick) + gauss() * dev_sig
         live = .not. (cand_sig(pick) < sig_min(pick) .or. cand_sig(pick) > sig_max(pick))
      end if

      if (live) then
         call format_model(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs, nlay, alpha, beta, rho, h, ok)
         live = ok
      end if
    end subroutine draw_candidate

    ! Metropolis rule of the temperature exchange; draws one uniform
    logical function swap_ok(t1, t2, l1, l2)
      real(8), intent(in) :: t1, t2, l1, l2
      swap_ok = (log(grnd()) <= (l2 - l1) * (1.d0 / t1 - 1.d0 / t2))
    end function swap_ok

    ! posterior bookkeeping of one non-tempered chain: the same bins, in the same
    ! order, as the record block of the reference's step routine, so that
    ! output_results (mcmc_out) produces identical files.
    subroutine record_sample(jc, trace)
      integer, intent(in) :: jc
      real(8), intent(in) :: trace(nsmp, ntrc)
      integer :: itrc, ibin, il, iz, iz1, iz2, ivp, ivs, ivpvs, nl, ismp
      real(8) :: a(nlay_max), b(nlay_max), rh(nlay_max), th(nlay_max), tmpz
      logical :: ok

      nmod = nmod + 1
      all_likelihood(nmod) = log_likelihood(jc)
      nk(k(jc)) = nk(k(jc)) + 1
      do itrc = 1, ntrc
         if (sig_mode(itrc) == 1) then
            ibin = int((sig(itrc, jc) - sig_min(itrc)) / dbin_sig(itrc)) + 1
            nsig(ibin, itrc) = nsig(ibin, itrc) + 1
         end if
      end do
      do il = 1, k(jc) - 1
         ibin = int((z(il, jc) - z_min) / dbin_z) + 1
         nz(ibin) = nz(ibin) + 1
      end do

      call format_model(k(jc), z(1:k_max-1, jc), dvp(1:k_max, jc), dvs(1:k_max, jc), &
           & nl, a, b, rh, th, ok)
      tmpz = 0.d0
      do il = 1, nl
         iz1 = int(tmpz / dbin_z) + 1
         if (il < nl) then
            iz2 = int((tmpz + th(il)) / dbin_z) + 1
         else
            iz2 = nbin_z + 1
         end if
         ivp = int((a(il) - vp_min) / dbin_vp) + 1
         ivs = max(1, int((b(il) - vs_min) / dbin_vs) + 1)
         ivpvs = int(((a(il) / b(il)) - vpvs_min) / dbin_vpvs) + 1
         ivpvs = min(max(1, ivpvs), nbin_vpvs)
         do iz = iz1, iz2 - 1
            nvpz(iz, ivp) = nvpz(iz, ivp) + 1
            vp_mean(iz) = vp_mean(iz) + a(il)
            vp_model(iz, nmod) = a(il)
            nvsz(iz, ivs) = nvsz(iz, ivs) + 1
            nvpvsz(iz, ivpvs) = nvpvsz(iz, ivpvs) + 1
            if (b(il) > 0.d0) then
               vpvs_mean(iz) = vpvs_mean(iz) + a(il) / b(il)
               vs_mean(iz) = vs_mean(iz) + b(il)
               vs_model(iz, nmod) = b(il)
            else
               vpvs_mean(iz) = vpvs_min
               vs_mean(iz) = vs_min
               vs_model(iz, nmod) = vs_min
            end if
            vp_model(iz, nmod) = a(il)
         end do
         tmpz = tmpz + th(il)
      end do

      do itrc = 1, ntrc
         do ismp = 1, nsmp
            ibin = int((trace(ismp, itrc) - amp_min) / dbin_amp) + 1
            if (ibin < 1) then
               write(0,*) "Warning: RF amp. out of range"
               ibin = 1
            else if (ibin > nbin_amp) then
               write(0,*) "Warning: RF amp. out of range"
               ibin = nbin_amp
            end if
            namp(ibin, ismp, itrc) = namp(ibin, ismp, itrc) + 1
         end do
      end do
    end subroutine record_sample

  end subroutine pt_control_batched

end module pt_mcmc_batched
