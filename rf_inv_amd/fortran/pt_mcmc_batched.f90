!=======================================================================
! module pt_mcmc_batched -- throughput form of the reference sampler loop.
!
!   call pt_control_batched(verb)     ! instead of  call pt_control(verb)
!
! One iteration = propose every chain -> ONE batched format_model + forward +
! likelihood call on the GPU (rf_eval_models: the proposals go down as
! (k, z, dVp, dVs), ~0.75 KB per chain at k_max 30, from pinned host arrays
! the proposal step writes in place -- no layer stacks are packed or
! uploaded) -> accept/reject every chain -> rf_commit (does not wait for the
! device) -> temperature swap.  The host keeps format_model only for the
! validity verdict its random stream depends on (src/pt_mcmc.f90:163-169).
! It shares all state with the reference's modules (model: k, z, dvp, dvs; likelihood: sig, log_likelihood; params: temps;
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
! the host copy likelihood::rft is NOT kept up to date by this loop.  The
! posterior accumulators (src/pt_mcmc.f90:204-286) are kept on the device too
! (rf_post_record) and copied into module pt_mcmc's arrays after the last
! iteration, so the recorded chains' traces never cross PCIe.
!=======================================================================
module pt_mcmc_batched
  use iso_c_binding
  use rfgpu_c
  implicit none
  public pt_control_batched
  private
  ! Which RCCL build librfgpu loads for the temperature exchange ("" = its default search: librccl.so.1), set before
  ! pt_control_batched; and whether ranks that share a GPU may join its communicator -- functional tests only, over a
  ! test double of RCCL (real RCCL refuses two ranks on one device).
  character(len=1024), public :: rf_rccl_library = ""
  logical, public :: rf_exchange_shared_gpu_ok = .false.
  ! wall-clock seconds of the last pt_control_batched by phase: 1 proposals (host), 2 rf_eval_models (copies, kernels,
  ! wait), 3 accept / reject + rf_commit, 4 counters + posterior records, 5 temperature swap (incl. waiting for peers)
  real(8), public :: rf_phase_seconds(5) = 0.d0
  ! ... and of the iteration loop as a whole (first proposal to the last swap decision; set-up -- device tables, pinned
  ! arrays, the exchange's bootstrap -- and the read-back of the posterior accumulators excluded)
  real(8), public :: rf_loop_seconds = 0.d0
  ! 2 (default): the chains of a rank are worked in two halves, one being evaluated on the GPU while the host judges and
  ! re-proposes the other (same trajectory: see the loop); 1: propose all, evaluate all, judge all
  integer, public :: rf_pipeline_segments = 2
  ! .true. (default): the engine keeps only samples 1 .. nsmp of every trace while this loop runs (rf_set_option
  ! "trace_window") -- all the likelihood and the amplitude histograms read (src/likelihood.f90:88,
  ! src/pt_mcmc.f90:273-274); the trace kernels write nfft / nsmp times less and the resident traces shrink as much.
  ! The chains' current traces are rebuilt by one batched evaluation when the loop starts (bit-identical values).
  logical, public :: rf_windowed_traces = .true.

contains

  subroutine pt_control_batched(verb)
    use params
    use mt19937, only: grnd
    use math, only: gauss
    use prior, only: laplace, log_prior_ratio
    use model
    use likelihood, only: sig, log_likelihood
    use forward, only: rf_ctx
    use rf_model_check, only: proposal_is_valid
    use pt_mcmc
    include "mpif.h"
    logical, intent(in) :: verb
    integer :: nproc, rank, ierr, it, n_tot_iter, n_all, ichain, nb, i, j
    integer :: itype, nlay, cand_k
    logical :: live, yn
    real(8) :: alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max)
    real(8) :: lpr, r, del_s, t_cold
    ! per-chain proposals of the current iteration.  The arrays the engine reads live in pinned host memory
    ! (rf_host_alloc): they travel to the GPU by DMA as the proposal step left them.
    integer, allocatable :: p_type(:)
    ! column jc of p_z / p_dvp / p_dvs equals the chain's state except in rows p_lo(jc) .. p_hi(jc) (none: 1 .. 0), p_sig
    ! except where p_sigd(jc): a proposal touches one interface (a death: those from the removed one on), so the
    ! candidate is built, undone and adopted by copying that range instead of 3 k_max + ntrc numbers each time
    integer, allocatable :: p_lo(:), p_hi(:)
    logical, allocatable :: p_sigd(:)
    logical, allocatable :: p_live(:), p_acc(:)
    real(8), allocatable :: p_lp(:), p_logr(:)
    integer(c_int32_t), pointer, contiguous :: p_k(:), b_id(:), b_fwd(:)
    real(c_double), pointer, contiguous :: p_z(:,:), p_dvp(:,:), p_dvs(:,:), p_sig(:,:), b_logl(:)
    type(c_ptr) :: pin(8)
    integer(c_int32_t), allocatable :: b_acc(:)
    ! device-side posterior accumulation
    integer(c_int32_t), allocatable :: r_id(:)
    real(c_double), allocatable, target :: r_temps(:)
    logical :: record_now
    ! temperature exchange between ranks: RCCL (one GPU per rank) or MPI (ranks sharing a GPU)
    logical :: over_rccl
    real(8) :: tick(6)
    ! the pipeline: segments of the chains, the evaluation in flight for each, the swap proposal drawn but not yet decided
    integer :: nseg, iseg, lo, hi, seg_lo(2), seg_hi(2)
    integer(c_int32_t) :: seg_ticket(2)
    logical :: seg_busy(2), swap_drawn
    integer(c_int32_t) :: sw_pick(2)
    real(8) :: sw_logu

    rf_phase_seconds = 0.d0
    call mpi_comm_size(MPI_COMM_WORLD, nproc, ierr)
    call mpi_comm_rank(MPI_COMM_WORLD, rank, ierr)
    n_all = nproc * nchains
    n_tot_iter = nburn + niter
    t_cold = 1.d0 + 1.0e-6

    allocate(p_type(nchains), p_live(nchains), p_acc(nchains), p_lp(nchains), p_logr(nchains), b_acc(nchains))
    allocate(p_lo(nchains), p_hi(nchains), p_sigd(nchains))
    call pinned_i32(pin(1), p_k, nchains)
    call pinned_i32(pin(2), b_id, nchains)
    call pinned_i32(pin(3), b_fwd, nchains)
    call pinned_f64_2d(pin(4), p_z, k_max, nchains)
    call pinned_f64_2d(pin(5), p_dvp, k_max, nchains)
    call pinned_f64_2d(pin(6), p_dvs, k_max, nchains)
    call pinned_f64_2d(pin(7), p_sig, ntrc, nchains)
    call pinned_f64_1d(pin(8), b_logl, nchains)
    p_z = 0.d0;  p_dvp = 0.d0;  p_dvs = 0.d0
    allocate(r_id(nchains), r_temps(nchains))
    call setup_device_posterior()
    call open_temperature_exchange()

    ! the first evaluation of every chain (init_likelihood) becomes its current trace
    do ichain = 1, nchains
       b_id(ichain) = ichain - 1            ! (constant: every iteration hands over all chains, null proposals flagged)
       b_acc(ichain) = 1
    end do
    ! the proposal columns start as copies of the chains' states (draw_candidate keeps them that way)
    do ichain = 1, nchains
       p_k(ichain) = k(ichain)
       p_z(1:k_max-1, ichain) = z(1:k_max-1, ichain)
       p_dvp(:, ichain) = dvp(1:k_max, ichain)
       p_dvs(:, ichain) = dvs(1:k_max, ichain)
       p_sig(:, ichain) = sig(1:ntrc, ichain)
       p_lo(ichain) = 1
       p_hi(ichain) = 0
       p_sigd(ichain) = .false.
       b_fwd(ichain) = 1
    end do
    if (rf_windowed_traces) then
       ! windowed trace storage: switching drops every stored trace, so the chains' current models are evaluated once
       ! more, all at once -- the same kernels on the same inputs: the log-likelihoods must come back bit for bit
       call rfgpu_check(rf_set_option(rf_ctx, "trace_window" // c_null_char, 1.0_c_double), "rf_set_option")
       call rfgpu_check(rf_eval_models(rf_ctx, int(nchains, c_int32_t), b_id, b_fwd, p_k, p_z, int(k_max, c_int32_t), &
            & p_dvp, p_dvs, p_sig, b_logl, c_null_ptr), "rf_eval_models")
       do ichain = 1, nchains
          if (b_logl(ichain) /= log_likelihood(ichain) .and. &
               & (b_logl(ichain) == b_logl(ichain) .or. log_likelihood(ichain) == log_likelihood(ichain))) then   ! (NaN = NaN here)
             write(0,*) "ERROR: pt_control_batched: chain", ichain, " re-evaluated to", b_logl(ichain), " not", log_likelihood(ichain)
             call rfgpu_check(1_c_int, "windowed re-evaluation of the initial models")
          end if
       end do
    end if
    call rfgpu_check(rf_commit(rf_ctx, int(nchains, c_int32_t), b_id, b_acc), "rf_commit")

    ! ------------------------------------------------------------------------------------------------------
    ! The loop, software-pipelined over nseg segments of the chains (rf_pipeline_segments; 1 = no overlap).
    ! While segment s of iteration `it` is being evaluated on the GPU the host finishes and re-proposes the
    ! next segment.  Order of events, with S = nseg:
    !   slot (it, s):  wait for + accept + commit segment s of iteration it-1
    !                  [s = S: posterior records of it-1, then the swap DECISION of it-1]
    !                  propose segment s of iteration it, hand it to the engine (rf_eval_models_begin)
    !                  [s = S: the swap DRAWS of iteration it]
    ! Every random draw sits where the reference's sequential loop makes it (chains 1 .. n of an iteration, then
    ! the swap's draws, src/pt_mcmc.f90:488-571) and none depends on a likelihood; a chain is re-proposed only
    ! after its previous proposal has been judged; an iteration's acceptance tests see the temperatures left by
    ! the previous iteration's swap and its swap sees every chain's state after that iteration.  The trajectory
    ! is the reference's.
    ! ------------------------------------------------------------------------------------------------------
    nseg = max(1, min(rf_pipeline_segments, 2, nchains))
    do i = 1, nseg
       seg_lo(i) = (i - 1) * nchains / nseg + 1
       seg_hi(i) = i * nchains / nseg
    end do
    seg_busy = .false.
    swap_drawn = .false.
    call mpi_barrier(MPI_COMM_WORLD, ierr)
    rf_loop_seconds = mpi_wtime()

    do it = 1, n_tot_iter + 1
       if (verb .and. it <= n_tot_iter .and. mod(it, ncorr) == 0) write(*,*) "Iteration #:", it, "/", n_tot_iter
       do iseg = 1, nseg
          lo = seg_lo(iseg)
          hi = seg_hi(iseg)
          !----------------------------------------------------------------
          ! a. the segment's previous proposals: results, Metropolis-Hastings decisions, state update
          !----------------------------------------------------------------
          tick(1) = mpi_wtime()
          if (it > 1) then
             if (seg_busy(iseg)) then
                call rfgpu_check(rf_eval_wait(rf_ctx, seg_ticket(iseg), b_logl(lo), c_null_ptr), "rf_eval_wait")
             end if
             tick(2) = mpi_wtime()
             do ichain = lo, hi
                b_acc(ichain) = 0
                if (.not. p_live(ichain)) cycle
                del_s = (b_logl(ichain) - log_likelihood(ichain)) / temps(ichain) + p_lp(ichain)
                yn = (p_logr(ichain) <= del_s)
                if (yn) then
                   b_acc(ichain) = 1
                   log_likelihood(ichain) = b_logl(ichain)
                   ! the candidate becomes the state: it differs from it in rows p_lo .. p_hi only
                   k(ichain) = p_k(ichain)
                   i = p_lo(ichain)
                   j = p_hi(ichain)
                   if (j >= i) then
                      dvp(i:j, ichain) = p_dvp(i:j, ichain)
                      dvs(i:j, ichain) = p_dvs(i:j, ichain)
                      j = min(j, k_max - 1)
                      if (j >= i) z(i:j, ichain) = p_z(i:j, ichain)
                   end if
                   if (p_sigd(ichain)) sig(1:ntrc, ichain) = p_sig(1:ntrc, ichain)
                end if
                p_acc(ichain) = yn
             end do
             if (seg_busy(iseg)) then
                call rfgpu_check(rf_commit(rf_ctx, int(hi - lo + 1, c_int32_t), b_id(lo), b_acc(lo)), "rf_commit")
             end if
             seg_busy(iseg) = .false.
             ! counters of the non-tempered chains (iteration it-1)
             do ichain = lo, hi
                if (temps(ichain) <= t_cold) then
                   nprop(p_type(ichain)) = nprop(p_type(ichain)) + 1
                   if (p_acc(ichain)) naccept(p_type(ichain)) = naccept(p_type(ichain)) + 1
                   likelihood_hist(it - 1) = likelihood_hist(it - 1) + log_likelihood(ichain)
                end if
             end do
             tick(3) = mpi_wtime()
             rf_phase_seconds(2) = rf_phase_seconds(2) + (tick(2) - tick(1))
             rf_phase_seconds(3) = rf_phase_seconds(3) + (tick(3) - tick(2))
             if (iseg == nseg) then
                ! every chain has finished iteration it-1: posterior records, then its temperature swap
                record_now = (it - 1 > nburn .and. mod(it - 1, ncorr) == 0)
                if (record_now) then
                   ! every chain's state goes down; the device keeps the non-tempered ones (temps filter)
                   ! and reads their current traces where the evaluation left them
                   r_temps(1:nchains) = temps(1:nchains)
                   call rfgpu_check(rf_post_record(rf_ctx, int(nchains, c_int32_t), r_id, k, z, dvp, dvs, sig, &
                        & log_likelihood, c_loc(r_temps)), "rf_post_record")
                end if
                tick(4) = mpi_wtime()
                if (swap_drawn) call decide_temperature_swap()
                swap_drawn = .false.
                tick(5) = mpi_wtime()
                rf_phase_seconds(4) = rf_phase_seconds(4) + (tick(4) - tick(3))
                rf_phase_seconds(5) = rf_phase_seconds(5) + (tick(5) - tick(4))
             end if
          end if
          if (it > n_tot_iter) cycle          ! (the last pass only drains the pipeline)

          !----------------------------------------------------------------
          ! b. the segment's proposals of iteration `it`, chain by chain, in the reference's draw order
          !----------------------------------------------------------------
          tick(1) = mpi_wtime()
          nb = 0
          do ichain = lo, hi
             call draw_candidate(ichain)       ! writes the candidate into column ichain of p_k, p_z, p_dvp, p_dvs, p_sig
             p_type(ichain) = itype
             p_live(ichain) = live
             p_acc(ichain) = .false.
             b_fwd(ichain) = -1                ! a null proposal: the engine skips the item
             if (.not. live) cycle
             ! the acceptance uniform of the Metropolis-Hastings test, drawn at its place in the
             ! reference's stream (it does not depend on the likelihood)
             do
                r = grnd()
                if (r >= epsilon(1.d0)) exit
             end do
             p_logr(ichain) = log(r)
             p_lp(ichain) = lpr
             nb = nb + 1
             b_fwd(ichain) = merge(0, 1, itype == itype_sig)   ! noise-level move: the chain's stored trace is re-used
          end do
          tick(2) = mpi_wtime()
          !----------------------------------------------------------------
          ! c. format_model + forward + likelihood of the segment: enqueued, collected at this segment's next slot
          !----------------------------------------------------------------
          if (nb > 0) then
             call rfgpu_check(rf_eval_models_begin(rf_ctx, int(hi - lo + 1, c_int32_t), b_id(lo), b_fwd(lo), p_k(lo), &
                  & p_z(1, lo), int(k_max, c_int32_t), p_dvp(1, lo), p_dvs(1, lo), p_sig(1, lo), 0_c_int32_t, &
                  & seg_ticket(iseg)), "rf_eval_models_begin")
             seg_busy(iseg) = .true.
          end if
          tick(3) = mpi_wtime()
          rf_phase_seconds(1) = rf_phase_seconds(1) + (tick(2) - tick(1))
          rf_phase_seconds(2) = rf_phase_seconds(2) + (tick(3) - tick(2))
          !----------------------------------------------------------------
          ! d. the draws of this iteration's temperature-swap proposal (decided once every chain is through)
          !----------------------------------------------------------------
          if (iseg == nseg .and. nchains >= 2) then
             call draw_temperature_swap()
             swap_drawn = .true.
             rf_phase_seconds(5) = rf_phase_seconds(5) + (mpi_wtime() - tick(3))
          end if
       end do
    end do

    call mpi_barrier(MPI_COMM_WORLD, ierr)
    rf_loop_seconds = mpi_wtime() - rf_loop_seconds
    if (over_rccl) call rfgpu_check(rf_comm_destroy(rf_ctx), "rf_comm_destroy")
    call fetch_device_posterior()
    do i = 1, size(pin)
       call rfgpu_check(rf_host_free(pin(i)), "rf_host_free")
    end do

  contains

    ! arrays in pinned host memory (rf_host_alloc)
    subroutine pinned_i32(handle, a, n)
      type(c_ptr), intent(out) :: handle
      integer(c_int32_t), pointer, intent(out) :: a(:)
      integer, intent(in) :: n
      call rfgpu_check(rf_host_alloc(int(4 * max(n, 1), c_size_t), handle), "rf_host_alloc")
      call c_f_pointer(handle, a, [n])
    end subroutine pinned_i32

    subroutine pinned_f64_1d(handle, a, n)
      type(c_ptr), intent(out) :: handle
      real(c_double), pointer, intent(out) :: a(:)
      integer, intent(in) :: n
      call rfgpu_check(rf_host_alloc(int(8 * max(n, 1), c_size_t), handle), "rf_host_alloc")
      call c_f_pointer(handle, a, [n])
    end subroutine pinned_f64_1d

    subroutine pinned_f64_2d(handle, a, m, n)
      type(c_ptr), intent(out) :: handle
      real(c_double), pointer, intent(out) :: a(:,:)
      integer, intent(in) :: m, n
      call rfgpu_check(rf_host_alloc(int(8, c_size_t) * int(max(m * n, 1), c_size_t), handle), "rf_host_alloc")
      call c_f_pointer(handle, a, [m, n])
    end subroutine pinned_f64_2d

    ! One chain's trans-dimensional proposal.  Sets (host-associated) itype, cand_*, lpr,
    ! live and -- for live candidates -- the formatted layer stack nlay/alpha/beta/rho/h.
    ! Every grnd()/gauss()/laplace() call sits where the reference's step routine makes it,
    ! so the stream position after this call is the reference's.
    subroutine draw_candidate(jc)
      integer, intent(in) :: jc
      integer :: pick, ulo, uhi
      logical :: ok

      ! the candidate is built in place, in column jc of the pinned proposal arrays
      real(c_double), pointer :: cand_z(:), cand_dvp(:), cand_dvs(:), cand_sig(:)

      cand_z => p_z(:, jc);  cand_dvp => p_dvp(:, jc);  cand_dvs => p_dvs(:, jc);  cand_sig => p_sig(:, jc)
      cand_k = k(jc)
      ! the column back to the chain's state: only the rows the previous candidate touched can differ (after an
      ! acceptance they do not either)
      ulo = p_lo(jc)
      uhi = p_hi(jc)
      if (uhi >= ulo) then
         cand_dvp(ulo:uhi) = dvp(ulo:uhi, jc)
         cand_dvs(ulo:uhi) = dvs(ulo:uhi, jc)
         uhi = min(uhi, k_max - 1)
         if (uhi >= ulo) cand_z(ulo:uhi) = z(ulo:uhi, jc)
      end if
      if (p_sigd(jc)) cand_sig(:) = sig(1:ntrc, jc)
      p_lo(jc) = 1
      p_hi(jc) = 0
      p_sigd(jc) = .false.
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
            p_lo(jc) = cand_k
            p_hi(jc) = cand_k
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
            p_lo(jc) = pick
            p_hi(jc) = cand_k + 1
         end if
      else if (itype == itype_z) then
         pick = int(grnd() * cand_k) + 1
         cand_z(pick) = cand_z(pick) + gauss() * dev_z
         p_lo(jc) = pick
         p_hi(jc) = pick
         live = .not. (cand_z(pick) < z_min .or. cand_z(pick) > z_max)
      else if (itype == itype_dvs .or. itype == itype_dvp) then
         ! velocity perturbation of one layer (the last index addresses the half-space slot)
         pick = int(grnd() * (cand_k + 1)) + 1
         if (pick == cand_k + 1) pick = k_max
         p_lo(jc) = pick
         p_hi(jc) = pick
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
         p_sigd(jc) = .true.
         live = .not. (cand_sig(pick) < sig_min(pick) .or. cand_sig(pick) > sig_max(pick))
      end if

      ! only format_model's VERDICT is needed here (the engine formats the model itself, bit for bit the same):
      ! rf_model_check gives it without the sort of three arrays, the densities and the five output arrays
      if (live) live = proposal_is_valid(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs)
      p_k(jc) = cand_k
    end subroutine draw_candidate

    ! ------------------------------------------------------------------------------------------------
    ! Temperature exchange (what src/pt_mcmc.f90:498-571 does over MPI).  Temperatures move, states stay.
    !
    ! Transport between ranks: RCCL over xGMI through librfgpu (rf_comm_*, rf_pt_swap_exchange) when every
    ! rank drives its own GPU; RCCL cannot put two ranks on one device, so ranks that share a GPU (tests on
    ! a one-GPU box) keep MPI.  The decision is made once, unanimously, before the collective
    ! ncclCommInitRank.  Either way a cross-rank proposal is ONE symmetric exchange of (T, logL, log u):
    ! both sides evaluate the same Metropolis rule with the uniform of the rank that owns the first walker
    ! (the rank that judges in the reference, :544-556), instead of a message there and a temperature back.
    ! ------------------------------------------------------------------------------------------------
    subroutine open_temperature_exchange()
      integer(c_int64_t) :: my_key
      integer(c_int64_t), allocatable :: keys(:)
      integer(c_int8_t) :: token(RF_COMM_ID_BYTES)
      integer :: usable, all_usable, ia, ib2

      over_rccl = .false.
      if (nproc < 2) return
      usable = 0
      if (len_trim(rf_rccl_library) > 0) then
         call rfgpu_check(rf_comm_set_library(trim(rf_rccl_library) // c_null_char), "rf_comm_set_library")
      end if
      ! which GPU every rank drives (costs nothing); RCCL itself -- a second to load -- only if each has its own
      call rfgpu_check(rf_comm_device_key(rf_ctx, my_key), "rf_comm_device_key")
      allocate(keys(nproc))
      call mpi_allgather(my_key, 1, MPI_INTEGER8, keys, 1, MPI_INTEGER8, MPI_COMM_WORLD, ierr)
      usable = 1
      do ia = 1, nproc - 1
         do ib2 = ia + 1, nproc
            if (keys(ia) == keys(ib2) .and. .not. rf_exchange_shared_gpu_ok) usable = 0      ! two ranks on one GPU
         end do
      end do
      if (usable == 1) then
         if (rf_comm_probe(rf_ctx, my_key) /= 0) usable = 0
      end if
      token = 0
      if (rank == 0 .and. usable == 1) then
         if (rf_comm_get_unique_id(token) /= 0) usable = 0
      end if
      call mpi_allreduce(usable, all_usable, 1, MPI_INTEGER4, MPI_MIN, MPI_COMM_WORLD, ierr)
      if (all_usable == 1) then
         call mpi_bcast(token, RF_COMM_ID_BYTES, MPI_BYTE, 0, MPI_COMM_WORLD, ierr)
         call rfgpu_check(rf_comm_init(rf_ctx, token, int(rank, c_int32_t), int(nproc, c_int32_t)), "rf_comm_init")
         over_rccl = .true.
      end if
      ! (always said, on rank 0's error unit: a silent fall-back to MPI on a multi-GPU node would be a performance bug
      ! nobody sees; the GPU identities compared above carry the host name, so ranks of different nodes never
      ! count as sharing a device)
      if (rank == 0) then
         if (over_rccl) then
            write(0,'(A,I0,A)') " Temperature exchange: RCCL (", nproc, " ranks, one GPU each)"
         else
            write(0,'(A)') " Temperature exchange: MPI (ranks share a GPU, or RCCL is not available)"
         end if
      end if
    end subroutine open_temperature_exchange

    ! The swap proposal of an iteration in two parts.  draw_temperature_swap: everything that touches the random
    ! stream or names the pair -- rank 0 draws two distinct walkers of the whole ensemble by global id, everybody
    ! learns the pair, and the rank of the first walker draws the uniform of the test (src/pt_mcmc.f90:501-519,
    ! :529,:548: none of it depends on a likelihood).  decide_temperature_swap: the Metropolis test itself, once
    ! every chain of this rank has finished the iteration.
    subroutine draw_temperature_swap()
      integer :: owner(2)

      sw_pick = 0
      sw_logu = 0.d0
      if (rank == 0) then
         sw_pick(1) = int(grnd() * n_all, c_int32_t)
         do
            sw_pick(2) = int(grnd() * n_all, c_int32_t)
            if (sw_pick(2) /= sw_pick(1)) exit
         end do
      end if
      if (over_rccl) then
         call rfgpu_check(rf_comm_bcast_i32(rf_ctx, sw_pick, 2_c_int32_t, 0_c_int32_t), "rf_comm_bcast_i32")
      else if (nproc > 1) then
         call mpi_bcast(sw_pick, 2, MPI_INTEGER4, 0, MPI_COMM_WORLD, ierr)
      end if
      owner = sw_pick / nchains              ! global id -> (rank, chain), src/pt_mcmc.f90:508-511
      if (owner(1) == rank) sw_logu = log(grnd())   ! the first walker's rank supplies the uniform (both local, or cross-rank)
    end subroutine draw_temperature_swap

    subroutine decide_temperature_swap()
      integer(c_int32_t) :: verdict
      integer :: owner(2), slot(2), mine, theirs
      real(8) :: gain, t_now

      owner = sw_pick / nchains
      slot = mod(sw_pick, nchains) + 1
      if (owner(1) /= rank .and. owner(2) /= rank) return

      if (owner(1) == owner(2)) then
         ! both walkers live here
         gain = (log_likelihood(slot(2)) - log_likelihood(slot(1))) * (1.d0 / temps(slot(1)) - 1.d0 / temps(slot(2)))
         if (sw_logu <= gain) then
            t_now = temps(slot(1))
            temps(slot(1)) = temps(slot(2))
            temps(slot(2)) = t_now
         end if
         return
      end if

      mine = merge(1, 2, owner(1) == rank)   ! which walker of the pair is on this rank
      theirs = 3 - mine
      if (over_rccl) then
         call rfgpu_check(rf_pt_swap_exchange(rf_ctx, int(owner(theirs), c_int32_t), int(2 - mine, c_int32_t), &
              & temps(slot(mine)), log_likelihood(slot(mine)), sw_logu, t_now, verdict), "rf_pt_swap_exchange")
      else
         call exchange_over_mpi(owner(theirs), mine == 1, temps(slot(mine)), log_likelihood(slot(mine)), sw_logu, t_now)
      end if
      temps(slot(mine)) = t_now
    end subroutine decide_temperature_swap

    ! the same symmetric exchange as rf_pt_swap_exchange, for ranks that share a GPU
    subroutine exchange_over_mpi(peer, first, t_mine, l_mine, logu, t_after)
      integer, intent(in) :: peer
      logical, intent(in) :: first
      real(8), intent(in) :: t_mine, l_mine, logu
      real(8), intent(out) :: t_after
      real(8) :: outgoing(3), incoming(3), t1, t2, l1, l2, u
      integer :: st(MPI_STATUS_SIZE)

      outgoing = [t_mine, l_mine, logu]
      call mpi_sendrecv(outgoing, 3, MPI_REAL8, peer, 77, incoming, 3, MPI_REAL8, peer, 77, MPI_COMM_WORLD, st, ierr)
      if (first) then
         t1 = t_mine; l1 = l_mine; t2 = incoming(1); l2 = incoming(2); u = logu
      else
         t2 = t_mine; l2 = l_mine; t1 = incoming(1); l1 = incoming(2); u = incoming(3)
      end if
      t_after = t_mine
      if (u <= (l2 - l1) * (1.d0 / t1 - 1.d0 / t2)) t_after = incoming(1)
    end subroutine exchange_over_mpi

    ! Hands format_model's tables and the histogram layout of init_pt_mcmc
    ! (src/pt_mcmc.f90:394-430) to the engine.
    subroutine setup_device_posterior()
      type(rf_model_config) :: mc
      type(rf_post_config) :: pc
      real(c_double), allocatable, target, save :: t_vp(:), t_vs(:), t_smin(:), t_smax(:)
      integer(c_int32_t), allocatable, target, save :: t_smode(:)
      integer :: ic

      if (allocated(t_vp)) deallocate(t_vp, t_vs, t_smin, t_smax, t_smode)
      allocate(t_vp(size(vp_ref)), t_vs(size(vs_ref)), t_smin(ntrc), t_smax(ntrc), t_smode(ntrc))
      t_vp = vp_ref
      t_vs = vs_ref
      t_smin = sig_min(1:ntrc)
      t_smax = sig_max(1:ntrc)
      t_smode = sig_mode(1:ntrc)
      mc%k_max = k_max;  mc%vp_mode = vp_mode;  mc%nref = size(vp_ref)
      mc%z_max = z_max;  mc%h_min = h_min;  mc%z_ref_min = z_ref_min;  mc%dz_ref = dz_ref
      mc%vp_min = vp_min;  mc%vp_max = vp_max;  mc%vs_min = vs_min;  mc%vs_max = vs_max
      mc%vpvs_min = vpvs_min;  mc%vpvs_max = vpvs_max
      mc%vp_ref = c_loc(t_vp);  mc%vs_ref = c_loc(t_vs)
      call rfgpu_check(rf_set_model(rf_ctx, mc), "rf_set_model")
      pc%nbin_z = nbin_z;  pc%nbin_vs = nbin_vs;  pc%nbin_vp = nbin_vp;  pc%nbin_vpvs = nbin_vpvs
      pc%nbin_sig = nbin_sig;  pc%nbin_amp = nbin_amp
      pc%amp_min = amp_min;  pc%amp_max = amp_max;  pc%z_min = z_min
      pc%sig_min = c_loc(t_smin);  pc%sig_max = c_loc(t_smax);  pc%sig_mode = c_loc(t_smode)
      pc%max_models = size(all_likelihood)
      call rfgpu_check(rf_post_create(rf_ctx, pc), "rf_post_create")
      do ic = 1, nchains
         r_id(ic) = ic - 1
      end do
    end subroutine setup_device_posterior

    ! Device accumulators -> the arrays of module pt_mcmc that output_results reads.
    subroutine fetch_device_posterior()
      type(rf_post_result) :: pr
      integer(c_int32_t), target :: t_nmod
      integer(c_int64_t), target :: t_oor
      integer(c_int32_t), allocatable, target :: t_nk(:), t_nz(:), t_nsig(:,:), t_namp(:,:,:)
      integer(c_int32_t), allocatable, target :: t_nvpz(:,:), t_nvsz(:,:), t_nvpvsz(:,:)
      real(c_double), allocatable, target :: t_vpm(:), t_vsm(:), t_vpvsm(:), t_vpmod(:,:), t_vsmod(:,:), t_all(:)
      integer :: nm

      ! how many model slots are in use (the profile arrays are sized for every chain, only the
      ! non-tempered ones record)
      pr%nmod = c_loc(t_nmod)
      pr%nk = c_null_ptr;  pr%nz = c_null_ptr;  pr%nsig = c_null_ptr;  pr%namp = c_null_ptr
      pr%nvpz = c_null_ptr;  pr%nvsz = c_null_ptr;  pr%nvpvsz = c_null_ptr
      pr%vp_mean = c_null_ptr;  pr%vs_mean = c_null_ptr;  pr%vpvs_mean = c_null_ptr
      pr%vp_model = c_null_ptr;  pr%vs_model = c_null_ptr;  pr%all_likelihood = c_null_ptr
      pr%amp_out_of_range = c_null_ptr
      call rfgpu_check(rf_post_read(rf_ctx, pr), "rf_post_read")
      nm = max(1, min(int(t_nmod), size(all_likelihood)))
      allocate(t_nk(k_max), t_nz(nbin_z), t_nsig(nbin_sig, ntrc), t_namp(nbin_amp, nsmp, ntrc))
      allocate(t_nvpz(nbin_z, nbin_vp), t_nvsz(nbin_z, nbin_vs), t_nvpvsz(nbin_z, nbin_vpvs))
      allocate(t_vpm(nbin_z), t_vsm(nbin_z), t_vpvsm(nbin_z), t_vpmod(nbin_z, nm), t_vsmod(nbin_z, nm), t_all(nm))
      pr%nmod = c_loc(t_nmod);  pr%nk = c_loc(t_nk);  pr%nz = c_loc(t_nz);  pr%nsig = c_loc(t_nsig)
      pr%namp = c_loc(t_namp);  pr%nvpz = c_loc(t_nvpz);  pr%nvsz = c_loc(t_nvsz);  pr%nvpvsz = c_loc(t_nvpvsz)
      pr%vp_mean = c_loc(t_vpm);  pr%vs_mean = c_loc(t_vsm);  pr%vpvs_mean = c_loc(t_vpvsm)
      pr%vp_model = c_loc(t_vpmod);  pr%vs_model = c_loc(t_vsmod);  pr%all_likelihood = c_loc(t_all)
      pr%amp_out_of_range = c_loc(t_oor)
      call rfgpu_check(rf_post_read(rf_ctx, pr), "rf_post_read")
      nmod = t_nmod
      nk = t_nk;  nz = t_nz;  nsig = t_nsig;  namp = t_namp
      nvpz = t_nvpz;  nvsz = t_nvsz;  nvpvsz = t_nvpvsz
      vp_mean = t_vpm;  vs_mean = t_vsm;  vpvs_mean = t_vpvsm
      if (t_nmod > 0) then
         nm = min(int(t_nmod), size(all_likelihood))
         vp_model(:, 1:nm) = t_vpmod(:, 1:nm)
         vs_model(:, 1:nm) = t_vsmod(:, 1:nm)
         all_likelihood(1:nm) = t_all(1:nm)
      end if
      if (t_oor > 0) write(0,*) "Warning: RF amp. out of range (", t_oor, " samples)"
    end subroutine fetch_device_posterior

  end subroutine pt_control_batched

end module pt_mcmc_batched
