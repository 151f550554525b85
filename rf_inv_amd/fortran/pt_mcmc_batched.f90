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
  ! ... and, within phase 2, of a GPU group's steps on this rank: 1 rf_eval_wait (first rank), 2 the barrier after it,
  ! 3 the barrier before the engine calls, 4 / 5 / 6 the engine calls: rf_commit, rf_post_record, rf_eval_models_begin (first rank)
  real(8), public :: rf_group_seconds(6) = 0.d0
  ! 2: the chains of a rank are worked in two halves, one being evaluated on the GPU while the host judges and
  ! re-proposes the other (same trajectory: see the loop); 1: propose all, evaluate all, judge all; 0 (default): 2 when
  ! a half is at least 1024 chains, else 1 -- below that two small launches per iteration cost the GPU more than the
  ! overlap returns (8 ranks x 1024 chains on one GPU, C4 shape: 4.9 M steps/s with 1 segment against 4.0 M with 2)
  integer, public :: rf_pipeline_segments = 0
  ! .true. (default): the engine keeps only samples 1 .. nsmp of every trace while this loop runs (rf_set_option
  ! "trace_window") -- all the likelihood and the amplitude histograms read (src/likelihood.f90:88,
  ! src/pt_mcmc.f90:273-274); the trace kernels write nfft / nsmp times less and the resident traces shrink as much.
  ! The chains' current traces are rebuilt by one batched evaluation when the loop starts (bit-identical values).
  logical, public :: rf_windowed_traces = .true.
  ! .true.: ranks that drive the same GPU form a group whose first rank evaluates the chains of all of them on ONE
  ! context -- one full-size launch per pipeline segment and iteration instead of one small launch per rank (the
  ! proposals travel through shared memory registered with the GPU, rf_host_alloc_shared; every rank keeps its own
  ! random stream, state and posterior accumulators: same trajectories, same result files).  .false. (default): every
  ! rank launches for itself.  Measured (profiles/r04_sampler_rate_shapes.txt): the group loses -- one stream runs a
  ! segment's copies, format_model, stage and commit kernels one after the other, where independent ranks' streams
  ! overlap them with each other's main kernels (C4 shape, 4 ranks: 4.5 M against 4.8 M steps/s; C3: 12.0 M against 13.7 M).
  logical, public :: rf_share_gpu = .false.

  ! .true.: HIP-event times of the loop's kernels (rf_profile_enable on the evaluating context): totals in
  ! rf_kernel_ms (main kernel(s), trace kernel, likelihood kernel), batches and launches in rf_kernel_launches
  logical, public :: rf_time_kernels = .false.
  real(c_double), public :: rf_kernel_ms(3) = 0.d0
  integer(c_int64_t), public :: rf_kernel_launches(4) = 0
  ! a GPU group's other ranks give their own queues back once the group's context has their chains (rf_release_gpu)
  logical, public :: rf_group_release_gpu = .true.
  interface
     integer(c_int) function c_getpid() bind(C, name="getpid")
       import :: c_int
     end function c_getpid
  end interface

contains

  subroutine pt_control_batched(verb)
    use params
    use mt19937, only: grnd
    use math, only: gauss
    use prior, only: laplace, log_prior_ratio
    use model
    use likelihood, only: sig, log_likelihood
    use forward, only: rf_ctx, rfgpu_new_context
    use rf_model_check, only: proposal_is_valid, velocity_move_is_valid, interface_move_is_valid, interface_removal_is_valid
    use pt_mcmc
    include "mpif.h"
    logical, intent(in) :: verb
    integer :: nproc, rank, ierr, it, n_tot_iter, n_all, ichain, nb, i, j
    integer :: itype, nlay, cand_k
    logical :: live, yn
    real(8) :: alpha(nlay_max), beta(nlay_max), rho(nlay_max), h(nlay_max)
    real(8) :: lpr, r, del_s, t_cold
    ! Per-chain proposals of the current iteration, one set of arrays per pipeline segment.  The arrays the engine
    ! reads live in pinned host memory (rf_host_alloc) and travel to the GPU by DMA as the proposal step left them --
    ! or, when several ranks share a GPU, in shared memory registered with the GPU by ONE of them (rf_host_alloc_shared):
    ! sg(s) are the arrays of segment s for every rank of the GPU group, rank after rank; my(s) this rank's columns of
    ! them, indexed by chain.  Without sharing the two coincide.
    type seg_arrays
       integer(c_int32_t), pointer :: id(:) => null(), fwd(:) => null(), k(:) => null(), acc(:) => null()
       real(c_double), pointer :: z(:,:) => null(), dvp(:,:) => null(), dvs(:,:) => null(), sig(:,:) => null()
       real(c_double), pointer :: logl(:) => null()
    end type seg_arrays
    type(seg_arrays) :: sg(2), my(2)
    integer(c_int32_t), pointer :: ck(:), cfwd(:), cacc(:)                ! my(iseg) of the segment being worked on
    real(c_double), pointer :: cz(:,:), cdvp(:,:), cdvs(:,:), csig(:,:), clogl(:)
    integer, allocatable :: p_type(:)
    ! column jc of the proposal arrays equals the chain's state except in rows p_lo(jc) .. p_hi(jc) (none: 1 .. 0), the
    ! noise levels except where p_sigd(jc): a proposal touches one interface (a death: those from the removed one on), so
    ! the candidate is built, undone and adopted by copying that range instead of 3 k_max + ntrc numbers each time
    integer, allocatable :: p_lo(:), p_hi(:)
    logical, allocatable :: p_sigd(:)
    logical, allocatable :: p_live(:), p_acc(:)
    real(8), allocatable :: p_lp(:), p_logr(:)
    type(c_ptr) :: pin(32)
    logical :: pin_shared(32)
    integer :: npin
    ! device-side posterior accumulation: what a record call hands over (the module arrays themselves, or -- ranks
    ! sharing a GPU -- every rank's copy of them in shared memory, rank after rank)
    integer(c_int32_t), pointer :: rec_id(:) => null(), rec_k(:) => null()
    real(c_double), pointer :: rec_z(:,:) => null(), rec_dvp(:,:) => null(), rec_dvs(:,:) => null(), rec_sig(:,:) => null()
    real(c_double), pointer :: rec_logl(:) => null(), rec_temps(:) => null()
    logical :: record_now, record_pending
    ! temperature exchange between ranks: RCCL (one GPU per rank) or MPI (ranks sharing a GPU)
    logical :: over_rccl
    real(8) :: tick(6)
    ! the pipeline: segments of the chains, the evaluation in flight for each, the swap proposal drawn but not yet decided
    integer :: nseg, iseg, lo, hi, seg_lo(2), seg_hi(2), seg_n(2)
    integer(c_int32_t) :: seg_ticket(2)
    logical :: seg_busy(2), swap_drawn, launch
    integer(c_int32_t) :: sw_pick(2)
    real(8) :: sw_logu
    ! ranks that share a GPU: the group, its communicator, this rank's place in it, and the context that evaluates
    ! the chains of all of them (the group's first rank owns it)
    logical :: shared, leader, alone_on_gpu
    integer :: g_size, g_rank, g_first, node_comm, n_node
    type(c_ptr) :: gctx
    character(len=40) :: shm_tag

    rf_phase_seconds = 0.d0
    rf_group_seconds = 0.d0
    call mpi_comm_size(MPI_COMM_WORLD, nproc, ierr)
    call mpi_comm_rank(MPI_COMM_WORLD, rank, ierr)
    n_all = nproc * nchains
    n_tot_iter = nburn + niter
    t_cold = 1.d0 + 1.0e-6
    nseg = rf_pipeline_segments
    if (nseg <= 0) nseg = merge(2, 1, nchains >= 2048)
    nseg = max(1, min(nseg, 2, nchains))
    do i = 1, nseg
       seg_lo(i) = (i - 1) * nchains / nseg + 1
       seg_hi(i) = i * nchains / nseg
       seg_n(i) = seg_hi(i) - seg_lo(i) + 1
    end do

    call find_gpu_group()
    n_node = nchains * g_size
    gctx = rf_ctx
    if (shared .and. leader) call create_group_context()

    allocate(p_type(nchains), p_live(nchains), p_acc(nchains), p_lp(nchains), p_logr(nchains))
    allocate(p_lo(nchains), p_hi(nchains), p_sigd(nchains))
    npin = 0
    do i = 1, nseg
       call host_i32(sg(i)%id, seg_n(i) * g_size, "id", i)
       call host_i32(sg(i)%fwd, seg_n(i) * g_size, "fw", i)
       call host_i32(sg(i)%k, seg_n(i) * g_size, "k", i)
       call host_i32(sg(i)%acc, seg_n(i) * g_size, "ac", i)
       call host_f64_2d(sg(i)%z, k_max, seg_n(i) * g_size, "z", i)
       call host_f64_2d(sg(i)%dvp, k_max, seg_n(i) * g_size, "vp", i)
       call host_f64_2d(sg(i)%dvs, k_max, seg_n(i) * g_size, "vs", i)
       call host_f64_2d(sg(i)%sig, ntrc, seg_n(i) * g_size, "sg", i)
       call host_f64_1d(sg(i)%logl, seg_n(i) * g_size, "ll", i)
       ! this rank's columns, indexed by chain
       j = g_rank * seg_n(i)
       my(i)%id(seg_lo(i):) => sg(i)%id(j+1:j+seg_n(i))
       my(i)%fwd(seg_lo(i):) => sg(i)%fwd(j+1:j+seg_n(i))
       my(i)%k(seg_lo(i):) => sg(i)%k(j+1:j+seg_n(i))
       my(i)%acc(seg_lo(i):) => sg(i)%acc(j+1:j+seg_n(i))
       my(i)%z(1:, seg_lo(i):) => sg(i)%z(:, j+1:j+seg_n(i))
       my(i)%dvp(1:, seg_lo(i):) => sg(i)%dvp(:, j+1:j+seg_n(i))
       my(i)%dvs(1:, seg_lo(i):) => sg(i)%dvs(:, j+1:j+seg_n(i))
       my(i)%sig(1:, seg_lo(i):) => sg(i)%sig(:, j+1:j+seg_n(i))
       my(i)%logl(seg_lo(i):) => sg(i)%logl(j+1:j+seg_n(i))
    end do
    if (shared) then
       call host_i32(rec_id, n_node, "ri", 0)
       call host_i32(rec_k, n_node, "rk", 0)
       call host_f64_2d(rec_z, k_max - 1, n_node, "rz", 0)
       call host_f64_2d(rec_dvp, k_max, n_node, "rp", 0)
       call host_f64_2d(rec_dvs, k_max, n_node, "rs", 0)
       call host_f64_2d(rec_sig, ntrc, n_node, "rg", 0)
       call host_f64_1d(rec_logl, n_node, "rl", 0)
       call host_f64_1d(rec_temps, n_node, "rt", 0)
    else
       allocate(rec_id(nchains), rec_temps(nchains))
    end if
    do ichain = 1, nchains
       rec_id(g_rank * nchains + ichain) = g_rank * nchains + ichain - 1
    end do
    if (leader) call setup_device_posterior()
    call open_temperature_exchange()

    ! the proposal columns start as copies of the chains' states (draw_candidate keeps them that way); walker ids
    ! within the context: this rank's chains after those of the group's ranks before it
    do iseg = 1, nseg
       do ichain = seg_lo(iseg), seg_hi(iseg)
          my(iseg)%id(ichain) = g_rank * nchains + ichain - 1
          my(iseg)%k(ichain) = k(ichain)
          my(iseg)%z(:, ichain) = 0.d0
          my(iseg)%z(1:k_max-1, ichain) = z(1:k_max-1, ichain)
          my(iseg)%dvp(:, ichain) = dvp(1:k_max, ichain)
          my(iseg)%dvs(:, ichain) = dvs(1:k_max, ichain)
          my(iseg)%sig(:, ichain) = sig(1:ntrc, ichain)
          my(iseg)%fwd(ichain) = 1
          my(iseg)%acc(ichain) = 1
          p_lo(ichain) = 1
          p_hi(ichain) = 0
          p_sigd(ichain) = .false.
       end do
    end do
    ! The first evaluation of every chain (init_likelihood, on the rank's own context) is its current trace.  With
    ! windowed trace storage (switching drops every stored trace) and in a GPU group (the group's context has seen
    ! none of the chains) the current models are evaluated once more, all at once -- the same kernels on the same
    ! inputs: the log-likelihoods must come back bit for bit.
    ! a rank that has its GPU to itself lets the engine transfer a segment's proposals under the other segment's
    ! kernels (a stream of their own); ranks sharing a GPU do not: one more queue per process and the hardware
    ! scheduler time-slices them (a GPU group's first rank launches alone, but its context is shared work)
    if (leader .and. alone_on_gpu) then
       call rfgpu_check(rf_set_option(gctx, "copy_stream" // c_null_char, 1.0_c_double), "rf_set_option")
    end if
    if (leader .and. rf_windowed_traces) then
       call rfgpu_check(rf_set_option(gctx, "trace_window" // c_null_char, 1.0_c_double), "rf_set_option")
    end if
    if (shared .or. rf_windowed_traces) then
       call group_barrier()
       if (leader) then
          do iseg = 1, nseg
             call rfgpu_check(rf_eval_models(gctx, int(seg_n(iseg) * g_size, c_int32_t), sg(iseg)%id, sg(iseg)%fwd, &
                  & sg(iseg)%k, sg(iseg)%z, int(k_max, c_int32_t), sg(iseg)%dvp, sg(iseg)%dvs, sg(iseg)%sig, &
                  & sg(iseg)%logl, c_null_ptr), "rf_eval_models")
          end do
       end if
       call group_barrier()
       do iseg = 1, nseg
          do ichain = seg_lo(iseg), seg_hi(iseg)
             r = my(iseg)%logl(ichain)
             if (r /= log_likelihood(ichain) .and. (r == r .or. log_likelihood(ichain) == log_likelihood(ichain))) then   ! (NaN = NaN here)
                ! A context picks its kernels from its capacity, and two plans agree to rounding, not bit for bit (the
                ! group's context holds g_size times the chains of the rank's own): inside the parity tolerance the
                ! group's value is the chain's from here on -- every later value comes from the same kernels.
                if (shared .and. abs(r - log_likelihood(ichain)) <= max(1.0d-9, 1.0d-12 * abs(r))) then
                   log_likelihood(ichain) = r
                else
                   write(0,*) "ERROR: pt_control_batched: chain", ichain, " re-evaluated to", r, " not", log_likelihood(ichain)
                   call rfgpu_check(1_c_int, "re-evaluation of the initial models")
                end if
             end if
          end do
       end do
    end if
    if (leader) then
       do iseg = 1, nseg
          call rfgpu_check(rf_commit(gctx, int(seg_n(iseg) * g_size, c_int32_t), sg(iseg)%id, sg(iseg)%acc), "rf_commit")
       end do
    end if

    ! ------------------------------------------------------------------------------------------------------
    ! The loop, software-pipelined over nseg segments of the chains (rf_pipeline_segments; 1 = no overlap).
    ! While segment s of iteration `it` is being evaluated on the GPU the host finishes and re-proposes the
    ! next segment.  Order of events, with S = nseg:
    !   slot (it, s):  wait for + accept segment s of iteration it-1
    !                  [s = S: the posterior records of it-1 are staged, then the swap DECISION of it-1]
    !                  propose segment s of iteration it
    !                  commit segment s of it-1, [s = S: record], hand segment s of it to the engine
    !                  [s = S: the swap DRAWS of iteration it]
    ! Every random draw sits where the reference's sequential loop makes it (chains 1 .. n of an iteration, then
    ! the swap's draws, src/pt_mcmc.f90:488-571) and none depends on a likelihood; a chain is re-proposed only
    ! after its previous proposal has been judged; an iteration's acceptance tests see the temperatures left by
    ! the previous iteration's swap and its swap sees every chain's state after that iteration.  The trajectory
    ! is the reference's.
    ! Ranks that share a GPU (a GPU group) run the same slots in step: each writes its columns of the shared arrays,
    ! the group's first rank makes the engine calls for all of them -- ONE full-size launch per segment instead
    ! of one small launch per rank -- and two barriers of the group per slot order the two: after the wait (every
    ! rank reads its results) and before the engine calls (every rank's flags and proposals are in place).
    ! ------------------------------------------------------------------------------------------------------
    seg_busy = .false.
    swap_drawn = .false.
    record_pending = .false.
    ! a rank whose chains the group's first rank evaluates needs its own context no longer (nothing after the loop
    ! touches the GPU either: output_results is host code)
    if (shared .and. .not. leader .and. rf_group_release_gpu .and. .not. over_rccl) then
       call rfgpu_check(rf_ctx_destroy(rf_ctx), "rf_ctx_destroy")
       rf_ctx = c_null_ptr
       call rfgpu_check(rf_release_gpu(), "rf_release_gpu")
    end if
    if (leader .and. rf_time_kernels) call rfgpu_check(rf_profile_enable(gctx, 1_c_int32_t), "rf_profile_enable")
    call mpi_barrier(MPI_COMM_WORLD, ierr)
    rf_loop_seconds = mpi_wtime()

    do it = 1, n_tot_iter + 1
       if (verb .and. it <= n_tot_iter .and. mod(it, ncorr) == 0) write(*,*) "Iteration #:", it, "/", n_tot_iter
       do iseg = 1, nseg
          lo = seg_lo(iseg)
          hi = seg_hi(iseg)
          ck => my(iseg)%k;  cfwd => my(iseg)%fwd;  cacc => my(iseg)%acc;  clogl => my(iseg)%logl
          cz => my(iseg)%z;  cdvp => my(iseg)%dvp;  cdvs => my(iseg)%dvs;  csig => my(iseg)%sig
          !----------------------------------------------------------------
          ! a. the segment's previous proposals: results, Metropolis-Hastings decisions, state update
          !----------------------------------------------------------------
          tick(1) = mpi_wtime()
          if (it > 1) then
             if (seg_busy(iseg)) then
                if (leader) call rfgpu_check(rf_eval_wait(gctx, seg_ticket(iseg), sg(iseg)%logl, c_null_ptr), "rf_eval_wait")
                tick(6) = mpi_wtime()
                call group_barrier()
                rf_group_seconds(1) = rf_group_seconds(1) + (tick(6) - tick(1))
                rf_group_seconds(2) = rf_group_seconds(2) + (mpi_wtime() - tick(6))
             end if
             tick(2) = mpi_wtime()
             do ichain = lo, hi
                cacc(ichain) = 0
                if (.not. p_live(ichain)) cycle
                del_s = (clogl(ichain) - log_likelihood(ichain)) / temps(ichain) + p_lp(ichain)
                yn = (p_logr(ichain) <= del_s)
                if (yn) then
                   ! the candidate becomes the state: it differs from it in rows p_lo .. p_hi only
                   cacc(ichain) = 1
                   log_likelihood(ichain) = clogl(ichain)
                   k(ichain) = ck(ichain)
                   i = p_lo(ichain)
                   j = p_hi(ichain)
                   if (j >= i) then
                      dvp(i:j, ichain) = cdvp(i:j, ichain)
                      dvs(i:j, ichain) = cdvs(i:j, ichain)
                      j = min(j, k_max - 1)
                      if (j >= i) z(i:j, ichain) = cz(i:j, ichain)
                   end if
                   if (p_sigd(ichain)) sig(1:ntrc, ichain) = csig(1:ntrc, ichain)
                end if
                p_acc(ichain) = yn
             end do
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
                   ! and reads their current traces where the evaluation left them.  The temperatures are
                   ! those BEFORE this iteration's swap (src/pt_mcmc.f90:204); the call itself follows the
                   ! segment's commit below.
                   j = g_rank * nchains
                   rec_temps(j+1:j+nchains) = temps(1:nchains)
                   if (shared) then
                      rec_k(j+1:j+nchains) = k(1:nchains)
                      rec_z(:, j+1:j+nchains) = z(1:k_max-1, 1:nchains)
                      rec_dvp(:, j+1:j+nchains) = dvp(1:k_max, 1:nchains)
                      rec_dvs(:, j+1:j+nchains) = dvs(1:k_max, 1:nchains)
                      rec_sig(:, j+1:j+nchains) = sig(1:ntrc, 1:nchains)
                      rec_logl(j+1:j+nchains) = log_likelihood(1:nchains)
                   end if
                   record_pending = .true.
                end if
                tick(4) = mpi_wtime()
                if (swap_drawn) call decide_temperature_swap()
                swap_drawn = .false.
                tick(5) = mpi_wtime()
                rf_phase_seconds(4) = rf_phase_seconds(4) + (tick(4) - tick(3))
                rf_phase_seconds(5) = rf_phase_seconds(5) + (tick(5) - tick(4))
             end if
          end if

          !----------------------------------------------------------------
          ! b. the segment's proposals of iteration `it`, chain by chain, in the reference's draw order
          !----------------------------------------------------------------
          tick(1) = mpi_wtime()
          nb = 0
          if (it <= n_tot_iter) then
             do ichain = lo, hi
                call draw_candidate(ichain)       ! writes the candidate into column ichain of the segment's arrays
                p_type(ichain) = itype
                p_live(ichain) = live
                p_acc(ichain) = .false.
                cfwd(ichain) = -1                 ! a null proposal: the engine skips the item
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
                cfwd(ichain) = merge(0, 1, itype == itype_sig)   ! noise-level move: the chain's stored trace is re-used
             end do
          end if
          tick(2) = mpi_wtime()
          !----------------------------------------------------------------
          ! c. the engine calls of the slot (the group's first rank, for every rank of the group): commit of the
          !    segment's decisions, [posterior records], format_model + forward + likelihood of its new proposals --
          !    enqueued, collected at this segment's next slot
          !----------------------------------------------------------------
          if (shared) call group_barrier()
          tick(6) = mpi_wtime()
          rf_group_seconds(3) = rf_group_seconds(3) + (tick(6) - tick(2))
          launch = it <= n_tot_iter .and. (shared .or. nb > 0)
          if (leader) then
             if (it > 1 .and. seg_busy(iseg)) then
                call rfgpu_check(rf_commit(gctx, int(seg_n(iseg) * g_size, c_int32_t), sg(iseg)%id, sg(iseg)%acc), "rf_commit")
             end if
             tick(4) = mpi_wtime()
             rf_group_seconds(4) = rf_group_seconds(4) + (tick(4) - tick(6))
             if (record_pending .and. iseg == nseg) call record_posterior()
             tick(5) = mpi_wtime()
             rf_group_seconds(5) = rf_group_seconds(5) + (tick(5) - tick(4))
             if (launch) then
                call rfgpu_check(rf_eval_models_begin(gctx, int(seg_n(iseg) * g_size, c_int32_t), sg(iseg)%id, sg(iseg)%fwd, &
                     & sg(iseg)%k, sg(iseg)%z, int(k_max, c_int32_t), sg(iseg)%dvp, sg(iseg)%dvs, sg(iseg)%sig, &
                     & 0_c_int32_t, seg_ticket(iseg)), "rf_eval_models_begin")
             end if
          end if
          if (iseg == nseg) record_pending = .false.
          seg_busy(iseg) = launch
          tick(3) = mpi_wtime()
          if (leader) rf_group_seconds(6) = rf_group_seconds(6) + (tick(3) - tick(5))
          rf_phase_seconds(1) = rf_phase_seconds(1) + (tick(2) - tick(1))
          rf_phase_seconds(2) = rf_phase_seconds(2) + (tick(3) - tick(2))
          !----------------------------------------------------------------
          ! d. the draws of this iteration's temperature-swap proposal (decided once every chain is through)
          !----------------------------------------------------------------
          if (it <= n_tot_iter .and. iseg == nseg .and. nchains >= 2) then
             call draw_temperature_swap()
             swap_drawn = .true.
             rf_phase_seconds(5) = rf_phase_seconds(5) + (mpi_wtime() - tick(3))
          end if
       end do
    end do

    call mpi_barrier(MPI_COMM_WORLD, ierr)
    rf_loop_seconds = mpi_wtime() - rf_loop_seconds
    if (leader .and. rf_time_kernels) then
       call rfgpu_check(rf_profile_read(gctx, rf_kernel_ms, rf_kernel_launches, 1_c_int32_t), "rf_profile_read")
       call rfgpu_check(rf_profile_enable(gctx, 0_c_int32_t), "rf_profile_enable")
    end if
    if (over_rccl) call rfgpu_check(rf_comm_destroy(rf_ctx), "rf_comm_destroy")
    call fetch_device_posterior()
    if (shared .and. leader) call rfgpu_check(rf_ctx_destroy(gctx), "rf_ctx_destroy")
    call group_barrier()
    do i = 1, npin
       if (pin_shared(i)) then
          call rfgpu_check(rf_host_free_shared(pin(i)), "rf_host_free_shared")
       else
          call rfgpu_check(rf_host_free(pin(i)), "rf_host_free")
       end if
    end do
    if (.not. shared) deallocate(rec_id, rec_temps)
    if (shared) call mpi_comm_free(node_comm, ierr)

  contains

    ! ------------------------------------------------------------------------------------------------
    ! Which ranks drive the same GPU as this one (rf_comm_device_key: host + boot id + PCI address)?  More than one:
    ! with rf_share_gpu they form a GPU group, whose first rank evaluates the chains of all of them on one context;
    ! otherwise (default) every rank keeps launching for itself.
    ! ------------------------------------------------------------------------------------------------
    subroutine find_gpu_group()
      integer(c_int64_t) :: my_key
      integer(c_int64_t), allocatable :: keys(:)
      integer :: ia, token(2)
      integer(c_int) :: pid

      shared = .false.
      leader = .true.
      g_size = 1
      g_rank = 0
      g_first = rank
      node_comm = MPI_COMM_NULL
      alone_on_gpu = .true.
      if (nproc < 2) return
      call rfgpu_check(rf_comm_device_key(rf_ctx, my_key), "rf_comm_device_key")
      allocate(keys(nproc))
      call mpi_allgather(my_key, 1, MPI_INTEGER8, keys, 1, MPI_INTEGER8, MPI_COMM_WORLD, ierr)
      g_size = 0
      g_first = -1
      do ia = 1, nproc
         if (keys(ia) /= my_key) cycle
         if (g_first < 0) g_first = ia - 1
         if (ia - 1 < rank) g_rank = g_rank + 1
         g_size = g_size + 1
      end do
      alone_on_gpu = g_size == 1
      if (.not. rf_share_gpu) then
         g_size = 1
         g_rank = 0
         g_first = rank
      end if
      shared = g_size > 1
      leader = g_rank == 0
      if (.not. shared) return
      call mpi_comm_split(MPI_COMM_WORLD, g_first, rank, node_comm, ierr)
      ! a name for the group's shared-memory blocks that no other job on the node uses
      pid = c_getpid()
      token = [int(pid), int(mod(mpi_wtime() * 1.0d3, 1.0d9))]
      call mpi_bcast(token, 2, MPI_INTEGER4, 0, node_comm, ierr)
      write(shm_tag, '(A,I0,A,I0,A,I0)') "/rfgpu_", token(1), "_", token(2), "_", g_first
      if (rank == 0) write(0,'(A,I0,A)') " GPU groups: ranks that share a GPU hand their proposals to the group's first rank (", &
           & g_size, " ranks on rank 0's GPU)"
    end subroutine find_gpu_group

    subroutine group_barrier()
      if (shared) call mpi_barrier(node_comm, ierr)
    end subroutine group_barrier

    ! the context that holds the chains of every rank of the group: same tables, same pseudo-inverse
    subroutine create_group_context()
      real(c_double), allocatable :: rinv(:)
      call rfgpu_new_context(n_node, gctx)
      allocate(rinv(nsmp * nsmp * ntrc))
      call rfgpu_check(rf_get_r_inv(rf_ctx, rinv), "rf_get_r_inv")
      call rfgpu_check(rf_set_r_inv(gctx, rinv), "rf_set_r_inv")
    end subroutine create_group_context

    ! Host arrays the engine reads or writes: pinned (rf_host_alloc), or -- in a GPU group -- shared by the group's
    ! ranks and registered with the GPU by its first rank (created there, opened by the others after a barrier).
    subroutine host_block(bytes, tag, iseg_tag, handle)
      integer(c_size_t), intent(in) :: bytes
      character(*), intent(in) :: tag
      integer, intent(in) :: iseg_tag
      type(c_ptr), intent(out) :: handle
      character(len=64) :: name

      npin = npin + 1
      pin_shared(npin) = shared
      if (.not. shared) then
         call rfgpu_check(rf_host_alloc(max(bytes, 8_c_size_t), handle), "rf_host_alloc")
      else
         write(name, '(A,A,A,I0)') trim(shm_tag), "_", tag, iseg_tag
         if (leader) call rfgpu_check(rf_host_alloc_shared(trim(name) // c_null_char, max(bytes, 8_c_size_t), 1_c_int32_t, &
              & 1_c_int32_t, handle), "rf_host_alloc_shared")
         call group_barrier()
         if (.not. leader) call rfgpu_check(rf_host_alloc_shared(trim(name) // c_null_char, max(bytes, 8_c_size_t), &
              & 0_c_int32_t, 0_c_int32_t, handle), "rf_host_alloc_shared")
         ! every rank has it mapped: the name can go (nothing is left in /dev/shm if the job dies later)
         call group_barrier()
         if (leader) call rfgpu_check(rf_host_unlink_shared(handle), "rf_host_unlink_shared")
      end if
      pin(npin) = handle
    end subroutine host_block

    subroutine host_i32(a, n, tag, iseg_tag)
      integer(c_int32_t), pointer, intent(out) :: a(:)
      integer, intent(in) :: n, iseg_tag
      character(*), intent(in) :: tag
      type(c_ptr) :: handle
      call host_block(int(4, c_size_t) * int(max(n, 1), c_size_t), tag, iseg_tag, handle)
      call c_f_pointer(handle, a, [n])
    end subroutine host_i32

    subroutine host_f64_1d(a, n, tag, iseg_tag)
      real(c_double), pointer, intent(out) :: a(:)
      integer, intent(in) :: n, iseg_tag
      character(*), intent(in) :: tag
      type(c_ptr) :: handle
      call host_block(int(8, c_size_t) * int(max(n, 1), c_size_t), tag, iseg_tag, handle)
      call c_f_pointer(handle, a, [n])
    end subroutine host_f64_1d

    subroutine host_f64_2d(a, m, n, tag, iseg_tag)
      real(c_double), pointer, intent(out) :: a(:,:)
      integer, intent(in) :: m, n, iseg_tag
      character(*), intent(in) :: tag
      type(c_ptr) :: handle
      call host_block(int(8, c_size_t) * int(max(m * n, 1), c_size_t), tag, iseg_tag, handle)
      call c_f_pointer(handle, a, [m, n])
    end subroutine host_f64_2d

    ! rf_post_record for every rank of the group, each into its own set of accumulators (rf_post_select): a rank ends
    ! the run with exactly the arrays it would have filled alone
    subroutine record_posterior()
      integer :: ig, o
      if (.not. shared) then
         call rfgpu_check(rf_post_record(gctx, int(nchains, c_int32_t), rec_id, k, z, dvp, dvs, sig, &
              & log_likelihood, c_loc(rec_temps)), "rf_post_record")
         return
      end if
      do ig = 0, g_size - 1
         o = ig * nchains + 1
         call rfgpu_check(rf_post_select(gctx, int(ig, c_int32_t)), "rf_post_select")
         call rfgpu_check(rf_post_record(gctx, int(nchains, c_int32_t), rec_id(o:), rec_k(o:), rec_z(:, o:), rec_dvp(:, o:), &
              & rec_dvs(:, o:), rec_sig(:, o:), rec_logl(o:), c_loc(rec_temps(o))), "rf_post_record")
      end do
    end subroutine record_posterior

    ! One chain's trans-dimensional proposal.  Sets (host-associated) itype, cand_*, lpr,
    ! live and -- for live candidates -- the formatted layer stack nlay/alpha/beta/rho/h.
    ! Every grnd()/gauss()/laplace() call sits where the reference's step routine makes it,
    ! so the stream position after this call is the reference's.
    subroutine draw_candidate(jc)
      integer, intent(in) :: jc
      integer :: pick, ulo, uhi
      logical :: ok

      ! the candidate is built in place, in column jc of the segment's proposal arrays
      real(c_double), pointer :: cand_z(:), cand_dvp(:), cand_dvs(:), cand_sig(:)

      cand_z => cz(:, jc);  cand_dvp => cdvp(:, jc);  cand_dvs => cdvs(:, jc);  cand_sig => csig(:, jc)
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
      ! rf_model_check gives it without the sort of three arrays, the densities and the five output arrays -- and,
      ! the chain's current model being valid, a noise-level move needs no look at the model at all (it is unchanged)
      ! and any other move a look at the one or two layers it changes
      if (live) then
         if (itype == itype_sig) then
            continue
         else if (itype == itype_dvs .or. itype == itype_dvp) then
            live = velocity_move_is_valid(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs, pick)
         else if (itype == itype_z) then
            live = interface_move_is_valid(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs, pick, z(pick, jc), .true.)
         else if (itype == itype_birth) then
            live = interface_move_is_valid(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs, cand_k, 0.d0, .false.)
         else if (itype == itype_death) then
            live = interface_removal_is_valid(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs, z(pick, jc))
         else
            live = proposal_is_valid(cand_k, cand_z(1:k_max-1), cand_dvp, cand_dvs)
         end if
      end if
      ck(jc) = cand_k
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
      call rfgpu_check(rf_set_model(gctx, mc), "rf_set_model")
      pc%nbin_z = nbin_z;  pc%nbin_vs = nbin_vs;  pc%nbin_vp = nbin_vp;  pc%nbin_vpvs = nbin_vpvs
      pc%nbin_sig = nbin_sig;  pc%nbin_amp = nbin_amp
      pc%amp_min = amp_min;  pc%amp_max = amp_max;  pc%z_min = z_min
      pc%sig_min = c_loc(t_smin);  pc%sig_max = c_loc(t_smax);  pc%sig_mode = c_loc(t_smode)
      pc%max_models = size(all_likelihood)
      ! one set of accumulators per rank of the group
      if (shared) call rfgpu_check(rf_post_sets(gctx, int(g_size, c_int32_t)), "rf_post_sets")
      call rfgpu_check(rf_post_create(gctx, pc), "rf_post_create")
    end subroutine setup_device_posterior

    ! Device accumulators -> the arrays of module pt_mcmc that output_results reads.  In a GPU group the first rank
    ! reads every rank's set and mails it to its owner (end of the run, once): each rank then holds exactly what it would
    ! have accumulated alone and the reference's output_results reduces / gathers as ever (src/mcmc_out.f90:52-93).
    subroutine fetch_device_posterior()
      integer :: ig, st(MPI_STATUS_SIZE)
      real(8), allocatable :: vp0(:,:), vs0(:,:), al0(:)

      if (.not. shared) then
         call read_selected_set()
         return
      end if
      if (leader) then
         ! (rows beyond a set's models keep what init_pt_mcmc put there: start every set from that state)
         allocate(vp0, source=vp_model)
         allocate(vs0, source=vs_model)
         allocate(al0, source=all_likelihood)
         do ig = g_size - 1, 0, -1
            vp_model = vp0;  vs_model = vs0;  all_likelihood = al0
            call rfgpu_check(rf_post_select(gctx, int(ig, c_int32_t)), "rf_post_select")
            call read_selected_set()
            if (ig == 0) exit
            call mpi_send(nmod, 1, MPI_INTEGER4, ig, 901, node_comm, ierr)
            call mpi_send(nk, size(nk), MPI_INTEGER4, ig, 902, node_comm, ierr)
            call mpi_send(nz, size(nz), MPI_INTEGER4, ig, 903, node_comm, ierr)
            call mpi_send(nsig, size(nsig), MPI_INTEGER4, ig, 904, node_comm, ierr)
            call mpi_send(namp, size(namp), MPI_INTEGER4, ig, 905, node_comm, ierr)
            call mpi_send(nvpz, size(nvpz), MPI_INTEGER4, ig, 906, node_comm, ierr)
            call mpi_send(nvsz, size(nvsz), MPI_INTEGER4, ig, 907, node_comm, ierr)
            call mpi_send(nvpvsz, size(nvpvsz), MPI_INTEGER4, ig, 908, node_comm, ierr)
            call mpi_send(vp_mean, size(vp_mean), MPI_REAL8, ig, 909, node_comm, ierr)
            call mpi_send(vs_mean, size(vs_mean), MPI_REAL8, ig, 910, node_comm, ierr)
            call mpi_send(vpvs_mean, size(vpvs_mean), MPI_REAL8, ig, 911, node_comm, ierr)
            call mpi_send(vp_model, size(vp_model), MPI_REAL8, ig, 912, node_comm, ierr)
            call mpi_send(vs_model, size(vs_model), MPI_REAL8, ig, 913, node_comm, ierr)
            call mpi_send(all_likelihood, size(all_likelihood), MPI_REAL8, ig, 914, node_comm, ierr)
         end do
      else
         call mpi_recv(nmod, 1, MPI_INTEGER4, 0, 901, node_comm, st, ierr)
         call mpi_recv(nk, size(nk), MPI_INTEGER4, 0, 902, node_comm, st, ierr)
         call mpi_recv(nz, size(nz), MPI_INTEGER4, 0, 903, node_comm, st, ierr)
         call mpi_recv(nsig, size(nsig), MPI_INTEGER4, 0, 904, node_comm, st, ierr)
         call mpi_recv(namp, size(namp), MPI_INTEGER4, 0, 905, node_comm, st, ierr)
         call mpi_recv(nvpz, size(nvpz), MPI_INTEGER4, 0, 906, node_comm, st, ierr)
         call mpi_recv(nvsz, size(nvsz), MPI_INTEGER4, 0, 907, node_comm, st, ierr)
         call mpi_recv(nvpvsz, size(nvpvsz), MPI_INTEGER4, 0, 908, node_comm, st, ierr)
         call mpi_recv(vp_mean, size(vp_mean), MPI_REAL8, 0, 909, node_comm, st, ierr)
         call mpi_recv(vs_mean, size(vs_mean), MPI_REAL8, 0, 910, node_comm, st, ierr)
         call mpi_recv(vpvs_mean, size(vpvs_mean), MPI_REAL8, 0, 911, node_comm, st, ierr)
         call mpi_recv(vp_model, size(vp_model), MPI_REAL8, 0, 912, node_comm, st, ierr)
         call mpi_recv(vs_model, size(vs_model), MPI_REAL8, 0, 913, node_comm, st, ierr)
         call mpi_recv(all_likelihood, size(all_likelihood), MPI_REAL8, 0, 914, node_comm, st, ierr)
      end if
    end subroutine fetch_device_posterior

    ! the selected set of accumulators of the group's context -> module pt_mcmc's arrays
    subroutine read_selected_set()
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
      call rfgpu_check(rf_post_read(gctx, pr), "rf_post_read")
      nm = max(1, min(int(t_nmod), size(all_likelihood)))
      allocate(t_nk(k_max), t_nz(nbin_z), t_nsig(nbin_sig, ntrc), t_namp(nbin_amp, nsmp, ntrc))
      allocate(t_nvpz(nbin_z, nbin_vp), t_nvsz(nbin_z, nbin_vs), t_nvpvsz(nbin_z, nbin_vpvs))
      allocate(t_vpm(nbin_z), t_vsm(nbin_z), t_vpvsm(nbin_z), t_vpmod(nbin_z, nm), t_vsmod(nbin_z, nm), t_all(nm))
      pr%nmod = c_loc(t_nmod);  pr%nk = c_loc(t_nk);  pr%nz = c_loc(t_nz);  pr%nsig = c_loc(t_nsig)
      pr%namp = c_loc(t_namp);  pr%nvpz = c_loc(t_nvpz);  pr%nvsz = c_loc(t_nvsz);  pr%nvpvsz = c_loc(t_nvpvsz)
      pr%vp_mean = c_loc(t_vpm);  pr%vs_mean = c_loc(t_vsm);  pr%vpvs_mean = c_loc(t_vpvsm)
      pr%vp_model = c_loc(t_vpmod);  pr%vs_model = c_loc(t_vsmod);  pr%all_likelihood = c_loc(t_all)
      pr%amp_out_of_range = c_loc(t_oor)
      call rfgpu_check(rf_post_read(gctx, pr), "rf_post_read")
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
    end subroutine read_selected_set

  end subroutine pt_control_batched

end module pt_mcmc_batched
