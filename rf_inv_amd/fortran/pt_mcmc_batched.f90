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
  ! ... and, within phases 2 / 3, of the engine calls: 1 rf_eval_wait, 2 rf_commit, 3 rf_post_record, 4 rf_eval_models_begin
  real(8), public :: rf_call_seconds(4) = 0.d0
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
  ! .true.: HIP-event times of the loop's kernels (rf_profile_enable on the evaluating context): totals in
  ! rf_kernel_ms (main kernel(s), trace kernel, likelihood kernel), batches and launches in rf_kernel_launches
  logical, public :: rf_time_kernels = .false.
  real(c_double), public :: rf_kernel_ms(3) = 0.d0
  integer(c_int64_t), public :: rf_kernel_launches(4) = 0

contains

  subroutine pt_control_batched(verb)
    use params
    use mt19937, only: grnd
    use math, only: gauss
    use prior, only: laplace, log_prior_ratio
    use model
    use likelihood, only: sig, log_likelihood
    use forward, only: rf_ctx
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
    ! reads live in pinned host memory (rf_host_alloc) and travel to the GPU by DMA as the proposal step left them:
    ! sg(s) are the arrays of segment s as the engine takes them (1 .. seg_n(s)); my(s) the same memory indexed by chain.
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
    integer :: npin
    ! device-side posterior accumulation: a record call hands over the module arrays themselves, plus these two
    integer(c_int32_t), pointer :: rec_id(:) => null()
    real(c_double), pointer :: rec_temps(:) => null()
    logical :: record_now, record_pending
    ! temperature exchange between ranks: RCCL (one GPU per rank) or MPI (ranks sharing a GPU)
    logical :: over_rccl
    real(8) :: tick(5)
    ! the pipeline: segments of the chains, the evaluation in flight for each, the swap proposal drawn but not yet decided
    integer :: nseg, iseg, lo, hi, seg_lo(2), seg_hi(2), seg_n(2)
    integer(c_int32_t) :: seg_ticket(2)
    logical :: seg_busy(2), swap_drawn, launch
    integer(c_int32_t) :: sw_pick(2)
    real(8) :: sw_logu
    logical :: alone_on_gpu      ! no other rank of this run drives this rank's GPU

    rf_phase_seconds = 0.d0
    rf_call_seconds = 0.d0
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

    call count_ranks_on_my_gpu()

    allocate(p_type(nchains), p_live(nchains), p_acc(nchains), p_lp(nchains), p_logr(nchains))
    allocate(p_lo(nchains), p_hi(nchains), p_sigd(nchains))
    npin = 0
    do i = 1, nseg
       call host_i32(sg(i)%id, seg_n(i))
       call host_i32(sg(i)%fwd, seg_n(i))
       call host_i32(sg(i)%k, seg_n(i))
       call host_i32(sg(i)%acc, seg_n(i))
       call host_f64_2d(sg(i)%z, k_max, seg_n(i))
       call host_f64_2d(sg(i)%dvp, k_max, seg_n(i))
       call host_f64_2d(sg(i)%dvs, k_max, seg_n(i))
       call host_f64_2d(sg(i)%sig, ntrc, seg_n(i))
       call host_f64_1d(sg(i)%logl, seg_n(i))
       ! the same columns, indexed by chain
       j = 0
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
    allocate(rec_id(nchains), rec_temps(nchains))
    do ichain = 1, nchains
       rec_id(ichain) = ichain - 1
    end do
    call setup_device_posterior()
    call open_temperature_exchange()

    ! the proposal columns start as copies of the chains' states (draw_candidate keeps them that way)
    do iseg = 1, nseg
       do ichain = seg_lo(iseg), seg_hi(iseg)
          my(iseg)%id(ichain) = ichain - 1
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
    ! The first evaluation of every chain (init_likelihood) is its current trace.  With windowed trace storage
    ! (switching drops every stored trace) the current models are evaluated once more, all at once -- the same kernels
    ! on the same inputs: the log-likelihoods must come back bit for bit.
    ! A rank that has its GPU to itself lets the engine transfer a segment's proposals under the other segment's
    ! kernels (a stream of their own); ranks sharing a GPU do not: one more queue per process and the hardware
    ! scheduler time-slices them.
    if (alone_on_gpu) then
       call rfgpu_check(rf_set_option(rf_ctx, "copy_stream" // c_null_char, 1.0_c_double), "rf_set_option")
    end if
    if (rf_windowed_traces) then
       call rfgpu_check(rf_set_option(rf_ctx, "trace_window" // c_null_char, 1.0_c_double), "rf_set_option")
       do iseg = 1, nseg
          call rfgpu_check(rf_eval_models(rf_ctx, int(seg_n(iseg), c_int32_t), sg(iseg)%id, sg(iseg)%fwd, &
               & sg(iseg)%k, sg(iseg)%z, int(k_max, c_int32_t), sg(iseg)%dvp, sg(iseg)%dvs, sg(iseg)%sig, &
               & sg(iseg)%logl, c_null_ptr), "rf_eval_models")
       end do
       do iseg = 1, nseg
          do ichain = seg_lo(iseg), seg_hi(iseg)
             r = my(iseg)%logl(ichain)
             if (r /= log_likelihood(ichain) .and. (r == r .or. log_likelihood(ichain) == log_likelihood(ichain))) then   ! (NaN = NaN here)
                write(0,*) "ERROR: pt_control_batched: chain", ichain, " re-evaluated to", r, " not", log_likelihood(ichain)
                call rfgpu_check(1_c_int, "re-evaluation of the initial models")
             end if
          end do
       end do
    end if
    do iseg = 1, nseg
       call rfgpu_check(rf_commit(rf_ctx, int(seg_n(iseg), c_int32_t), sg(iseg)%id, sg(iseg)%acc), "rf_commit")
    end do

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
    ! ------------------------------------------------------------------------------------------------------
    seg_busy = .false.
    swap_drawn = .false.
    record_pending = .false.
    if (rf_time_kernels) call rfgpu_check(rf_profile_enable(rf_ctx, 1_c_int32_t), "rf_profile_enable")
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
                call rfgpu_check(rf_eval_wait(rf_ctx, seg_ticket(iseg), sg(iseg)%logl, c_null_ptr), "rf_eval_wait")
                rf_call_seconds(1) = rf_call_seconds(1) + (mpi_wtime() - tick(1))
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
                   rec_temps(1:nchains) = temps(1:nchains)
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
          ! c. the engine calls of the slot: commit of the segment's decisions, [posterior records], format_model +
          !    forward + likelihood of its new proposals -- enqueued, collected at this segment's next slot
          !----------------------------------------------------------------
          launch = it <= n_tot_iter .and. nb > 0
          if (it > 1 .and. seg_busy(iseg)) then
             call rfgpu_check(rf_commit(rf_ctx, int(seg_n(iseg), c_int32_t), sg(iseg)%id, sg(iseg)%acc), "rf_commit")
          end if
          tick(4) = mpi_wtime()
          rf_call_seconds(2) = rf_call_seconds(2) + (tick(4) - tick(2))
          if (record_pending .and. iseg == nseg) then
             call rfgpu_check(rf_post_record(rf_ctx, int(nchains, c_int32_t), rec_id, k, z, dvp, dvs, sig, &
                  & log_likelihood, c_loc(rec_temps)), "rf_post_record")
          end if
          tick(5) = mpi_wtime()
          rf_call_seconds(3) = rf_call_seconds(3) + (tick(5) - tick(4))
          if (launch) then
             call rfgpu_check(rf_eval_models_begin(rf_ctx, int(seg_n(iseg), c_int32_t), sg(iseg)%id, sg(iseg)%fwd, &
                  & sg(iseg)%k, sg(iseg)%z, int(k_max, c_int32_t), sg(iseg)%dvp, sg(iseg)%dvs, sg(iseg)%sig, &
                  & 0_c_int32_t, seg_ticket(iseg)), "rf_eval_models_begin")
          end if
          if (iseg == nseg) record_pending = .false.
          seg_busy(iseg) = launch
          tick(3) = mpi_wtime()
          rf_call_seconds(4) = rf_call_seconds(4) + (tick(3) - tick(5))
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
    if (rf_time_kernels) then
       call rfgpu_check(rf_profile_read(rf_ctx, rf_kernel_ms, rf_kernel_launches, 1_c_int32_t), "rf_profile_read")
       call rfgpu_check(rf_profile_enable(rf_ctx, 0_c_int32_t), "rf_profile_enable")
    end if
    if (over_rccl) call rfgpu_check(rf_comm_destroy(rf_ctx), "rf_comm_destroy")
    call read_device_posterior()
    do i = 1, npin
       call rfgpu_check(rf_host_free(pin(i)), "rf_host_free")
    end do
    deallocate(rec_id, rec_temps)

  contains

    ! ------------------------------------------------------------------------------------------------
    ! Does another rank of this run drive the same GPU (rf_comm_device_key: host + boot id + PCI address)?
    ! ------------------------------------------------------------------------------------------------
    subroutine count_ranks_on_my_gpu()
      integer(c_int64_t) :: my_key
      integer(c_int64_t), allocatable :: keys(:)

      alone_on_gpu = .true.
      if (nproc < 2) return
      call rfgpu_check(rf_comm_device_key(rf_ctx, my_key), "rf_comm_device_key")
      allocate(keys(nproc))
      call mpi_allgather(my_key, 1, MPI_INTEGER8, keys, 1, MPI_INTEGER8, MPI_COMM_WORLD, ierr)
      alone_on_gpu = count(keys == my_key) == 1
    end subroutine count_ranks_on_my_gpu

    ! Host arrays the engine reads or writes: pinned (rf_host_alloc)
    subroutine host_block(bytes, handle)
      integer(c_size_t), intent(in) :: bytes
      type(c_ptr), intent(out) :: handle
      npin = npin + 1
      call rfgpu_check(rf_host_alloc(max(bytes, 8_c_size_t), handle), "rf_host_alloc")
      pin(npin) = handle
    end subroutine host_block

    subroutine host_i32(a, n)
      integer(c_int32_t), pointer, intent(out) :: a(:)
      integer, intent(in) :: n
      type(c_ptr) :: handle
      call host_block(int(4, c_size_t) * int(max(n, 1), c_size_t), handle)
      call c_f_pointer(handle, a, [n])
    end subroutine host_i32

    subroutine host_f64_1d(a, n)
      real(c_double), pointer, intent(out) :: a(:)
      integer, intent(in) :: n
      type(c_ptr) :: handle
      call host_block(int(8, c_size_t) * int(max(n, 1), c_size_t), handle)
      call c_f_pointer(handle, a, [n])
    end subroutine host_f64_1d

    subroutine host_f64_2d(a, m, n)
      real(c_double), pointer, intent(out) :: a(:,:)
      integer, intent(in) :: m, n
      type(c_ptr) :: handle
      call host_block(int(8, c_size_t) * int(max(m * n, 1), c_size_t), handle)
      call c_f_pointer(handle, a, [m, n])
    end subroutine host_f64_2d

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
      call rfgpu_check(rf_set_model(rf_ctx, mc), "rf_set_model")
      pc%nbin_z = nbin_z;  pc%nbin_vs = nbin_vs;  pc%nbin_vp = nbin_vp;  pc%nbin_vpvs = nbin_vpvs
      pc%nbin_sig = nbin_sig;  pc%nbin_amp = nbin_amp
      pc%amp_min = amp_min;  pc%amp_max = amp_max;  pc%z_min = z_min
      pc%sig_min = c_loc(t_smin);  pc%sig_max = c_loc(t_smax);  pc%sig_mode = c_loc(t_smode)
      pc%max_models = size(all_likelihood)
      call rfgpu_check(rf_post_create(rf_ctx, pc), "rf_post_create")
    end subroutine setup_device_posterior

    ! Device accumulators -> the arrays of module pt_mcmc that output_results reads
    subroutine read_device_posterior()
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
    end subroutine read_device_posterior

  end subroutine pt_control_batched

end module pt_mcmc_batched
