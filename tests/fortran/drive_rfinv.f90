!=======================================================================
! End-to-end drop-in run: the reference's OWN sampler (pt_mcmc.f90, compiled
! unmodified) on top of OUR forward / likelihood modules (GPU).  Follows the
! init order of the reference main program (src/rf_inv.f90:45-98) except for
! init_fftw, which the replaced modules no longer need, and dumps the
! per-iteration mean T=1 log-likelihood (what mcmc_out writes to
! rslt/likelihood, src/mcmc_out.f90:142) and the proposal counters.
!   usage: drive_rfinv params.in n_burn n_iter mode [out [rccl_library]]  (mode 0: the reference's
!          pt_control, 1: our pt_control_batched, 2: the same without its two-segment pipeline, 4: segments by the chain count (the module's default); a fifth argument makes the reference's own
!          output_results (src/mcmc_out.f90, compiled unmodified) write its result files
!          into params.in's output directory)
!=======================================================================
program drive_rfinv
  use params
  use mt19937
  use model
  use likelihood
  use forward
  use pt_mcmc
  use pt_mcmc_batched
  use mcmc_out
  implicit none
  include "mpif.h"
  integer :: nproc, rank, ierr, it, u, n_it, mode
  real(8) :: t_loop0, t_loop1
  character(clen_max) :: param_file, arg, dump_file

  call mpi_init(ierr)
  call mpi_comm_size(MPI_COMM_WORLD, nproc, ierr)
  call mpi_comm_rank(MPI_COMM_WORLD, rank, ierr)
  param_file = "params.in"
  if (command_argument_count() > 0) call get_command_argument(1, param_file)
  call get_params(.false., param_file)
  mode = 0
  if (command_argument_count() > 2) then
     call get_command_argument(2, arg)
     read(arg, *) nburn
     call get_command_argument(3, arg)
     read(arg, *) niter
  end if
  if (command_argument_count() > 3) then
     call get_command_argument(4, arg)
     read(arg, *) mode
  end if
  if (command_argument_count() > 5) then
     ! sixth argument: an RCCL library for the temperature exchange (the test double), ranks may share a GPU
     call get_command_argument(6, rf_rccl_library)
     rf_exchange_shared_gpu_ok = .true.
  end if
  call read_obs(.false.)
  iseed = iseed + rank * rank * 10000 + 23 * rank
  call sgrnd(iseed)
  call init_forward(.false.)
  call read_ref_model(.false.)
  call init_model(.false.)
  call init_likelihood(.false.)
  call init_pt_mcmc(.false.)
  call mpi_barrier(MPI_COMM_WORLD, ierr)
  t_loop0 = mpi_wtime()
  if (mode == 0) then
     call pt_control(.false.)
  else
     rf_pipeline_segments = 2                    ! mode 1: the two-segment pipeline whatever the chain count
     if (mode == 2) rf_pipeline_segments = 1     ! propose all, evaluate all, judge all (no host / GPU overlap)
     if (mode == 4) rf_pipeline_segments = 0     ! the module's default: by the number of chains
     call get_environment_variable("RFINV_TIME_KERNELS", arg)
     rf_time_kernels = len_trim(arg) > 0
     call pt_control_batched(.false.)
  end if
  call mpi_barrier(MPI_COMM_WORLD, ierr)
  t_loop1 = mpi_wtime()
  ! (tests/tools/sampler_rate*.sh: wall time of the sampler loop alone, all ranks)
  if (rank == 0) write(*,'(A,F12.6,A,I0,A,I0,A,I0)') " drive_rfinv: loop seconds ", t_loop1 - t_loop0, " ranks ", nproc, &
       & " chains_per_rank ", nchains, " iterations ", nburn + niter
  if (rank == 0 .and. mode /= 0) write(*,'(A,F12.6)') " drive_rfinv: batched loop seconds (set-up excluded) ", rf_loop_seconds
  if (rank == 0 .and. mode /= 0) write(*,'(A,5F10.4)') " drive_rfinv: phase seconds (propose, eval, accept+commit, record, swap) ", &
       & rf_phase_seconds
  if (rank == 0 .and. mode /= 0 .and. rf_time_kernels) write(*,'(A,3F12.3,4I9)') " drive_rfinv: kernel ms (main, trace, likelihood), batches and launches ", &
       & rf_kernel_ms, rf_kernel_launches
  if (rank == 0 .and. mode /= 0) write(*,'(A,4F10.4)') " drive_rfinv: engine call seconds (wait, commit, record, begin) ", rf_call_seconds

  u = 79
  if (nproc == 1) then
     dump_file = "rfinv_dump.txt"
  else
     write(dump_file, '(a,i0,a)') "rfinv_dump_", rank, ".txt"
  end if
  open(u, file = trim(dump_file), status = "unknown")
  write(u, *) nburn + niter, ntype, ncool
  do it = 1, nburn + niter
     write(u, '(es25.17)') likelihood_hist(it) / dble(ncool * nproc)
  end do
  write(u, *) nprop(1:ntype)
  write(u, *) naccept(1:ntype)
  write(u, *) nmod
  ! checksums of the posterior histograms (what mcmc_out would write)
  write(u, *) sum(nk), sum(nz), sum(namp), sum(nvpz), sum(nvsz), sum(nvpvsz)
  write(u, *) sum(int(nk, 8) * [(int(it, 8), it = 1, k_max)])
  write(u, '(es25.17)') sum(vp_mean), sum(vs_mean), sum(vpvs_mean), sum(all_likelihood(1:nmod))
  write(u, '(es25.17)') sum(temps), sum(log_likelihood)
  write(u, '(es25.17)') temps(1:nchains)
  close(u)
  if (command_argument_count() > 4) call output_results(nproc, rank, .false.)
  call mpi_finalize(ierr)
  write(*,*) "drive_rfinv: ok"
end program drive_rfinv
