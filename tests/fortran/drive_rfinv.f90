!=======================================================================
! End-to-end drop-in run: the reference's OWN sampler (pt_mcmc.f90, compiled
! unmodified) on top of OUR forward / likelihood modules (GPU).  Follows the
! init order of the reference main program (src/rf_inv.f90:45-98) except for
! init_fftw, which the replaced modules no longer need, and dumps the
! per-iteration mean T=1 log-likelihood (what mcmc_out writes to
! rslt/likelihood, src/mcmc_out.f90:142) and the proposal counters.
!   usage: drive_rfinv params.in n_iter
!=======================================================================
program drive_rfinv
  use params
  use mt19937
  use model
  use likelihood
  use forward
  use pt_mcmc
  implicit none
  include "mpif.h"
  integer :: nproc, rank, ierr, it, u, n_it
  character(clen_max) :: param_file, arg

  call mpi_init(ierr)
  call mpi_comm_size(MPI_COMM_WORLD, nproc, ierr)
  call mpi_comm_rank(MPI_COMM_WORLD, rank, ierr)
  param_file = "params.in"
  if (command_argument_count() > 0) call get_command_argument(1, param_file)
  call get_params(.false., param_file)
  if (command_argument_count() > 1) then
     call get_command_argument(2, arg)
     read(arg, *) n_it
     nburn = 0          ! run exactly n_it iterations, all "sampling"
     niter = n_it
  end if
  call read_obs(.false.)
  iseed = iseed + rank * rank * 10000 + 23 * rank
  call sgrnd(iseed)
  call init_forward(.false.)
  call read_ref_model(.false.)
  call init_model(.false.)
  call init_likelihood(.false.)
  call init_pt_mcmc(.false.)
  call pt_control(.false.)

  u = 79
  open(u, file = "rfinv_dump.txt", status = "unknown")
  write(u, *) nburn + niter, ntype, ncool
  do it = 1, nburn + niter
     write(u, '(es25.17)') likelihood_hist(it) / dble(ncool * nproc)
  end do
  write(u, *) nprop(1:ntype)
  write(u, *) naccept(1:ntype)
  write(u, *) nmod
  close(u)
  call mpi_finalize(ierr)
  write(*,*) "drive_rfinv: ok"
end program drive_rfinv
