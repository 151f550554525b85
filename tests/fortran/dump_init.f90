!=======================================================================
! Dumps the state the REFERENCE's own host modules (params, mt19937, model;
! compiled unmodified from /root/reference/src) reach after
!   get_params -> sgrnd(iseed) -> read_ref_model -> init_model,
! plus the next few grnd() values, so that the Python mirrors
! (rf_inv_amd/mt19937.py, mcmc.py: init_model) can be checked bit-for-bit.
!=======================================================================
program dump_init
  use params
  use mt19937
  use model
  implicit none
  character(clen_max) :: param_file
  integer :: ichain, i, u
  real(8) :: r

  param_file = "params.in"
  if (command_argument_count() > 0) call get_command_argument(1, param_file)
  call get_params(.false., param_file)
  call sgrnd(iseed)
  call read_ref_model(.false.)
  call init_model(.false.)
  u = 78
  open(u, file = "init_dump.txt", status = "unknown")
  write(u, *) nchains, k_max
  do ichain = 1, nchains
     write(u, *) k(ichain)
     write(u, '(es25.17)') (z(i, ichain), i = 1, k_max - 1)
     write(u, '(es25.17)') (dvp(i, ichain), i = 1, k_max)
     write(u, '(es25.17)') (dvs(i, ichain), i = 1, k_max)
  end do
  do i = 1, 8
     r = grnd()
     write(u, '(es25.17)') r
  end do
  close(u)
  write(*,*) "dump_init: ok"
end program dump_init
