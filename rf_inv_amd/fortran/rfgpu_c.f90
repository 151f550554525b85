!=======================================================================
! rfgpu_c -- ISO_C_BINDING view of include/rfgpu.h (librfgpu C ABI).
! Written from scratch for the rf_inv_amd project; mirrors the header
! one-to-one (same names, same argument order).
!=======================================================================
module rfgpu_c
  use iso_c_binding
  implicit none
  public

  ! struct rf_config (include/rfgpu.h)
  type, bind(C) :: rf_config
     integer(c_int32_t) :: nfft, ntrc, nsmp, deconv_mode
     real(c_double)     :: delta, t_start, sdep
     type(c_ptr)        :: rayps, a_gus, ipha
     type(c_ptr)        :: obs
     integer(c_int32_t) :: ldobs
     type(c_ptr)        :: r_inv
     integer(c_int32_t) :: max_walkers, nlay_max, device
  end type rf_config

  ! struct rf_model_config
  type, bind(C) :: rf_model_config
     integer(c_int32_t) :: k_max, vp_mode, nref
     real(c_double)     :: z_max, h_min, z_ref_min, dz_ref
     real(c_double)     :: vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max
     type(c_ptr)        :: vp_ref, vs_ref
  end type rf_model_config

  ! struct rf_post_config
  type, bind(C) :: rf_post_config
     integer(c_int32_t) :: nbin_z, nbin_vs, nbin_vp, nbin_vpvs, nbin_sig, nbin_amp
     real(c_double)     :: amp_min, amp_max, z_min
     type(c_ptr)        :: sig_min, sig_max, sig_mode
     integer(c_int64_t) :: max_models
  end type rf_post_config

  ! struct rf_post_result (c_null_ptr = not wanted)
  type, bind(C) :: rf_post_result
     type(c_ptr) :: nmod
     type(c_ptr) :: nk, nz, nsig, namp, nvpz, nvsz, nvpvsz
     type(c_ptr) :: vp_mean, vs_mean, vpvs_mean
     type(c_ptr) :: vp_model, vs_model, all_likelihood
     type(c_ptr) :: amp_out_of_range
  end type rf_post_result

  integer, parameter :: RF_COMM_ID_BYTES = 128

  interface
     integer(c_int) function rf_ctx_create(cfg, ctx_out) bind(C, name="rf_ctx_create")
       import :: c_int, c_ptr, rf_config
       type(rf_config), intent(in) :: cfg
       type(c_ptr), intent(out) :: ctx_out
     end function rf_ctx_create

     integer(c_int) function rf_ctx_destroy(ctx) bind(C, name="rf_ctx_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function rf_ctx_destroy

     type(c_ptr) function rf_last_error() bind(C, name="rf_last_error")
       import :: c_ptr
     end function rf_last_error

     integer(c_int) function rf_get_flt(ctx, flt) bind(C, name="rf_get_flt")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: flt(*)
     end function rf_get_flt

     integer(c_int) function rf_get_is_ray_common(ctx, flag) bind(C, name="rf_get_is_ray_common")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: flag
     end function rf_get_is_ray_common

     integer(c_int) function rf_get_r_inv(ctx, r_inv) bind(C, name="rf_get_r_inv")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: r_inv(*)
     end function rf_get_r_inv

     integer(c_int) function rf_set_r_inv(ctx, r_inv) bind(C, name="rf_set_r_inv")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: r_inv(*)
     end function rf_set_r_inv

     integer(c_int) function rf_calc_rf(ctx, nlay, alpha, beta, rho, h, rft) bind(C, name="rf_calc_rf")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nlay
       real(c_double), intent(in) :: alpha(*), beta(*), rho(*), h(*)
       real(c_double), intent(out) :: rft(*)
     end function rf_calc_rf

     integer(c_int) function rf_calc_likelihood(ctx, walker, fwd_flag, nlay, alpha, beta, rho, h, &
          & sig, prop_log_likelihood, prop_rft) bind(C, name="rf_calc_likelihood")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: walker, fwd_flag, nlay
       real(c_double), intent(in) :: alpha(*), beta(*), rho(*), h(*), sig(*)
       real(c_double), intent(out) :: prop_log_likelihood
       real(c_double), intent(out) :: prop_rft(*)
     end function rf_calc_likelihood

     integer(c_int) function rf_calc_likelihood_of_trace(ctx, rft, sig, logl) &
          & bind(C, name="rf_calc_likelihood_of_trace")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: rft(*), sig(*)
       real(c_double), intent(out) :: logl
     end function rf_calc_likelihood_of_trace
     integer(c_int) function rf_eval_batch(ctx, nb, walker_ids, fwd_flag, nlay, nlay_pad, layers, sig, logl) &
          & bind(C, name="rf_eval_batch")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nb, nlay_pad
       integer(c_int32_t), intent(in) :: walker_ids(*), fwd_flag(*), nlay(*)
       real(c_double), intent(in) :: layers(*), sig(*)
       real(c_double), intent(out) :: logl(*)
     end function rf_eval_batch

     ! format_model + calc_likelihood for nb proposals (k, z, dVp, dVs) in the sampler's own column-per-chain arrays
     integer(c_int) function rf_eval_models(ctx, nb, walker_ids, fwd_flag, k, z, ldz, dvp, dvs, sig, logl, valid) &
          & bind(C, name="rf_eval_models")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nb, ldz
       integer(c_int32_t), intent(in) :: walker_ids(*), fwd_flag(*), k(*)
       real(c_double), intent(in) :: z(*), dvp(*), dvs(*), sig(*)
       real(c_double), intent(out) :: logl(*)
       type(c_ptr), value :: valid          ! c_null_ptr: not wanted
     end function rf_eval_models

     ! the asynchronous pair: enqueue an evaluation, collect it later (up to 4 in flight, executed in order)
     integer(c_int) function rf_eval_models_begin(ctx, nb, walker_ids, fwd_flag, k, z, ldz, dvp, dvs, sig, &
          & want_valid, ticket) bind(C, name="rf_eval_models_begin")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nb, ldz, want_valid
       integer(c_int32_t), intent(in) :: walker_ids(*), fwd_flag(*), k(*)
       real(c_double), intent(in) :: z(*), dvp(*), dvs(*), sig(*)
       integer(c_int32_t), intent(out) :: ticket
     end function rf_eval_models_begin

     integer(c_int) function rf_eval_wait(ctx, ticket, logl, valid) bind(C, name="rf_eval_wait")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: ticket
       real(c_double), intent(out) :: logl(*)
       type(c_ptr), value :: valid          ! c_null_ptr: not wanted
     end function rf_eval_wait

     integer(c_int) function rf_set_option(ctx, name, value) bind(C, name="rf_set_option")
       import :: c_int, c_ptr, c_char, c_double
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)     ! NUL-terminated
       real(c_double), value :: value
     end function rf_set_option

     ! the reference's two FFTW plans executed on the GPU (rf_inv_amd/fortran/fftw.f90)
     integer(c_int) function rf_fft_c2r(nfft, cx, rx) bind(C, name="rf_fft_c2r")
       import :: c_int, c_int32_t, c_double, c_double_complex
       integer(c_int32_t), value :: nfft
       complex(c_double_complex), intent(in) :: cx(*)
       real(c_double), intent(out) :: rx(*)
     end function rf_fft_c2r

     integer(c_int) function rf_fft_r2c(nfft, rx, cx) bind(C, name="rf_fft_r2c")
       import :: c_int, c_int32_t, c_double, c_double_complex
       integer(c_int32_t), value :: nfft
       real(c_double), intent(in) :: rx(*)
       complex(c_double_complex), intent(inout) :: cx(*)
     end function rf_fft_r2c

     ! pinned host memory (arrays from it travel to the GPU by DMA as they are)
     integer(c_int) function rf_host_alloc(bytes, ptr) bind(C, name="rf_host_alloc")
       import :: c_int, c_ptr, c_size_t
       integer(c_size_t), value :: bytes
       type(c_ptr), intent(out) :: ptr
     end function rf_host_alloc

     integer(c_int) function rf_host_free(ptr) bind(C, name="rf_host_free")
       import :: c_int, c_ptr
       type(c_ptr), value :: ptr
     end function rf_host_free

     integer(c_int) function rf_commit(ctx, nb, walker_ids, accept) bind(C, name="rf_commit")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: nb
       integer(c_int32_t), intent(in) :: walker_ids(*), accept(*)
     end function rf_commit

     integer(c_int) function rf_get_rft(ctx, walker, which, n, out) bind(C, name="rf_get_rft")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: walker, which, n
       real(c_double), intent(out) :: out(*)
     end function rf_get_rft
     integer(c_int) function rf_get_rft_batch(ctx, n, walker_ids, which, nout, out) &
          & bind(C, name="rf_get_rft_batch")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n, which, nout
       integer(c_int32_t), intent(in) :: walker_ids(*)
       real(c_double), intent(out) :: out(*)
     end function rf_get_rft_batch

     integer(c_int) function rf_set_model(ctx, m) bind(C, name="rf_set_model")
       import :: c_int, c_ptr, rf_model_config
       type(c_ptr), value :: ctx
       type(rf_model_config), intent(in) :: m
     end function rf_set_model

     integer(c_int) function rf_post_create(ctx, cfg) bind(C, name="rf_post_create")
       import :: c_int, c_ptr, rf_post_config
       type(c_ptr), value :: ctx
       type(rf_post_config), intent(in) :: cfg
     end function rf_post_create

     integer(c_int) function rf_profile_enable(ctx, on) bind(C, name="rf_profile_enable")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
     end function rf_profile_enable

     integer(c_int) function rf_profile_read(ctx, ms, launches, reset) bind(C, name="rf_profile_read")
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: ms(3)
       integer(c_int64_t), intent(out) :: launches(4)
       integer(c_int32_t), value :: reset
     end function rf_profile_read

     integer(c_int) function rf_post_reset(ctx) bind(C, name="rf_post_reset")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function rf_post_reset

     ! temps = c_null_ptr: no temperature filter
     integer(c_int) function rf_post_record(ctx, n, walker_ids, k, z, dvp, dvs, sig, logl, temps) &
          & bind(C, name="rf_post_record")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: n
       integer(c_int32_t), intent(in) :: walker_ids(*), k(*)
       real(c_double), intent(in) :: z(*), dvp(*), dvs(*), sig(*), logl(*)
       type(c_ptr), value :: temps
     end function rf_post_record

     ! ---- temperature exchange over RCCL (include/rfgpu.h, "multi-GPU") ----
     integer(c_int) function rf_comm_device_key(ctx, device_key) bind(C, name="rf_comm_device_key")
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(out) :: device_key
     end function rf_comm_device_key

     integer(c_int) function rf_comm_probe(ctx, device_key) bind(C, name="rf_comm_probe")
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), intent(out) :: device_key
     end function rf_comm_probe

     integer(c_int) function rf_comm_set_library(path) bind(C, name="rf_comm_set_library")
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: path(*)     ! NUL-terminated
     end function rf_comm_set_library

     integer(c_int) function rf_comm_get_unique_id(id) bind(C, name="rf_comm_get_unique_id")
       import :: c_int, c_int8_t
       integer(c_int8_t), intent(out) :: id(*)
     end function rf_comm_get_unique_id

     integer(c_int) function rf_comm_init(ctx, id, rank, nranks) bind(C, name="rf_comm_init")
       import :: c_int, c_ptr, c_int8_t, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int8_t), intent(in) :: id(*)
       integer(c_int32_t), value :: rank, nranks
     end function rf_comm_init

     integer(c_int) function rf_comm_destroy(ctx) bind(C, name="rf_comm_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function rf_comm_destroy

     integer(c_int) function rf_comm_bcast_i32(ctx, buf, n, root) bind(C, name="rf_comm_bcast_i32")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(inout) :: buf(*)
       integer(c_int32_t), value :: n, root
     end function rf_comm_bcast_i32

     ! options of the context's communicator: "sequential_reduce" 0 | 1
     integer(c_int) function rf_comm_set_option(ctx, name, value) bind(C, name="rf_comm_set_option")
       import :: c_int, c_ptr, c_char, c_double
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: name(*)
       real(c_double), value :: value
     end function rf_comm_set_option

     ! end-of-run merge of the device accumulators (src/mcmc_out.f90:52-93) over the RCCL communicator
     integer(c_int) function rf_comm_post_reduce(ctx, root, nmod_sum) bind(C, name="rf_comm_post_reduce")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: root
       integer(c_int32_t), intent(out) :: nmod_sum
     end function rf_comm_post_reduce

     integer(c_int) function rf_comm_post_gather(ctx, root, nmod_rank, vp_model_all, vs_model_all, all_likelihood_all) &
          & bind(C, name="rf_comm_post_gather")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: root
       integer(c_int32_t), intent(out) :: nmod_rank(*)
       type(c_ptr), value :: vp_model_all, vs_model_all, all_likelihood_all   ! c_loc of the root's arrays, or c_null_ptr
     end function rf_comm_post_gather

     integer(c_int) function rf_pt_swap_exchange(ctx, peer, judge, temp, logl, log_u, new_temp, accepted) &
          & bind(C, name="rf_pt_swap_exchange")
       import :: c_int, c_ptr, c_int32_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: peer, judge
       real(c_double), value :: temp, logl, log_u
       real(c_double), intent(out) :: new_temp
       integer(c_int32_t), intent(out) :: accepted
     end function rf_pt_swap_exchange

     integer(c_int) function rf_post_read(ctx, res) bind(C, name="rf_post_read")
       import :: c_int, c_ptr, rf_post_result
       type(c_ptr), value :: ctx
       type(rf_post_result), intent(in) :: res
     end function rf_post_read
  end interface

contains

  ! Print-finalize-stop, the reference's convention for fatal errors
  ! (e.g. /root/reference/src/likelihood.f90:206-210).
  subroutine rfgpu_check(ierr, where)
    include "mpif.h"
    integer(c_int), intent(in) :: ierr
    character(*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    type(c_ptr) :: p
    integer :: i, ierr2
    logical :: mpi_up
    if (ierr == 0) return
    p = rf_last_error()
    write(0, '(3a)', advance='no') "ERROR: librfgpu ", where, ": "
    if (c_associated(p)) then
       call c_f_pointer(p, msg, [512])
       do i = 1, 512
          if (msg(i) == c_null_char) exit
          write(0, '(a)', advance='no') msg(i)
       end do
    end if
    write(0, *)
    ! (abort, not finalize: the other ranks may be waiting for this one in a barrier or a collective; a host that never
    ! initialised MPI -- the reference's make_syn -- just stops)
    call mpi_initialized(mpi_up, ierr2)
    if (mpi_up) call mpi_abort(MPI_COMM_WORLD, 1, ierr2)
    stop 1
  end subroutine rfgpu_check

end module rfgpu_c
