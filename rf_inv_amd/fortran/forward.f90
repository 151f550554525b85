!=======================================================================
! module forward -- drop-in replacement of RF_INV's src/forward.f90.
!
! Same public interface as the reference module (src/forward.f90:28-41):
!     real(8), allocatable, public :: flt(:,:)
!     logical :: is_ray_common
!     subroutine init_forward(verb)
!     subroutine calc_rf(chain_id, nlay, n, ntrc, rayps, alpha, beta, rho, h, rft)
! but every evaluation is delegated to librfgpu (hand-written HIP kernels for
! gfx950) through the C ABI of include/rfgpu.h.  Written from scratch; nothing
! here is taken from the reference implementation.  The host keeps using its
! own `params` module; `fftw` is not needed by this module (a host that still says
! `use fftw` / `call init_fftw()` gets rf_inv_amd/fortran/fftw.f90).
!=======================================================================
module forward
  use iso_c_binding
  use rfgpu_c
  implicit none

  real(8), allocatable, public :: flt(:,:)
  logical :: is_ray_common

  ! the engine context, shared with module likelihood
  type(c_ptr), public :: rf_ctx = c_null_ptr
  ! HIP device ordinal; a host that runs several MPI ranks per node sets it
  ! (e.g. rank modulo GPUs per node) before init_forward.
  integer, public :: rf_device = 0
  ! the tables a context is created from (what init_forward copied out of module params)
  integer(c_int32_t), allocatable, target, save, private :: ipha_c(:)
  real(c_double), allocatable, target, save, private :: rayps_c(:), a_gus_c(:), obs_c(:,:)

  public init_forward, calc_rf

contains

  !---------------------------------------------------------------------
  subroutine init_forward(verb)
    use params, only: nfft, ntrc, rayps, a_gus, ipha, obs, npts_max, nchains
    logical, intent(in) :: verb
    integer(c_int32_t) :: flag
    integer :: nh

    allocate(ipha_c(ntrc), rayps_c(ntrc), a_gus_c(ntrc), obs_c(npts_max, ntrc))
    ipha_c = int(ipha, c_int32_t)
    rayps_c = rayps
    a_gus_c = a_gus
    obs_c = obs

    call rfgpu_new_context(nchains, rf_ctx)

    nh = nfft / 2 + 1
    allocate(flt(nh, ntrc))
    call rfgpu_check(rf_get_flt(rf_ctx, flt), "rf_get_flt")
    call rfgpu_check(rf_get_is_ray_common(rf_ctx, flag), "rf_get_is_ray_common")
    is_ray_common = (flag /= 0)

    if (verb .and. ntrc > 1) then
       write(*,*) "--- check ray parameters ---"
       if (is_ray_common) then
          write(*,*) "Ray geometries are common among traces"
          write(*,*) "-> Single FWD mode"
       else
          write(*,*) "Ray geometries are not common among traces"
          write(*,*) "-> Multiple FWD mode"
       end if
       write(*,*)
    end if
  end subroutine init_forward

  !---------------------------------------------------------------------
  ! An engine context on module params' tables with room for max_walkers chains
  subroutine rfgpu_new_context(max_walkers, ctx)
    use params, only: nfft, ntrc, nsmp, deconv_mode, delta, t_start, sdep, npts_max, k_max
    integer, intent(in) :: max_walkers
    type(c_ptr), intent(out) :: ctx
    type(rf_config) :: cfg

    cfg%nfft = nfft
    cfg%ntrc = ntrc
    cfg%nsmp = nsmp
    cfg%deconv_mode = deconv_mode
    cfg%delta = delta
    cfg%t_start = t_start
    cfg%sdep = sdep
    cfg%rayps = c_loc(rayps_c)
    cfg%a_gus = c_loc(a_gus_c)
    cfg%ipha = c_loc(ipha_c)
    cfg%obs = c_loc(obs_c)
    cfg%ldobs = npts_max
    cfg%r_inv = c_null_ptr          ! library default; init_likelihood may override
    cfg%max_walkers = max_walkers
    cfg%nlay_max = k_max + 2        ! ocean + k_max - 1 interfaces + half-space
    cfg%device = rf_device
    call rfgpu_check(rf_ctx_create(cfg, ctx), "rf_ctx_create")
  end subroutine rfgpu_new_context

  !---------------------------------------------------------------------
  ! n, ntrc and rayps are part of the reference's interface; like every call site of the
  ! reference (src/likelihood.f90:78-79, src/forward_test.f90:63) they must be params' nfft, ntrc
  ! and rayps -- the engine was created from those -- and anything else is refused, not ignored.
  subroutine calc_rf(chain_id, nlay, n, ntrc, rayps, alpha, beta, rho, h, rft)
    use params, only: p_nfft => nfft, p_ntrc => ntrc, p_rayps => rayps
    integer, intent(in) :: nlay, n, ntrc, chain_id
    real(8), intent(in) :: rayps(ntrc)
    real(8), intent(in) :: alpha(nlay), beta(nlay), rho(nlay), h(nlay)
    real(8), intent(out) :: rft(n, ntrc)

    if (n /= p_nfft .or. ntrc /= p_ntrc) then
       write(0,*) "ERROR: calc_rf called with n, ntrc =", n, ntrc, " but the engine was set up for", p_nfft, p_ntrc
       call rfgpu_check(1_c_int, "calc_rf argument check")
    else if (any(rayps(1:ntrc) /= p_rayps(1:ntrc))) then
       write(0,*) "ERROR: calc_rf called with ray parameters other than params' rayps"
       call rfgpu_check(1_c_int, "calc_rf argument check")
    end if
    call rfgpu_check(rf_calc_rf(rf_ctx, int(nlay, c_int32_t), alpha, beta, rho, h, rft), "rf_calc_rf")
  end subroutine calc_rf

end module forward
