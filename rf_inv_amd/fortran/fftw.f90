!=======================================================================
! module fftw -- drop-in replacement of RF_INV's src/fftw.f90 without FFTW3.
!
! Same public names and kinds as the reference module (src/fftw.f90:28-48):
!     complex(kind(0d0)), allocatable :: cx(:)      ! (nfft)
!     real(kind(0d0)),    allocatable :: rx(:)      ! (nfft)
!     integer(8) :: ifft, ifft2                     ! plan handles: c2r, r2c
!     subroutine init_fftw()
! so that the reference's main programs, which say `use fftw` and
! `call init_fftw()` (src/rf_inv.f90:31,79; src/make_syn.f90:33,61), compile
! and link unmodified against the GPU drop-in modules.  The forward model no
! longer touches cx / rx (its inverse transform is part of the trace kernels
! of librfgpu); the buffers and handles exist for hosts that execute the
! plans themselves -- src/make_syn.f90:91-95,107-111 filters its noise with
! `call dfftw_execute(ifft2)`, a product with flt, `call dfftw_execute(ifft)`
! -- and the external subroutine below gives them FFTW's meaning on the GPU
! (rf_fft_r2c / rf_fft_c2r, include/rfgpu_ext.h).
!
! A plan handle is an opaque integer(8) in FFTW's Fortran interface; here it
! is a tag that says which transform of the module's buffers is meant.
! Written from scratch for rf_inv_amd.
!=======================================================================
module fftw
  use params, only: nfft
  implicit none

  complex(kind(0d0)), allocatable :: cx(:)
  real(kind(0d0)), allocatable :: rx(:)
  integer(8) :: ifft, ifft2

  integer(8), parameter :: rf_plan_c2r = 7305001_8, rf_plan_r2c = 7305002_8

contains

  subroutine init_fftw()
    if (allocated(cx)) deallocate(cx)
    if (allocated(rx)) deallocate(rx)
    allocate(cx(nfft), rx(nfft))
    cx = (0.d0, 0.d0)
    rx = 0.d0
    ifft = rf_plan_c2r      ! cx(1:nfft/2+1) -> rx(1:nfft), unnormalised (FFTW_BACKWARD c2r)
    ifft2 = rf_plan_r2c     ! rx(1:nfft) -> cx(1:nfft/2+1)
  end subroutine init_fftw

end module fftw

!-----------------------------------------------------------------------
! FFTW's legacy Fortran entry point, for the two plans of module fftw only:
! executes the transform the handle stands for on the module's buffers, on
! the GPU.  Any other handle is an error (there is no FFTW behind it).
!-----------------------------------------------------------------------
subroutine dfftw_execute(plan)
  use iso_c_binding, only: c_int32_t
  use params, only: nfft
  use fftw, only: cx, rx, rf_plan_c2r, rf_plan_r2c
  use rfgpu_c, only: rf_fft_c2r, rf_fft_r2c, rfgpu_check
  implicit none
  integer(8), intent(in) :: plan

  if (.not. allocated(cx)) then
     write(0,*) "ERROR: dfftw_execute before init_fftw"
     call rfgpu_check(1, "dfftw_execute")
  else if (plan == rf_plan_c2r) then
     call rfgpu_check(rf_fft_c2r(int(nfft, c_int32_t), cx, rx), "rf_fft_c2r")
  else if (plan == rf_plan_r2c) then
     call rfgpu_check(rf_fft_r2c(int(nfft, c_int32_t), rx, cx), "rf_fft_r2c")
  else
     write(0,*) "ERROR: dfftw_execute: not a plan of module fftw (ifft, ifft2):", plan
     call rfgpu_check(1, "dfftw_execute")
  end if
end subroutine dfftw_execute
