!     The ONE name the reference takes from FFTW3's Fortran header (src/fftw.f90:31 includes it, :44-45 use it): the
!     planner flag FFTW_ESTIMATE, 64 in FFTW3's public API (1 << 6).  The image has FFTW3's Fortran INTERFACE (Intel
!     MKL's dfftw_* wrappers in /opt/conda/lib) but not its header file, so oracle/Makefile.ref puts this directory
!     on the include path.  A stand-in for a header: builds that use it are supplementary evidence, not the formal
!     pin (DESIGN.md section 5).  Test infrastructure.
      INTEGER FFTW_ESTIMATE
      PARAMETER (FFTW_ESTIMATE=64)
