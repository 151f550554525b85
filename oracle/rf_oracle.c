/*
 * rf_oracle.c -- CPU restatement of RF_INV's forward + likelihood hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it may be
 * imported / linked / executed only by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg, and there only as the checker or the reported
 * CPU baseline -- never as (a fallback of) the product path.
 *
 * It restates, function by function and in the reference's own evaluation
 * order, the scalar fp64 / complex128 algorithm of
 *     /root/reference/src/forward.f90    (calc_rf, calc_seis, e_inverse,
 *                                         layer_matrix_sol, layer_matrix_liq,
 *                                         water_level_decon, direct_arrival,
 *                                         init_filter)
 *     /root/reference/src/likelihood.f90 (calc_likelihood quadratic form)
 *     /root/reference/src/model.f90      (format_model, vp_to_rho)
 *     /root/reference/src/sort.f90       (quick_sort)
 * Every function cites the reference file:line it follows.  Nothing here is
 * copied from the reference (which is Fortran); it is a re-expression in C.
 *
 * Third-party arithmetic that is NOT in /root/reference:
 *   - FFTW3 (unpinned version, reference Makefile:18): only its c2r plan is on
 *     the path (src/fftw.f90:44, executed at src/forward.f90:172,200).  Its
 *     published definition is the unnormalised inverse real DFT
 *         rx[j] = sum_{k=0}^{n-1} X[k] exp(+2 pi i j k / n),
 *     with X[n-k] = conj(X[k]) supplied implicitly from X[0..n/2] and the
 *     imaginary parts of X[0] and X[n/2] ignored.  rfo_c2r() restates that
 *     definition (radix-2 FFT) and rfo_c2r_naive() is the O(n^2) long-double
 *     sum of the definition used to check it.
 *   - LAPACK dgesvd (src/likelihood.f90:197-205, init only): R^-1 is an INPUT
 *     to this file; oracle/rf_oracle.py builds it with LAPACK dgesvd through
 *     scipy exactly as src/likelihood.f90:168-222 does.
 *
 * PARITY PINNING.
 *   (0) [round 6] THE WHOLE REFERENCE, BUILT ON THE CPU.  The image holds FFTW3's Fortran interface and LAPACK (Intel MKL
 *       under /opt/conda/lib: dfftw_plan_dft_c2r_1d_, dfftw_plan_dft_r2c_1d_, dfftw_execute_, dgesvd_), so all twelve of
 *       the reference's sources -- src/fftw.f90, forward.f90, likelihood.f90 included -- compile unmodified and run without
 *       a GPU and without any product code (oracle/Makefile.ref -> oracle/_ref/cpu_o0, cpu_o2).  oracle/gen_golden.py
 *       freezes their outputs as committed fixtures (tests/golden/ref/): calc_rf on the matrix of SURVEY.md section 8c
 *       (P / S, deconvolution, sea floor, common rays, nfft 256 .. 8192 and odd lengths, 2 .. 31 layers, the DC bin, a NaN
 *       trace), calc_likelihood with fwd_flag true and false on bench.py's own walkers of nine workloads, format_model on
 *       2100 proposals, the npre shifts, R^-1 as init_r_inv forms it, the main program's result files.
 *       tests/test_reference_fixtures.py checks THIS FILE against them in the CPU suite (traces <= 8e-15 of their scale,
 *       logL <= 0.026 of max(1e-9, 1e-12 |logL|), layer stacks / shifts bit for bit); tests/test_reference_live.py runs
 *       fresh random contexts through the same build at test time.  The one thing that build does not take from the
 *       image is FFTW3's header file (src/fftw.f90:31 includes it): oracle/fftw3_include/fftw3.f supplies the single
 *       constant read from it, a stand-in for a header -- which is why these fixtures are supplementary evidence and the
 *       FORMAL pin stays (1) below.  (Round 5 had run the reference's forward / likelihood modules on the product's GPU
 *       transform instead; that build is now the drop-in module's own test, tests/test_reference_forward.py.)
 * and, on the CPU, the reference's own fixtures (tests/test_oracle_kat.py):
 *   (1) sample_syn/true/true.velmod + sample_syn/params.in geometry (land)
 *       -> sample_syn/data/sample_{1,2}.trc, to float32 quantisation
 *       (RMS ~2e-9 / ~5e-9): pins init_filter, e_inverse, layer_matrix_sol,
 *       the matrix chain, the free-surface boundary condition, conj/sign,
 *       direct_arrival, c2r, the P time-shift map and the vertical-max
 *       normalisation (deconv_mode = 0, P phase).
 *   (2) true.velmod row 1 (Vp 5.0 -> rho 2.5347508187769563): pins
 *       vp_to_rho's single-precision literals bit-exactly.
 *   (3) tests/test_mcmc_driver.py: the end-to-end value recorded in SURVEY.md
 *       section 8c(4) from a run of the unmodified reference (rslt/likelihood,
 *       iteration 1 = -1044.33907794324; shipped sample_syn params.in: ocean
 *       layer, 2 traces, 5 chains, seed 12345678, 1 rank).  Driving this oracle
 *       with the reference-order RNG restatement (rf_inv_amd/mcmc.py, itself
 *       checked bit-for-bit against the reference's compiled mt19937/model
 *       modules) reproduces it to 2.7e-11: pins the ocean boundary condition
 *       (layer_matrix_liq), the 2-trace non-common-ray path, R^-1 and the
 *       log-likelihood.  Provenance caveat: that number comes from the
 *       surveyor's probe build (FFT/LAPACK link shims), not from a fixture the
 *       reference ships.
 * Branches with no reference-SHIPPED known answers: S-phase traces, water-level
 * deconvolution, the sea floor beyond (3).  Pin (0) covers them with the
 * reference's own code, on the CPU; independently of the reference's code they
 * are also checked against another FORMULATION (tests/analytic_layered.py,
 * tests/test_analytic_pins.py):
 * the receiver function of a layer stack by the reflectivity method --
 * scattering matrices from numerically solved boundary conditions, Kennett's
 * addition rules, the reverberation operator; no propagator product -- plus
 * the textbook receiver-function processing, with the reference's integer
 * quirks listed one by one.  This file and the HIP path both agree with it to
 * 1e-11 of the trace scale for P and S, with and without deconvolution,
 * 2..29 layers, on models where the water level clips bins; and, with the
 * free surface replaced by the sea floor's boundary conditions solved
 * together with the acoustic waves of the water column, for the ocean-bottom
 * case (layer_matrix_liq and the sea-floor rows, forward.f90:276-287,
 * 424-442) -- which therefore no longer rests on (3) alone.  Also kept: an
 * independent numpy restatement (calc_seis_numpy) and the half-space
 * apparent-angle relations (tests/test_oracle_kat.py).  The likelihood's
 * quadratic form + logL rest on (0) and (3).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* src/forward.f90:33 */
static const double RF_PI = 3.1415926535897931;

typedef struct { double re, im; } cplx;

static inline cplx c_make(double re, double im) { cplx z = {re, im}; return z; }
static inline cplx c_add(cplx a, cplx b) { return c_make(a.re + b.re, a.im + b.im); }
static inline cplx c_sub(cplx a, cplx b) { return c_make(a.re - b.re, a.im - b.im); }
static inline cplx c_neg(cplx a) { return c_make(-a.re, -a.im); }
static inline cplx c_conj(cplx a) { return c_make(a.re, -a.im); }
/* Fortran complex multiply: (ac - bd, ad + bc), no NaN recovery. */
static inline cplx c_mul(cplx a, cplx b)
{
    return c_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static inline cplx c_scale(cplx a, double s) { return c_make(a.re * s, a.im * s); }
/* Fortran complex divide with range reduction (Smith), the form gfortran and
 * flang emit for complex(8) a / b. */
static inline cplx c_div(cplx a, cplx b)
{
    double r, d;
    if (fabs(b.re) >= fabs(b.im)) {
        r = b.im / b.re;
        d = b.re + b.im * r;
        return c_make((a.re + a.im * r) / d, (a.im - a.re * r) / d);
    }
    r = b.re / b.im;
    d = b.im + b.re * r;
    return c_make((a.re * r + a.im) / d, (a.im * r - a.re) / d);
}

/* 4x4 complex matmul, c = a * b, column-major a(i,k) = a[i + 4k]
 * (Fortran matmul at src/forward.f90:262,264: sum over k = 1..4 in order). */
static void mat4_mul(const cplx *a, const cplx *b, cplx *c)
{
    cplx t[16];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            cplx s = c_make(0.0, 0.0);
            for (int k = 0; k < 4; ++k)
                s = c_add(s, c_mul(a[i + 4 * k], b[k + 4 * j]));
            t[i + 4 * j] = s;
        }
    memcpy(c, t, sizeof t);
}

#define M4(m, i, j) ((m)[((i) - 1) + 4 * ((j) - 1)])

/* ------------------------------------------------------------------ */
/* src/forward.f90:95-119  init_filter                                */
/* flt is (nh, ntrc) column-major.                                    */
void rfo_init_filter(int nfft, int ntrc, double delta, const double *a_gus,
                     double *flt)
{
    int nh = nfft / 2 + 1;
    double df = 1.0 / (delta * nfft);                    /* :103 */
    for (int itrc = 0; itrc < ntrc; ++itrc) {
        double fac_norm = nfft * a_gus[itrc] * delta / sqrt(RF_PI); /* :108 */
        for (int i = 1; i <= nh; ++i) {
            double omega = (i - 1) * 2.0 * RF_PI * df;   /* :110 */
            double q = omega / (2.0 * a_gus[itrc]);
            flt[(i - 1) + (size_t)nh * itrc] = exp(-(q * q)) / fac_norm; /* :112-113 */
        }
    }
}

/* src/forward.f90:350-380  e_inverse (Aki & Richards eq. 5.71) */
static void e_inverse(double omega, double rho, double alpha, double beta,
                      double p, cplx *e_inv)
{
    for (int i = 0; i < 16; ++i) e_inv[i] = c_make(0.0, 0.0);
    double eta = sqrt(1.0 / (beta * beta) - p * p);      /* :358 */
    double xi = sqrt(1.0 / (alpha * alpha) - p * p);     /* :359 */
    double bp = 1.0 - 2.0 * beta * beta * p * p;         /* :360 */

    M4(e_inv, 1, 1) = c_make(beta * beta * p / alpha, 0.0);               /* :362 */
    M4(e_inv, 1, 2) = c_make(bp / (2.0 * alpha * xi), 0.0);               /* :363 */
    M4(e_inv, 1, 3) = c_make(0.0, -p / (2.0 * omega * rho * alpha * xi)); /* :364 */
    M4(e_inv, 1, 4) = c_make(0.0, -1.0 / (2.0 * omega * rho * alpha));    /* :365 */
    M4(e_inv, 2, 1) = c_make(bp / (2.0 * beta * eta), 0.0);               /* :366 */
    M4(e_inv, 2, 2) = c_make(-beta * p, 0.0);                             /* :367 */
    M4(e_inv, 2, 3) = c_make(0.0, -1.0 / (2.0 * omega * rho * beta));     /* :368 */
    M4(e_inv, 2, 4) = c_make(0.0, p / (2.0 * omega * rho * beta * eta));  /* :369 */
    M4(e_inv, 3, 1) = M4(e_inv, 1, 1);                                    /* :370 */
    M4(e_inv, 3, 2) = c_neg(M4(e_inv, 1, 2));
    M4(e_inv, 3, 3) = c_neg(M4(e_inv, 1, 3));
    M4(e_inv, 3, 4) = M4(e_inv, 1, 4);
    M4(e_inv, 4, 1) = M4(e_inv, 2, 1);
    M4(e_inv, 4, 2) = c_neg(M4(e_inv, 2, 2));
    M4(e_inv, 4, 3) = c_neg(M4(e_inv, 2, 3));
    M4(e_inv, 4, 4) = M4(e_inv, 2, 4);                                    /* :377 */
}

/* src/forward.f90:385-421  layer_matrix_sol (Aki & Richards Box 9.1 eq. 3) */
static void layer_matrix_sol(double omega, double rho, double alpha,
                             double beta, double p, double z, cplx *pm)
{
    double beta2 = beta * beta;                          /* :392 */
    double p2 = p * p;
    double bp = 1.0 - 2.0 * beta2 * p2;
    double eta = sqrt(1.0 / beta2 - p2);                 /* :395 */
    double xi = sqrt(1.0 / (alpha * alpha) - p2);        /* :396 */
    double cos_xi = cos(omega * xi * z);                 /* :397 */
    double cos_eta = cos(omega * eta * z);
    double sin_xi = sin(omega * xi * z);
    double sin_eta = sin(omega * eta * z);               /* :400 */

    M4(pm, 1, 1) = c_make(2.0 * beta2 * p2 * cos_xi + bp * cos_eta, 0.0);              /* :403 */
    M4(pm, 2, 1) = c_make(0.0, p * (2.0 * beta2 * xi * sin_xi - bp / eta * sin_eta)); /* :404 */
    M4(pm, 3, 1) = c_make(omega * rho * (-4.0 * beta2 * beta2 * p2 * xi * sin_xi
                                         - bp * bp / eta * sin_eta), 0.0);             /* :405 */
    M4(pm, 4, 1) = c_make(0.0, 2.0 * omega * beta2 * rho * p * bp * (cos_xi - cos_eta)); /* :406 */
    M4(pm, 1, 2) = c_make(0.0, p * (bp / xi * sin_xi - 2.0 * beta2 * eta * sin_eta)); /* :407 */
    M4(pm, 2, 2) = c_make(bp * cos_xi + 2.0 * beta2 * p2 * cos_eta, 0.0);              /* :408 */
    M4(pm, 3, 2) = M4(pm, 4, 1);                                                       /* :409 */
    M4(pm, 4, 2) = c_make(-omega * rho * (bp * bp / xi * sin_xi
                                          + 4.0 * beta2 * beta2 * p2 * eta * sin_eta), 0.0); /* :410 */
    M4(pm, 1, 3) = c_make((p2 / xi * sin_xi + eta * sin_eta) / (omega * rho), 0.0);    /* :411 */
    M4(pm, 2, 3) = c_make(0.0, p * (-cos_xi + cos_eta) / (omega * rho));               /* :412 */
    M4(pm, 3, 3) = M4(pm, 1, 1);
    M4(pm, 4, 3) = M4(pm, 1, 2);
    M4(pm, 1, 4) = M4(pm, 2, 3);
    M4(pm, 2, 4) = c_make((xi * sin_xi + p2 / eta * sin_eta) / (omega * rho), 0.0);    /* :416 */
    M4(pm, 3, 4) = M4(pm, 2, 1);
    M4(pm, 4, 4) = M4(pm, 2, 2);                                                       /* :418 */
}

/* src/forward.f90:424-442  layer_matrix_liq; returns (1,1) and (2,1) only,
 * the two entries the caller consumes (src/forward.f90:278-285). */
static void layer_matrix_liq(double omega, double rho, double alpha, double p,
                             double z, double *lq11, double *lq21)
{
    double xi = sqrt(1.0 / (alpha * alpha) - p * p);     /* :431 */
    double cos_xi = cos(omega * xi * z);
    double sin_xi = sin(omega * xi * z);
    double g = rho * omega / xi;                         /* :434 */
    *lq11 = cos_xi;                                      /* :436 */
    *lq21 = -g * sin_xi;                                 /* :438 */
}

/* src/forward.f90:212-344  calc_seis.  ur/uz receive bins 1..nh (0-based
 * 0..nh-1).  Optional dumps for stage-level tests: sl_dump (nh*16 cplx). */
void rfo_calc_seis(int nlay, int npts, double delta, double rayp, int ipha,
                   const double *alpha, const double *beta, const double *rho,
                   const double *h, cplx *ur_freq, cplx *uz_freq)
{
    int sea_flag = beta[0] < 0.0;                        /* :229-233 */
    int nhalf = npts / 2 + 1;
    int ilay0 = sea_flag ? 2 : 1;                        /* :235-239 */
    double domg = 2.0 * RF_PI / (npts * delta);          /* :241 */

    for (int iomg = 1; iomg <= nhalf; ++iomg) {          /* :244 */
        double omg = (double)(iomg - 1) * domg;          /* :245 */
        if (iomg == 1) omg = (double)1.0e-5f;            /* :246-248 single literal */
        cplx e_inv[16], p_prod[16], p_mat[16], sl[16];
        e_inverse(omg, rho[nlay - 1], alpha[nlay - 1], beta[nlay - 1], rayp, e_inv); /* :250 */
        for (int i = 0; i < 16; ++i) p_prod[i] = c_make(0.0, 0.0);                   /* :255 */
        for (int j = 1; j <= 4; ++j) M4(p_prod, j, j) = c_make(1.0, 0.0);
        for (int ilay = ilay0; ilay <= nlay - 1; ++ilay) {                           /* :259 */
            layer_matrix_sol(omg, rho[ilay - 1], alpha[ilay - 1], beta[ilay - 1],
                             rayp, h[ilay - 1], p_mat);
            mat4_mul(p_mat, p_prod, p_prod);                                         /* :262 */
        }
        mat4_mul(e_inv, p_prod, sl);                                                 /* :264 */

        cplx ur, uz;
        if (!sea_flag) {                                                             /* :267 */
            cplx denom = c_sub(c_mul(M4(sl, 3, 1), M4(sl, 4, 2)),
                               c_mul(M4(sl, 3, 2), M4(sl, 4, 1)));                   /* :268 */
            if (ipha >= 0) {
                ur = c_div(M4(sl, 4, 2), denom);                                     /* :270 */
                uz = c_div(c_neg(M4(sl, 4, 1)), denom);                              /* :271 */
            } else {
                ur = c_div(c_neg(M4(sl, 3, 2)), denom);                              /* :273 */
                uz = c_div(M4(sl, 3, 1), denom);                                     /* :274 */
            }
        } else {
            double l11, l21;
            layer_matrix_liq(omg, rho[0], alpha[0], rayp, h[0], &l11, &l21);         /* :277 */
            cplx lq11 = c_make(l11, 0.0), lq21 = c_make(l21, 0.0);
            cplx a = c_add(c_mul(M4(sl, 4, 2), lq11), c_mul(M4(sl, 4, 4), lq21));    /* :278 */
            cplx b = c_add(c_mul(M4(sl, 3, 2), lq11), c_mul(M4(sl, 3, 4), lq21));    /* :279 */
            cplx d1 = c_sub(c_mul(a, M4(sl, 3, 1)), c_mul(b, M4(sl, 4, 1)));         /* a*sl31 - b*sl41 */
            cplx d2 = c_sub(c_mul(b, M4(sl, 4, 1)), c_mul(a, M4(sl, 3, 1)));         /* b*sl41 - a*sl31 */
            if (ipha >= 0) {
                ur = c_div(a, d1);                                                   /* :281 */
                uz = c_div(c_mul(lq11, M4(sl, 4, 1)), d2);                           /* :282 */
            } else {
                ur = c_div(c_neg(b), d1);                                            /* :284 */
                uz = c_div(c_mul(c_neg(lq11), M4(sl, 3, 1)), d2);                    /* :285 */
            }
        }
        ur_freq[iomg - 1] = ur;
        uz_freq[iomg - 1] = uz;
    }
}

/* src/forward.f90:447-470  water_level_decon: z = y * conj(x) / max(|x|^2, wl) */
static void water_level_decon(const cplx *y, const cplx *x, cplx *z, int n,
                              double pcnt)
{
    double *amp = (double *)malloc(sizeof(double) * (size_t)n);
    double mx = -HUGE_VAL;
    for (int i = 0; i < n; ++i) {
        amp[i] = c_mul(x[i], c_conj(x[i])).re;           /* :458 */
        if (amp[i] > mx) mx = amp[i];
    }
    double wlvl = pcnt * mx;                             /* :460 */
    for (int i = 0; i < n; ++i) {
        double d = amp[i] > wlvl ? amp[i] : wlvl;        /* max(amp, wlvl) :464 */
        cplx num = c_mul(y[i], c_conj(x[i]));
        z[i] = c_make(num.re / d, num.im / d);
    }
    free(amp);
}

/* src/forward.f90:474-519  direct_arrival (sequential sum feeds nint) */
double rfo_direct_arrival(int nlay, const double *h, const double *v,
                          double rayp, double sdep)
{
    double t = 0.0;
    int i0 = (sdep > 0.0) ? 2 : 1;                       /* :484-488 keyed on sdep */
    for (int i = i0; i <= nlay - 1; ++i)                 /* :489-491 */
        t = t + h[i - 1] * sqrt(1.0 / (v[i - 1] * v[i - 1]) - rayp * rayp);
    return t;
}

/* ------------------------------------------------------------------ */
/* FFTW c2r (src/fftw.f90:44, src/forward.f90:172,200): see file header. */
void rfo_c2r_naive(int n, const cplx *cx, double *rx)
{
    int nh = n / 2 + 1;
    const long double tw = 2.0L * 3.14159265358979323846264338327950288L / n;
    for (int j = 0; j < n; ++j) {
        long double s = cx[0].re;
        for (int k = 1; k < nh; ++k) {
            long long m = ((long long)j * k) % n;
            long double c = cosl(tw * m), sn = sinl(tw * m);
            if (2 * k == n)
                s += (long double)cx[k].re * c;
            else
                s += 2.0L * ((long double)cx[k].re * c - (long double)cx[k].im * sn);
        }
        rx[j] = (double)s;
    }
}

typedef struct { int n; cplx *tw; int *rev; long double *ct, *st; } fft_plan;

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static void fft_plan_init(fft_plan *pl, int n)
{
    pl->n = n;
    pl->ct = pl->st = NULL;
    if (!is_pow2(n)) {
        /* any other length: the table cos / sin(2 pi m / n), m = 0 .. n-1, in long double, for the direct sum of the
         * definition (c2r_direct) -- the same values rfo_c2r_naive evaluates inside its loops */
        const long double w = 2.0L * 3.14159265358979323846264338327950288L / n;
        pl->ct = (long double *)malloc(sizeof(long double) * (size_t)n);
        pl->st = (long double *)malloc(sizeof(long double) * (size_t)n);
        for (int m = 0; m < n; ++m) { pl->ct[m] = cosl(w * m); pl->st[m] = sinl(w * m); }
    }
    pl->tw = (cplx *)malloc(sizeof(cplx) * (size_t)(n / 2 > 0 ? n / 2 : 1));
    pl->rev = (int *)malloc(sizeof(int) * (size_t)n);
    for (int k = 0; k < n / 2; ++k) {
        /* e^{+2 pi i k / n}; quadrant-exact evaluation for accuracy */
        long double a = 2.0L * 3.14159265358979323846264338327950288L * k / n;
        pl->tw[k] = c_make((double)cosl(a), (double)sinl(a));
    }
    int lg = 0;
    while ((1 << lg) < n) ++lg;
    for (int i = 0; i < n; ++i) {
        int r = 0;
        for (int b = 0; b < lg; ++b)
            if (i & (1 << b)) r |= 1 << (lg - 1 - b);
        pl->rev[i] = r;
    }
}
static void fft_plan_free(fft_plan *pl) { free(pl->tw); free(pl->rev); free(pl->ct); free(pl->st); }

/* rfo_c2r_naive with the plan's table: the O(n^2) long-double sum of the definition of FFTW's c2r */
static void c2r_direct(const fft_plan *pl, const cplx *cx, double *rx)
{
    int n = pl->n, nh = n / 2 + 1;
    for (int j = 0; j < n; ++j) {
        long double s = cx[0].re;
        int m = 0;
        for (int k = 1; k < nh; ++k) {
            m += j;
            if (m >= n) m -= n;                    /* (j * k) mod n */
            if (2 * k == n)
                s += (long double)cx[k].re * pl->ct[m];
            else
                s += 2.0L * ((long double)cx[k].re * pl->ct[m] - (long double)cx[k].im * pl->st[m]);
        }
        rx[j] = (double)s;
    }
}

/* in-place radix-2 DIT, sign +, unnormalised; n must be a power of two */
static void fft_pow2_inverse(const fft_plan *pl, cplx *a)
{
    int n = pl->n;
    for (int i = 0; i < n; ++i) {
        int r = pl->rev[i];
        if (r > i) { cplx t = a[i]; a[i] = a[r]; a[r] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int s = 0; s < n; s += len)
            for (int k = 0; k < half; ++k) {
                cplx w = pl->tw[k * step];
                cplx u = a[s + k], v = c_mul(a[s + k + half], w);
                a[s + k] = c_add(u, v);
                a[s + k + half] = c_sub(u, v);
            }
    }
}

/* c2r through the full Hermitian-extended complex transform */
static void c2r_exec(const fft_plan *pl, const cplx *cx, double *rx, cplx *work)
{
    int n = pl->n, nh = n / 2 + 1;
    if (!is_pow2(n)) { c2r_direct(pl, cx, rx); return; }
    work[0] = c_make(cx[0].re, 0.0);
    for (int k = 1; k < nh; ++k) {
        if (2 * k == n) { work[k] = c_make(cx[k].re, 0.0); continue; }
        work[k] = cx[k];
        work[n - k] = c_conj(cx[k]);
    }
    fft_pow2_inverse(pl, work);
    for (int j = 0; j < n; ++j) rx[j] = work[j].re;
}

void rfo_c2r(int n, const cplx *cx, double *rx)
{
    fft_plan pl;
    fft_plan_init(&pl, n);
    cplx *work = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    c2r_exec(&pl, cx, rx, work);
    free(work);
    fft_plan_free(&pl);
}

/* ------------------------------------------------------------------ */
/* Fortran nint: round half away from zero */
static inline int f_nint(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : -floor(0.5 - x)); }

typedef struct {
    int nfft, ntrc, nsmp, deconv_mode;
    double delta, t_start, sdep;
    const double *rayps;   /* ntrc */
    const double *a_gus;   /* ntrc */
    const int *ipha;       /* ntrc : +1 P, -1 S */
} rfo_cfg;

static int rays_common(const rfo_cfg *c)
{
    /* src/forward.f90:59-91 check_ray */
    int common = 1;
    for (int i = 1; i < c->ntrc; ++i)
        if (c->rayps[i] != c->rayps[0] || c->ipha[i] != c->ipha[0]) common = 0;
    return common;
}

/* scratch of one calc_rf evaluation (the reference keeps these on the stack / in module fftw);
 * the batched driver allocates one per thread instead of one per evaluation */
typedef struct {
    cplx *freq_r, *freq_v, *rff, *cx, *work;
    double *rx, *rft, *misfits, *phi1;
} rfo_scratch;

static void scratch_init(rfo_scratch *w, int n, int ntrc, int nsmp)
{
    int nh = n / 2 + 1;
    w->freq_r = (cplx *)malloc(sizeof(cplx) * (size_t)nh);
    w->freq_v = (cplx *)malloc(sizeof(cplx) * (size_t)nh);
    w->rff = (cplx *)malloc(sizeof(cplx) * (size_t)nh);
    w->cx = (cplx *)malloc(sizeof(cplx) * (size_t)nh);
    w->work = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    w->rx = (double *)malloc(sizeof(double) * (size_t)n);
    w->rft = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(ntrc > 0 ? ntrc : 1));
    w->misfits = (double *)malloc(sizeof(double) * (size_t)(nsmp > 0 ? nsmp : 1));
    w->phi1 = (double *)malloc(sizeof(double) * (size_t)(nsmp > 0 ? nsmp : 1));
}

static void scratch_free(rfo_scratch *w)
{
    free(w->freq_r); free(w->freq_v); free(w->rff); free(w->cx); free(w->work);
    free(w->rx); free(w->rft); free(w->misfits); free(w->phi1);
}

/* src/forward.f90:123-208  calc_rf.  rft is (nfft, ntrc) column-major.
 * flt (nh, ntrc) from rfo_init_filter.  Optional outputs (may be NULL):
 * npre_out[ntrc] (the integer shift), spec_out[(2*nh)*ntrc] cplx = rff then
 * freq_v per trace (stage-level checks). */
static void calc_rf_impl(const rfo_cfg *c, const fft_plan *pl, const double *flt,
                         int nlay, const double *alpha, const double *beta,
                         const double *rho, const double *h, double *rft,
                         int *npre_out, cplx *spec_out, rfo_scratch *w, double *kappa_out)
{
    int n = c->nfft, nh = n / 2 + 1;
    if (kappa_out) *kappa_out = 1.0;
    cplx *freq_r = w->freq_r, *freq_v = w->freq_v, *rff = w->rff, *cx = w->cx, *work = w->work;
    double *rx = w->rx;
    int common = rays_common(c);
    double tp = 0.0;

    for (int itrc = 0; itrc < c->ntrc; ++itrc) {                           /* :140 */
        int ipha = c->ipha[itrc];
        if (itrc == 0 || !common) {                                        /* :141 */
            rfo_calc_seis(nlay, n, c->delta, c->rayps[itrc], ipha, alpha, beta,
                          rho, h, freq_r, freq_v);                         /* :142 */
            for (int i = 0; i < nh; ++i) {
                freq_r[i] = c_conj(freq_r[i]);                             /* :145 */
                freq_v[i] = c_neg(c_conj(freq_v[i]));                      /* :146 */
            }
            if (c->deconv_mode == 1 && ipha == 1) {                        /* :148 */
                water_level_decon(freq_r, freq_v, rff, nh, 0.001);
                tp = 0.0;
            } else if (c->deconv_mode == 1 && ipha == -1) {                /* :151 */
                water_level_decon(freq_v, freq_r, rff, nh, 0.001);
                tp = 0.0;
            } else if (ipha == 1) {                                        /* :155 */
                memcpy(rff, freq_r, sizeof(cplx) * (size_t)nh);
                tp = rfo_direct_arrival(nlay, h, alpha, c->rayps[itrc], c->sdep);
            } else {                                                       /* :159 */
                memcpy(rff, freq_v, sizeof(cplx) * (size_t)nh);
                tp = rfo_direct_arrival(nlay, h, beta, c->rayps[itrc], c->sdep);
            }
        }
        if (spec_out) {
            memcpy(spec_out + (size_t)(2 * itrc) * nh, rff, sizeof(cplx) * (size_t)nh);
            memcpy(spec_out + (size_t)(2 * itrc + 1) * nh, freq_v, sizeof(cplx) * (size_t)nh);
        }
        const double *f = flt + (size_t)nh * itrc;
        for (int i = 0; i < nh; ++i) cx[i] = c_scale(rff[i], f[i]);        /* :168 */
        c2r_exec(pl, cx, rx, work);                                        /* :172 */

        double *out = rft + (size_t)n * itrc;
        int npre;
        if (ipha == 1) {                                                   /* :176 */
            npre = f_nint((-c->t_start - tp) / c->delta);                  /* :177 */
            for (int i = 1; i <= n; ++i) {
                int j = (n - npre + i) % n;                                /* :179 Fortran mod */
                if (j == 0) j = n;
                out[i - 1] = rx[j - 1];
            }
        } else {                                                           /* :185 */
            npre = f_nint((-c->t_start + tp) / c->delta);                  /* :186 */
            for (int i = 1; i <= n; ++i) {
                int j = (n + npre - i + 1) % n;                            /* :188 */
                if (j == 0) j = n;
                out[i - 1] = -rx[j - 1];
            }
        }
        if (npre_out) npre_out[itrc] = npre;

        if (c->deconv_mode == 0) {                                         /* :197 */
            for (int i = 0; i < nh; ++i) cx[i] = c_scale(freq_v[i], f[i]);
            c2r_exec(pl, cx, rx, work);
            double fac_norm = rx[0];
            for (int i = 1; i < n; ++i) fac_norm = rx[i] > fac_norm ? rx[i] : fac_norm; /* maxval :201 */
            if (kappa_out) {
                /* diagnostic of the tests' conditioning rule (not part of the reference): how much of the
                 * vertical trace's scale cancels in the SIGNED maximum it is divided by */
                double amax = 0.0;
                for (int i = 0; i < n; ++i) amax = fabs(rx[i]) > amax ? fabs(rx[i]) : amax;
                double kap = fac_norm == 0.0 ? HUGE_VAL : amax / fabs(fac_norm);
                if (kap > *kappa_out || kap != kap) *kappa_out = kap;
            }
            for (int i = 0; i < n; ++i) out[i] = out[i] / fac_norm;        /* :202 */
        }
    }
}

/* NOTE on Fortran mod with negative first argument: mod(a, n) keeps the sign
 * of a, exactly like C's %.  (n - npre + i) can be negative only if
 * npre > n + i, which the reference does not guard either; j <= 0 would then
 * index out of bounds in the reference.  The oracle reproduces the in-range
 * behaviour only and the tests keep |npre| < nfft. */

void rfo_calc_rf(int nfft, int ntrc, int deconv_mode, double delta, double t_start,
                 double sdep, const double *rayps, const double *a_gus,
                 const int *ipha, int nlay, const double *alpha, const double *beta,
                 const double *rho, const double *h, double *rft, int *npre_out,
                 cplx *spec_out)
{
    rfo_cfg c = {nfft, ntrc, 0, deconv_mode, delta, t_start, sdep, rayps, a_gus, ipha};
    int nh = nfft / 2 + 1;
    double *flt = (double *)malloc(sizeof(double) * (size_t)nh * ntrc);
    rfo_init_filter(nfft, ntrc, delta, a_gus, flt);
    fft_plan pl;
    fft_plan_init(&pl, nfft);
    rfo_scratch w;
    scratch_init(&w, nfft, 0, 0);
    calc_rf_impl(&c, &pl, flt, nlay, alpha, beta, rho, h, rft, npre_out, spec_out, &w, NULL);
    scratch_free(&w);
    fft_plan_free(&pl);
    free(flt);
}

/* src/likelihood.f90:84-98: the misfit / quadratic form part of
 * calc_likelihood.  obs has leading dimension ldobs (src/params.f90:413 uses
 * npts_max = 2000), r_inv is (nsmp, nsmp, ntrc) column-major. */
static double log_likelihood_impl(int nfft, int ntrc, int nsmp, const double *rft,
                                  const double *obs, int ldobs, const double *r_inv,
                                  const double *sig, double *misfits, double *phi1)
{
    double ll = 0.0;                                                       /* :86 */
    for (int itrc = 0; itrc < ntrc; ++itrc) {
        for (int i = 0; i < nsmp; ++i)
            misfits[i] = rft[i + (size_t)nfft * itrc] - obs[i + (size_t)ldobs * itrc]; /* :88 */
        double s = sig[itrc];                                              /* :91 */
        const double *ri = r_inv + (size_t)nsmp * nsmp * itrc;
        for (int j = 0; j < nsmp; ++j) {                                   /* :92 row-vector x matrix */
            double acc = 0.0;
            for (int i = 0; i < nsmp; ++i) acc += misfits[i] * ri[i + (size_t)nsmp * j];
            phi1[j] = acc;
        }
        double phi = 0.0;                                                  /* :93 */
        for (int j = 0; j < nsmp; ++j) phi += phi1[j] * misfits[j];
        ll = ll - 0.5 * phi / (s * s) - (double)nsmp * log(s);             /* :94-96 */
    }
    return ll;
}

double rfo_log_likelihood(int nfft, int ntrc, int nsmp, const double *rft,
                          const double *obs, int ldobs, const double *r_inv,
                          const double *sig)
{
    double *misfits = (double *)malloc(sizeof(double) * (size_t)nsmp);
    double *phi1 = (double *)malloc(sizeof(double) * (size_t)nsmp);
    double ll = log_likelihood_impl(nfft, ntrc, nsmp, rft, obs, ldobs, r_inv, sig, misfits, phi1);
    free(misfits); free(phi1);
    return ll;
}

/* ------------------------------------------------------------------ */
/* src/model.f90:298-314 vp_to_rho: coefficients are default-real (single
 * precision) literals promoted to double. */
double rfo_vp_to_rho(double a1)
{
    double a2 = a1 * a1, a3 = a2 * a1, a4 = a3 * a1, a5 = a4 * a1;
    return (double)1.6612f * a1 - (double)0.4721f * a2 + (double)0.0671f * a3
           - (double)0.0043f * a4 + (double)0.000106f * a5;                /* :308-309 */
}

/* src/sort.f90:34-68 quick_sort on three parallel arrays, 1-based il..ir */
static void swap_d(double *a, int i, int j) { double t = a[i - 1]; a[i - 1] = a[j - 1]; a[j - 1] = t; }
static void quick_sort3(double *a, int il, int ir, double *b, double *c)
{
    if (ir - il <= 0) return;
    int ipiv = (il + ir) / 2;
    double piv = a[ipiv - 1];
    swap_d(a, ipiv, ir); swap_d(b, ipiv, ir); swap_d(c, ipiv, ir);
    int i = il;
    for (int j = il; j <= ir; ++j)
        if (a[j - 1] < piv) {
            swap_d(a, i, j); swap_d(b, i, j); swap_d(c, i, j);
            ++i;
        }
    swap_d(a, i, ir); swap_d(b, i, ir); swap_d(c, i, ir);
    quick_sort3(a, il, i, b, c);
    quick_sort3(a, i + 1, ir, b, c);
}

typedef struct {
    int k_max, vp_mode, nref;
    double sdep, z_max, h_min, z_ref_min, dz_ref;
    double vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max;
    const double *vp_ref, *vs_ref;
} rfo_model_cfg;

/* src/model.f90:175-290 format_model.  Returns nlay; *is_valid as reference.
 * prop_z has k_max-1 entries, prop_dvp/prop_dvs k_max.  Outputs sized
 * >= k_max + 2. */
int rfo_format_model(const rfo_model_cfg *m, int prop_k, const double *prop_z,
                     const double *prop_dvp, const double *prop_dvs,
                     double *alpha, double *beta, double *rho, double *h,
                     int *is_valid)
{
    int kmax = m->k_max;
    double *tz = (double *)malloc(sizeof(double) * (size_t)(kmax > 1 ? kmax : 2));
    double *tvp = (double *)malloc(sizeof(double) * (size_t)kmax);
    double *tvs = (double *)malloc(sizeof(double) * (size_t)kmax);
    memcpy(tz, prop_z, sizeof(double) * (size_t)(kmax - 1));
    memcpy(tvp, prop_dvp, sizeof(double) * (size_t)kmax);
    memcpy(tvs, prop_dvs, sizeof(double) * (size_t)kmax);
    *is_valid = 1;
    quick_sort3(tz, 1, prop_k, tvp, tvs);                                  /* :197-198 */

    int i = 0;
    if (m->sdep > 0.0) {                                                   /* :201-207 */
        alpha[i] = 1.5; beta[i] = -999.0; rho[i] = 1.0; h[i] = m->sdep;
        ++i;
    }
#define RFO_RANGE_CHECK(ii)                                                           \
    if (alpha[ii] < m->vp_min || alpha[ii] > m->vp_max || beta[ii] < m->vs_min ||     \
        beta[ii] > m->vs_max || alpha[ii] / beta[ii] < m->vpvs_min ||                 \
        alpha[ii] / beta[ii] > m->vpvs_max)                                           \
        *is_valid = 0;
    /* top layer :210-231 */
    {
        double zc = 0.5 * (m->sdep + tz[0]);
        int iz = f_nint((zc - m->z_ref_min) / m->dz_ref) + 1;
        beta[i] = m->vs_ref[iz - 1] + tvs[0];
        alpha[i] = m->vp_mode == 1 ? m->vp_ref[iz - 1] + tvp[0] : m->vp_ref[iz - 1];
        RFO_RANGE_CHECK(i)
        rho[i] = rfo_vp_to_rho(alpha[i]);
        h[i] = tz[0] - m->sdep;
        if (h[i] < (double)0.125f * alpha[i]) *is_valid = 0;               /* :229 */
        ++i;
    }
    for (int j = 2; j <= prop_k; ++j) {                                    /* :235-262 */
        double zc = 0.5 * (tz[j - 1] + tz[j - 2]);
        int iz = f_nint((zc - m->z_ref_min) / m->dz_ref) + 1;
        beta[i] = m->vs_ref[iz - 1] + tvs[j - 1];
        alpha[i] = m->vp_mode == 1 ? m->vp_ref[iz - 1] + tvp[j - 1] : m->vp_ref[iz - 1];
        RFO_RANGE_CHECK(i)
        rho[i] = rfo_vp_to_rho(alpha[i]);
        h[i] = tz[j - 1] - tz[j - 2];
        if (h[i] < m->h_min) *is_valid = 0;                                /* :256 */
        ++i;
    }
    {   /* half space :265-282 */
        double zc = 0.5 * (m->z_max + tz[prop_k - 1]);
        int iz = f_nint((zc - m->z_ref_min) / m->dz_ref) + 1;
        beta[i] = m->vs_ref[iz - 1] + tvs[kmax - 1];
        alpha[i] = m->vp_mode == 1 ? m->vp_ref[iz - 1] + tvp[kmax - 1] : m->vp_ref[iz - 1];
        RFO_RANGE_CHECK(i)
        rho[i] = rfo_vp_to_rho(alpha[i]);
        h[i] = 999.0;
        ++i;
    }
#undef RFO_RANGE_CHECK
    free(tz); free(tvp); free(tvs);
    return i;                                                              /* :285 */
}

/* ------------------------------------------------------------------ */
/* Batched driver used by tests and by bench.py's cpu_baseline leg: nb
 * independent calc_likelihood(fwd_flag=.true.) evaluations
 * (src/likelihood.f90:56-101 with the layer stack already formatted).
 * layers is [nb][4][nlay_pad] (alpha, beta, rho, h rows), sig [nb][ntrc].
 * rft_out (may be NULL) is [nb][ntrc][nfft].  nthreads <= 1 -> scalar. */
/* kappa_out (may be NULL) [nb]: max over traces of max|rx| / |maxval(rx)| of the filtered vertical trace the
 * item is normalised by (src/forward.f90:197-202) -- the conditioning number of the tests' kappa rule; 1 with
 * deconvolution. */
void rfo_eval_batch_kappa(int nfft, int ntrc, int nsmp, int deconv_mode, double delta,
                          double t_start, double sdep, const double *rayps,
                          const double *a_gus, const int *ipha, const double *obs,
                          int ldobs, const double *r_inv, int nb, const int *nlay,
                          int nlay_pad, const double *layers, const double *sig,
                          double *logl_out, double *rft_out, int nthreads, double *kappa_out)
{
    rfo_cfg c = {nfft, ntrc, nsmp, deconv_mode, delta, t_start, sdep, rayps, a_gus, ipha};
    int nh = nfft / 2 + 1;
    double *flt = (double *)malloc(sizeof(double) * (size_t)nh * ntrc);
    rfo_init_filter(nfft, ntrc, delta, a_gus, flt);
    fft_plan pl;
    fft_plan_init(&pl, nfft);
    if (nthreads < 1) nthreads = 1;
    /* one scratch set per thread for the whole batch: no allocation inside the evaluation loop */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        rfo_scratch w;
        scratch_init(&w, nfft, ntrc, nsmp);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int b = 0; b < nb; ++b) {
            const double *L = layers + (size_t)b * 4 * nlay_pad;
            calc_rf_impl(&c, &pl, flt, nlay[b], L, L + nlay_pad, L + 2 * nlay_pad,
                         L + 3 * nlay_pad, w.rft, NULL, NULL, &w, kappa_out ? kappa_out + b : NULL);
            logl_out[b] = log_likelihood_impl(nfft, ntrc, nsmp, w.rft, obs, ldobs, r_inv,
                                              sig + (size_t)b * ntrc, w.misfits, w.phi1);
            if (rft_out)
                memcpy(rft_out + (size_t)b * nfft * ntrc, w.rft, sizeof(double) * (size_t)nfft * ntrc);
        }
        scratch_free(&w);
    }
    fft_plan_free(&pl);
    free(flt);
    (void)nthreads;
}

void rfo_eval_batch(int nfft, int ntrc, int nsmp, int deconv_mode, double delta,
                    double t_start, double sdep, const double *rayps,
                    const double *a_gus, const int *ipha, const double *obs,
                    int ldobs, const double *r_inv, int nb, const int *nlay,
                    int nlay_pad, const double *layers, const double *sig,
                    double *logl_out, double *rft_out, int nthreads)
{
    rfo_eval_batch_kappa(nfft, ntrc, nsmp, deconv_mode, delta, t_start, sdep, rayps, a_gus, ipha, obs, ldobs, r_inv,
                         nb, nlay, nlay_pad, layers, sig, logl_out, rft_out, nthreads, NULL);
}

int rfo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
