/*
 * rfgpu_ext.h -- extensions of librfgpu's C ABI beyond the drop-in contract of rfgpu.h (the
 * steps right before and after the hot path, kept on the device), plus instrumentation.  Same library, same
 * conventions, same RFGPU_ABI_VERSION.
 *   - format_model on the device and the evaluation of proposals given as (k, z, dVp, dVs)   [row f-2]
 *   - pinned host memory for those arrays
 *   - the two transforms of the reference's `module fftw` for hosts that execute its plans themselves  [rows a12, f-4]
 *   - posterior accumulation on the device and its end-of-run merge over RCCL               [row f-3]
 *   - launch-plan options, the launch plan, HIP-event timing
 */
#ifndef RFGPU_EXT_H
#define RFGPU_EXT_H

#include "rfgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- format_model on the device (the step right before the path) ------------ */
/* What format_model (src/model.f90:175-290) reads from `module params` / `module model`
 * besides the proposal itself; sdep comes from the context. */
typedef struct rf_model_config {
    int32_t k_max;        /* params k_max                                                  */
    int32_t vp_mode;      /* params vp_mode: 1 = dVp solved, 0 = Vp from the reference model */
    int32_t nref;         /* entries of the reference velocity table                       */
    double z_max, h_min;  /* params z_max, h_min                                            */
    double z_ref_min, dz_ref;                         /* model z_ref_min, dz_ref (src/model.f90:36) */
    double vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max; /* params validity limits   */
    const double *vp_ref, *vs_ref;                    /* [nref] model vp_ref, vs_ref       */
} rf_model_config;
int rf_set_model(rf_ctx *ctx, const rf_model_config *m);

/* subroutine format_model(prop_k, prop_z, prop_dvp, prop_dvs, nlay, alpha, beta, rho, h, is_valid)
 * for nb proposals at once, all pointers device pointers: k[nb], z[nb][k_max-1],
 * dvp[nb][k_max], dvs[nb][k_max] -> nlay[nb], layers[nb][4][nlay_pad] (nlay_pad >= k_max + 2),
 * valid[nb] (may be NULL).  Bit-exact with the reference: same quick_sort permutation
 * (src/sort.f90:34-68), nint look-ups, vp_to_rho with its single-precision literals. */
int rf_format_models_device(rf_ctx *ctx, int32_t nb, const int32_t *d_k, const double *d_z,
                            const double *d_dvp, const double *d_dvs, int32_t *d_nlay, double *d_layers,
                            int32_t nlay_pad, int32_t *d_valid, void *stream);

/* format_model + calc_likelihood for nb proposals given as (k, z, dVp, dVs): what
 * src/pt_mcmc.f90:163-180 does per chain.  Items whose model is invalid are not evaluated
 * (the reference turns them into null proposals): valid[i] = 0, logl[i] = NaN.
 * fwd_flag[i] = 0 items (sigma-only) skip format_model like the reference does. */
int rf_eval_models_device(rf_ctx *ctx, int32_t nb, const int32_t *d_walker_ids, const int32_t *d_fwd_flag,
                          const int32_t *d_k, const double *d_z, const double *d_dvp, const double *d_dvs,
                          const double *d_sig, double *d_logl, int32_t *d_valid, void *stream);

/* the same from HOST arrays in the layout the batched sampler keeps its proposals in (one column per chain, as
 * module model's z(k_max-1, nchains), dvp/dvs(k_max, nchains), src/model.f90:33-34): k[nb]; z(ldz, nb), rows 1 .. k_max-1
 * used, ldz = k_max - 1 or k_max; dvp(k_max, nb) (read only when vp_mode = 1); dvs(k_max, nb); sig(ntrc, nb);
 * fwd_flag[nb] or NULL (1 forward model, 0 sigma-only, < 0 skip: the reference's null proposals); logl[nb] out;
 * valid[nb] out, may be NULL.  Synchronous at return.  One iteration of pt_control_batched is one such call: the
 * host keeps format_model only for the validity verdict its random stream depends on (src/pt_mcmc.f90:163-169). */
int rf_eval_models(rf_ctx *ctx, int32_t nb, const int32_t *walker_ids, const int32_t *fwd_flag, const int32_t *k,
                   const double *z, int32_t ldz, const double *dvp, const double *dvs, const double *sig,
                   double *logl, int32_t *valid);

/* Asynchronous form, for a host that overlaps its own work with the evaluation (pt_control_batched proposes one half
 * of its chains while the other half is being evaluated): rf_eval_models_begin enqueues the transfers and kernels on
 * the context's stream and returns a ticket; rf_eval_wait blocks until that evaluation has finished and delivers
 * logl[nb] (and valid[nb] if want_valid was set).  Evaluations of a context execute in submission order; up to
 * RF_EVAL_MAX_IN_FLIGHT may be outstanding.  Pageable input arrays are copied before rf_eval_models_begin returns;
 * PINNED ones (rf_host_alloc) are read by DMA afterwards and must stay untouched until the matching rf_eval_wait.
 * Other host-buffer calls on the context (rf_commit, rf_post_record ...) may be issued in between: stream-ordered. */
#define RF_EVAL_MAX_IN_FLIGHT 4
int rf_eval_models_begin(rf_ctx *ctx, int32_t nb, const int32_t *walker_ids, const int32_t *fwd_flag, const int32_t *k,
                         const double *z, int32_t ldz, const double *dvp, const double *dvs, const double *sig,
                         int32_t want_valid, int32_t *ticket);
int rf_eval_wait(rf_ctx *ctx, int32_t ticket, double *logl, int32_t *valid);

/* ---- the transforms of `module fftw` outside the hot path ------------------------
 * What the reference's two FFTW plans compute when a host executes them itself -- src/make_syn.f90:91-95,107-111 filters
 * its noise series with dfftw_execute(ifft2) (r2c), a product with flt, dfftw_execute(ifft) (c2r) -- so that the drop-in
 * `module fftw` (rf_inv_amd/fortran/fftw.f90) needs no FFTW3.  Host pointers, synchronous, on the calling thread's
 * current HIP device; no context.  nfft: any length 2 .. 1048576 (the transform's definition, exact-table twiddles,
 * compensated sums: O(nfft^2 / 2), an init-time utility -- inside evaluations the c2r is part of the trace kernels).
 *   rf_fft_c2r: the plan `ifft` of src/fftw.f90:44 -- rx(1:nfft) from cx(1:nfft/2+1) (complex128, interleaved re, im),
 *               unnormalised, Hermitian extension implied, Im cx(1) and Im cx(nfft/2+1) ignored as FFTW's c2r does
 *   rf_fft_r2c: the plan `ifft2` of :45 -- cx(1:nfft/2+1) from rx(1:nfft); cx beyond nfft/2+1 is not touched */
int rf_fft_c2r(int32_t nfft, const double *cx, double *rx);
int rf_fft_r2c(int32_t nfft, const double *rx, double *cx);

/* Pinned (page-locked, device-mapped) host memory.  Host arrays handed to rf_eval_batch / rf_eval_models from such
 * memory go to the device by DMA as they are; pageable arrays are first copied into the context's own pinned staging
 * area (one host memcpy per array and call: ~1 KB per chain at k_max 30).  Optional; any host memory works. */
int rf_host_alloc(size_t bytes, void **ptr);
int rf_host_free(void *ptr);

/* ---- posterior accumulation ---------------------------- */
/* The "record sampled model" block of subroutine mcmc (src/pt_mcmc.f90:204-286) with the
 * accumulators of module pt_mcmc (allocated/zeroed src/pt_mcmc.f90:394-421) kept on the
 * device, so that the traces of the recorded chains never leave HBM.  Bin widths are
 * formed as src/pt_mcmc.f90:423-430.  Needs rf_set_model (the V-z profile runs
 * format_model, :240-242).  Layouts are the reference's (column-major):
 * nk[k_max], nz[nbin_z], nsig[ntrc][nbin_sig], namp[ntrc][nsmp][nbin_amp],
 * nvpz[nbin_vp][nbin_z], nvsz[nbin_vs][nbin_z], nvpvsz[nbin_vpvs][nbin_z],
 * vp_mean/vs_mean/vpvs_mean[nbin_z], vp_model/vs_model[max_models][nbin_z],
 * all_likelihood[max_models] -- i.e. exactly the memory of the Fortran arrays
 * nsig(nbin_sig,ntrc), namp(nbin_amp,nsmp,ntrc), nvpz(nbin_z,nbin_vp), vp_model(nbin_z,:) ... */
typedef struct rf_post_config {
    int32_t nbin_z, nbin_vs, nbin_vp, nbin_vpvs, nbin_sig, nbin_amp;  /* params, src/params.f90:293-314 */
    double amp_min, amp_max;                                          /* src/params.f90:317           */
    double z_min;                                                     /* params z_min (nz bins, :229) */
    const double *sig_min, *sig_max;                                  /* [ntrc]                        */
    const int32_t *sig_mode;                                          /* [ntrc] 1 = sigma solved       */
    int64_t max_models;   /* capacity of vp_model / vs_model / all_likelihood (the reference sizes it
                             int(nchains*niter/ncorr), :407-409); 0 = do not keep per-model profiles */
} rf_post_config;
int rf_post_create(rf_ctx *ctx, const rf_post_config *cfg);
int rf_post_reset(rf_ctx *ctx);
/* Record n chains, in order, exactly as n consecutive passes through src/pt_mcmc.f90:204-286
 * would (the fp64 sums vp_mean / vs_mean / vpvs_mean -- and the ocean-layer ASSIGNMENTS
 * :263-264 -- are applied in that order, so they are bit-identical to the reference's).
 * walker_ids[n] name the chains whose CURRENT trace feeds the amplitude histogram;
 * k[n], z[n][k_max-1], dvp[n][k_max], dvs[n][k_max], sig[n][ntrc], logl[n] are the chains'
 * current state.  temps (may be NULL) applies the reference's filter temp <= 1 + 1e-6 (:204)
 * on the device: chains above it are skipped.  The caller applies the iteration filter
 * (iter > nburn, mod(iter, ncorr) == 0).  _device: all pointers are device pointers and the
 * call is asynchronous on `stream`; the host variant copies pageable arrays before it returns (they may be changed at
 * once; arrays in PINNED memory -- rf_host_alloc -- are read by DMA in place after it returns:
 * leave them alone until a later call on the context has waited for work issued after it, e.g. rf_eval_wait) and, like
 * rf_commit, does not wait for the device: stream-ordered between the calls issued before and after it.
 * Departures (the reference has undefined behaviour there): histogram indices outside an
 * array are clamped to its edge bins; models beyond max_models are counted but their
 * profile rows are dropped.  Amplitudes outside [amp_min, amp_max) go to the edge bins as in
 * :274-281, counted in amp_out_of_range instead of the reference's warning line. */
int rf_post_record(rf_ctx *ctx, int32_t n, const int32_t *walker_ids, const int32_t *k, const double *z,
                   const double *dvp, const double *dvs, const double *sig, const double *logl,
                   const double *temps);
int rf_post_record_device(rf_ctx *ctx, int32_t n, const int32_t *d_walker_ids, const int32_t *d_k,
                          const double *d_z, const double *d_dvp, const double *d_dvs, const double *d_sig,
                          const double *d_logl, const double *d_temps, void *stream);

/* Copy the accumulators to host arrays (any pointer may be NULL = not wanted).  Synchronises
 * the context stream; after rf_post_record_device on another stream the caller synchronises
 * that stream first. */
typedef struct rf_post_result {
    int32_t *nmod;                 /* [1]                                     */
    int32_t *nk, *nz, *nsig, *namp, *nvpz, *nvsz, *nvpvsz;
    double *vp_mean, *vs_mean, *vpvs_mean;
    double *vp_model, *vs_model;   /* [max_models][nbin_z]: only the first min(nmod, max_models) rows    */
    double *all_likelihood;        /* [max_models]          are written; the caller's later rows keep     */
                                   /* what init_pt_mcmc put there (vs_model(1,:) = -999.9, :419)          */
    int64_t *amp_out_of_range;     /* [1]                                     */
} rf_post_result;
int rf_post_read(rf_ctx *ctx, const rf_post_result *out);

/* End-of-run merge of the accumulators over the context's RCCL communicator (rf_comm_init), replacing the MPI calls
 * at the top of output_results for the arrays kept on the device.  Both are collective and synchronise the context
 * stream.
 * rf_comm_post_reduce: src/mcmc_out.f90:58-71,74-79 -- mpi_reduce(SUM -> root) of nk, namp, nvpz, nvsz, nvpvsz, nz,
 *   nsig (int32) and vp_mean, vs_mean, vpvs_mean (f64), plus amp_out_of_range -- as one group of ncclReduce IN PLACE
 *   into the root's device accumulators: rf_post_read on the root then returns the merged arrays; other ranks keep their
 *   own.  nmod (:52) is summed into *nmod_sum (root only; may be NULL) and NOT into the root's accumulator, whose
 *   count goes on naming the root's own model rows.  Call it once per run.  (nprop / naccept / likelihood_hist,
 *   :54-57,72-73, are host-module arrays: the host reduces them as before.)
 * rf_comm_post_gather: src/mcmc_out.f90:88-93 -- mpi_gather of vs_model, vp_model (and all_likelihood, which the
 *   reference allocates per rank, :84, but never gathers).  nmod_rank[nranks] (every rank; may be NULL) = models each
 *   rank recorded; on the root, host arrays vp_model_all / vs_model_all [nranks][max_models][nbin_z] and
 *   all_likelihood_all [nranks][max_models] (any may be NULL; ignored on other ranks) receive per rank block the first
 *   min(nmod_rank[r], max_models) rows, later rows untouched (as rf_post_read). */
int rf_comm_post_reduce(rf_ctx *ctx, int32_t root, int32_t *nmod_sum);
int rf_comm_post_gather(rf_ctx *ctx, int32_t root, int32_t *nmod_rank, double *vp_model_all, double *vs_model_all,
                        double *all_likelihood_all);

/* Options of the context's communicator (after rf_comm_init).
 *   "sequential_reduce"  0 (default): rf_comm_post_reduce issues its twelve ncclReduce calls as ONE group | 1: one after the
 *                        other on the stream -- the same sums; a one-flag way around an RCCL build that mishandles in-place
 *                        reductions inside a group */
int rf_comm_set_option(rf_ctx *ctx, const char *name, double value);

/* ---- instrumentation ----------------------------------------------------- */
/* Launch-plan options.  librfgpu reads NO environment variables; a knob is set here, validated,
 * and reported by rf_get_launch_plan.  Every option re-partitions or re-orders the same work:
 * results do not depend on them (tests/test_gpu_parity.py), except "bin_cutoff", which is opt-in
 * and off by default, and "block_threads", whose two kernels factorise the FFT differently (last-bit differences
 * between the two settings, never within one).  Call between evaluations (the call synchronises the device).
 *   "fused"            -1 by shape (default) | 0 split spectra -> trace kernels | 1 one fused kernel (per (walker,
 *                      trace), or per walker when the rays are common: plan[0])
 *   "chain"            -1 by shape (default) | 0, 2, 3, 4, 8 bins per phase chain
 *   "lpt"              1 (default) longest-first dispatch order | 0
 *   "order_reuse"      1 (default) a launch prepares the next launch's order | 0 order kernel every time
 *   "nsplit"           0 by batch size (default) | 1..64 bin-splits per walker (split spectra kernel)
 *   "waves_per_block"  1..4 (default 4) waves sharing a staged layer stack (split spectra kernel)
 *   "defer_logl"       -1 by batch size (default) | 0 quadratic form + logL inside the main kernel | 1 follow-up kernel
 *   "block_threads"    0 by the context's capacity (default: max_walkers * ntrc blocks within three rounds of the
 *                      GPU -> 512; fixed per context, never per launch) | 256 fused_kernel | 512 fused8_kernel
 *                      (nfft 4096 on land only)
 *   "gemm_tile"        0 (default) = 64 | 128: the block tile of the long-window plan's GEMM (plan[12]): 128 x 64
 *                      (walkers x columns, four blocks per CU) or 128 x 128 (two); same values
 *   "copy_stream"      0 (default) | 1: rf_eval_models_begin transfers its host arrays on a stream of the context's own,
 *                      so that they run under the kernels of the evaluation before it (a sampler's two pipeline
 *                      segments: one rank at the C4 shape 4.8 -> 5.2 M steps/s).  For a process that has the GPU to
 *                      ITSELF: with several processes on one GPU the extra queue per process makes the hardware
 *                      scheduler time-slice them (4 ranks: 5.0 -> 3.3 M).  pt_control_batched sets it accordingly.
 *   "gemm_triangle"    1 (default): the long-window GEMM runs on the quadratic form's upper triangle T(i, j) = R^-1(i, j)
 *                      + R^-1(j, i) (i < j), R^-1(j, j), 0 below -- m R m^T = sum_j m_j sum_{i<=j} m_i T(i, j) for ANY R, half
 *                      the multiply-adds | 0: the full product m . R^-1 in the reference's row order.  Results agree to
 *                      rounding (both within the parity tolerance of the oracle), not bit for bit.
 *   "trace_window"     0 (default: every trace is kept as the reference's rft(nfft, ntrc, chain), filled completely) | 1:
 *                      only samples 1 .. nsmp are stored -- all the likelihood, the histograms and make_syn ever read
 *                      (src/likelihood.f90:88, src/pt_mcmc.f90:273-274): the trace array shrinks nfft / nsmp-fold (C5:
 *                      8.6 GB -> 0.2 GB) and the trace kernels write 40x less.  Same logL and same samples 1 .. nsmp, bit
 *                      for bit.  rf_get_rft* then refuse n > nsmp, rf_calc_rf and rf_calc_likelihood with prop_rft
 *                      refuse.  Switching it re-allocates the array: EVERY stored trace is dropped (set it before the
 *                      first evaluation, or re-evaluate and commit the chains afterwards, as pt_control_batched does)
 *   "bin_cutoff"       0 (default: every bin like the reference) | tol in (0, 1): bins whose Gaussian filter
 *                      weight is below tol * flt(1) are not propagated (contexts with one
 *                      forward computation per trace only: common-ray contexts share one pass between filters
 *                      of different width and ignore it)
 * A library built with -DRFGPU_DIAGNOSTICS (tools/ablate.sh; never the shipped one) also accepts
 * "ablate" = N: blocks stop after phase N, results are invalid. */
int rf_set_option(rf_ctx *ctx, const char *name, double value);

/* how rf_eval_batch* will launch, plan[16] (entries beyond those listed are 0):
 *  [0] 1 when spectra + trace run as ONE fused kernel (contexts with one forward computation per trace), 2 when
 *      they run as the common-ray fused kernel (several traces of one ray, nfft 4096 on land: one block per walker,
 *      one propagator pass, ntrc trace tails); then ms[0] of rf_profile_read is that kernel and ms[1] stays 0
 *  [1] bins per phase chain (0: direct sincos)   [2] waves per block of the split spectra kernel
 *  [3] bin-splits per walker at a full batch     [4] lpt   [5] order_reuse   [6] defer_logl (-1 / 0 / 1)
 *  [7] 1 when a bin cut-off is active            [8] number of options away from their defaults
 *  [9] 0 production build | 1 RFGPU_DIAGNOSTICS build | 2 diagnostics build with "ablate" set (results invalid)
 *  [10] the "block_threads" option (0 = by capacity)   [11] threads per block of the context's fused kernel
 *  [12] 0 | 2 (1 with "gemm_triangle" = 0) on the long-window plan (nsmp > 191; the reference allows npts_max = 2000, src/params.f90:44): every trace
 *       kernel leaves its misfits in HBM and the quadratic forms misfit . R^-1 . misfit of the whole batch run as ONE
 *       tiled GEMM on the FP64 matrix cores (v_mfma_f64_16x16x4_f64) followed by logL; "defer_logl" is then ignored.
 *       Fixed per context from nsmp; ms[2] of rf_profile_read is the GEMM + logL pair
 *  [13] the "trace_window" option
 *  [14] host arrays of the last rf_eval_batch / rf_eval_models(_begin) call that were copied into the context's pinned
 *       staging area (pageable memory); 0 = all of them travelled by DMA from the caller's own pinned arrays (rf_host_alloc)
 *  [15] the "copy_stream" option */
int rf_get_launch_plan(const rf_ctx *ctx, int32_t *plan);

/* HIP-event timing (on the streams the kernels are launched on) of the three kernels
 * of rf_eval_batch*, accumulated while enabled.  on = 1 times every batch, on = k > 1 every k-th
 * batch (an event record costs ~4 us of stream time: sampling keeps a timed loop undisturbed),
 * 0 switches it off.  ms[3] = spectra, trace, logl totals over the timed batches;
 * launches[4] = batches timed, then spectra / trace / logl kernel launches (a batch is
 * pipelined in chunks, so there can be several spectra / trace launches per batch). */
int rf_profile_enable(rf_ctx *ctx, int32_t on);
int rf_profile_read(rf_ctx *ctx, double *ms, int64_t *launches, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* RFGPU_EXT_H */
