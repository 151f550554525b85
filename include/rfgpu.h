/*
 * rfgpu.h -- C ABI of librfgpu: the MI355X (gfx950) forward + likelihood engine
 * that replaces RF_INV's src/forward.f90 + src/likelihood.f90.
 *
 * The reference has no FFI for this path: the boundary is two Fortran module
 * interfaces (`module forward`, `module likelihood`).  The entry points below
 * are exactly what a bind(C) replacement of those two modules binds; each one
 * cites the reference interface it replaces.  The Fortran shim modules that do
 * the binding live in rf_inv_amd/fortran/ and INTEGRATION.md shows the build
 * change a maintainer makes.
 *
 * Conventions
 *   - plain pointers and sizes only; all reals are IEEE binary64;
 *   - arrays are Fortran column-major exactly as the reference passes them;
 *   - every call returns 0 on success, non-zero on error (rf_last_error()
 *     returns the message); evaluations never trap on out-of-domain physics,
 *     they propagate NaN like the reference;
 *   - "walker" = one chain slot owned by the context, 0-based;
 *   - host-buffer calls are synchronous at return and run on the context's own
 *     stream; *_device calls take device pointers + a hipStream_t (as void*)
 *     and are asynchronous on that stream: synchronise it before a host-buffer
 *     call that depends on their result (rf_get_rft, rf_commit, ...).
 *   - a context is not thread-safe; use one per host thread / process / GPU.
 *   - there is NO CPU fallback: if no gfx950 device is usable rf_ctx_create
 *     fails.
 *
 * This header is the CONTRACT of SURVEY.md section 8b: context, the tables the
 * reference exports, the single-evaluation drop-ins, the batched evaluation,
 * commit / trace read-back, the temperature swap and its RCCL transport.
 * Everything else -- format_model on the device, the asynchronous evaluation of
 * proposals, pinned host memory, the posterior accumulators and their merge,
 * launch-plan options and timing -- is in rfgpu_ext.h (same library, same ABI
 * version).
 */
#ifndef RFGPU_H
#define RFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFGPU_ABI_VERSION 6

typedef struct rf_ctx rf_ctx;

/* Everything `init_forward` / `init_likelihood` read from `module params`
 * (reference src/params.f90:34-96) plus capacity hints. */
typedef struct rf_config {
    int32_t nfft;        /* params nfft >= 8 (FFTW plans any n, src/fftw.f90:44).  Powers of two: in-LDS FFT up to
                            8192, four-step transform through HBM up to 65536.  Any other length: direct DFT up
                            to 2048, Bluestein's algorithm (two power-of-two transforms) up to 32768 */
    int32_t ntrc;        /* params ntrc                                         */
    int32_t nsmp;        /* params nsmp (src/params.f90:449-451)                */
    int32_t deconv_mode; /* params deconv_mode: 0 = normalise by vertical, 1 = water-level decon */
    double delta;        /* params delta = dble(float32 SAC delta) (src/params.f90:452) */
    double t_start;      /* params t_start                                      */
    double sdep;         /* params sdep (> 0: ocean layer; keys direct_arrival, src/forward.f90:484) */
    const double *rayps; /* [ntrc] params rayps                                 */
    const double *a_gus; /* [ntrc] params a_gus                                 */
    const int32_t *ipha; /* [ntrc] params ipha: +1 P, -1 S                      */
    const double *obs;   /* obs(ldobs, ntrc) column-major, rows 1..nsmp used (src/params.f90:413,458) */
    int32_t ldobs;       /* leading dimension of obs (reference: npts_max = 2000) */
    const double *r_inv; /* r_inv(nsmp, nsmp, ntrc) column-major as built by src/likelihood.f90:168-222,
                            or NULL: the library builds it (rf_compute_r_inv)    */
    int32_t max_walkers; /* chain slots to allocate (reference: nchains)        */
    int32_t nlay_max;    /* max layers incl. ocean + half-space (reference nlay_max = 200, src/params.f90:44) */
    int32_t device;      /* HIP device ordinal                                  */
} rf_config;

/* ---- lifecycle ------------------------------------------------------- */
/* replaces init_fftw (src/fftw.f90:41-48) + init_forward (src/forward.f90:47-55)
 * + init_r_inv (src/likelihood.f90:168-241) + allocation of rft/log_likelihood
 * state (src/likelihood.f90:150-151). */
/* Device memory of a context (W = max_walkers + 1): traces 16 W ntrc nfft bytes (option "trace_window": nsmp instead of
 * nfft); per-item constants 200 W nfwd nlay_max bytes; contexts on the split launch plan (common rays off nfft 4096,
 * nfft 8192 and beyond, nfft not a power of two, or after rf_set_option("fused", 0)) add the spectra, 32 W nfwd (nfft/2+1)
 * bytes, allocated here or by that rf_set_option call; long series (nfft > 8192, or > 2048 and not a power of two) add
 * two scratch rows of 16 M bytes (M = the transform length, <= 65536) per resident block, at most 2 blocks per CU
 * (1 GB at M = 65536 on 256 CUs); long windows (nsmp > 191) add the padded R^-1 image and misfit rows of ~8 nsmp bytes. */
int rf_ctx_create(const rf_config *cfg, rf_ctx **ctx_out);
int rf_ctx_destroy(rf_ctx *ctx);
const char *rf_last_error(void);
int rf_abi_version(void);

/* ---- tables the reference exports ------------------------------------ */
/* `flt(nh, ntrc)` public array of module forward (src/forward.f90:30,95-119) */
int rf_get_flt(const rf_ctx *ctx, double *flt);
/* `is_ray_common` of module forward (src/forward.f90:36,59-91) */
int rf_get_is_ray_common(const rf_ctx *ctx, int32_t *flag);
/* private r_inv(nsmp, nsmp, ntrc) of module likelihood (src/likelihood.f90:34) */
int rf_get_r_inv(const rf_ctx *ctx, double *r_inv);
/* init_r_inv for one trace (src/likelihood.f90:183-222): Gaussian-correlated
 * noise matrix, SVD, pseudo-inverse with cut-off s > 1e-3.  Host-only helper
 * (one-sided Jacobi SVD, fp64); r_inv is (nsmp, nsmp) column-major.
 * rank_out (may be NULL): singular values kept.  cut_gap_out (may be NULL): min |s - 1e-3| / 1e-3 over
 * the singular values -- how far the nearest one is from the hard rank cut-off of :214.  The reference's
 * LAPACK dgesvd and any other correct SVD agree on the rank only while that gap is wide compared with
 * their rounding (~1e-11 relative to the cut-off); rf_ctx_create refuses to build r_inv itself below
 * RF_R_INV_MIN_CUT_GAP and asks for the host's own r_inv instead. */
#define RF_R_INV_MIN_CUT_GAP 1.0e-7
int rf_compute_r_inv(int32_t nsmp, double a_gus, double delta, double *r_inv, int32_t *rank_out,
                     double *cut_gap_out);
/* per trace: rank and cut-off gap of the pseudo-inverse the library built (rank -1 / gap NaN where the
 * caller supplied r_inv); rank[ntrc], cut_gap[ntrc], either may be NULL */
int rf_get_r_inv_info(const rf_ctx *ctx, int32_t *rank, double *cut_gap);

/* replace the noise-covariance pseudo-inverse after creation, e.g. with the one the
 * host built through its own LAPACK dgesvd exactly as src/likelihood.f90:183-222
 * (r_inv(nsmp, nsmp, ntrc) column-major).  Every device image of the matrix is replaced.  Set it BEFORE the first
 * evaluation: the quadratic forms cached with the stored traces (re-used by sigma-only proposals, fwd_flag = 0) are
 * invalidated -- such a proposal returns NaN until its chain has been evaluated with fwd_flag = 1 and committed again. */
int rf_set_r_inv(rf_ctx *ctx, const double *r_inv);

/* ---- single-evaluation drop-ins --------------------------------------- */
/* subroutine calc_rf(chain_id, nlay, n, ntrc, rayps, alpha, beta, rho, h, rft)
 * (src/forward.f90:123-208).  n, ntrc, rayps come from the context.
 * rft is rft(nfft, ntrc), filled completely. */
int rf_calc_rf(rf_ctx *ctx, int32_t nlay, const double *alpha, const double *beta,
               const double *rho, const double *h, double *rft);

/* subroutine calc_likelihood(chain_id, fwd_flag, prop_k, prop_z, prop_dvp, prop_dvs,
 *                            sig, prop_log_likelihood, prop_rft)
 * (src/likelihood.f90:56-101) with the layer stack already formatted by the
 * host's format_model (src/likelihood.f90:75-76 stays on the Fortran side).
 * fwd_flag = 0 re-uses the walker's stored trace (src/likelihood.f90:81).
 * prop_rft(nfft, ntrc) may be NULL (trace stays device-resident). */
int rf_calc_likelihood(rf_ctx *ctx, int32_t walker, int32_t fwd_flag, int32_t nlay,
                       const double *alpha, const double *beta, const double *rho,
                       const double *h, const double *sig, double *prop_log_likelihood,
                       double *prop_rft);

/* the fwd_flag = .false. branch of calc_likelihood for a trace the HOST owns
 * (src/likelihood.f90:81-98: prop_rft = rft(:,:,chain_id), then the misfit loop).
 * In the per-call drop-in the Fortran host keeps `rft(nfft, ntrc, nchains)` itself
 * (src/pt_mcmc.f90:190 writes it without calling into this module), so the shim
 * passes the stored trace in: rft is rft(nfft, ntrc), sig(ntrc). */
int rf_calc_likelihood_of_trace(rf_ctx *ctx, const double *rft, const double *sig, double *logl);

/* ---- batched evaluation (the throughput path) ------------------------- */
/* nb independent calc_likelihood calls (the sequential chain loop of
 * src/pt_mcmc.f90:493-496 turned into one launch).
 *   walker_ids[nb]  distinct walker slots
 *   fwd_flag[nb]    or NULL (= all 1); 1 = forward model, 0 = sigma-only (stored trace re-used),
 *                   < 0 = skip the item (logl = NaN, nothing changes)
 *   nlay[nb]        layers of each proposed model
 *   layers          [nb][4][nlay_pad]: alpha, beta, rho, h rows (C order)
 *   sig             [nb][ntrc]
 *   logl            [nb] out
 * The proposed traces stay on the device until rf_commit / rf_get_rft. */
int rf_eval_batch(rf_ctx *ctx, int32_t nb, const int32_t *walker_ids, const int32_t *fwd_flag,
                  const int32_t *nlay, int32_t nlay_pad, const double *layers,
                  const double *sig, double *logl);
/* same with every pointer a device pointer; asynchronous on `stream`.  The arrays cannot be checked on the host, so
 * the first kernel of the batch checks every item on the device: nlay outside [2, nlay_pad], a walker id outside
 * [0, max_walkers] or a fwd_flag > 1 makes that ITEM refused -- logl = NaN, nothing of it evaluated or written, the rest
 * of the batch unaffected -- and the next call on the context that returns a status after the work has been
 * synchronised (rf_commit*, rf_get_rft*, rf_eval_batch, rf_eval_wait, rf_calc_likelihood, rf_profile_read, rf_post_read,
 * rf_ctx_destroy) fails once, rf_last_error naming the item and the reason.  d_logl may also be pinned
 * (device-mapped) host memory.  Calls on one context share per-context state (trace slots, the
 * dispatch order a launch prepares for the next one): issue them on one stream, or synchronise
 * between streams yourself. */
int rf_eval_batch_device(rf_ctx *ctx, int32_t nb, const int32_t *d_walker_ids,
                         const int32_t *d_fwd_flag, const int32_t *d_nlay, int32_t nlay_pad,
                         const double *d_layers, const double *d_sig, double *d_logl,
                         void *stream);

/* accept step of src/pt_mcmc.f90:182-191: for accept[i] != 0 the proposed trace
 * of walker_ids[i] becomes its current trace (`rft(:,:,ichain) = prop_rft`).
 * rf_commit returns nothing from the device and does not wait for it: the arrays are copied before it returns (the
 * caller may reuse them at once) and every later call on the context -- host-buffer or *_device, on any stream --
 * is ordered behind the flip. */
int rf_commit(rf_ctx *ctx, int32_t nb, const int32_t *walker_ids, const int32_t *accept);
int rf_commit_device(rf_ctx *ctx, int32_t nb, const int32_t *d_walker_ids,
                     const int32_t *d_accept, void *stream);

/* read back `rft(1:n, 1:ntrc, walker)` (which = 0) or the last proposed trace
 * (which = 1); out is out(n, ntrc) column-major.  Used for histogram recording
 * (src/pt_mcmc.f90:272-285). */
int rf_get_rft(rf_ctx *ctx, int32_t walker, int32_t which, int32_t n, double *out);

/* the same for n walkers in one device gather + one copy: out is out(nout, ntrc, n)
 * column-major (walker i at out[i * ntrc * nout]).  The batched sampler uses it once per
 * recording iteration instead of one copy per chain. */
int rf_get_rft_batch(rf_ctx *ctx, int32_t n, const int32_t *walker_ids, int32_t which, int32_t nout,
                     double *out);

/* ---- parallel tempering ------------------------------------------------ */
/* judge_pt (src/pt_mcmc.f90:580-595) for npairs DISJOINT chain pairs: swap
 * temps[i1] <-> temps[i2] iff log(u) <= (L2-L1)(1/T1-1/T2).  temps/logl are device
 * arrays indexed by (local or gathered-global) walker; pairs [npairs][2]; no walker
 * may appear twice (the reference proposes a single pair per iteration,
 * src/pt_mcmc.f90:501-506).  log_u[npairs] are the host-drawn log(grnd()) values
 * (the RNG stays on the host; temperatures move, states stay: src/pt_mcmc.f90:532-535). */
int rf_pt_swap_device(rf_ctx *ctx, int32_t npairs, const int32_t *d_pairs, const double *d_log_u,
                      double *d_temps, const double *d_logl, int32_t *d_accepted, void *stream);

/* ---- multi-GPU: the temperature exchange over RCCL (xGMI) -------------------------------
 * Replaces the MPI traffic of the swap step (src/pt_mcmc.f90:498-571).  One process per GPU, one context
 * per process; walkers shard across ranks in contiguous blocks (global id = rank * nchains + chain, :508-511)
 * and never migrate: temperatures move (:532-535).  RCCL is loaded at run time (librccl.so.1); single-GPU
 * runs never need it.  Bootstrap: rank 0 calls rf_comm_get_unique_id, the host distributes the 128 bytes
 * by whatever it already has (the Fortran host: one mpi_bcast at start-up), every rank calls rf_comm_init.
 * RCCL needs one GPU per rank (it refuses two ranks on one device): rf_comm_init then fails and the host keeps
 * its own transport (the Fortran batched host: mpi_sendrecv). */
#define RF_COMM_ID_BYTES 128
/* device_key = the physical GPU of the context (hash of host name + boot id over PCI domain/bus/device, >= 0); loads
 * nothing: hosts compare the keys of their ranks first and only probe RCCL when every rank has a GPU of its own. */
int rf_comm_device_key(rf_ctx *ctx, int64_t *device_key);
/* the same key, and 0 when this rank can join an RCCL communicator (librccl.so.1 loads; takes ~1 s).  ncclCommInitRank
 * is collective: the host gathers (result, key) of all ranks and calls rf_comm_init only if every rank returned 0 and
 * all keys differ. */
int rf_comm_probe(rf_ctx *ctx, int64_t *device_key);
/* optional, process-wide, before anything else of this section: load RCCL from this file instead of the default search
 * (librccl.so.1 by soname -- inside a PyTorch process the RCCL torch loaded -- then /opt/rocm/lib/librccl.so.1) */
int rf_comm_set_library(const char *path);
int rf_comm_get_unique_id(uint8_t *id /* [RF_COMM_ID_BYTES] */);
int rf_comm_init(rf_ctx *ctx, const uint8_t *id, int32_t rank, int32_t nranks);
int rf_comm_destroy(rf_ctx *ctx);
/* rank / size of the context's communicator (0 of 1 without one) and the RCCL version in use (ncclGetVersion:
 * major * 10000 + minor * 100 + patch; 0 when librccl.so.1 cannot be loaded); any pointer may be NULL */
int rf_comm_info(rf_ctx *ctx, int32_t *rank, int32_t *nranks, int32_t *rccl_version);
/* mpi_bcast(ipack, 4, MPI_INTEGER4, 0, ...) of :518-519 (the pair rank 0 drew): buf[n] host, n <= 8,
 * 0 <= root < nranks */
int rf_comm_bcast_i32(rf_ctx *ctx, int32_t *buf, int32_t n, int32_t root);
/* the cross-rank branch :542-571 as ONE grouped ncclSend + ncclRecv with `peer`: both ranks exchange
 * (T, logL, log u) of their chain and form the same judge_pt decision (:580-595) from the uniform of the rank
 * that owns chain 1 (judge != 0 there; the reference lets that rank judge and mail the temperature back).
 * temp / logl: this rank's chain; log_u: log(grnd()) on the judge, ignored on the peer.
 * new_temp: the temperature this rank's chain holds afterwards; accepted (may be NULL). */
int rf_pt_swap_exchange(rf_ctx *ctx, int32_t peer, int32_t judge, double temp, double logl, double log_u,
                        double *new_temp, int32_t *accepted);
/* throughput form: npairs DISJOINT pairs of GLOBAL walker ids per iteration: ONE RCCL group (two all-gathers
 * straight from d_temps / d_logl: 16 B per walker, no staging copies), then ONE kernel that judges every pair on
 * the gathered snapshot and writes this rank's own temperatures, d_temps[nchains], in place.
 * d_pairs[npairs][2], d_log_u[npairs] replicated on every rank; a pair with an id outside [0, nranks * nchains) is
 * ignored (nothing read, nothing moved). */
int rf_pt_swap_allgather_device(rf_ctx *ctx, int32_t nchains, int32_t npairs, const int32_t *d_pairs,
                                const double *d_log_u, double *d_temps, const double *d_logl, void *stream);
/* the kernel of the form above on arrays some other transport gathered (the host's MPI / gloo when ranks share a
 * GPU and RCCL cannot form a communicator): d_g_temps / d_g_logl [nranks * nchains] indexed by global id =
 * rank * nchains + chain (:508-511), read only; d_temps[nchains] = this rank's temperatures, updated in place;
 * d_accepted[npairs] may be NULL.  Needs no communicator. */
int rf_pt_swap_gathered_device(rf_ctx *ctx, int32_t nchains, int32_t rank, int32_t nranks, int32_t npairs,
                               const int32_t *d_pairs, const double *d_log_u, const double *d_g_temps,
                               const double *d_g_logl, double *d_temps, int32_t *d_accepted, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RFGPU_H */
