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
 */
#ifndef RFGPU_H
#define RFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFGPU_ABI_VERSION 5

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
 * (r_inv(nsmp, nsmp, ntrc) column-major). */
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

/* Pinned (page-locked, device-mapped) host memory.  Host arrays handed to rf_eval_batch / rf_eval_models from such
 * memory go to the device by DMA as they are; pageable arrays are first copied into the context's own pinned staging
 * area (one host memcpy per array and call: ~1 KB per chain at k_max 30).  Optional; any host memory works. */
int rf_host_alloc(size_t bytes, void **ptr);
int rf_host_free(void *ptr);
/* The same for host ranks that SHARE a GPU: POSIX shared memory mapped by every rank of the node (same `name`, which
 * starts with '/'; exactly one rank passes create != 0 and returns before the others call -- the host's barrier) and
 * registered with the GPU in the rank whose context reads / writes it (gpu != 0), so that one rank can hand the
 * proposals of all of them to its context in ONE call, by DMA.  rf_host_free_shared unmaps (the creator also unlinks). */
int rf_host_alloc_shared(const char *name, size_t bytes, int32_t create, int32_t gpu, void **ptr);
int rf_host_unlink_shared(void *ptr);   /* creator, once every rank has mapped the block: the name goes, the mappings stay */
int rf_host_free_shared(void *ptr);
/* This process will issue no more GPU work: give its queues back (hipDeviceReset).  Every context of the process must
 * have been destroyed.  For host ranks of a GPU group other than its first: many processes holding idle queues on one
 * GPU make the hardware scheduler time-slice them, and the rank that launches waits for its turn. */
int rf_release_gpu(void);

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
int rf_post_reset(rf_ctx *ctx);                 /* every set */
/* Accumulator SETS: a context that evaluates the chains of several host ranks (ranks sharing a GPU) keeps one set of
 * accumulators per rank, so that each rank ends with the arrays it would have filled alone and the reference's
 * output_results (src/mcmc_out.f90:52-93) reduces and gathers them unchanged.  rf_post_sets(n) before rf_post_create
 * (default 1); rf_post_select(i) names the set that the following rf_post_record* / rf_post_read / rf_comm_post_*
 * calls address (stream-ordered like the calls themselves). */
int rf_post_sets(rf_ctx *ctx, int32_t nsets);
int rf_post_select(rf_ctx *ctx, int32_t set);

/* Record n chains, in order, exactly as n consecutive passes through src/pt_mcmc.f90:204-286
 * would (the fp64 sums vp_mean / vs_mean / vpvs_mean -- and the ocean-layer ASSIGNMENTS
 * :263-264 -- are applied in that order, so they are bit-identical to the reference's).
 * walker_ids[n] name the chains whose CURRENT trace feeds the amplitude histogram;
 * k[n], z[n][k_max-1], dvp[n][k_max], dvs[n][k_max], sig[n][ntrc], logl[n] are the chains'
 * current state.  temps (may be NULL) applies the reference's filter temp <= 1 + 1e-6 (:204)
 * on the device: chains above it are skipped.  The caller applies the iteration filter
 * (iter > nburn, mod(iter, ncorr) == 0).  _device: all pointers are device pointers and the
 * call is asynchronous on `stream`; the host variant copies pageable arrays before it returns (they may be changed at
 * once; arrays in PINNED memory -- rf_host_alloc, rf_host_alloc_shared -- are read by DMA in place after it returns:
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
 *       staging area (pageable memory); 0 = all of them travelled by DMA from the caller's own pinned arrays (rf_host_alloc) */
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
#endif /* RFGPU_H */
