// Internal launcher interface between the C-ABI host layer (rfgpu_api.cpp) and the
// gfx950 kernels (rfgpu_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rfgpu {

struct CommState;   // rfgpu_comm.cpp

// doubles of per-layer constants written by stage_kernel (layout: the comment above K1 in rfgpu_kernels.hip)
constexpr int NCOEF = 24;
// doubles per (batch item, forward-trace) of walker constants written by stage_kernel
constexpr int GTAIL = 24;

struct DeviceTables {
    int nfft, nh, ntrc, nfwd, nsmp, deconv_mode, ray_common;
    double delta, t_start, sdep, domg, omg_dc;
    const double *flt;      // [ntrc][nh]
    const double *obs;      // [ntrc][nsmp]
    const double *r_inv_t;  // [ntrc][nsmp*nsmp]  r_inv(i,j) at [i*nsmp + j] (transposed image)
    const double *rayps;    // [ntrc]
    const int *ipha;        // [ntrc]
    const double2 *twiddle; // [nfft]  exp(+2 pi i k / nfft), the full turn (power-of-two nfft: the in-LDS FFT)
    const double2 *twiddle_any; // [nfft] exp(+2 pi i k / nfft) for any other nfft (direct DFT, trace_anyn_kernel), else nullptr
    const int *nh_active;   // [ntrc] bins with a non-negligible filter weight, or nullptr (all bins)
    // Long time windows (nsmp beyond what phi_deferred_kernel's LDS holds): every kernel that ends with a trace writes
    // its misfits to HBM (rows zero-padded to mis_stride) and the quadratic forms of the whole batch are ONE tiled
    // GEMM on the FP64 matrix cores (phi_gemm_kernel).  The trace kernels then keep no misfits / partial sums in LDS.
    int phi_gemm;           // 1: the plan above (fixed per context from nsmp, never per launch)
    int lds_nsmp;           // nsmp as far as the trace kernels' LDS layouts are concerned: nsmp, or 0 with phi_gemm
    int mis_stride;         // doubles between consecutive misfit rows of WalkerState::misfit: nsmp, or kp with phi_gemm
};

// tables of the long-window plan (phi_gemm_kernel)
struct PhiGemmTables {
    const double *rg;       // [ntrc][kp][np] R^-1(i, j) at [i][j], zero-padded: kp = nsmp rounded up to 16, np to 128
    const double *rt;       // the same quadratic form's upper triangle: R^-1(i, j) + R^-1(j, i) for i < j, the diagonal, 0 below
    double *part;           // [ntrc][nchunk][pstride] per 64-column chunk partial sums of misfit . R^-1 . misfit
    int kp, np, nchunk, pstride;
    int triangle;           // "gemm_triangle": 1 (default) the triangular image and half the K loop | 0 the full product
    int tile;               // "gemm_tile": 0 (default) / 64: 128 x 64 blocks | 128: 128 x 128 blocks (same values)
    int num_cu;
};

struct BatchArgs {
    int nb, nlay_pad;
    const int *walker_ids; // [nb]
    const int *fwd_flag;   // [nb] or nullptr
    const int *nlay;       // [nb]
    const double *layers;  // [nb][4][nlay_pad]
    const double *sig;     // [nb][ntrc]
    double *logl;          // [nb]
    const int *order;      // [nb] dispatch order (deepest walkers first) or nullptr
};

struct WalkerState {
    double *rft;      // [2][nslots][ntrc][trace_len]
    int trace_len;    // samples kept of every trace: nfft, or nsmp with the "trace_window" option
    double *phi;      // [2][nslots][ntrc]
    int *cur_slot;    // [nslots] 0/1: which half holds the current trace
    int *prop_fwd;    // [nslots] last proposal ran the forward model
    int *done;        // [nslots] per batch item: traces finished (last one forms logL), self-resetting
    double *misfit;   // [nslots][ntrc][mis_stride] per batch item: misfits handed to phi_deferred_kernel / phi_gemm_kernel
    double *gcoef;    // [nslots * nfwd][nlay_max][NCOEF] per batch item: stage_kernel's per-layer constants
    double *gtail;    // [nslots * nfwd][GTAIL]           ... walker constants + direct-arrival time
    int *gflag;       // [nslots * nfwd]                  ... bit 0 sea, bit 1 phases beyond the fast sincos range
    int *item_state;  // [nslots] per batch item, by stage_kernel: 1 evaluate, 0 sigma-only, -1 skipped, -2 refused (bad input)
    int *err;         // [4] error word in device-mapped host memory: {reason (0 none), batch item, offending value, -}
    int nslots;
};

// K1: propagator-matrix spectra  -> spec[nb][nfwd][2][nh] (freq_r, freq_v after
// the conj / -conj of forward.f90:145-146)
void launch_spectra(const DeviceTables &t, const BatchArgs &b, double2 *spec, int nsplit, int chain,
                    int waves_per_block, int *slow_list, int *slow_count, const WalkerState &w, hipStream_t s);
// K2: decon / filter / c2r / shift / normalise / write trace / quadratic form
// (xbuf: nullptr, or the scratch of the long non-power-of-two variant, trace_anyn_scratch_entries per block)
void launch_trace(const DeviceTables &t, const BatchArgs &b, const double2 *spec,
                  const WalkerState &w, int *slow_count, double2 *xbuf, int defer_logl, hipStream_t s);
// log-likelihood from cached quadratic forms (used for host-owned traces; the batched path
// forms logL inside trace_kernel)
void launch_logl(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, hipStream_t s);
// fused K1+K2 (contexts with one forward computation per trace)
void launch_fused(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, int chain, int *slow_count,
                  int ablate, int defer_logl, int *order_next, double *extra_out, hipStream_t s);
// the fused kernel with 512-thread blocks (nfft 4096, land): see fused8_kernel
void launch_fused8(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, int *slow_count, int ablate,
                   int defer_logl, int *order_next, double *extra_out, hipStream_t s);
size_t fused8_lds_bytes(int nsmp, int nlay_pad);
// "single FWD mode" (common rays, several traces) in one launch: one 512-thread block per walker, one propagator
// pass, ntrc trace tails from registers (nfft 4096, land): see fusedc_kernel
void launch_fusedc(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, int *slow_count, int ablate,
                   int defer_logl, int *order_next, double *extra_out, hipStream_t s);
// K0: per-(item, forward-trace) constants of the propagator, once per batch item, in front of K1 / the fused kernel
void launch_stage(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, hipStream_t s);
// logL of a batch launched with defer_logl (one thread per batch item, after the fused kernel)
void launch_logl_deferred(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, hipStream_t s);
size_t phi_deferred_lds_bytes(int nsmp);
// long windows: quadratic forms of the batch as one FP64-MFMA GEMM + logL (phi_gemm_kernel, phi_gemm_finish_kernel)
void launch_phi_gemm(const DeviceTables &t, const BatchArgs &b, const WalkerState &w, const PhiGemmTables &g, hipStream_t s);
// misfits of the scratch walker's proposal slot -> misfit row of batch item 0 (host-owned traces, long windows)
void launch_misfit_of_trace(const DeviceTables &t, const WalkerState &w, int walker, hipStream_t s);
size_t fused_lds_bytes(int nfft, int nsmp, int nlay_pad);
void launch_phi(const DeviceTables &t, const WalkerState &w, int walker, hipStream_t s);
struct ModelConfig {
    int k_max, vp_mode, nref;
    double sdep, z_max, h_min, z_ref_min, dz_ref;
    double vp_min, vp_max, vs_min, vs_max, vpvs_min, vpvs_max;
    const double *vp_ref, *vs_ref;   // device
};
struct FormatParams {
    ModelConfig m;
    int nb, nlay_pad;
    const int *k;
    const double *z, *dvp, *dvs;
    const int *fwd_in;
    int *nlay;
    double *layers;
    int *flag;
    int *valid;
    double *scratch;
    int ldz;            // doubles between the z rows of consecutive items; 0 = k_max - 1 (packed)
};
void launch_format_model(const FormatParams &P, hipStream_t s);
void launch_order(int nb, const int *nlay, const int *fwd_flag, int *order, hipStream_t s);
void launch_gather_rft(const WalkerState &w, int ntrc, int nfft, int n, const int *walker_ids, int which, int nout,
                       double *out, hipStream_t s);
void launch_commit(const WalkerState &w, int nb, const int *walker_ids, const int *accept,
                   int ntrc, hipStream_t s);
void launch_pt_swap(int npairs, const int *pairs, const double *log_u, double *temps,
                    const double *logl, int *accepted, hipStream_t s);
void launch_pt_swap_gathered(int npairs, const int *pairs, const double *log_u, const double *g_temps,
                             const double *g_logl, int nchains, int rank, int nranks, double *temps, int *accepted,
                             hipStream_t s);

// ---- posterior accumulation (rfgpu_posterior.hip) --------------------------------
struct PostConfig {
    int nbin_z, nbin_vs, nbin_vp, nbin_vpvs, nbin_sig, nbin_amp;
    int k_max, ntrc, nsmp, nfft;
    double amp_min, dbin_amp, z_min, dbin_z, dbin_vp, dbin_vs, dbin_vpvs;
    double vp_min, vs_min, vpvs_min;
    const double *sig_min, *dbin_sig;   // [ntrc] device
    const int *sig_mode;                // [ntrc] device
    long long max_models;
};
struct PostState {
    int *nmod;                          // [2]: models recorded, base index of the batch in flight
    int *nk, *nz, *nsig, *namp, *nvpz, *nvsz, *nvpvsz;
    double *vp_mean, *vs_mean, *vpvs_mean, *vp_model, *vs_model, *all_likelihood;
    long long *amp_oor;
    int *sel, *nsel;                    // [nslots], [1]: batch items that pass the temperature filter
    double *row_a, *row_b;              // [nslots][nbin_z] per-model profile rows of the batch
};
struct PostBatch {
    int n;
    const int *walker_ids, *k;
    const double *z, *sig, *logl, *temps;
    const int *nlay;                    // from format_model
    const double *layers;
    int nlay_pad;
};
void launch_post_record(const PostConfig &c, const PostState &st, const PostBatch &b, const WalkerState &w,
                        hipStream_t s);
void launch_post_mark_unused(const PostConfig &c, const PostState &st, hipStream_t s);

// ---- long series (trace_long_kernel): a power of two beyond 8192, or Bluestein for any other long nfft --------
struct LongTables {
    int m, log2n2;               // M = 4096 << log2n2; log2n2 = 1 .. 4
    int bluestein;               // 0: nfft == M (four-step transform only); 1: Bluestein on top
    const double2 *tw_m;         // [M] exp(+2 pi i k / M), second half the exact negative of the first
    const double2 *chirp;        // [nfft] exp(+i pi m^2 / nfft)            (Bluestein)
    const double2 *bhat;         // [M] DFT_M of the wrapped conj chirp       (Bluestein)
    double2 *scratch;            // [rows][2][row_entries]
    size_t row_entries;          // >= max(M, fft_pad(nfft) + 1, 2 nsmp)
};
size_t long_row_entries(int nfft, int m, int nsmp);
size_t trace_long_lds_bytes(int nsmp);
void launch_trace_long(const DeviceTables &t, const BatchArgs &b, const double2 *spec, const WalkerState &w, int *slow_count,
                       const LongTables &L, int rows, int defer_logl, hipStream_t s);

size_t spectra_lds_bytes(int nlay_pad);
size_t trace_anyn_lds_bytes(int nfft, int nsmp, int nlay_pad);
size_t trace_anyn_big_lds_bytes(int nfft, int nsmp);
__host__ __device__ size_t trace_anyn_scratch_entries(int nfft, int nsmp);
size_t trace_lds_bytes(int nfft, int nsmp, int nlay_pad);

} // namespace rfgpu
