// rfgpu_api.cpp -- host side of the C ABI declared in include/rfgpu.h and rfgpu_ext.h.
// Context management, init-time tables (filter, twiddles, R^-1), staging of host
// buffers and kernel launches.  No CPU fallback: every evaluation runs the gfx950
// kernels of rfgpu_kernels.hip.
#include "rfgpu_internal.h"
#include "../../include/rfgpu_ext.h"


#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace rfgpu;

static thread_local std::string g_err;

static int fail(const std::string &msg)
{
    g_err = msg;
    return 1;
}
namespace rfgpu {
int comm_fail(const std::string &msg) { return fail(msg); }
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

struct rf_ctx {
    rfgpu::CommState *comm = nullptr;   // RCCL communicator of the temperature exchange (rfgpu_comm.cpp), or none
    rf_config cfg{};
    int device = 0;
    int nh = 0, nfwd = 1, ray_common = 1, nslots = 0;
    DeviceTables tab{};
    PhiGemmTables pg{};       // long windows (tab.phi_gemm): the zero-padded R^-1 image and the partial sums of phi_gemm_kernel
    WalkerState ws{};
    hipStream_t stream = nullptr;
    // owned device allocations
    std::vector<void *> owned;
    double2 *spec = nullptr; // [nslots][nfwd][2][nh]
    double2 *anyn_scratch = nullptr; // long non-power-of-two nfft only: [nslots * ntrc][trace_anyn_scratch_entries]
    bool long_series = false;        // trace_long_kernel: 2^n beyond 8192 (four-step transform) or Bluestein
    LongTables longt{};
    int long_rows = 0;               // resident blocks of trace_long_kernel = scratch rows
    int *slow_list = nullptr, *slow_count = nullptr; // walkers deferred to the generic-sincos kernel
    // staging for host-buffer calls
    int *d_ids = nullptr, *d_fwd = nullptr, *d_nlay = nullptr, *d_acc = nullptr;
    double *d_layers = nullptr, *d_sig = nullptr, *d_logl = nullptr;
    int stage_nb = 0, stage_pad = 0;
    // Host buffers of the host-pointer entry points travel by DMA from PINNED memory: the caller's own arrays when
    // they come from rf_host_alloc, else a copy in this arena (a bump allocator, rewound by every call that uses it).
    // Outputs (logL, valid) are written by the kernels straight into device-mapped pinned memory (h_out).
    // One arena and one output region per evaluation that can be in flight (rf_eval_models_begin / rf_eval_wait)
    // plus one for the synchronous calls and rf_commit (index RF_EVAL_MAX_IN_FLIGHT).
    struct Arena {
        char *p = nullptr;
        size_t cap = 0, off = 0;
        hipEvent_t ev = nullptr;  // recorded after the last asynchronous use (an evaluation in flight; rf_commit returns early)
        bool pending = false;
        int staged = 0;           // host arrays of the current call that were copied here (pageable), not DMA'd in place
    } arena[RF_EVAL_MAX_IN_FLIGHT + 2];   // [.. + 1]: rf_post_record's own (it returns early too)
    struct Ticket {
        bool busy = false;
        int nb = 0;
        bool want_valid = false;
    } ticket[RF_EVAL_MAX_IN_FLIGHT];
    int ticket_next = 0;
    double *h_out = nullptr, *d_out = nullptr;   // [RF_EVAL_MAX_IN_FLIGHT + 1][2][out_cap]: logL | valid (as int) per region; host pointer and its device alias
    int out_cap = 0;
    // rf_eval_models_begin: one set of device input arrays per evaluation that can be in flight, filled on a stream of
    // its own -- the transfers of one evaluation run under the kernels of the one before it (two pipeline segments of a
    // sampler: ~85 us of DMA and gaps per 4096-chain segment at the C4 shape no longer sit between the kernels)
    struct SlotIn {
        int *ids = nullptr, *fwd = nullptr, *k = nullptr;
        double *z = nullptr, *dvp = nullptr, *dvs = nullptr, *sig = nullptr;
        int cap = 0;
        hipEvent_t copied = nullptr;
    } slot_in[RF_EVAL_MAX_IN_FLIGHT];
    hipStream_t copy_stream = nullptr;
    bool use_copy_stream = false;     // "copy_stream"
    // device images of the proposals of rf_eval_models_device / rf_format_models_device: k | z(ldz = k_max) | dvp | dvs
    int *d_m_k = nullptr;
    double *d_m_z = nullptr, *d_m_dvp = nullptr, *d_m_dvs = nullptr;
    // device format_model (row f-2): model tables + per-batch outputs
    bool have_model = false;
    ModelConfig model{};
    int fm_pad = 0;
    int *d_fm_nlay = nullptr, *d_fm_flag = nullptr;
    double *d_fm_layers = nullptr, *d_fm_scratch = nullptr;
    int *d_nh_active = nullptr;   // [ntrc] "bin_cutoff": bins with a non-negligible filter weight
    int *h_err = nullptr;     // [4] the error word stage_kernel's input check writes (device-mapped host memory; ws.err is its device alias)
    int *d_order = nullptr;   // [nslots] LPT dispatch order of the current batch
    int *d_order_alt = nullptr;   // [nslots] the order the running launch computes for the next one
    int order_next_nb = 0;    // d_order_alt holds an order for a batch of this size (0: none)
    bool order_reuse = true;  // "order_reuse" = 0: a fresh order_kernel before every launch
    // posterior accumulators (row f-3)
    bool have_post = false;
    PostConfig post{};
    PostState pst{};                  // the accumulators rf_post_record* / rf_post_read / rf_comm_post_* address
    std::vector<std::pair<void *, size_t>> post_zero;   // accumulators cleared by rf_post_reset
    int *d_post_nlay = nullptr, *d_post_flag = nullptr, *d_post_k = nullptr;
    double *d_post_layers = nullptr, *d_post_scratch = nullptr;
    double *d_post_in = nullptr;   // staging of rf_post_record's host arrays
    // launch-plan options (rf_set_option; every combination computes the same results)
    bool lpt = true;
    int nsplit_override = 0;  // "nsplit"
    int chain_override = -1;  // "chain": -1 = by shape
    bool fused_allowed = false;   // the context's shape admits the fused kernel
    bool fusedc_allowed = false;  // ... the common-ray fused kernel (several traces of one ray: nfft 4096, land)
    bool fusedc = false;          // one launch per batch: a block per walker, one propagator pass, ntrc trace tails
    int fused_override = -1;  // "fused": -1 = by shape
    int defer_logl = -1;      // "defer_logl": -1 = by batch size, 0 / 1 = never / always
    int block_threads = 0;    // "block_threads": 0 = by batch size, 256 / 512 = fused_kernel / fused8_kernel
    int fused8_max_rounds = 3;   // contexts of up to this many rounds of blocks (2 blocks per CU) take fused8_kernel
                                 // (measured at the end of round 3, 8-wave vs 4-wave: two rounds +2.7 %, three +1.5 %,
                                 // four -3.8 %, six -4 %, C3's sixteen and C4's forty-eight -3..4 %)
    double bin_cutoff = 0.0;  // "bin_cutoff": opt-in filter-support cut-off (0 = off: every bin like the reference)
    int trace_window = 0;     // "trace_window": 1 = only samples 1 .. nsmp of every trace are stored
    int last_staged = 0;      // input arrays of the last rf_eval_batch / rf_eval_models(_begin) that went through the pinned
                              // arena (pageable memory); 0 = every one travelled by DMA from the caller's own pinned array
    int n_overrides = 0;      // options set away from their defaults (echoed by rf_get_launch_plan)
    int ablate = 0;           // RFGPU_DIAGNOSTICS builds only ("ablate"): stops the kernel early, results invalid
    double *h_single_in = nullptr, *h_single_out = nullptr;   // pinned staging of the per-call drop-in
    double *d_single_in = nullptr, *d_single_out = nullptr;
    double *single_trace_out = nullptr;   // set by rf_calc_likelihood around its run_batch: extra trace copy
    double *d_gather = nullptr;
    size_t gather_bytes = 0;
    // host copies of tables
    std::vector<double> flt, r_inv;
    std::vector<int> r_inv_rank;        // per trace: rank of the library-built pseudo-inverse (-1: caller's r_inv)
    std::vector<double> r_inv_gap;      // per trace: relative gap at the 1e-3 cut-off (NaN: caller's r_inv)
    // launch policy
    bool fused = false;       // one launch for spectra + trace (needs one forward computation per trace)
    int chain = 0;            // bins per phase chain in the spectra kernel (0: direct sincos)
    int waves_per_block = 4;  // waves of one walker sharing a staged layer stack
    int num_cu = 256;
    // profiling
    // profiling: a pool of event quads so that timing never synchronises inside a
    // timed loop (flushed lazily / when the pool is exhausted)
    bool prof = false;
    int prof_every = 1;        // time every prof_every-th batch (event records cost ~4 us each on the stream)
    long long prof_batch = 0;  // batches seen while profiling
    bool prof_this = false;    // the batch being launched is a timed one
    struct Timed { int kind; hipEvent_t e0, e1; };
    std::vector<Timed> ev_pool;
    size_t ev_used = 0;
    double prof_ms[3] = {0, 0, 0};
    int64_t prof_n[4] = {0, 0, 0, 0};   // batches, spectra / trace / logl kernel launches
};

namespace rfgpu {
hipStream_t ctx_stream(rf_ctx *c) { return c->stream; }
int ctx_device(rf_ctx *c) { return c->device; }
CommState *&ctx_comm(rf_ctx *c) { return c->comm; }
bool ctx_post(rf_ctx *c, const PostConfig **q, const PostState **st)
{
    *q = &c->post;
    *st = &c->pst;
    return c->have_post;
}
}

static int device_error(rf_ctx *c, const char *where);
static int ensure_spec(rf_ctx *c);

extern "C" const char *rf_last_error(void) { return g_err.c_str(); }
extern "C" int rf_abi_version(void) { return RFGPU_ABI_VERSION; }

// ---------------------------------------------------------------------------
// init_r_inv for one trace (reference src/likelihood.f90:183-222) with a one-sided
// Jacobi SVD (Hestenes) in fp64.  r_inv(i,j) column-major.
// ---------------------------------------------------------------------------
extern "C" int rf_compute_r_inv(int32_t nsmp, double a_gus, double delta, double *r_inv, int32_t *rank_out,
                                double *cut_gap_out)
{
    if (nsmp <= 0 || !r_inv) return fail("rf_compute_r_inv: bad arguments");
    const int n = nsmp;
    const double r = std::exp(-(a_gus * a_gus) * (delta * delta)); // :183
    // U holds the working columns (starts as R), V accumulates the rotations
    std::vector<double> U((size_t)n * n), V((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int d = (i - j) * (i - j);
            U[(size_t)j + (size_t)n * i] = std::pow(r, (double)d); // :185-190
        }
    for (int i = 0; i < n; ++i) V[(size_t)i + (size_t)n * i] = 1.0;
    const double eps = 1e-15;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double *up = &U[(size_t)n * p], *uq = &U[(size_t)n * q];
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < n; ++i) {
                    alpha += up[i] * up[i];
                    beta += uq[i] * uq[i];
                    gamma += up[i] * uq[i];
                }
                if (gamma == 0.0) continue;
                const double lim = eps * std::sqrt(alpha * beta);
                if (std::fabs(gamma) <= lim) continue;
                off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                double *vp = &V[(size_t)n * p], *vq = &V[(size_t)n * q];
                for (int i = 0; i < n; ++i) {
                    const double a = up[i], b = uq[i];
                    up[i] = c * a - s * b;
                    uq[i] = s * a + c * b;
                    const double va = vp[i], vb = vq[i];
                    vp[i] = c * va - s * vb;
                    vq[i] = s * va + c * vb;
                }
            }
        if (off < 1e-14) break;
    }
    // singular values = column norms; R+ = sum_{s_k > 1e-3} v_k u_k^T / s_k  (:212-222)
    std::fill(r_inv, r_inv + (size_t)n * n, 0.0);
    int rank = 0;
    // process in descending singular-value order for a deterministic summation order
    std::vector<std::pair<double, int>> sv(n);
    for (int k = 0; k < n; ++k) {
        double nn = 0;
        for (int i = 0; i < n; ++i) nn += U[(size_t)n * k + i] * U[(size_t)n * k + i];
        sv[k] = {std::sqrt(nn), k};
    }
    std::sort(sv.begin(), sv.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) {
        return a.first > b.first;
    });
    // relative distance of the nearest singular value to the hard cut-off s > 1e-3 (:214): the rank -- and with
    // it every value of the pseudo-inverse -- is only as well defined as this gap is wide compared with the
    // rounding of an SVD (~1e-15 * s_max absolute, ~1e-11 relative to the cut-off for these matrices)
    double gap = HUGE_VAL;
    for (int kk = 0; kk < n; ++kk) gap = std::min(gap, std::fabs(sv[kk].first - 1.0e-3) / 1.0e-3);
    if (cut_gap_out) *cut_gap_out = gap;
    for (int kk = 0; kk < n; ++kk) {
        const double s = sv[kk].first;
        const int k = sv[kk].second;
        if (!(s > 1.0e-3)) continue; // :214
        ++rank;
        const double *uk = &U[(size_t)n * k], *vk = &V[(size_t)n * k];
        const double inv = 1.0 / (s * s); // u_k (unit) = U[:,k] / s, times 1/s
        for (int j = 0; j < n; ++j) {
            const double uj = uk[j] * inv;
            for (int i = 0; i < n; ++i) r_inv[(size_t)i + (size_t)n * j] += vk[i] * uj;
        }
    }
    if (rank_out) *rank_out = rank;
    return 0;
}

// ---------------------------------------------------------------------------
static int dev_alloc(rf_ctx *c, void **p, size_t bytes)
{
    HIP_TRY(hipMalloc(p, bytes ? bytes : 8));
    c->owned.push_back(*p);
    return 0;
}

template <class T>
static int upload(rf_ctx *c, const std::vector<T> &h, const T **d)
{
    void *p = nullptr;
    if (dev_alloc(c, &p, h.size() * sizeof(T))) return 1;
    HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *d = static_cast<const T *>(p);
    return 0;
}

// device image of r_inv: r_inv(i,j,t) at [(t*nsmp + i)*nsmp + j] (row-major per trace)
static std::vector<double> transpose_r_inv(const std::vector<double> &r, int ntrc, int nsmp)
{
    std::vector<double> out(r.size());
    for (int t = 0; t < ntrc; ++t)
        for (int j = 0; j < nsmp; ++j)
            for (int i = 0; i < nsmp; ++i)
                out[((size_t)t * nsmp + i) * nsmp + j] = r[((size_t)t * nsmp + j) * nsmp + i];
    return out;
}

// long windows: R^-1(i, j, t) at [(t * kp + i) * np + j], zero beyond nsmp (phi_gemm_kernel's B operand)
static std::vector<double> pad_r_inv(const std::vector<double> &r, int ntrc, int nsmp, int kp, int np)
{
    std::vector<double> out((size_t)ntrc * kp * np, 0.0);
    for (int t = 0; t < ntrc; ++t)
        for (int j = 0; j < nsmp; ++j)
            for (int i = 0; i < nsmp; ++i)
                out[((size_t)t * kp + i) * np + j] = r[((size_t)t * nsmp + j) * nsmp + i];
    return out;
}

// The same quadratic form through its upper triangle: m R m^T = sum_j m_j sum_{i <= j} m_i T(i, j) with T(i, j) =
// R(i, j) + R(j, i) above the diagonal, R(j, j) on it, 0 below -- exact for ANY R (only the symmetric part of a matrix
// enters its quadratic form), so nothing is assumed about how symmetric the SVD left r_inv.  Column block c of the
// product then needs rows 0 .. 64 (c + 1) - 1 only: half the multiply-adds and half the matrix traffic.
static std::vector<double> triangle_r_inv(const std::vector<double> &rg, int ntrc, int kp, int np)
{
    std::vector<double> out(rg.size(), 0.0);
    for (int t = 0; t < ntrc; ++t) {
        const double *r = rg.data() + (size_t)t * kp * np;
        double *o = out.data() + (size_t)t * kp * np;
        for (int i = 0; i < kp; ++i)
            for (int j = i; j < kp; ++j) o[(size_t)i * np + j] = i == j ? r[(size_t)i * np + i] : r[(size_t)i * np + j] + r[(size_t)j * np + i];
    }
    return out;
}

static int ensure_stage(rf_ctx *c, int nb, int pad)
{
    if (nb <= c->stage_nb && pad <= c->stage_pad) return 0;
    const int nnb = std::max(nb, c->stage_nb), npad = std::max(pad, c->stage_pad);
    // (old staging buffers stay owned by the context until destroy; growth is rare)
    void *p;
    if (dev_alloc(c, &p, sizeof(int) * nnb)) return 1;
    c->d_ids = (int *)p;
    if (dev_alloc(c, &p, sizeof(int) * nnb)) return 1;
    c->d_fwd = (int *)p;
    if (dev_alloc(c, &p, sizeof(int) * nnb)) return 1;
    c->d_nlay = (int *)p;
    if (dev_alloc(c, &p, sizeof(int) * nnb)) return 1;
    c->d_acc = (int *)p;
    if (dev_alloc(c, &p, sizeof(double) * (size_t)nnb * 4 * npad)) return 1;
    c->d_layers = (double *)p;
    if (dev_alloc(c, &p, sizeof(double) * (size_t)nnb * c->cfg.ntrc)) return 1;
    c->d_sig = (double *)p;
    if (dev_alloc(c, &p, sizeof(double) * nnb)) return 1;
    c->d_logl = (double *)p;
    c->stage_nb = nnb;
    c->stage_pad = npad;
    return 0;
}

// ---- pinned staging of host buffers ---------------------------------------------------------------------------
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();       // pageable memory is "an invalid value" to this query: not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

// Start of a host-pointer call: makes room for `bytes`, rewinds.  An earlier asynchronous user of the arena (its copies
// may still be queued behind running kernels) is waited for only when this call really writes into the arena -- the
// first arena_take: a call whose arrays are all pinned transfers them in place and never touches it (a GPU group's
// first rank records the posterior of every rank of the group back to back: each call used to wait for the previous
// one's copies, i.e. for the evaluation running in front of them).
static int arena_begin(rf_ctx::Arena &A, size_t bytes)
{
    bytes += 8 * 256;                  // alignment slack of up to eight takes
    if (bytes > A.cap) {
        if (A.pending) {
            HIP_TRY(hipEventSynchronize(A.ev));
            A.pending = false;
        }
        if (A.p) HIP_TRY(hipHostFree(A.p));
        A.p = nullptr;
        A.cap = 0;
        const size_t cap = std::max(bytes + bytes / 2, (size_t)1 << 16);
        HIP_TRY(hipHostMalloc((void **)&A.p, cap, hipHostMallocDefault));
        A.cap = cap;
    }
    A.off = 0;
    A.staged = 0;
    return 0;
}

static void *arena_take(rf_ctx::Arena &A, size_t bytes)
{
    if (A.pending) {
        (void)hipEventSynchronize(A.ev);
        A.pending = false;
    }
    const size_t at = (A.off + 255) & ~(size_t)255;
    A.off = at + bytes;
    return A.p + at;                   // (arena_begin sized the arena for every take of the call)
}

// after the last stream operation that reads the arena: its next user waits for this
static int arena_mark(rf_ctx::Arena &A, hipStream_t s)
{
    if (!A.ev) HIP_TRY(hipEventCreateWithFlags(&A.ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(A.ev, s));
    A.pending = true;
    return 0;
}

// host -> device on stream s: DMA straight from the caller's array when it is pinned, else through the arena
static int h2d(rf_ctx::Arena &A, void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (!bytes) return 0;
    const void *from = src;
    if (!is_pinned_host(src)) {
        void *stage = arena_take(A, bytes);
        std::memcpy(stage, src, bytes);
        from = stage;
        ++A.staged;
    }
    HIP_TRY(hipMemcpyAsync(dst, from, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

// device-mapped pinned memory the kernels write a batch's logL (and validity flags) into: one region per evaluation
// that can be in flight + one for the synchronous calls
static int ensure_out(rf_ctx *c, int nb)
{
    if (nb <= c->out_cap) return 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_out) HIP_TRY(hipHostFree(c->h_out));
    c->h_out = nullptr;
    c->out_cap = 0;
    const int cap = std::max(nb, c->nslots);
    HIP_TRY(hipHostMalloc((void **)&c->h_out, sizeof(double) * 2 * (size_t)cap * (RF_EVAL_MAX_IN_FLIGHT + 1), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&c->d_out, c->h_out, 0));
    c->out_cap = cap;
    return 0;
}
static size_t out_region(const rf_ctx *c, int region) { return (size_t)region * 2 * c->out_cap; }   // in doubles

// The walkers' traces: [2][nslots][ntrc][len] -- len = nfft (the reference's rft(nfft, ntrc, nchains), filled
// completely), or nsmp with the "trace_window" option: the samples the likelihood, the histograms and make_syn ever
// read (src/likelihood.f90:88, src/pt_mcmc.f90:273-274).  (Re)allocated zero-filled: every stored trace is gone.
static int alloc_traces(rf_ctx *c, int len)
{
    if (c->ws.rft) {
        c->owned.erase(std::remove(c->owned.begin(), c->owned.end(), (void *)c->ws.rft), c->owned.end());
        HIP_TRY(hipFree(c->ws.rft));
        c->ws.rft = nullptr;
    }
    const size_t elems = 2 * (size_t)c->nslots * c->cfg.ntrc * (size_t)len;
    void *p = nullptr;
    if (dev_alloc(c, &p, sizeof(double) * elems)) return 1;
    c->ws.rft = (double *)p;
    c->ws.trace_len = len;
    HIP_TRY(hipMemset(p, 0, sizeof(double) * elems));
    HIP_TRY(hipDeviceSynchronize());   // (a memset of device memory may return early; c->stream is non-blocking: nothing else orders it)
    return 0;
}

// Launch plan from the context's shape, then the explicit options on top.
static void default_plan(rf_ctx *c)
{
    // chained phases pay off when a wave owns many 64-bin iterations (nfft 4096: 33); with
    // few iterations (nfft 256: 3) the direct path with more bin-splits is faster
    // (measured on MI355X: c4 spectra 4.18 -> 3.17 ms, c2 0.133 -> 0.119 ms, c1 0.035 vs 0.058 ms)
    const int niter = (c->nh + 63) / 64;
    c->chain = niter >= 16 ? 4 : 0;
    c->fused = c->fused_allowed && c->fused_override != 0;
    // common rays ("single FWD mode", forward.f90:59-91,141): ONE propagator pass feeds ntrc traces.  nfft 4096 on
    // land: fusedc_kernel keeps the spectra in registers (measured at the C4 shape: one launch instead of
    // spectra_kernel -> 537 MB of spectra in HBM -> trace_kernel); other shapes, or an explicit phase-chain
    // length, keep the split plan
    c->fusedc = c->fusedc_allowed && c->fused_override != 0 && c->chain_override < 0;
    // fused kernel: chains of 8 when that gives each of the block's four waves whole chunks (nfft 4096: 4 chunks of 8
    // iterations; the Nyquist bin comes from stage_kernel); measured C4 +7 %, C2 +2 % over 4.  Ocean (3 propagated
    // columns) too since round 3: the 8-bin loop fits 256 VGPRs without a spill inside it once the tail's twiddles
    // are formed as products (33 VGPRs spilled around it, block start and finish only) -- measured C5 shape +3.5 % over
    // chains of 4 (two per wave, each paying its own sincos).
    if (c->fused && c->chain == 4 && (niter / 8) >= 4 && (niter / 8) % 4 == 0) c->chain = 8;
    if (c->chain_override >= 0) c->chain = c->chain_override;
}

extern "C" int rf_ctx_create(const rf_config *cfg, rf_ctx **ctx_out)
{
    if (!cfg || !ctx_out) return fail("rf_ctx_create: null argument");
    *ctx_out = nullptr;
    const int n = cfg->nfft;
    if (n < 8) return fail("rf_ctx_create: nfft must be >= 8");
    const bool pow2 = (n & (n - 1)) == 0;
    if (cfg->ntrc < 1 || cfg->nsmp < 1 || cfg->nsmp > n) return fail("rf_ctx_create: bad ntrc / nsmp");
    if (cfg->deconv_mode != 0 && cfg->deconv_mode != 1) return fail("rf_ctx_create: deconv_mode must be 0 or 1");
    if (!cfg->rayps || !cfg->a_gus || !cfg->ipha || !cfg->obs) return fail("rf_ctx_create: null table");
    if (cfg->ldobs < cfg->nsmp) return fail("rf_ctx_create: ldobs < nsmp");
    if (cfg->max_walkers < 1 || cfg->nlay_max < 2) return fail("rf_ctx_create: bad capacity");
    for (int i = 0; i < cfg->ntrc; ++i)
        if (cfg->ipha[i] != 1 && cfg->ipha[i] != -1) return fail("rf_ctx_create: ipha must be +1 or -1");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail("rf_ctx_create: no HIP device available (librfgpu has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("rf_ctx_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(std::string("rf_ctx_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);

    rf_ctx *c = new rf_ctx();
    c->cfg = *cfg;
    c->device = cfg->device;
    c->num_cu = prop.multiProcessorCount;
    const int ntrc = cfg->ntrc, nsmp = cfg->nsmp, nh = n / 2 + 1;
    c->nh = nh;
    // check_ray (forward.f90:59-91)
    c->ray_common = 1;
    for (int i = 1; i < ntrc; ++i)
        if (cfg->rayps[i] != cfg->rayps[0] || cfg->ipha[i] != cfg->ipha[0]) c->ray_common = 0;
    c->nfwd = c->ray_common ? 1 : ntrc;
    c->nslots = cfg->max_walkers + 1; // last slot: scratch walker of rf_calc_rf

    auto cleanup = [&](int rc) {
        rf_ctx_destroy(c);
        return rc;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
        return cleanup(fail("hipStreamCreate failed"));

    // init_filter (forward.f90:95-119), fp64 on the host, same expression order
    const double pi = 3.1415926535897931;
    c->flt.resize((size_t)nh * ntrc);
    {
        const double df = 1.0 / (cfg->delta * n);
        for (int t = 0; t < ntrc; ++t) {
            const double fac_norm = n * cfg->a_gus[t] * cfg->delta / std::sqrt(pi);
            for (int i = 1; i <= nh; ++i) {
                const double omega = (i - 1) * 2.0 * pi * df;
                const double q = omega / (2.0 * cfg->a_gus[t]);
                c->flt[(size_t)(i - 1) + (size_t)nh * t] = std::exp(-(q * q)) / fac_norm;
            }
        }
    }
    // r_inv: supplied by the host (its own LAPACK dgesvd, the same products as src/likelihood.f90:212-222) or built here
    c->r_inv.resize((size_t)nsmp * nsmp * ntrc);
    c->r_inv_rank.assign(ntrc, -1);      // -1 / NaN: r_inv supplied by the caller
    c->r_inv_gap.assign(ntrc, std::nan(""));
    if (cfg->r_inv) {
        std::memcpy(c->r_inv.data(), cfg->r_inv, sizeof(double) * c->r_inv.size());
    } else {
        for (int t = 0; t < ntrc; ++t) {
            int same = -1;
            for (int u = 0; u < t; ++u)
                if (cfg->a_gus[u] == cfg->a_gus[t]) same = u;
            double *dst = c->r_inv.data() + (size_t)nsmp * nsmp * t;
            if (same >= 0) {
                std::memcpy(dst, c->r_inv.data() + (size_t)nsmp * nsmp * same, sizeof(double) * nsmp * nsmp);
                c->r_inv_rank[t] = c->r_inv_rank[same];
                c->r_inv_gap[t] = c->r_inv_gap[same];
            }
            else {
                int rank = 0;
                double gap = 0.0;
                if (rf_compute_r_inv(nsmp, cfg->a_gus[t], cfg->delta, dst, &rank, &gap)) return cleanup(1);
                if (!(gap >= RF_R_INV_MIN_CUT_GAP)) {
                    char msg[256];
                    snprintf(msg, sizeof msg,
                             "rf_ctx_create: trace %d: a singular value of the noise matrix lies within %.1e (relative) "
                             "of the 1e-3 rank cut-off (src/likelihood.f90:214): the pseudo-inverse is not well defined; "
                             "pass the host's own r_inv in rf_config", t + 1, gap);
                    return cleanup(fail(msg));
                }
                c->r_inv_rank[t] = rank;
                c->r_inv_gap[t] = gap;
            }
        }
    }
    std::vector<double> obs((size_t)nsmp * ntrc);
    for (int t = 0; t < ntrc; ++t)
        for (int i = 0; i < nsmp; ++i) obs[(size_t)i + (size_t)nsmp * t] = cfg->obs[(size_t)i + (size_t)cfg->ldobs * t];
    // exp(+2 pi i k / n), the full turn.  Power-of-two n (the in-LDS FFT): the second half is the exact negative of
    // the first, so that a butterfly reads its twiddle straight from the table (no sign select).
    std::vector<double2> tw((size_t)n);
    for (size_t k = 0; k < tw.size(); ++k) {
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * k / n;
        tw[k] = make_double2((double)cosl(a), (double)sinl(a));
    }
    if (pow2)
        for (size_t k = 0; k < (size_t)n / 2; ++k) tw[k + (size_t)n / 2] = make_double2(-tw[k].x, -tw[k].y);
    std::vector<double> rayps(cfg->rayps, cfg->rayps + ntrc);
    std::vector<int> ipha(cfg->ipha, cfg->ipha + ntrc);

    DeviceTables &T = c->tab;
    T.nfft = n; T.nh = nh; T.ntrc = ntrc; T.nfwd = c->nfwd; T.nsmp = nsmp;
    T.deconv_mode = cfg->deconv_mode; T.ray_common = c->ray_common;
    T.delta = cfg->delta; T.t_start = cfg->t_start; T.sdep = cfg->sdep;
    T.domg = 2.0 * pi / (n * cfg->delta);   // forward.f90:241
    T.omg_dc = (double)1.0e-5f;             // forward.f90:247 single-precision literal
    std::vector<double> r_inv_t = transpose_r_inv(c->r_inv, ntrc, nsmp);
    T.nh_active = nullptr;   // rf_set_option("bin_cutoff"): opt-in, off by default
    // Windows beyond what phi_deferred_kernel's LDS holds (nsmp > 191; the reference allows npts_max = 2000,
    // src/params.f90:44): the quadratic forms of a batch run as one GEMM on the FP64 matrix cores (phi_gemm_kernel).
    // Decided from nsmp alone, once: a chain evaluated alone, in a partial or in a full batch takes the same plan.
    T.phi_gemm = phi_deferred_lds_bytes(nsmp) > 60 * 1024 ? 1 : 0;
    T.lds_nsmp = T.phi_gemm ? 0 : nsmp;
    T.mis_stride = T.phi_gemm ? (nsmp + 15) / 16 * 16 : nsmp;
    if (T.phi_gemm) {
        PhiGemmTables &g = c->pg;
        g.kp = T.mis_stride;
        g.np = (nsmp + 127) / 128 * 128;
        g.nchunk = g.np / 64;
        g.pstride = c->nslots;
        g.tile = 0;
        g.triangle = 1;
        g.num_cu = c->num_cu;
        const std::vector<double> rg = pad_r_inv(c->r_inv, ntrc, nsmp, g.kp, g.np);
        const std::vector<double> rt = triangle_r_inv(rg, ntrc, g.kp, g.np);
        void *q = nullptr;
        if (upload(c, rg, &g.rg) || upload(c, rt, &g.rt) || dev_alloc(c, &q, sizeof(double) * (size_t)ntrc * g.nchunk * g.pstride)) return cleanup(1);
        g.part = (double *)q;
    }
    if (upload(c, c->flt, &T.flt) || upload(c, obs, &T.obs) || upload(c, r_inv_t, &T.r_inv_t) ||
        upload(c, rayps, &T.rayps) || upload(c, ipha, &T.ipha) || upload(c, tw, &T.twiddle))
        return cleanup(1);
    T.twiddle_any = nullptr;
    if (!pow2) {
        T.twiddle_any = T.twiddle;
        T.twiddle = nullptr;
    }

    // walker state
    void *p;
    c->ws.rft = nullptr;
    if (alloc_traces(c, n)) return cleanup(1);
    if (dev_alloc(c, &p, sizeof(double) * 2 * (size_t)c->nslots * ntrc)) return cleanup(1);
    c->ws.phi = (double *)p;
    (void)hipMemset(p, 0, sizeof(double) * 2 * (size_t)c->nslots * ntrc);
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return cleanup(1);
    c->ws.cur_slot = (int *)p;
    (void)hipMemset(p, 0, sizeof(int) * c->nslots);
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return cleanup(1);
    c->ws.prop_fwd = (int *)p;
    (void)hipMemset(p, 0, sizeof(int) * c->nslots);
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return cleanup(1);
    c->ws.done = (int *)p;
    (void)hipMemset(p, 0, sizeof(int) * c->nslots);
    if (dev_alloc(c, &p, sizeof(double) * (size_t)c->nslots * ntrc * T.mis_stride)) return cleanup(1);
    c->ws.misfit = (double *)p;
    // (long windows: the padding of a row is never written and must read as zero)
    if (hipMemset(p, 0, sizeof(double) * (size_t)c->nslots * ntrc * T.mis_stride) != hipSuccess) return cleanup(fail("hipMemset failed"));
    // stage_kernel's output (fused path): nlay_max * NCOEF doubles per (walker, forward-trace)
    if (dev_alloc(c, &p, sizeof(double) * (size_t)c->nslots * c->nfwd * cfg->nlay_max * NCOEF)) return cleanup(1);
    c->ws.gcoef = (double *)p;
    if (dev_alloc(c, &p, sizeof(double) * (size_t)c->nslots * c->nfwd * GTAIL)) return cleanup(1);
    c->ws.gtail = (double *)p;
    if (dev_alloc(c, &p, sizeof(int) * (size_t)c->nslots * c->nfwd)) return cleanup(1);
    c->ws.gflag = (int *)p;
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return cleanup(1);
    c->ws.item_state = (int *)p;
    (void)hipMemset(p, 0xff, sizeof(int) * c->nslots);     // "skipped" until a batch's stage_kernel says otherwise
    if (hipHostMalloc((void **)&c->h_err, sizeof(int) * 4, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->ws.err, c->h_err, 0) != hipSuccess)
        return cleanup(fail("rf_ctx_create: error word allocation failed"));
    c->h_err[0] = c->h_err[1] = c->h_err[2] = c->h_err[3] = 0;
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return cleanup(1);
    c->d_order = (int *)p;
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return cleanup(1);
    c->d_order_alt = (int *)p;
    c->ws.nslots = c->nslots;
    // (spec, the split launch plan's intermediate -- 8.6 GB at the C5 capacity -- is allocated by the first launch
    // that takes that plan: run_batch)
    if (dev_alloc(c, &p, sizeof(int) * ((size_t)c->nslots * c->nfwd + 1))) return cleanup(1);
    c->slow_count = (int *)p;      // [1]
    c->slow_list = (int *)p + 1;   // [nslots * nfwd]
    (void)hipMemset(p, 0, sizeof(int));

    // Long series: a power of two beyond 8192 runs the four-step transform (4096-point in-LDS transforms + a radix
    // n / 4096 stage through a scratch row per resident block); any other nfft beyond 2048 runs Bluestein's algorithm
    // on top of it (two power-of-two transforms of length M >= 2 nfft - 1) instead of the O(n^2) direct DFT
    c->long_series = (pow2 && n > 8192) || (!pow2 && n > 2048);
    if (c->long_series) {
        int m = n;
        if (!pow2) {
            m = 8192;
            while (m < 2 * n - 1) m <<= 1;
        }
        if (m > 65536)
            return cleanup(fail("rf_ctx_create: nfft beyond 65536 (power of two) / 32768 (any other length) is not supported"));
        LongTables &L = c->longt;
        L.m = m;
        L.log2n2 = 0;
        while ((4096 << L.log2n2) < m) ++L.log2n2;
        L.bluestein = pow2 ? 0 : 1;
        const long double PI_L = 3.14159265358979323846264338327950288L;
        std::vector<double2> twm((size_t)m);
        for (int k = 0; k < m / 2; ++k) {
            const long double a = 2.0L * PI_L * k / m;
            twm[k] = make_double2((double)cosl(a), (double)sinl(a));
            twm[(size_t)k + m / 2] = make_double2(-twm[k].x, -twm[k].y);
        }
        if (upload(c, twm, &L.tw_m)) return cleanup(1);
        if (L.bluestein) {
            // chirp c[j] = exp(+i pi j^2 / n): the argument reduced exactly, j^2 mod 2n in integers
            std::vector<double2> ch((size_t)n), bh((size_t)m);
            std::vector<long double> br((size_t)m, 0.0L), bi((size_t)m, 0.0L);
            for (int j = 0; j < n; ++j) {
                const long long q = ((long long)j * j) % (2LL * n);
                const long double a = PI_L * (long double)q / (long double)n;
                const long double cr = cosl(a), ci = sinl(a);
                ch[j] = make_double2((double)cr, (double)ci);
                // b[j] = conj(c[|j|]) wrapped onto 0 .. M-1
                br[j] = cr; bi[j] = -ci;
                if (j) { br[(size_t)m - j] = cr; bi[(size_t)m - j] = -ci; }
            }
            // Bhat = DFT_M(b) (sign -), radix-2 in long double on the host
            {
                int lg = 0;
                while ((1 << lg) < m) ++lg;
                for (int i = 0; i < m; ++i) {
                    int r = 0;
                    for (int bb = 0; bb < lg; ++bb)
                        if (i & (1 << bb)) r |= 1 << (lg - 1 - bb);
                    if (r > i) { std::swap(br[i], br[r]); std::swap(bi[i], bi[r]); }
                }
                for (int len = 2; len <= m; len <<= 1) {
                    const int half = len >> 1;
                    for (int k = 0; k < half; ++k) {
                        const long double a = -2.0L * PI_L * k / len;
                        const long double wr = cosl(a), wi = sinl(a);
                        for (int s0 = 0; s0 < m; s0 += len) {
                            const long double xr = br[s0 + k + half] * wr - bi[s0 + k + half] * wi;
                            const long double xi = br[s0 + k + half] * wi + bi[s0 + k + half] * wr;
                            br[s0 + k + half] = br[s0 + k] - xr; bi[s0 + k + half] = bi[s0 + k] - xi;
                            br[s0 + k] += xr; bi[s0 + k] += xi;
                        }
                    }
                }
                for (int k = 0; k < m; ++k) bh[k] = make_double2((double)br[k], (double)bi[k]);
            }
            if (upload(c, ch, &L.chirp) || upload(c, bh, &L.bhat)) return cleanup(1);
        }
        L.row_entries = long_row_entries(n, m, nsmp);
        c->long_rows = 2 * c->num_cu;
        const long long units = (long long)c->nslots * ntrc;
        if (units < c->long_rows) c->long_rows = (int)units;
        void *q = nullptr;
        if (dev_alloc(c, &q, sizeof(double2) * 2 * L.row_entries * (size_t)c->long_rows)) return cleanup(1);
        L.scratch = (double2 *)q;
    }
    const int lnsmp = T.lds_nsmp;   // what the trace kernels keep of the window in LDS (nothing on the long-window plan)
    if (!c->long_series && !pow2 && trace_anyn_lds_bytes(n, lnsmp, cfg->nlay_max) > 160 * 1024) {
        // longer series: only the filtered spectra stay in LDS, the time series goes through a scratch row per block
        if (trace_anyn_big_lds_bytes(n, lnsmp) > 160 * 1024)
            return cleanup(fail("rf_ctx_create: an nfft that is not a power of two is transformed by a direct DFT whose "
                                "spectra must fit the 160 KiB LDS of a CU (nfft up to ~9000); use a power of two for "
                                "longer series"));
        if (dev_alloc(c, &p, sizeof(double2) * (size_t)c->nslots * ntrc * trace_anyn_scratch_entries(n, nsmp)))
            return cleanup(1);
        c->anyn_scratch = (double2 *)p;
    }
    if (spectra_lds_bytes(cfg->nlay_max) > 160 * 1024 ||
        (pow2 && !c->long_series && trace_lds_bytes(n, lnsmp, cfg->nlay_max) > 160 * 1024) ||
        (c->long_series && trace_long_lds_bytes(lnsmp) > 160 * 1024) ||
        (!T.phi_gemm && sizeof(double) * (size_t)(5 * ((nsmp + 1) & ~1) + 8) > 160 * 1024))   // phi_kernel (host-owned traces)
        return cleanup(fail("rf_ctx_create: nfft / nsmp / nlay_max exceed the 160 KiB LDS of a gfx950 CU"));
    c->fused_allowed = pow2 && (c->nfwd == ntrc) && fused_lds_bytes(n, lnsmp, cfg->nlay_max) <= 160 * 1024;
    c->fusedc_allowed = n == 4096 && c->ray_common && ntrc > 1 && cfg->sdep <= 0.0 &&
                        fused8_lds_bytes(lnsmp, cfg->nlay_max) <= 80 * 1024;
    default_plan(c);
    if (!c->fused && !c->fusedc && ensure_spec(c)) return cleanup(1);
    // the creation-time memsets above went to the NULL stream and may return early; the context's stream is non-blocking
    if (hipDeviceSynchronize() != hipSuccess) return cleanup(fail("rf_ctx_create: hipDeviceSynchronize failed"));
    *ctx_out = c;
    return 0;
}

extern "C" int rf_ctx_destroy(rf_ctx *c)
{
    if (!c) return 0;
    (void)rf_comm_destroy(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    const int pending = device_error(c, "rf_ctx_destroy");   // (everything is released all the same)
    if (c->h_err) (void)hipHostFree(c->h_err);
    for (void *p : c->owned) (void)hipFree(p);
    if (c->h_single_in) (void)hipHostFree(c->h_single_in);
    if (c->h_single_out) (void)hipHostFree(c->h_single_out);
    for (auto &A : c->arena) {
        if (A.p) (void)hipHostFree(A.p);
        if (A.ev) (void)hipEventDestroy(A.ev);
    }
    if (c->h_out) (void)hipHostFree(c->h_out);
    for (auto &q : c->ev_pool) {
        (void)hipEventDestroy(q.e0);
        (void)hipEventDestroy(q.e1);
    }
    for (auto &q : c->slot_in)
        if (q.copied) (void)hipEventDestroy(q.copied);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return pending;
}

extern "C" int rf_get_flt(const rf_ctx *c, double *flt)
{
    if (!c || !flt) return fail("rf_get_flt: null argument");
    std::memcpy(flt, c->flt.data(), sizeof(double) * c->flt.size());
    return 0;
}
extern "C" int rf_get_is_ray_common(const rf_ctx *c, int32_t *flag)
{
    if (!c || !flag) return fail("rf_get_is_ray_common: null argument");
    *flag = c->ray_common;
    return 0;
}
extern "C" int rf_get_r_inv_info(const rf_ctx *c, int32_t *rank, double *cut_gap)
{
    if (!c) return fail("rf_get_r_inv_info: null context");
    for (int t = 0; t < c->cfg.ntrc; ++t) {
        if (rank) rank[t] = c->r_inv_rank[t];
        if (cut_gap) cut_gap[t] = c->r_inv_gap[t];
    }
    return 0;
}
extern "C" int rf_get_r_inv(const rf_ctx *c, double *r_inv)
{
    if (!c || !r_inv) return fail("rf_get_r_inv: null argument");
    std::memcpy(r_inv, c->r_inv.data(), sizeof(double) * c->r_inv.size());
    return 0;
}

// The input check of the *_device entry points runs on the device (stage_kernel): what it refused is reported by the
// next call that looks here -- after its own synchronisation where it has one (rf_eval_batch, rf_eval_wait,
// rf_get_rft*, rf_calc_likelihood, rf_profile_read, rf_post_read, rf_ctx_destroy), on entry otherwise (rf_commit*: the
// caller has synchronised the evaluation's stream to read logL).  Reading clears the word.
static int device_error(rf_ctx *c, const char *where)
{
    if (!c->h_err || !c->h_err[0]) return 0;
    const int why = c->h_err[0], item = c->h_err[1], what = c->h_err[2];
    c->h_err[0] = 0;
    static const char *reason[] = {"", "nlay outside [2, nlay_pad]", "walker id outside the context's slots", "fwd_flag > 1"};
    char msg[320];
    snprintf(msg, sizeof msg, "%s: an earlier batch carried a bad item: batch item %d: %s (value %d); the item was not "
             "evaluated (logL = NaN), the rest of its batch was", where, item, reason[why >= 1 && why <= 3 ? why : 0], what);
    return fail(msg);
}

// ---------------------------------------------------------------------------
static void flush_profile(rf_ctx *c)
{
    for (size_t u = 0; u < c->ev_used; ++u) {
        rf_ctx::Timed &q = c->ev_pool[u];
        (void)hipEventSynchronize(q.e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, q.e0, q.e1);
        c->prof_ms[q.kind] += ms;
        c->prof_n[1 + q.kind] += 1;
    }
    c->ev_used = 0;
}

// begin timing one kernel launch of `kind` on stream s; returns the stop event to
// record after the launch (nullptr when profiling is off)
static hipEvent_t prof_begin(rf_ctx *c, int kind, hipStream_t s)
{
    if (!c->prof || !c->prof_this) return nullptr;
    constexpr size_t kMaxPool = 4096;
    if (c->ev_used == c->ev_pool.size()) {
        if (c->ev_pool.size() >= kMaxPool) {
            flush_profile(c);
        } else {
            rf_ctx::Timed q{};
            if (hipEventCreate(&q.e0) != hipSuccess || hipEventCreate(&q.e1) != hipSuccess) return nullptr;
            c->ev_pool.push_back(q);
        }
    }
    rf_ctx::Timed &q = c->ev_pool[c->ev_used++];
    q.kind = kind;
    (void)hipEventRecord(q.e0, s);
    return q.e1;
}

static int pick_nsplit(const rf_ctx *c, int nb)
{
    // Walkers differ in depth (2..k_max layers), so one wave per walker leaves a long
    // tail; splitting the bin iterations of a walker over several waves (interleaved, so
    // every split sees the same layer count) evens it out.  Aim at >= 32 waves per CU
    // (measured on MI355X: c2 1024 walkers 2.6 -> 4.1 M evals/s from 1 -> 8 splits; c4 with
    // 24576 (walker, trace) waves is best unsplit); never below one 64-bin iteration per wave.
    // (with chained phases the unit of splitting is a chunk of `chain` iterations)
    const int niter = c->chain > 1 ? std::max(1, ((c->nh + 63) / 64) / c->chain) : (c->nh + 63) / 64;
    const long waves = (long)nb * c->nfwd;
    const long want = 32L * c->num_cu;
    int ns = (int)std::min<long>(niter, std::max<long>(1, (want + waves - 1) / waves));
    if (c->nsplit_override > 0) ns = std::min(niter, c->nsplit_override);
    return ns;
}

// 512-thread blocks (fused8_kernel: nfft 4096, land, default phase chains) or 256-thread blocks (fused_kernel).
// The 8-wave block has 4-bin phase chains started from a block-shared anchor table, keeps four waves per SIMD and
// halves a block's latency-bound tail.  Measured A/B on MI355X at the end of round 2 (tools/ab_opt.sh block_threads
// 256 512, and bench.py --walkers): two rounds of blocks (C2) +1.6 % throughput, with deconvolution +3.5 %;
// three and four rounds -2 %; C3 (16 rounds) and C4 (48 rounds) level to -2 %.
// The two factorise the FFT differently (8^4 / 16^3), so a chain's trace differs in the last bits between them.
// The choice is therefore a property of the CONTEXT, made from its capacity (max_walkers * ntrc blocks at a full
// batch: up to three rounds -> the 8-wave kernel), never of a launch's batch size: a chain evaluated alone, in a
// partial batch or in a full one gets bit-identical results (tests/test_gpu_parity.py).
static bool use_fused8(const rf_ctx *c)
{
    const bool can = c->fused && c->cfg.nfft == 4096 && c->cfg.sdep <= 0.0 && c->chain_override < 0 &&
                     fused8_lds_bytes(c->tab.lds_nsmp, c->cfg.nlay_max) <= 80 * 1024;
    if (!can || c->block_threads == 256) return false;
    if (c->block_threads == 512) return true;
    const long long blocks = (long long)c->cfg.max_walkers * c->cfg.ntrc;
    return blocks <= (long long)c->fused8_max_rounds * 2 * c->num_cu;
}

// The split plan's intermediate, spec[nslots][nfwd][2][nh] complex128 (8.6 GB at the C5 capacity): allocated when a
// context first NEEDS it -- at creation or at the rf_set_option call that makes the split plan the context's plan, so
// that running out of memory surfaces at a configuration call, not in the middle of a run.
static int ensure_spec(rf_ctx *c)
{
    if (c->spec) return 0;
    void *p = nullptr;
    if (dev_alloc(c, &p, sizeof(double2) * (size_t)c->nslots * c->nfwd * 2 * c->nh)) return 1;
    c->spec = (double2 *)p;
    return 0;
}

// what follows a trace kernel that left its misfits in HBM: the quadratic forms and logL of the batch
// (defer 1: phi_deferred_kernel, 8 items per block; defer 2, long windows: one GEMM on the FP64 matrix cores)
static int finish_likelihood(rf_ctx *c, const BatchArgs &b, int defer, hipStream_t s)
{
    if (!defer) return 0;
    hipEvent_t e = prof_begin(c, 2, s);
    if (defer == 2)
        launch_phi_gemm(c->tab, b, c->ws, c->pg, s);
    else
        launch_logl_deferred(c->tab, b, c->ws, s);
    if (e) (void)hipEventRecord(e, s);
    return 0;
}

// rf_commit and rf_post_record (host arrays) return before the device has done their work; a *_device call on ANOTHER
// stream that reads or writes walker state or the accumulators is put behind them
static int order_after_commit(rf_ctx *c, hipStream_t s)
{
    for (int i = RF_EVAL_MAX_IN_FLIGHT; i < RF_EVAL_MAX_IN_FLIGHT + 2; ++i) {      // rf_commit's and rf_post_record's
        const rf_ctx::Arena &A = c->arena[i];
        if (A.pending && s != c->stream) HIP_TRY(hipStreamWaitEvent(s, A.ev, 0));
    }
    return 0;
}

static int run_batch(rf_ctx *c, const BatchArgs &b_in, hipStream_t s)
{
    if (b_in.nb <= 0) return 0;
    if (order_after_commit(c, s)) return 1;
    if (b_in.nb > c->nslots) return fail("batch larger than max_walkers + 1");
    if (b_in.nlay_pad > c->cfg.nlay_max) return fail("nlay_pad exceeds nlay_max of the context");
    HIP_TRY(hipSetDevice(c->device));
    BatchArgs b = b_in;
    c->prof_this = c->prof && (c->prof_batch++ % c->prof_every) == 0;
    int *order_next = nullptr;
    const bool one_launch = c->fused || c->fusedc;   // spectra + traces + (small batches) logL in one kernel
    if (c->lpt && !b.order && b.nb >= 2 * c->num_cu) {
        // deepest walkers first (worth it once blocks outnumber the CUs).  Fused path: the previous
        // launch of a batch of this size left the order in d_order_alt (sorted by its own depths -- one
        // proposal step stale for this launch, which costs a little balance, never correctness), and
        // this launch does the same for the next; otherwise a ~4 us order_kernel in front.
        if (one_launch && c->order_reuse && c->order_next_nb == b.nb) {
            std::swap(c->d_order, c->d_order_alt);
        } else {
            launch_order(b.nb, b.nlay, b.fwd_flag, c->d_order, s);
        }
        b.order = c->d_order;
        if (one_launch && c->order_reuse) order_next = c->d_order_alt;
        c->order_next_nb = order_next ? b.nb : 0;
    } else {
        c->order_next_nb = 0;
    }
    // the per-(item, forward-trace) constants of the propagator, once per batch item, for whichever plan follows
    launch_stage(c->tab, b, c->ws, s);
    if (c->fusedc) {
        // one block per walker carries all its traces: the same "misfits to HBM, quadratic forms by the follow-up
        // kernels" rule as below, counted in (walker, trace) units of work
        const long long units = (long long)b.nb * c->cfg.ntrc, round = 2LL * c->num_cu;
        const int defer = c->tab.phi_gemm ? 2 : (c->defer_logl >= 0 ? c->defer_logl : units >= 2 * round);
        hipEvent_t e = prof_begin(c, 0, s);
        launch_fusedc(c->tab, b, c->ws, c->slow_count, c->ablate, defer, order_next, c->single_trace_out, s);
        if (e) (void)hipEventRecord(e, s);
        if (finish_likelihood(c, b, defer, s)) return 1;
    } else if (c->fused) {
        // several traces per walker and at least two rounds of blocks: the block ends with the trace store;
        // misfits go to HBM (808 B per trace) and ONE small follow-up kernel forms the quadratic forms -- the
        // rows of R^-1 fetched once per 8 walkers instead of once per block, same arithmetic -- and logL.  That removes from
        // every block the R^-1 read (80 KB from L2), the cross-block hand-off of phi and ~5 us of holding
        // its CU slot (measured C4 +5 %, C5 +6 %, C1 shape +13 %).  Smaller batches (the per-call drop-in), single-trace
        // contexts and long windows (misfits of 8 walkers must fit the follow-up kernel's LDS) keep the
        // single launch.
        // (single trace: it pays from four rounds of blocks on -- measured C3 +6 %,
        // C2, two rounds, -2 %)
        // Long windows (phi_gemm): always the misfits to HBM, then the batch's quadratic forms as one MFMA GEMM.
        const long long blocks = (long long)b.nb * c->cfg.ntrc, round = 2LL * c->num_cu;
        const int defer = c->tab.phi_gemm ? 2 : (c->defer_logl >= 0 ? c->defer_logl
                          : (c->cfg.ntrc > 1 ? blocks >= 2 * round : blocks >= 4 * round));
        hipEvent_t e = prof_begin(c, 0, s);
        if (use_fused8(c))
            launch_fused8(c->tab, b, c->ws, c->slow_count, c->ablate, defer, order_next, c->single_trace_out, s);
        else
            launch_fused(c->tab, b, c->ws, c->chain, c->slow_count, c->ablate, defer, order_next, c->single_trace_out, s);
        if (e) (void)hipEventRecord(e, s);
        if (finish_likelihood(c, b, defer, s)) return 1;
    } else {
        if (ensure_spec(c)) return 1;
        hipEvent_t e = prof_begin(c, 0, s);
        launch_spectra(c->tab, b, c->spec, pick_nsplit(c, b.nb), c->chain, c->waves_per_block, c->slow_list,
                       c->slow_count, c->ws, s);
        if (e) (void)hipEventRecord(e, s);
        e = prof_begin(c, 1, s);
        const int defer = c->tab.phi_gemm ? 2 : 0;
        if (c->long_series)
            launch_trace_long(c->tab, b, c->spec, c->ws, c->slow_count, c->longt, c->long_rows, defer, s);
        else
            launch_trace(c->tab, b, c->spec, c->ws, c->slow_count, c->anyn_scratch, defer, s);   // also forms logL (short windows)
        if (e) (void)hipEventRecord(e, s);
        if (finish_likelihood(c, b, defer, s)) return 1;
    }
    if (c->prof_this) c->prof_n[0] += 1;
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rf_eval_batch_device(rf_ctx *c, int32_t nb, const int32_t *d_walker_ids,
                                    const int32_t *d_fwd_flag, const int32_t *d_nlay, int32_t nlay_pad,
                                    const double *d_layers, const double *d_sig, double *d_logl, void *stream)
{
    if (!c || !d_walker_ids || !d_nlay || !d_layers || !d_sig || !d_logl)
        return fail("rf_eval_batch_device: null argument");
    BatchArgs b{nb, nlay_pad, d_walker_ids, d_fwd_flag, d_nlay, d_layers, d_sig, d_logl, nullptr};
    return run_batch(c, b, (hipStream_t)stream);
}

extern "C" int rf_eval_batch(rf_ctx *c, int32_t nb, const int32_t *walker_ids, const int32_t *fwd_flag,
                             const int32_t *nlay, int32_t nlay_pad, const double *layers, const double *sig,
                             double *logl)
{
    if (!c || !walker_ids || !nlay || !layers || !sig || !logl) return fail("rf_eval_batch: null argument");
    if (nb <= 0) return 0;
    for (int i = 0; i < nb; ++i) {
        if (walker_ids[i] < 0 || walker_ids[i] >= c->nslots) return fail("rf_eval_batch: walker id out of range");
        if (nlay[i] < 2 || nlay[i] > nlay_pad) return fail("rf_eval_batch: nlay out of range");
    }
    HIP_TRY(hipSetDevice(c->device));
    if (ensure_stage(c, nb, nlay_pad) || ensure_out(c, nb)) return 1;
    hipStream_t s = c->stream;
    // inputs: one DMA per array from pinned memory (the caller's, or the arena's copy of a pageable array); output:
    // the kernels write logL into device-mapped pinned memory -- no copy launch on either side of the evaluation
    const size_t b_lay = sizeof(double) * (size_t)nb * 4 * nlay_pad, b_sig = sizeof(double) * (size_t)nb * c->cfg.ntrc;
    rf_ctx::Arena &A = c->arena[RF_EVAL_MAX_IN_FLIGHT];
    double *h_logl = c->h_out + out_region(c, RF_EVAL_MAX_IN_FLIGHT), *d_logl = c->d_out + out_region(c, RF_EVAL_MAX_IN_FLIGHT);
    if (arena_begin(A, 4 * sizeof(int) * (size_t)nb + b_lay + b_sig)) return 1;
    if (h2d(A, c->d_ids, walker_ids, sizeof(int) * nb, s)) return 1;
    if (fwd_flag && h2d(A, c->d_fwd, fwd_flag, sizeof(int) * nb, s)) return 1;
    if (h2d(A, c->d_nlay, nlay, sizeof(int) * nb, s) || h2d(A, c->d_layers, layers, b_lay, s) ||
        h2d(A, c->d_sig, sig, b_sig, s))
        return 1;
    BatchArgs b{nb, nlay_pad, c->d_ids, fwd_flag ? c->d_fwd : nullptr, c->d_nlay, c->d_layers, c->d_sig, d_logl, nullptr};
    if (c->lpt && nb >= 2 * c->num_cu) {
        // host buffers: the longest-first order is a counting sort here (straight into the arena), no extra launch
        int *ord = static_cast<int *>(arena_take(A, sizeof(int) * nb));
        int hist[257] = {0};
        auto key = [&](int i) { return (fwd_flag && fwd_flag[i] != 1) ? 0 : std::min(nlay[i], 255); };
        for (int i = 0; i < nb; ++i) ++hist[256 - key(i)];          // descending keys first
        for (int k = 1; k <= 256; ++k) hist[k] += hist[k - 1];
        for (int i = nb - 1; i >= 0; --i) ord[--hist[256 - key(i)]] = i;
        HIP_TRY(hipMemcpyAsync(c->d_order, ord, sizeof(int) * nb, hipMemcpyHostToDevice, s));
        b.order = c->d_order;
    }
    if (run_batch(c, b, s)) return 1;
    c->last_staged = A.staged;
    HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(logl, h_logl, sizeof(double) * nb);
    return device_error(c, "rf_eval_batch");
}

extern "C" int rf_get_rft(rf_ctx *c, int32_t walker, int32_t which, int32_t nout, double *out)
{
    if (!c || !out) return fail("rf_get_rft: null argument");
    if (walker < 0 || walker >= c->nslots) return fail("rf_get_rft: walker out of range");
    if (nout < 1 || nout > c->cfg.nfft) return fail("rf_get_rft: n out of range");
    if (nout > c->ws.trace_len) return fail("rf_get_rft: the context keeps samples 1 .. nsmp of every trace only (option trace_window)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int cur = 0, pf = 0;
    HIP_TRY(hipMemcpy(&cur, c->ws.cur_slot + walker, sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&pf, c->ws.prop_fwd + walker, sizeof(int), hipMemcpyDeviceToHost));
    const int slot = (which == 1 && pf) ? 1 - cur : cur;
    const int ntrc = c->cfg.ntrc;
    const int len = c->ws.trace_len;
    const double *src = c->ws.rft + (((size_t)slot * c->nslots + walker) * ntrc) * (size_t)len;
    HIP_TRY(hipMemcpy2D(out, sizeof(double) * nout, src, sizeof(double) * len, sizeof(double) * nout, ntrc,
                        hipMemcpyDeviceToHost));
    return device_error(c, "rf_get_rft");
}

extern "C" int rf_get_rft_batch(rf_ctx *c, int32_t n, const int32_t *walker_ids, int32_t which, int32_t nout,
                                double *out)
{
    if (!c || !walker_ids || !out) return fail("rf_get_rft_batch: null argument");
    if (n <= 0) return 0;
    if (nout < 1 || nout > c->cfg.nfft) return fail("rf_get_rft_batch: n out of range");
    if (nout > c->ws.trace_len) return fail("rf_get_rft_batch: the context keeps samples 1 .. nsmp of every trace only (option trace_window)");
    for (int i = 0; i < n; ++i)
        if (walker_ids[i] < 0 || walker_ids[i] >= c->nslots) return fail("rf_get_rft_batch: walker out of range");
    HIP_TRY(hipSetDevice(c->device));
    if (ensure_stage(c, n, 2)) return 1;
    const size_t bytes = sizeof(double) * (size_t)n * c->cfg.ntrc * nout;
    if (bytes > c->gather_bytes) {
        void *p = nullptr;
        if (dev_alloc(c, &p, bytes)) return 1;
        c->d_gather = (double *)p;
        c->gather_bytes = bytes;
    }
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(c->d_ids, walker_ids, sizeof(int) * n, hipMemcpyHostToDevice, s));
    launch_gather_rft(c->ws, c->cfg.ntrc, c->cfg.nfft, n, c->d_ids, which, nout, c->d_gather, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->d_gather, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return device_error(c, "rf_get_rft_batch");
}

// Per-call drop-in (one chain per call, src/pt_mcmc.f90:178-180).  Latency matters here, not
// throughput: one pinned host->device copy of the packed inputs, the evaluation, a gather of the
// proposed trace next to logL, one device->host copy, one synchronisation.
extern "C" int rf_calc_likelihood(rf_ctx *c, int32_t walker, int32_t fwd_flag, int32_t nlay,
                                  const double *alpha, const double *beta, const double *rho, const double *h,
                                  const double *sig, double *prop_log_likelihood, double *prop_rft)
{
    if (!c || !sig || !prop_log_likelihood) return fail("rf_calc_likelihood: null argument");
    if (walker < 0 || walker >= c->nslots) return fail("rf_calc_likelihood: walker out of range");
    if (prop_rft && c->ws.trace_len < c->cfg.nfft)
        return fail("rf_calc_likelihood: prop_rft(nfft, ntrc) cannot be delivered: the context keeps samples 1 .. nsmp only "
                    "(option trace_window); pass NULL and read the window with rf_get_rft");
    if (fwd_flag) {
        if (!alpha || !beta || !rho || !h) return fail("rf_calc_likelihood: null layer arrays");
        if (nlay < 2 || nlay > c->cfg.nlay_max) return fail("rf_calc_likelihood: nlay out of range");
    }
    HIP_TRY(hipSetDevice(c->device));
    const int ntrc = c->cfg.ntrc, n = c->cfg.nfft, pad = c->cfg.nlay_max;
    // packed layout (doubles): [0] ids: walker, fwd, nlay as 3 ints (+1 pad int) | layers[4][pad] | sig[ntrc]
    const size_t in_doubles = 2 + (size_t)4 * pad + ntrc;
    const size_t out_doubles = 1 + (size_t)n * ntrc;
    if (!c->h_single_in) {
        // pinned, device-mapped staging: the kernels read the packed inputs and write logL and the gathered
        // trace straight through PCIe -- no copy launches on this latency-bound path
        HIP_TRY(hipHostMalloc((void **)&c->h_single_in, sizeof(double) * in_doubles, hipHostMallocMapped));
        HIP_TRY(hipHostMalloc((void **)&c->h_single_out, sizeof(double) * out_doubles, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&c->d_single_in, c->h_single_in, 0));
        HIP_TRY(hipHostGetDevicePointer((void **)&c->d_single_out, c->h_single_out, 0));
    }
    int *hi = reinterpret_cast<int *>(c->h_single_in);
    const int nl = fwd_flag ? nlay : 2;
    hi[0] = walker;
    hi[1] = fwd_flag ? 1 : 0;
    hi[2] = nl;
    hi[3] = 0;
    double *hl = c->h_single_in + 2;
    for (size_t i = 0; i < (size_t)4 * pad; ++i) hl[i] = 1.0;
    if (fwd_flag) {
        std::memcpy(hl, alpha, sizeof(double) * nlay);
        std::memcpy(hl + pad, beta, sizeof(double) * nlay);
        std::memcpy(hl + 2 * (size_t)pad, rho, sizeof(double) * nlay);
        std::memcpy(hl + 3 * (size_t)pad, h, sizeof(double) * nlay);
    }
    std::memcpy(hl + 4 * (size_t)pad, sig, sizeof(double) * ntrc);
    hipStream_t s = c->stream;
    const int *di = reinterpret_cast<const int *>(c->d_single_in);
    const double *dl = c->d_single_in + 2;
    BatchArgs b{1, pad, di, di + 1, di + 2, dl, dl + 4 * (size_t)pad, c->d_single_out, nullptr};
    // fused path with a forward evaluation: the kernel itself writes the proposed trace to the mapped
    // buffer; otherwise (split kernels, or the stored trace of a sigma-only call) a gather kernel does
    const bool direct = prop_rft && fwd_flag && (c->fused || c->fusedc);
    c->single_trace_out = direct ? c->d_single_out + 1 : nullptr;
    const int rc = run_batch(c, b, s);
    c->single_trace_out = nullptr;
    if (rc) return 1;
    if (prop_rft && !direct) {
        launch_gather_rft(c->ws, ntrc, n, 1, di, 1, n, c->d_single_out + 1, s);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(s));
    *prop_log_likelihood = c->h_single_out[0];
    if (prop_rft) std::memcpy(prop_rft, c->h_single_out + 1, sizeof(double) * (size_t)n * ntrc);
    return device_error(c, "rf_calc_likelihood");
}

extern "C" int rf_calc_rf(rf_ctx *c, int32_t nlay, const double *alpha, const double *beta, const double *rho,
                          const double *h, double *rft)
{
    if (!c || !rft) return fail("rf_calc_rf: null argument");
    std::vector<double> sig((size_t)c->cfg.ntrc, 1.0);
    double ll;
    return rf_calc_likelihood(c, c->nslots - 1, 1, nlay, alpha, beta, rho, h, sig.data(), &ll, rft);
}

extern "C" int rf_set_r_inv(rf_ctx *c, const double *r_inv)
{
    if (!c || !r_inv) return fail("rf_set_r_inv: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::memcpy(c->r_inv.data(), r_inv, sizeof(double) * c->r_inv.size());
    const std::vector<double> t = transpose_r_inv(c->r_inv, c->cfg.ntrc, c->cfg.nsmp);
    HIP_TRY(hipMemcpy(const_cast<double *>(c->tab.r_inv_t), t.data(), sizeof(double) * t.size(),
                      hipMemcpyHostToDevice));
    if (c->tab.phi_gemm) {
        const std::vector<double> rg = pad_r_inv(c->r_inv, c->cfg.ntrc, c->cfg.nsmp, c->pg.kp, c->pg.np);
        HIP_TRY(hipMemcpy(const_cast<double *>(c->pg.rg), rg.data(), sizeof(double) * rg.size(), hipMemcpyHostToDevice));
        // the triangular image is what the default plan ("gemm_triangle" 1) multiplies: it follows the new matrix too
        const std::vector<double> rt = triangle_r_inv(rg, c->cfg.ntrc, c->pg.kp, c->pg.np);
        HIP_TRY(hipMemcpy(const_cast<double *>(c->pg.rt), rt.data(), sizeof(double) * rt.size(), hipMemcpyHostToDevice));
    }
    // The cached quadratic forms of the stored traces (what a sigma-only proposal re-uses, src/likelihood.f90:81) belong
    // to the OLD matrix: they become NaN, so that such a proposal fails loudly until the chain has been re-evaluated and
    // committed (set the matrix before the first evaluation, as the Fortran shim does).
    // (on the context's own stream and waited for: c->stream is non-blocking, a NULL-stream memset would not be ordered
    // against the evaluation or commit the caller issues next)
    HIP_TRY(hipMemsetAsync(c->ws.phi, 0xFF, sizeof(double) * 2 * (size_t)c->nslots * (size_t)c->cfg.ntrc, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rf_calc_likelihood_of_trace(rf_ctx *c, const double *rft, const double *sig, double *logl)
{
    if (!c || !rft || !sig || !logl) return fail("rf_calc_likelihood_of_trace: null argument");
    HIP_TRY(hipSetDevice(c->device));
    const int n = c->cfg.nfft, ntrc = c->cfg.ntrc, wk = c->nslots - 1;   // scratch walker
    if (ensure_stage(c, 1, 2)) return 1;
    hipStream_t s = c->stream;
    HIP_TRY(hipStreamSynchronize(s));
    int cur = 0;
    HIP_TRY(hipMemcpy(&cur, c->ws.cur_slot + wk, sizeof(int), hipMemcpyDeviceToHost));
    const int len = c->ws.trace_len;     // (nfft, or the window's nsmp samples of each trace)
    double *dst = c->ws.rft + (((size_t)(1 - cur) * c->nslots + wk) * ntrc) * (size_t)len;
    HIP_TRY(hipMemcpy2DAsync(dst, sizeof(double) * len, rft, sizeof(double) * n, sizeof(double) * len, ntrc,
                             hipMemcpyHostToDevice, s));
    const int one = 1;
    HIP_TRY(hipMemcpyAsync(c->d_ids, &wk, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_fwd, &one, sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_sig, sig, sizeof(double) * ntrc, hipMemcpyHostToDevice, s));
    BatchArgs b{1, 2, c->d_ids, c->d_fwd, c->d_nlay, c->d_layers, c->d_sig, c->d_logl, nullptr};
    if (c->tab.phi_gemm) {
        // long windows: the same GEMM the batched path runs, on one row (bit-identical to the chain's own evaluation)
        launch_misfit_of_trace(c->tab, c->ws, wk, s);
        launch_phi_gemm(c->tab, b, c->ws, c->pg, s);
    } else {
        launch_phi(c->tab, c->ws, wk, s);
        launch_logl(c->tab, b, c->ws, s);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(logl, c->d_logl, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int rf_set_model(rf_ctx *c, const rf_model_config *m)
{
    if (!c || !m || !m->vp_ref || !m->vs_ref) return fail("rf_set_model: null argument");
    if (m->k_max < 2 || m->nref < 1) return fail("rf_set_model: bad k_max / nref");
    if (m->k_max + 2 > c->cfg.nlay_max) return fail("rf_set_model: k_max + 2 exceeds nlay_max of the context");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<double> vp(m->vp_ref, m->vp_ref + m->nref), vs(m->vs_ref, m->vs_ref + m->nref);
    const double *dvp = nullptr, *dvs = nullptr;
    if (upload(c, vp, &dvp) || upload(c, vs, &dvs)) return 1;
    c->model = ModelConfig{m->k_max, m->vp_mode, m->nref, c->cfg.sdep, m->z_max, m->h_min, m->z_ref_min, m->dz_ref,
                           m->vp_min, m->vp_max, m->vs_min, m->vs_max, m->vpvs_min, m->vpvs_max, dvp, dvs};
    c->fm_pad = m->k_max + 2;
    void *p;
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return 1;
    c->d_fm_nlay = (int *)p;
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return 1;
    c->d_fm_flag = (int *)p;
    if (dev_alloc(c, &p, sizeof(double) * (size_t)c->nslots * 4 * c->fm_pad)) return 1;
    c->d_fm_layers = (double *)p;
    if (dev_alloc(c, &p, sizeof(double) * (size_t)c->nslots * 3 * m->k_max)) return 1;
    c->d_fm_scratch = (double *)p;
    // device images of the host arrays of rf_eval_models
    if (dev_alloc(c, &p, sizeof(int) * c->nslots)) return 1;
    c->d_m_k = (int *)p;
    if (dev_alloc(c, &p, sizeof(double) * (size_t)c->nslots * 3 * m->k_max)) return 1;
    c->d_m_z = (double *)p;
    c->d_m_dvp = c->d_m_z + (size_t)c->nslots * m->k_max;
    c->d_m_dvs = c->d_m_dvp + (size_t)c->nslots * m->k_max;
    c->have_model = true;
    return 0;
}

extern "C" int rf_format_models_device(rf_ctx *c, int32_t nb, const int32_t *d_k, const double *d_z,
                                       const double *d_dvp, const double *d_dvs, int32_t *d_nlay, double *d_layers,
                                       int32_t nlay_pad, int32_t *d_valid, void *stream)
{
    if (!c || !d_k || !d_z || !d_dvp || !d_dvs || !d_nlay || !d_layers) return fail("rf_format_models_device: null argument");
    if (!c->have_model) return fail("rf_format_models_device: rf_set_model has not been called");
    if (nb <= 0) return 0;
    if (nb > c->nslots) return fail("rf_format_models_device: batch larger than max_walkers + 1");
    if (nlay_pad < c->model.k_max + 2) return fail("rf_format_models_device: nlay_pad < k_max + 2");
    HIP_TRY(hipSetDevice(c->device));
    FormatParams P{c->model, nb, nlay_pad, d_k, d_z, d_dvp, d_dvs, nullptr, d_nlay, d_layers, c->d_fm_flag, d_valid,
                   c->d_fm_scratch};
    launch_format_model(P, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rf_eval_models_device(rf_ctx *c, int32_t nb, const int32_t *d_walker_ids, const int32_t *d_fwd_flag,
                                     const int32_t *d_k, const double *d_z, const double *d_dvp, const double *d_dvs,
                                     const double *d_sig, double *d_logl, int32_t *d_valid, void *stream)
{
    if (!c || !d_walker_ids || !d_k || !d_z || !d_dvp || !d_dvs || !d_sig || !d_logl)
        return fail("rf_eval_models_device: null argument");
    if (!c->have_model) return fail("rf_eval_models_device: rf_set_model has not been called");
    if (nb <= 0) return 0;
    if (nb > c->nslots) return fail("rf_eval_models_device: batch larger than max_walkers + 1");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    FormatParams P{c->model, nb, c->fm_pad, d_k, d_z, d_dvp, d_dvs, d_fwd_flag, c->d_fm_nlay, c->d_fm_layers,
                   c->d_fm_flag, d_valid, c->d_fm_scratch};
    launch_format_model(P, s);
    BatchArgs b{nb, c->fm_pad, d_walker_ids, c->d_fm_flag, c->d_fm_nlay, c->d_fm_layers, d_sig, d_logl, nullptr};
    return run_batch(c, b, s);
}

// The same from HOST arrays in the batched sampler's own layout (one column per chain): what an iteration of
// pt_control_batched hands over.  Pinned arrays (rf_host_alloc) go down by DMA as they are.  rf_eval_models_begin
// enqueues transfers and kernels and returns; rf_eval_wait delivers the results of that ticket.
extern "C" int rf_eval_models_begin(rf_ctx *c, int32_t nb, const int32_t *walker_ids, const int32_t *fwd_flag,
                                    const int32_t *k, const double *z, int32_t ldz, const double *dvp, const double *dvs,
                                    const double *sig, int32_t want_valid, int32_t *ticket)
{
    if (!c || !walker_ids || !k || !z || !dvp || !dvs || !sig || !ticket) return fail("rf_eval_models_begin: null argument");
    if (!c->have_model) return fail("rf_eval_models: rf_set_model has not been called");
    if (nb <= 0 || nb > c->nslots) return fail("rf_eval_models: batch size must be 1 .. max_walkers + 1");
    const int kmax = c->model.k_max, ntrc = c->cfg.ntrc;
    if (ldz < kmax - 1 || ldz > kmax) return fail("rf_eval_models: ldz must be k_max - 1 or k_max");
    for (int i = 0; i < nb; ++i) {
        if (walker_ids[i] < 0 || walker_ids[i] >= c->nslots) return fail("rf_eval_models: walker id out of range");
        if ((!fwd_flag || fwd_flag[i] == 1) && (k[i] < 1 || k[i] >= kmax)) return fail("rf_eval_models: k out of range");
    }
    const int slot = c->ticket_next;
    if (c->ticket[slot].busy) return fail("rf_eval_models_begin: RF_EVAL_MAX_IN_FLIGHT evaluations are already in flight");
    HIP_TRY(hipSetDevice(c->device));
    if (ensure_stage(c, nb, 2) || ensure_out(c, nb)) return 1;
    hipStream_t s = c->stream;
    const size_t N = (size_t)nb;
    // the slot's own device inputs (its previous evaluation has been waited for: nothing reads them any more)
    rf_ctx::SlotIn &I = c->slot_in[slot];
    if (nb > I.cap) {
        const size_t cap = (size_t)std::max(nb, c->nslots);
        void *p = nullptr;
        if (dev_alloc(c, &p, sizeof(int) * 3 * cap)) return 1;
        I.ids = (int *)p; I.fwd = I.ids + cap; I.k = I.fwd + cap;
        if (dev_alloc(c, &p, sizeof(double) * cap * (3 * (size_t)kmax + ntrc))) return 1;
        I.z = (double *)p; I.dvp = I.z + cap * kmax; I.dvs = I.dvp + cap * kmax; I.sig = I.dvs + cap * kmax;
        I.cap = (int)cap;
    }
    // "copy_stream": the transfers on a stream of their own (a process that has the GPU to itself), or in line
    hipStream_t cs = s;
    if (c->use_copy_stream) {
        if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        if (!I.copied) HIP_TRY(hipEventCreateWithFlags(&I.copied, hipEventDisableTiming));
        cs = c->copy_stream;
    }
    rf_ctx::Arena &A = c->arena[slot];
    if (arena_begin(A, sizeof(int) * 3 * N + sizeof(double) * N * (ldz + 2 * (size_t)kmax + ntrc))) return 1;
    if (h2d(A, I.ids, walker_ids, sizeof(int) * N, cs) || h2d(A, I.k, k, sizeof(int) * N, cs)) return 1;
    if (fwd_flag && h2d(A, I.fwd, fwd_flag, sizeof(int) * N, cs)) return 1;
    if (h2d(A, I.z, z, sizeof(double) * N * ldz, cs) || h2d(A, I.dvs, dvs, sizeof(double) * N * kmax, cs) ||
        h2d(A, I.sig, sig, sizeof(double) * N * ntrc, cs))
        return 1;
    // (dVp enters format_model only when it is solved for, src/model.f90:216-217)
    if (c->model.vp_mode == 1 && h2d(A, I.dvp, dvp, sizeof(double) * N * kmax, cs)) return 1;
    if (cs != s) {
        HIP_TRY(hipEventRecord(I.copied, cs));
        HIP_TRY(hipStreamWaitEvent(s, I.copied, 0));
    }
    double *d_logl = c->d_out + out_region(c, slot);
    int *d_valid = reinterpret_cast<int *>(d_logl + c->out_cap);
    FormatParams P{c->model, nb, c->fm_pad, I.k, I.z, I.dvp, I.dvs, fwd_flag ? I.fwd : nullptr,
                   c->d_fm_nlay, c->d_fm_layers, c->d_fm_flag, want_valid ? d_valid : nullptr, c->d_fm_scratch, ldz};
    launch_format_model(P, s);
    BatchArgs b{nb, c->fm_pad, I.ids, c->d_fm_flag, c->d_fm_nlay, c->d_fm_layers, I.sig, d_logl, nullptr};
    // (consecutive evaluations of a context may be different sets of chains -- the halves of pt_control_batched's
    // pipeline: the dispatch order the previous launch prepared is not this batch's)
    c->order_next_nb = 0;
    if (run_batch(c, b, s)) return 1;
    if (arena_mark(A, s)) return 1;      // the ticket's completion event
    c->last_staged = A.staged;
    c->ticket[slot] = rf_ctx::Ticket{true, nb, want_valid != 0};
    c->ticket_next = (slot + 1) % RF_EVAL_MAX_IN_FLIGHT;
    *ticket = slot;
    return 0;
}

extern "C" int rf_eval_wait(rf_ctx *c, int32_t ticket, double *logl, int32_t *valid)
{
    if (!c || !logl) return fail("rf_eval_wait: null argument");
    if (ticket < 0 || ticket >= RF_EVAL_MAX_IN_FLIGHT || !c->ticket[ticket].busy) return fail("rf_eval_wait: no such evaluation in flight");
    rf_ctx::Ticket &T = c->ticket[ticket];
    if (valid && !T.want_valid) return fail("rf_eval_wait: validity flags were not asked for at rf_eval_models_begin");
    HIP_TRY(hipSetDevice(c->device));
    rf_ctx::Arena &A = c->arena[ticket];
    HIP_TRY(hipEventSynchronize(A.ev));
    A.pending = false;
    const double *h_logl = c->h_out + out_region(c, ticket);
    std::memcpy(logl, h_logl, sizeof(double) * (size_t)T.nb);
    if (valid) std::memcpy(valid, reinterpret_cast<const int *>(h_logl + c->out_cap), sizeof(int) * (size_t)T.nb);
    T.busy = false;
    return device_error(c, "rf_eval_wait");
}

extern "C" int rf_eval_models(rf_ctx *c, int32_t nb, const int32_t *walker_ids, const int32_t *fwd_flag, const int32_t *k,
                              const double *z, int32_t ldz, const double *dvp, const double *dvs, const double *sig,
                              double *logl, int32_t *valid)
{
    if (!logl) return fail("rf_eval_models: null argument");
    if (nb <= 0) return 0;
    int32_t t = -1;
    if (rf_eval_models_begin(c, nb, walker_ids, fwd_flag, k, z, ldz, dvp, dvs, sig, valid ? 1 : 0, &t)) return 1;
    return rf_eval_wait(c, t, logl, valid);
}

extern "C" int rf_host_alloc(size_t bytes, void **ptr)
{
    if (!ptr) return fail("rf_host_alloc: null argument");
    *ptr = nullptr;
    HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocMapped));
    return 0;
}

extern "C" int rf_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return 0;
}

extern "C" int rf_commit_device(rf_ctx *c, int32_t nb, const int32_t *d_walker_ids, const int32_t *d_accept,
                                void *stream)
{
    if (!c || !d_walker_ids || !d_accept) return fail("rf_commit_device: null argument");
    if (nb <= 0) return 0;
    if (device_error(c, "rf_commit_device")) return 1;
    HIP_TRY(hipSetDevice(c->device));
    if (order_after_commit(c, (hipStream_t)stream)) return 1;
    launch_commit(c->ws, nb, d_walker_ids, d_accept, c->cfg.ntrc, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rf_commit(rf_ctx *c, int32_t nb, const int32_t *walker_ids, const int32_t *accept)
{
    if (!c || !walker_ids || !accept) return fail("rf_commit: null argument");
    if (nb <= 0) return 0;
    for (int i = 0; i < nb; ++i)
        if (walker_ids[i] < 0 || walker_ids[i] >= c->nslots) return fail("rf_commit: walker id out of range");
    if (device_error(c, "rf_commit")) return 1;
    HIP_TRY(hipSetDevice(c->device));
    if (ensure_stage(c, nb, 2)) return 1;
    hipStream_t s = c->stream;
    // Nothing comes back from a commit, so the call does not wait for the device: the two arrays are copied into the
    // pinned arena (the caller may reuse its own at once) and everything else is stream-ordered -- every later call
    // on the context runs behind it.  The next user of the arena waits for this copy (pin_ev).
    rf_ctx::Arena &A = c->arena[RF_EVAL_MAX_IN_FLIGHT];
    if (arena_begin(A, 2 * sizeof(int) * (size_t)nb)) return 1;
    int *st = static_cast<int *>(arena_take(A, sizeof(int) * nb)), *sa = static_cast<int *>(arena_take(A, sizeof(int) * nb));
    std::memcpy(st, walker_ids, sizeof(int) * nb);
    std::memcpy(sa, accept, sizeof(int) * nb);
    HIP_TRY(hipMemcpyAsync(c->d_ids, st, sizeof(int) * nb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_acc, sa, sizeof(int) * nb, hipMemcpyHostToDevice, s));
    launch_commit(c->ws, nb, c->d_ids, c->d_acc, c->cfg.ntrc, s);
    HIP_TRY(hipGetLastError());
    return arena_mark(A, s);
}

extern "C" int rf_pt_swap_device(rf_ctx *c, int32_t npairs, const int32_t *d_pairs, const double *d_log_u,
                                 double *d_temps, const double *d_logl, int32_t *d_accepted, void *stream)
{
    if (!c || !d_pairs || !d_log_u || !d_temps || !d_logl) return fail("rf_pt_swap_device: null argument");
    if (npairs <= 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    launch_pt_swap(npairs, d_pairs, d_log_u, d_temps, d_logl, d_accepted, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rf_pt_swap_gathered_device(rf_ctx *c, int32_t nchains, int32_t rank, int32_t nranks, int32_t npairs,
                                          const int32_t *d_pairs, const double *d_log_u, const double *d_g_temps,
                                          const double *d_g_logl, double *d_temps, int32_t *d_accepted, void *stream)
{
    if (!c || !d_pairs || !d_log_u || !d_g_temps || !d_g_logl || !d_temps)
        return fail("rf_pt_swap_gathered_device: null argument");
    if (nchains < 1 || nranks < 1 || rank < 0 || rank >= nranks) return fail("rf_pt_swap_gathered_device: bad rank / nranks / nchains");
    if (npairs <= 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    launch_pt_swap_gathered(npairs, d_pairs, d_log_u, d_g_temps, d_g_logl, nchains, rank, nranks, d_temps, d_accepted,
                            (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// posterior accumulation (src/pt_mcmc.f90:204-286 on the device; kernels in rfgpu_posterior.hip)
// ---------------------------------------------------------------------------
template <class T>
static int post_alloc(rf_ctx *c, T **p, size_t count, bool zeroed)
{
    void *q = nullptr;
    if (dev_alloc(c, &q, sizeof(T) * count)) return 1;
    *p = static_cast<T *>(q);
    if (zeroed) c->post_zero.emplace_back(q, sizeof(T) * count);
    return 0;
}

extern "C" int rf_post_reset(rf_ctx *c)
{
    if (!c || !c->have_post) return fail("rf_post_reset: rf_post_create has not been called");
    HIP_TRY(hipSetDevice(c->device));
    for (auto &z : c->post_zero) HIP_TRY(hipMemsetAsync(z.first, 0, z.second, c->stream));
    launch_post_mark_unused(c->post, c->pst, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int rf_post_create(rf_ctx *c, const rf_post_config *p)
{
    if (!c || !p || !p->sig_min || !p->sig_max || !p->sig_mode) return fail("rf_post_create: null argument");
    if (!c->have_model) return fail("rf_post_create: rf_set_model has not been called");
    if (c->have_post) return fail("rf_post_create: already created for this context");
    if (p->nbin_z < 1 || p->nbin_vs < 1 || p->nbin_vp < 1 || p->nbin_vpvs < 1 || p->nbin_sig < 1 || p->nbin_amp < 1)
        return fail("rf_post_create: every nbin_* must be >= 1");
    if (p->max_models < 0) return fail("rf_post_create: max_models < 0");
    HIP_TRY(hipSetDevice(c->device));
    const int ntrc = c->cfg.ntrc, nsmp = c->cfg.nsmp, kmax = c->model.k_max;
    PostConfig &q = c->post;
    q = PostConfig{};
    q.nbin_z = p->nbin_z; q.nbin_vs = p->nbin_vs; q.nbin_vp = p->nbin_vp;
    q.nbin_vpvs = p->nbin_vpvs; q.nbin_sig = p->nbin_sig; q.nbin_amp = p->nbin_amp;
    q.k_max = kmax; q.ntrc = ntrc; q.nsmp = nsmp; q.nfft = c->cfg.nfft;
    // bin widths as src/pt_mcmc.f90:423-430 (double / integer)
    q.amp_min = p->amp_min;
    q.dbin_amp = (p->amp_max - p->amp_min) / p->nbin_amp;
    q.dbin_vp = (c->model.vp_max - c->model.vp_min) / p->nbin_vp;
    q.dbin_vs = (c->model.vs_max - c->model.vs_min) / p->nbin_vs;
    q.dbin_z = (c->model.z_max - 0.0) / p->nbin_z;
    q.dbin_vpvs = (c->model.vpvs_max - c->model.vpvs_min) / p->nbin_vpvs;
    q.z_min = p->z_min;
    q.vp_min = c->model.vp_min; q.vs_min = c->model.vs_min; q.vpvs_min = c->model.vpvs_min;
    q.max_models = p->max_models;
    std::vector<double> smin(p->sig_min, p->sig_min + ntrc), dsig(ntrc);
    std::vector<int> smode(p->sig_mode, p->sig_mode + ntrc);
    for (int t = 0; t < ntrc; ++t) dsig[t] = (p->sig_max[t] - p->sig_min[t]) / p->nbin_sig;
    if (upload(c, smin, &q.sig_min) || upload(c, dsig, &q.dbin_sig) || upload(c, smode, &q.sig_mode)) return 1;

    c->pst = PostState{};
    PostState &st = c->pst;
    const size_t nm = (size_t)std::max<int64_t>(p->max_models, 1);
    {
        PostState &a = st;
        if (post_alloc(c, &a.nmod, 2, true) || post_alloc(c, &a.nk, kmax, true) ||
            post_alloc(c, &a.nz, q.nbin_z, true) || post_alloc(c, &a.nsig, (size_t)ntrc * q.nbin_sig, true) ||
            post_alloc(c, &a.namp, (size_t)ntrc * nsmp * q.nbin_amp, true) ||
            post_alloc(c, &a.nvpz, (size_t)q.nbin_vp * q.nbin_z, true) ||
            post_alloc(c, &a.nvsz, (size_t)q.nbin_vs * q.nbin_z, true) ||
            post_alloc(c, &a.nvpvsz, (size_t)q.nbin_vpvs * q.nbin_z, true) ||
            post_alloc(c, &a.vp_mean, q.nbin_z, true) || post_alloc(c, &a.vs_mean, q.nbin_z, true) ||
            post_alloc(c, &a.vpvs_mean, q.nbin_z, true) || post_alloc(c, &a.vp_model, nm * q.nbin_z, true) ||
            post_alloc(c, &a.vs_model, nm * q.nbin_z, true) || post_alloc(c, &a.all_likelihood, nm, true) ||
            post_alloc(c, &a.amp_oor, 1, true))
            return 1;
    }
    // scratch of a record call
    if (post_alloc(c, &st.sel, c->nslots, false) ||
        post_alloc(c, &st.nsel, 1, true) || post_alloc(c, &st.row_a, (size_t)c->nslots * q.nbin_z, false) ||
        post_alloc(c, &st.row_b, (size_t)c->nslots * q.nbin_z, false) ||
        post_alloc(c, &c->d_post_nlay, c->nslots, false) || post_alloc(c, &c->d_post_flag, c->nslots, false) ||
        post_alloc(c, &c->d_post_k, 2 * (size_t)c->nslots, false) ||
        post_alloc(c, &c->d_post_layers, (size_t)c->nslots * 4 * c->fm_pad, false) ||
        post_alloc(c, &c->d_post_scratch, (size_t)c->nslots * 3 * kmax, false) ||
        post_alloc(c, &c->d_post_in, (size_t)c->nslots * (3 * (size_t)kmax + ntrc + 2), false))
        return 1;
    c->have_post = true;
    return rf_post_reset(c);
}

extern "C" int rf_post_record_device(rf_ctx *c, int32_t n, const int32_t *d_walker_ids, const int32_t *d_k,
                                     const double *d_z, const double *d_dvp, const double *d_dvs,
                                     const double *d_sig, const double *d_logl, const double *d_temps, void *stream)
{
    if (!c || !d_walker_ids || !d_k || !d_z || !d_dvp || !d_dvs || !d_sig || !d_logl)
        return fail("rf_post_record_device: null argument");
    if (!c->have_post) return fail("rf_post_record_device: rf_post_create has not been called");
    if (n <= 0) return 0;
    if (n > c->nslots) return fail("rf_post_record_device: batch larger than max_walkers + 1");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (order_after_commit(c, s)) return 1;
    // format_model of every chain in the batch (:240-242); rows of filtered-out chains are unused
    FormatParams P{c->model, n, c->fm_pad, d_k, d_z, d_dvp, d_dvs, nullptr, c->d_post_nlay, c->d_post_layers,
                   c->d_post_flag, nullptr, c->d_post_scratch};
    launch_format_model(P, s);
    PostBatch b{n, d_walker_ids, d_k, d_z, d_sig, d_logl, d_temps, c->d_post_nlay, c->d_post_layers, c->fm_pad};
    launch_post_record(c->post, c->pst, b, c->ws, s);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rf_post_record(rf_ctx *c, int32_t n, const int32_t *walker_ids, const int32_t *k, const double *z,
                              const double *dvp, const double *dvs, const double *sig, const double *logl,
                              const double *temps)
{
    if (!c || !walker_ids || !k || !z || !dvp || !dvs || !sig || !logl) return fail("rf_post_record: null argument");
    if (!c->have_post) return fail("rf_post_record: rf_post_create has not been called");
    if (n <= 0) return 0;
    if (n > c->nslots) return fail("rf_post_record: batch larger than max_walkers + 1");
    const int kmax = c->model.k_max, ntrc = c->cfg.ntrc;
    for (int i = 0; i < n; ++i) {
        if (walker_ids[i] < 0 || walker_ids[i] >= c->nslots) return fail("rf_post_record: walker out of range");
        if (k[i] < 1 || k[i] >= kmax) return fail("rf_post_record: k out of range");
    }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t N = (size_t)n;
    double *dz = c->d_post_in, *ddvp = dz + N * (kmax - 1), *ddvs = ddvp + N * kmax, *dsig = ddvs + N * kmax,
           *dlogl = dsig + N * ntrc, *dtemps = dlogl + N;
    int *dids = c->d_post_k, *dk = dids + n;
    // Like rf_commit, a record returns nothing from the device and does not wait for it: PAGEABLE host arrays are
    // copied into a pinned arena of their own before the call returns (the caller may change them at once); arrays in
    // pinned memory (rf_host_alloc) are read by DMA in place AFTER it returns and must stay untouched until a later call
    // on the context has waited for work issued behind this one (rf_eval_wait, rf_post_read ...).  The transfers and
    // kernels are stream-ordered behind everything issued before and in front of everything issued after.
    rf_ctx::Arena &A = c->arena[RF_EVAL_MAX_IN_FLIGHT + 1];
    if (arena_begin(A, sizeof(int) * 2 * N + sizeof(double) * N * (3 * (size_t)kmax + ntrc + 2))) return 1;
    if (h2d(A, dids, walker_ids, sizeof(int) * N, s) || h2d(A, dk, k, sizeof(int) * N, s) ||
        h2d(A, dz, z, sizeof(double) * N * (kmax - 1), s) || h2d(A, ddvp, dvp, sizeof(double) * N * kmax, s) ||
        h2d(A, ddvs, dvs, sizeof(double) * N * kmax, s) || h2d(A, dsig, sig, sizeof(double) * N * ntrc, s) ||
        h2d(A, dlogl, logl, sizeof(double) * N, s))
        return 1;
    if (temps && h2d(A, dtemps, temps, sizeof(double) * N, s)) return 1;
    if (rf_post_record_device(c, n, dids, dk, dz, ddvp, ddvs, dsig, dlogl, temps ? dtemps : nullptr, s)) return 1;
    return arena_mark(A, s);
}

extern "C" int rf_post_read(rf_ctx *c, const rf_post_result *o)
{
    if (!c || !o) return fail("rf_post_read: null argument");
    if (!c->have_post) return fail("rf_post_read: rf_post_create has not been called");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (device_error(c, "rf_post_read")) return 1;
    const PostConfig &q = c->post;
    const PostState &st = c->pst;
    const size_t nz = q.nbin_z, nm = (size_t)q.max_models;
    auto get = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        if (!dst || !bytes) return hipSuccess;
        return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    };
    int nmod = 0;
    HIP_TRY(hipMemcpy(&nmod, st.nmod, sizeof(int), hipMemcpyDeviceToHost));
    if (o->nmod) *o->nmod = nmod;
    const size_t nrows = std::min((size_t)std::max(nmod, 0), nm);   // model slots in use
    HIP_TRY(get(o->nk, st.nk, sizeof(int) * q.k_max));
    HIP_TRY(get(o->nz, st.nz, sizeof(int) * nz));
    HIP_TRY(get(o->nsig, st.nsig, sizeof(int) * (size_t)q.ntrc * q.nbin_sig));
    HIP_TRY(get(o->namp, st.namp, sizeof(int) * (size_t)q.ntrc * q.nsmp * q.nbin_amp));
    HIP_TRY(get(o->nvpz, st.nvpz, sizeof(int) * (size_t)q.nbin_vp * nz));
    HIP_TRY(get(o->nvsz, st.nvsz, sizeof(int) * (size_t)q.nbin_vs * nz));
    HIP_TRY(get(o->nvpvsz, st.nvpvsz, sizeof(int) * (size_t)q.nbin_vpvs * nz));
    HIP_TRY(get(o->vp_mean, st.vp_mean, sizeof(double) * nz));
    HIP_TRY(get(o->vs_mean, st.vs_mean, sizeof(double) * nz));
    HIP_TRY(get(o->vpvs_mean, st.vpvs_mean, sizeof(double) * nz));
    HIP_TRY(get(o->vp_model, st.vp_model, sizeof(double) * nrows * nz));
    HIP_TRY(get(o->vs_model, st.vs_model, sizeof(double) * nrows * nz));
    HIP_TRY(get(o->all_likelihood, st.all_likelihood, sizeof(double) * nrows));
    HIP_TRY(get(o->amp_out_of_range, st.amp_oor, sizeof(long long)));
    return 0;
}

// ---------------------------------------------------------------------------
// launch-plan options.  The library reads NO environment variables: every knob is an explicit
// call, validated, and visible in rf_get_launch_plan (bench.py echoes it).
// ---------------------------------------------------------------------------
static int upload_bin_cutoff(rf_ctx *c)
{
    const int ntrc = c->cfg.ntrc, nh = c->nh;
    if (!(c->bin_cutoff > 0.0)) {
        c->tab.nh_active = nullptr;
        return 0;
    }
    std::vector<int> act(ntrc);
    for (int t = 0; t < ntrc; ++t) {
        int last = 0;
        for (int k = 0; k < nh; ++k)
            if (c->flt[(size_t)k + (size_t)nh * t] >= c->bin_cutoff * c->flt[(size_t)nh * t]) last = k;
        act[t] = last + 1;
    }
    // one ntrc-sized buffer for the life of the context, overwritten in place (rf_set_option has synchronised the
    // device: nothing in flight reads it)
    if (!c->d_nh_active) {
        void *p = nullptr;
        if (dev_alloc(c, &p, sizeof(int) * ntrc)) return 1;
        c->d_nh_active = (int *)p;
    }
    HIP_TRY(hipMemcpy(c->d_nh_active, act.data(), sizeof(int) * ntrc, hipMemcpyHostToDevice));
    c->tab.nh_active = c->d_nh_active;
    return 0;
}

extern "C" int rf_set_option(rf_ctx *c, const char *name, double value)
{
    if (!c || !name) return fail("rf_set_option: null argument");
    const std::string k(name);
    const int iv = (int)value;
    const bool integral = (double)iv == value;
    HIP_TRY(hipSetDevice(c->device));
    // the *_device entry points run on the caller's streams: an option must not change the launch plan or a table
    // under a batch in flight on any of them
    HIP_TRY(hipDeviceSynchronize());
    if (k == "fused") {
        if (!integral || iv < -1 || iv > 1) return fail("rf_set_option: fused must be -1 (by shape), 0 or 1");
        if (iv == 1 && !c->fused_allowed && !c->fusedc_allowed)
            return fail("rf_set_option: this context cannot use the fused kernel (common rays or LDS footprint)");
        c->fused_override = iv;
    } else if (k == "chain") {
        if (!integral || !(iv == -1 || iv == 0 || iv == 2 || iv == 3 || iv == 4 || iv == 8))
            return fail("rf_set_option: chain must be -1 (by shape), 0, 2, 3, 4 or 8");
        c->chain_override = iv;
    } else if (k == "lpt") {
        if (!integral || iv < 0 || iv > 1) return fail("rf_set_option: lpt must be 0 or 1");
        c->lpt = iv != 0;
    } else if (k == "order_reuse") {
        if (!integral || iv < 0 || iv > 1) return fail("rf_set_option: order_reuse must be 0 or 1");
        c->order_reuse = iv != 0;
        c->order_next_nb = 0;
    } else if (k == "nsplit") {
        if (!integral || iv < 0 || iv > 64) return fail("rf_set_option: nsplit must be 0 (by batch size) .. 64");
        c->nsplit_override = iv;
    } else if (k == "waves_per_block") {
        if (!integral || iv < 1 || iv > 4) return fail("rf_set_option: waves_per_block must be 1 .. 4");
        c->waves_per_block = iv;
    } else if (k == "defer_logl") {
        if (!integral || iv < -1 || iv > 1) return fail("rf_set_option: defer_logl must be -1 (by batch size), 0 or 1");
        c->defer_logl = iv;
    } else if (k == "block_threads") {
        if (!integral || !(iv == 0 || iv == 256 || iv == 512))
            return fail("rf_set_option: block_threads must be 0 (by batch size), 256 or 512");
        c->block_threads = iv;
    } else if (k == "gemm_tile") {
        if (!integral || (iv != 0 && iv != 64 && iv != 128)) return fail("rf_set_option: gemm_tile must be 0 (by launch size), 64 or 128");
        c->pg.tile = iv;
    } else if (k == "copy_stream") {
        if (!integral || iv < 0 || iv > 1) return fail("rf_set_option: copy_stream must be 0 or 1");
        c->use_copy_stream = iv != 0;
    } else if (k == "gemm_triangle") {
        if (!integral || iv < 0 || iv > 1) return fail("rf_set_option: gemm_triangle must be 0 or 1");
        c->pg.triangle = iv;
    } else if (k == "trace_window") {
        if (!integral || iv < 0 || iv > 1) return fail("rf_set_option: trace_window must be 0 or 1");
        const int len = iv ? c->cfg.nsmp : c->cfg.nfft;
        if (len != c->ws.trace_len && alloc_traces(c, len)) return 1;
        c->trace_window = iv;
    } else if (k == "bin_cutoff") {
        if (!(value >= 0.0 && value < 1.0)) return fail("rf_set_option: bin_cutoff must be in [0, 1)");
        c->bin_cutoff = value;
        if (upload_bin_cutoff(c)) return 1;
#ifdef RFGPU_DIAGNOSTICS
    } else if (k == "ablate") {
        c->ablate = iv;   // timing diagnostics: the kernel stops early, results are INVALID
#endif
    } else {
        return fail("rf_set_option: unknown option '" + k + "'");
    }
    default_plan(c);
    if (!c->fused && !c->fusedc && ensure_spec(c)) return 1;
    c->n_overrides = (c->fused_override != -1) + (c->chain_override != -1) + (!c->lpt) + (!c->order_reuse) +
                     (c->nsplit_override != 0) + (c->waves_per_block != 4) + (c->defer_logl != -1) +
                     (c->block_threads != 0) + (c->bin_cutoff > 0.0) + (c->ablate != 0) + (c->trace_window != 0) +
                     (c->pg.tile != 0) + (c->tab.phi_gemm && c->pg.triangle != 1) + (c->use_copy_stream ? 1 : 0);
    return 0;
}

extern "C" int rf_get_launch_plan(const rf_ctx *c, int32_t *plan)
{
    if (!c || !plan) return fail("rf_get_launch_plan: null argument");
    plan[0] = c->fusedc ? 2 : (c->fused ? 1 : 0);
    plan[1] = c->chain;
    plan[2] = c->waves_per_block;
    plan[3] = pick_nsplit(c, c->cfg.max_walkers);
    plan[4] = c->lpt ? 1 : 0;
    plan[5] = c->order_reuse ? 1 : 0;
    plan[6] = c->defer_logl;
    plan[7] = c->bin_cutoff > 0.0 ? 1 : 0;
    plan[8] = c->n_overrides;
#ifdef RFGPU_DIAGNOSTICS
    plan[9] = 1 + (c->ablate != 0);
#else
    plan[9] = 0;
#endif
    plan[10] = c->block_threads;
    plan[11] = (c->fusedc || use_fused8(c)) ? 512 : 256;
    plan[12] = c->tab.phi_gemm ? (c->pg.triangle ? 2 : 1) : 0;
    plan[13] = c->trace_window;
    plan[14] = c->last_staged;
    plan[15] = c->use_copy_stream ? 1 : 0;
    return 0;
}

extern "C" int rf_profile_enable(rf_ctx *c, int32_t on)
{
    if (!c) return fail("rf_profile_enable: null context");
    if (!on) flush_profile(c);
    c->prof = on != 0;
    c->prof_every = on > 1 ? on : 1;
    c->prof_batch = 0;
    return 0;
}

extern "C" int rf_profile_read(rf_ctx *c, double *ms, int64_t *launches, int32_t reset)
{
    if (!c || !ms || !launches) return fail("rf_profile_read: null argument");
    flush_profile(c);
    if (device_error(c, "rf_profile_read")) return 1;
    for (int i = 0; i < 3; ++i) ms[i] = c->prof_ms[i];
    for (int i = 0; i < 4; ++i) launches[i] = c->prof_n[i];
    if (reset) {
        c->prof_ms[0] = c->prof_ms[1] = c->prof_ms[2] = 0;
        c->prof_n[0] = c->prof_n[1] = c->prof_n[2] = c->prof_n[3] = 0;
    }
    return 0;
}
