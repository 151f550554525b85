// rfgpu_comm.cpp -- the parallel-tempering temperature exchange over RCCL (xGMI): the C-ABI replacement of the
// MPI traffic of reference src/pt_mcmc.f90:498-571 (mpi_bcast of the chosen pair :518, the (T, logL) message
// :544-545 / :566-567 and the returned temperature :555-556 / :568-569).
//
// One process per GPU; walkers shard across ranks in contiguous blocks and never migrate -- only temperatures
// move (:532-535, :551-554, :570).  Payloads are 8-24 bytes, so everything here is latency-bound: each step is
// ONE grouped RCCL call.  RCCL is loaded at run time (dlopen of librccl.so.1 -- inside a PyTorch process that is
// the RCCL torch already loaded); librfgpu has no link-time dependency on it and single-GPU users never touch it.
#include "rfgpu_internal.h"
#include "../../include/rfgpu_ext.h"

#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h>

namespace rfgpu {
int comm_fail(const std::string &msg);   // sets rf_last_error (rfgpu_api.cpp)
hipStream_t ctx_stream(rf_ctx *c);
int ctx_device(rf_ctx *c);
CommState *&ctx_comm(rf_ctx *c);
bool ctx_post(rf_ctx *c, const PostConfig **q, const PostState **st);   // false: no rf_post_create yet
}
using namespace rfgpu;

namespace {
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

std::string g_rccl_path;   // rf_comm_set_library: this file instead of the default search
bool g_rccl_tried = false;

Rccl *rccl()
{
    static Rccl r;
    if (g_rccl_tried) return r.h ? &r : nullptr;
    g_rccl_tried = true;
    if (!g_rccl_path.empty()) {
        r.h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
    } else {
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
        }
    }
    if (!r.h) return nullptr;
#define RF_SYM(n) r.n = reinterpret_cast<decltype(r.n)>(dlsym(r.h, "nccl" #n))
    RF_SYM(GetUniqueId); RF_SYM(CommInitRank); RF_SYM(CommDestroy); RF_SYM(GetErrorString); RF_SYM(GetVersion); RF_SYM(Broadcast);
    RF_SYM(AllGather); RF_SYM(Reduce); RF_SYM(Send); RF_SYM(Recv); RF_SYM(GroupStart); RF_SYM(GroupEnd);
#undef RF_SYM
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Broadcast || !r.AllGather || !r.Reduce || !r.Send || !r.Recv ||
        !r.GroupStart || !r.GroupEnd) {
        dlclose(r.h);
        r.h = nullptr;
        return nullptr;
    }
    return &r;
}
}   // namespace

struct rfgpu::CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    // staging, three disjoint areas of 8 doubles (the host-value exchanges run on the communicator's own stream, the merge on
    // the evaluation stream: no two entry points share a word): [0..7] rf_pt_swap_exchange ([0..2] outgoing (T, logL,
    // log u), [4..6] incoming), [8..15] rf_comm_bcast_i32, [16..23] rf_comm_post_reduce
    double *d_buf = nullptr;
    double *h_buf = nullptr;     // pinned mirror
    double *d_gather = nullptr;  // [2][nranks * nchains]: every rank's T, then every rank's logL, by global walker id
    size_t gather_doubles = 0;
    // The host-value exchanges (rf_comm_bcast_i32, rf_pt_swap_exchange) run on a stream of the communicator's own: what
    // they carry are host scalars with no ordering against the evaluation stream, and a sampler calls them while a
    // segment's kernels are in flight there (pt_control_batched: the pair is drawn right after rf_eval_models_begin) --
    // on the evaluation stream the host would wait for the whole iteration's kernels before its 16 bytes moved.
    hipStream_t stream = nullptr;
    bool sequential_reduce = false;   // rf_comm_set_option("sequential_reduce")
};

#define RCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return comm_fail(std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(r_) : "RCCL error")); \
    } while (0)
#define HIPC_TRY(expr)                                                                  \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return comm_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Which RCCL to load: a file path instead of the default search (librccl.so.1 by soname, then /opt/rocm/lib).  For
// sites with several RCCL builds -- and for the two-ranks-on-one-GPU test, which points it at a host-staged test
// double (tests/c/rccl_double.cpp).  Process-wide; must come before the first call that needs RCCL.
extern "C" int rf_comm_set_library(const char *path)
{
    if (!path || !*path) return comm_fail("rf_comm_set_library: empty path");
    if (g_rccl_tried) return comm_fail("rf_comm_set_library: RCCL has already been loaded in this process");
    g_rccl_path = path;
    return 0;
}

// rf_comm_device_key: the context's physical GPU -- a hash of the machine (host name and boot id: ranks of different
// nodes with the same PCI address are NOT the same GPU) over PCI domain / bus / device.  Loads nothing.
// rf_comm_probe: the same key, and can this rank take part in an RCCL communicator?  0 when librccl.so.1 loads (that
// takes about a second: hosts compare the keys first and probe only when every rank has a GPU of its own).
// The host gathers the answers of all ranks and uses RCCL only when every rank answered 0 and all keys differ --
// ncclCommInitRank is collective, so the decision has to be unanimous BEFORE anyone enters it.
extern "C" int rf_comm_device_key(rf_ctx *c, int64_t *device_key)
{
    if (!c || !device_key) return comm_fail("rf_comm_device_key: null argument");
    hipDeviceProp_t prop;
    HIPC_TRY(hipGetDeviceProperties(&prop, ctx_device(c)));
    // the machine: host name + boot id (containers of different nodes may share a host name; a boot id they do not)
    char host[256] = {0};
    (void)gethostname(host, sizeof host - 1);
    uint32_t hh = 2166136261u;                         // FNV-1a
    for (const char *q = host; *q; ++q) hh = (hh ^ (uint8_t)*q) * 16777619u;
    if (FILE *fh = std::fopen("/proc/sys/kernel/random/boot_id", "r")) {
        char id[64] = {0};
        if (std::fgets(id, sizeof id, fh))
            for (const char *q = id; *q && *q != '\n'; ++q) hh = (hh ^ (uint8_t)*q) * 16777619u;
        std::fclose(fh);
    }
    *device_key = ((int64_t)(hh & 0x7fffffffu) << 32) | ((int64_t)(prop.pciDomainID & 0xffff) << 16) |
                  ((int64_t)(prop.pciBusID & 0xff) << 8) | (int64_t)(prop.pciDeviceID & 0xff);
    return 0;
}

extern "C" int rf_comm_probe(rf_ctx *c, int64_t *device_key)
{
    if (rf_comm_device_key(c, device_key)) return 1;
    if (!rccl()) return comm_fail("rf_comm_probe: librccl.so.1 cannot be loaded");
    return 0;
}

extern "C" int rf_comm_get_unique_id(uint8_t *id)
{
    if (!id) return comm_fail("rf_comm_get_unique_id: null argument");
    Rccl *R = rccl();
    if (!R) return comm_fail("rf_comm_get_unique_id: librccl.so.1 cannot be loaded");
    ncclUniqueId u;
    RCCL_TRY(R->GetUniqueId(&u));
    static_assert(sizeof(u) == RF_COMM_ID_BYTES, "RCCL unique id size");
    std::memcpy(id, &u, sizeof u);
    return 0;
}

extern "C" int rf_comm_init(rf_ctx *c, const uint8_t *id, int32_t rank, int32_t nranks)
{
    if (!c || !id) return comm_fail("rf_comm_init: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return comm_fail("rf_comm_init: bad rank / nranks");
    if (ctx_comm(c)) return comm_fail("rf_comm_init: the context already has a communicator");
    Rccl *R = rccl();
    if (!R) return comm_fail("rf_comm_init: librccl.so.1 cannot be loaded");
    HIPC_TRY(hipSetDevice(ctx_device(c)));
    CommState *s = new CommState();
    s->rank = rank;
    s->nranks = nranks;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t r = R->CommInitRank(&s->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete s;
        return comm_fail(std::string("rf_comm_init: ncclCommInitRank: ") + (R->GetErrorString ? R->GetErrorString(r) : "error") +
                         " (one rank per GPU is required: RCCL refuses two ranks on one device)");
    }
    if (hipMalloc((void **)&s->d_buf, sizeof(double) * 24) != hipSuccess ||
        hipHostMalloc((void **)&s->h_buf, sizeof(double) * 24, hipHostMallocDefault) != hipSuccess ||
        hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        R->CommDestroy(s->comm);
        if (s->d_buf) (void)hipFree(s->d_buf);
        if (s->h_buf) (void)hipHostFree(s->h_buf);
        delete s;
        return comm_fail("rf_comm_init: staging allocation failed");
    }
    ctx_comm(c) = s;
    return 0;
}

extern "C" int rf_comm_destroy(rf_ctx *c)
{
    if (!c) return 0;
    CommState *s = ctx_comm(c);
    if (!s) return 0;
    (void)hipSetDevice(ctx_device(c));
    (void)hipStreamSynchronize(ctx_stream(c));
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    Rccl *R = rccl();
    if (R && s->comm) R->CommDestroy(s->comm);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->d_buf) (void)hipFree(s->d_buf);
    if (s->h_buf) (void)hipHostFree(s->h_buf);
    if (s->d_gather) (void)hipFree(s->d_gather);
    delete s;
    ctx_comm(c) = nullptr;
    return 0;
}

// mpi_bcast(ipack, 4, MPI_INTEGER4, 0, ...) of src/pt_mcmc.f90:518-519: the pair rank 0 drew
extern "C" int rf_comm_bcast_i32(rf_ctx *c, int32_t *buf, int32_t n, int32_t root)
{
    if (!c || !buf) return comm_fail("rf_comm_bcast_i32: null argument");
    CommState *s = ctx_comm(c);
    if (!s) return comm_fail("rf_comm_bcast_i32: rf_comm_init has not been called");
    if (n < 1 || n > 8) return comm_fail("rf_comm_bcast_i32: n must be 1 .. 8");
    if (root < 0 || root >= s->nranks) return comm_fail("rf_comm_bcast_i32: root out of range");
    Rccl *R = rccl();
    hipStream_t st = s->stream;
    HIPC_TRY(hipSetDevice(ctx_device(c)));
    int32_t *hb = reinterpret_cast<int32_t *>(s->h_buf + 8), *db = reinterpret_cast<int32_t *>(s->d_buf + 8);
    if (s->rank == root) {
        std::memcpy(hb, buf, sizeof(int32_t) * n);
        HIPC_TRY(hipMemcpyAsync(db, hb, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    }
    RCCL_TRY(R->Broadcast(db, db, (size_t)n, ncclInt32, root, s->comm, st));
    HIPC_TRY(hipMemcpyAsync(hb, db, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIPC_TRY(hipStreamSynchronize(st));
    std::memcpy(buf, hb, sizeof(int32_t) * n);
    return 0;
}

// The cross-rank branch of the swap (src/pt_mcmc.f90:542-571) as ONE grouped send + receive: both ranks
// exchange (T, logL, log u) and form the same judge_pt decision (:580-595) from rank1's uniform -- the
// reference lets rank1 judge with its own RNG and mail the temperature back; here the second message is
// replaced by shipping rank1's log u along with the first.  judge != 0 on the rank that owns chain 1 (its
// log_u is the one used; the peer's argument is ignored).
extern "C" int rf_pt_swap_exchange(rf_ctx *c, int32_t peer, int32_t judge, double temp, double logl, double log_u,
                                   double *new_temp, int32_t *accepted)
{
    if (!c || !new_temp) return comm_fail("rf_pt_swap_exchange: null argument");
    CommState *s = ctx_comm(c);
    if (!s) return comm_fail("rf_pt_swap_exchange: rf_comm_init has not been called");
    if (peer < 0 || peer >= s->nranks) return comm_fail("rf_pt_swap_exchange: peer out of range");
    Rccl *R = rccl();
    hipStream_t st = s->stream;
    HIPC_TRY(hipSetDevice(ctx_device(c)));
    s->h_buf[0] = temp;
    s->h_buf[1] = logl;
    s->h_buf[2] = log_u;
    HIPC_TRY(hipMemcpyAsync(s->d_buf, s->h_buf, sizeof(double) * 3, hipMemcpyHostToDevice, st));
    RCCL_TRY(R->GroupStart());
    RCCL_TRY(R->Send(s->d_buf, 3, ncclDouble, peer, s->comm, st));
    RCCL_TRY(R->Recv(s->d_buf + 4, 3, ncclDouble, peer, s->comm, st));
    RCCL_TRY(R->GroupEnd());
    HIPC_TRY(hipMemcpyAsync(s->h_buf + 4, s->d_buf + 4, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
    HIPC_TRY(hipStreamSynchronize(st));
    const double t_peer = s->h_buf[4], l_peer = s->h_buf[5], u_peer = s->h_buf[6];
    // chain 1 = the judge's, chain 2 = the other one:  log u <= (L2 - L1) (1/T1 - 1/T2)
    const double t1 = judge ? temp : t_peer, t2 = judge ? t_peer : temp;
    const double l1 = judge ? logl : l_peer, l2 = judge ? l_peer : logl;
    const double lu = judge ? log_u : u_peer;
    const double del_s = (l2 - l1) * (1.0 / t1 - 1.0 / t2);
    const int yes = lu <= del_s;
    *new_temp = yes ? t_peer : temp;
    if (accepted) *accepted = yes;
    return 0;
}

// rank / size of the context's communicator and the RCCL version it runs on (major * 10000 + minor * 100 + patch,
// ncclGetVersion); any pointer may be NULL.  Without a communicator: rank 0 of 1, version of the loadable library.
extern "C" int rf_comm_info(rf_ctx *c, int32_t *rank, int32_t *nranks, int32_t *rccl_version)
{
    if (!c) return comm_fail("rf_comm_info: null context");
    CommState *s = ctx_comm(c);
    if (rank) *rank = s ? s->rank : 0;
    if (nranks) *nranks = s ? s->nranks : 1;
    if (rccl_version) {
        *rccl_version = 0;
        Rccl *R = rccl();
        int v = 0;
        if (R && R->GetVersion && R->GetVersion(&v) == ncclSuccess) *rccl_version = v;
    }
    return 0;
}

// Options of the context's communicator (after rf_comm_init).  "sequential_reduce" 0 (default) | 1: rf_comm_post_reduce
// issues its twelve ncclReduce calls one by one instead of as one group.
extern "C" int rf_comm_set_option(rf_ctx *c, const char *name, double value)
{
    if (!c || !name) return comm_fail("rf_comm_set_option: null argument");
    CommState *s = ctx_comm(c);
    if (!s) return comm_fail("rf_comm_set_option: rf_comm_init has not been called");
    const std::string k(name);
    if (k == "sequential_reduce") {
        if (value != 0.0 && value != 1.0) return comm_fail("rf_comm_set_option: sequential_reduce must be 0 or 1");
        s->sequential_reduce = value != 0.0;
        return 0;
    }
    return comm_fail("rf_comm_set_option: unknown option '" + k + "'");
}

// Throughput form: K DISJOINT pairs per iteration over global walker ids (rank * nchains + chain,
// src/pt_mcmc.f90:508-511).  ONE RCCL group -- two all-gathers straight from the caller's arrays, every rank's T
// into g_t[nranks * nchains] and every rank's logL into g_l: rank blocks in rank order ARE the global-id order --
// then ONE kernel that reads the gathered snapshot and writes this rank's own temperatures in place
// (pt_swap_gathered_kernel).  No staging copies: 16 bytes per walker cross xGMI, nothing else moves.
extern "C" int rf_pt_swap_allgather_device(rf_ctx *c, int32_t nchains, int32_t npairs, const int32_t *d_pairs,
                                           const double *d_log_u, double *d_temps, const double *d_logl, void *stream)
{
    if (!c || !d_pairs || !d_log_u || !d_temps || !d_logl) return comm_fail("rf_pt_swap_allgather_device: null argument");
    CommState *s = ctx_comm(c);
    if (!s) return comm_fail("rf_pt_swap_allgather_device: rf_comm_init has not been called");
    if (nchains < 1) return comm_fail("rf_pt_swap_allgather_device: nchains < 1");
    if (npairs <= 0) return 0;
    Rccl *R = rccl();
    hipStream_t st = (hipStream_t)stream;
    HIPC_TRY(hipSetDevice(ctx_device(c)));
    const size_t all = (size_t)s->nranks * (size_t)nchains, need = 2 * all;
    if (need > s->gather_doubles) {
        // (growth only: the first call of a run; the old buffer may still be read by work in flight on `st`)
        HIPC_TRY(hipStreamSynchronize(st));
        if (s->d_gather) (void)hipFree(s->d_gather);
        s->d_gather = nullptr;
        s->gather_doubles = 0;
        HIPC_TRY(hipMalloc((void **)&s->d_gather, sizeof(double) * need));
        s->gather_doubles = need;
    }
    double *g_t = s->d_gather, *g_l = s->d_gather + all;
    RCCL_TRY(R->GroupStart());
    RCCL_TRY(R->AllGather(d_temps, g_t, (size_t)nchains, ncclDouble, s->comm, st));
    RCCL_TRY(R->AllGather(d_logl, g_l, (size_t)nchains, ncclDouble, s->comm, st));
    RCCL_TRY(R->GroupEnd());
    launch_pt_swap_gathered(npairs, d_pairs, d_log_u, g_t, g_l, nchains, s->rank, s->nranks, d_temps, nullptr, st);
    HIPC_TRY(hipGetLastError());
    return 0;
}

// ---- end-of-run merge of the posterior accumulators (last sentence) -------------------------------
// The top of output_results (src/mcmc_out.f90:52-79) for the accumulators rf_post_* keeps on the device: its
// mpi_reduce(SUM -> 0) of nk, namp, nvpz, nvsz, nvpvsz, nz, nsig (int32) and vp_mean, vs_mean, vpvs_mean (f64) as ONE
// RCCL group of ncclReduce(sum) IN PLACE into the root's accumulators -- histograms go GPU to GPU, nothing is staged
// through the hosts; rf_post_read on the root then returns the merged arrays (the other ranks' accumulators are
// unchanged).  nmod (:52) is summed into *nmod_sum instead (root only): the root's own count stays what names its
// model rows.  amp_out_of_range (this library's counter for the reference's warning line) is summed too.  The proposal
// counters and likelihood_hist of :54-57,72-73 live in the host's modules and stay with the host's own reduce.
// Collective over the communicator; ONCE per run (a second call would add the other ranks' counts again).  fp64 sums:
// RCCL's order over ranks, like MPI's, is the library's; with two ranks it is the one possible order.
extern "C" int rf_comm_post_reduce(rf_ctx *c, int32_t root, int32_t *nmod_sum)
{
    if (!c) return comm_fail("rf_comm_post_reduce: null context");
    CommState *s = ctx_comm(c);
    if (!s) return comm_fail("rf_comm_post_reduce: rf_comm_init has not been called");
    if (root < 0 || root >= s->nranks) return comm_fail("rf_comm_post_reduce: root out of range");
    const PostConfig *q;
    const PostState *p;
    if (!ctx_post(c, &q, &p)) return comm_fail("rf_comm_post_reduce: rf_post_create has not been called");
    Rccl *R = rccl();
    hipStream_t st = ctx_stream(c);
    HIPC_TRY(hipSetDevice(ctx_device(c)));
    const size_t nz = q->nbin_z;
    const struct { void *ptr; size_t n; ncclDataType_t t; } items[] = {
        {p->nk, (size_t)q->k_max, ncclInt32},
        {p->nz, nz, ncclInt32},
        {p->nsig, (size_t)q->ntrc * q->nbin_sig, ncclInt32},
        {p->namp, (size_t)q->ntrc * q->nsmp * q->nbin_amp, ncclInt32},
        {p->nvpz, (size_t)q->nbin_vp * nz, ncclInt32},
        {p->nvsz, (size_t)q->nbin_vs * nz, ncclInt32},
        {p->nvpvsz, (size_t)q->nbin_vpvs * nz, ncclInt32},
        {p->vp_mean, nz, ncclDouble},
        {p->vs_mean, nz, ncclDouble},
        {p->vpvs_mean, nz, ncclDouble},
        {p->amp_oor, 1, ncclInt64},
    };
    int32_t *d_nmod = reinterpret_cast<int32_t *>(s->d_buf + 16), *h_nmod = reinterpret_cast<int32_t *>(s->h_buf + 16);
    // one group of twelve reductions (one launch); rf_comm_set_option("sequential_reduce", 1): twelve plain calls, one
    // after the other on the stream -- the same sums, for an RCCL build that mishandles in-place reductions in a group
    if (!s->sequential_reduce) RCCL_TRY(R->GroupStart());
    for (const auto &it : items) RCCL_TRY(R->Reduce(it.ptr, it.ptr, it.n, it.t, ncclSum, root, s->comm, st));
    RCCL_TRY(R->Reduce(p->nmod, d_nmod, 1, ncclInt32, ncclSum, root, s->comm, st));
    if (!s->sequential_reduce) RCCL_TRY(R->GroupEnd());
    if (s->rank == root) HIPC_TRY(hipMemcpyAsync(h_nmod, d_nmod, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPC_TRY(hipStreamSynchronize(st));
    if (s->rank == root && nmod_sum) *nmod_sum = h_nmod[0];
    return 0;
}

// The two mpi_gather of src/mcmc_out.f90:88-93 (vs_model, vp_model: every rank's per-model profile rows, rank blocks in
// rank order on rank 0), plus all_likelihood, which the reference allocates for every rank (:84) but never gathers.
// nmod_rank[nranks] (every rank, may be NULL): the models each rank recorded.  On the root the host arrays
// vp_model_all / vs_model_all [nranks][max_models][nbin_z] and all_likelihood_all [nranks][max_models] (any may be
// NULL) receive, per rank block, the first min(nmod_rank[r], max_models) rows -- as rf_post_read, the later rows keep what
// the caller put there (init_pt_mcmc's vs_model(1,:) = -999.9, src/pt_mcmc.f90:419).  Rows travel device to device (one
// ncclSend per array and rank, the root receives into a staging block and copies it out); the other ranks' pointers
// are ignored.  Collective; call it BEFORE rf_comm_post_reduce or after, the rows are not touched by the reduce.
extern "C" int rf_comm_post_gather(rf_ctx *c, int32_t root, int32_t *nmod_rank, double *vp_model_all, double *vs_model_all,
                                   double *all_likelihood_all)
{
    if (!c) return comm_fail("rf_comm_post_gather: null context");
    CommState *s = ctx_comm(c);
    if (!s) return comm_fail("rf_comm_post_gather: rf_comm_init has not been called");
    if (root < 0 || root >= s->nranks) return comm_fail("rf_comm_post_gather: root out of range");
    const PostConfig *q;
    const PostState *p;
    if (!ctx_post(c, &q, &p)) return comm_fail("rf_comm_post_gather: rf_post_create has not been called");
    Rccl *R = rccl();
    hipStream_t st = ctx_stream(c);
    HIPC_TRY(hipSetDevice(ctx_device(c)));
    const int nr = s->nranks;
    const size_t nz = q->nbin_z, nm = (size_t)q->max_models;
    // every rank's count first (it sizes the messages): an all-gather of one int32
    int32_t *d_cnt = nullptr;
    HIPC_TRY(hipMalloc((void **)&d_cnt, sizeof(int32_t) * nr));
    std::vector<int32_t> cnt(nr);
    auto done = [&](int rc) {
        (void)hipFree(d_cnt);
        return rc;
    };
    {
        ncclResult_t r_ = R->AllGather(p->nmod, d_cnt, 1, ncclInt32, s->comm, st);
        if (r_ != ncclSuccess) return done(comm_fail("rf_comm_post_gather: ncclAllGather failed"));
        if (hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int32_t) * nr, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return done(comm_fail("rf_comm_post_gather: reading the counts failed"));
    }
    if (nmod_rank) std::memcpy(nmod_rank, cnt.data(), sizeof(int32_t) * nr);
    auto rows_of = [&](int r) { return std::min((size_t)std::max(cnt[r], 0), nm); };
    const struct { const double *dev; double *host; size_t width; } arr[3] = {
        {p->vp_model, vp_model_all, nz}, {p->vs_model, vs_model_all, nz}, {p->all_likelihood, all_likelihood_all, 1}};
    if (s->rank != root) {
        // (the root may not want an array: every rank still sends all three, the message pattern does not depend on the
        // root's pointers, which this rank cannot see)
        const size_t rows = rows_of(s->rank);
        for (const auto &a : arr) {
            if (!rows) break;              // (nothing recorded: the root, which has the counts too, posts no receive)
            ncclResult_t r_ = R->Send(a.dev, rows * a.width, ncclDouble, root, s->comm, st);
            if (r_ != ncclSuccess) return done(comm_fail("rf_comm_post_gather: ncclSend failed"));
        }
        if (hipStreamSynchronize(st) != hipSuccess) return done(comm_fail("rf_comm_post_gather: stream synchronisation failed"));
        return done(0);
    }
    size_t most = 0;
    for (int r = 0; r < nr; ++r) most = std::max(most, rows_of(r));
    double *d_stage = nullptr;
    if (most && hipMalloc((void **)&d_stage, sizeof(double) * most * nz) != hipSuccess)
        return done(comm_fail("rf_comm_post_gather: staging allocation failed"));
    auto done2 = [&](int rc) {
        if (d_stage) (void)hipFree(d_stage);
        return done(rc);
    };
    for (int r = 0; r < nr; ++r) {
        const size_t rows = rows_of(r);
        for (const auto &a : arr) {
            const double *src = a.dev;
            if (r != root && rows) {
                ncclResult_t r_ = R->Recv(d_stage, rows * a.width, ncclDouble, r, s->comm, st);
                if (r_ != ncclSuccess) return done2(comm_fail("rf_comm_post_gather: ncclRecv failed"));
                src = d_stage;
            }
            if (a.host && rows &&
                hipMemcpyAsync(a.host + (size_t)r * nm * a.width, src, sizeof(double) * rows * a.width, hipMemcpyDeviceToHost, st) != hipSuccess)
                return done2(comm_fail("rf_comm_post_gather: copy to the host failed"));
            // (the staging block is reused by the next message)
            if (hipStreamSynchronize(st) != hipSuccess) return done2(comm_fail("rf_comm_post_gather: stream synchronisation failed"));
        }
    }
    return done2(0);
}
