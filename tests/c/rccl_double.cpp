// rccl_double.cpp -- TEST DOUBLE of the twelve RCCL entry points rfgpu_comm.cpp binds.  Test infrastructure only:
// the product loads the real librccl.so.1; tests/test_pt_swap.py points librfgpu at this file with
// rf_comm_set_library so that TWO RANKS ON ONE GPU -- which real RCCL refuses -- drive rf_comm_init,
// rf_comm_bcast_i32, rf_pt_swap_exchange, rf_pt_swap_allgather_device and rf_comm_post_reduce / _gather with nranks = 2
// through the C ABI.
//
// Data moves through a file in /dev/shm named after the unique id: a collective synchronises its stream, copies its
// contribution device -> shared memory, waits for the other ranks, copies their contributions back to the device.
// Same call semantics as RCCL at the level librfgpu relies on: results are in place for later work on the stream;
// all-gather lays rank blocks out in rank order; reduce sums in rank order into the root's buffer only (send == recv
// allowed); sends and receives of one group do not deadlock (a long message moves in BOX-sized pieces).  Every wait gives
// up after 120 s with an error instead of hanging a test.
//   hipcc -shared -fPIC -o librccl_double.so tests/c/rccl_double.cpp
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

namespace {
constexpr int MAXR = 8;
constexpr size_t SLOT = 1u << 20;   // bytes one rank may contribute to a collective
constexpr size_t BOX = 1u << 14;    // bytes of one piece of a point-to-point message

struct Shm {
    volatile long posted[MAXR], readn[MAXR];                  // collectives: op number a rank has posted / finished reading
    volatile long p_posted[MAXR][MAXR], p_read[MAXR][MAXR];   // mailboxes [src][dst]
    char coll[MAXR][SLOT];
    char box[MAXR][MAXR][BOX];
};

struct Comm {
    int rank, nranks;
    Shm *shm;
    long seq;
    char path[96];
};

struct Op {
    int kind;   // 0 all-gather, 1 broadcast, 2 send, 3 recv, 4 reduce(sum) to `peer`
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t type;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local int g_nq = 0;
thread_local Op g_q[32];

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 2;
    }
}

double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

template <class F>
bool wait_for(F ready)
{
    const double t0 = now();
    while (!ready()) {
        if (now() - t0 > 120.0) return false;
        sched_yield();
    }
    __sync_synchronize();
    return true;
}

ncclResult_t run(const Op &o)
{
    Comm *c = o.comm;
    Shm *s = c->shm;
    const size_t bytes = o.count * type_bytes(o.type);
    if (o.kind <= 1 || o.kind == 4) {
        if (bytes > SLOT) return ncclInvalidArgument;
        const long n = ++c->seq;
        // nobody still reads what this rank posted for the previous collective
        if (!wait_for([&] { for (int r = 0; r < c->nranks; ++r) if (s->readn[r] < n - 1) return false; return true; }))
            return ncclSystemError;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (o.kind != 1 || c->rank == o.peer)
            if (hipMemcpy(s->coll[c->rank], o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        __sync_synchronize();
        s->posted[c->rank] = n;
        if (!wait_for([&] { for (int r = 0; r < c->nranks; ++r) if (s->posted[r] < n) return false; return true; }))
            return ncclSystemError;
        if (o.kind == 0) {
            for (int r = 0; r < c->nranks; ++r)
                if (hipMemcpy((char *)o.recv + (size_t)r * bytes, s->coll[r], bytes, hipMemcpyHostToDevice) != hipSuccess)
                    return ncclUnhandledCudaError;
        } else if (o.kind == 4) {
            if (c->rank == o.peer) {
                static thread_local char acc[SLOT];
                std::memset(acc, 0, bytes);
                for (int r = 0; r < c->nranks; ++r)
                    for (size_t i = 0; i < o.count; ++i) {
                        if (o.type == ncclInt32) reinterpret_cast<int *>(acc)[i] += reinterpret_cast<const int *>(s->coll[r])[i];
                        else if (o.type == ncclInt64) reinterpret_cast<long long *>(acc)[i] += reinterpret_cast<const long long *>(s->coll[r])[i];
                        else if (o.type == ncclFloat64) reinterpret_cast<double *>(acc)[i] += reinterpret_cast<const double *>(s->coll[r])[i];
                        else return ncclInvalidArgument;
                    }
                if (hipMemcpy(o.recv, acc, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            }
        } else if (hipMemcpy(o.recv, s->coll[o.peer], bytes, hipMemcpyHostToDevice) != hipSuccess) {
            return ncclUnhandledCudaError;
        }
        __sync_synchronize();
        s->readn[c->rank] = n;
        return ncclSuccess;
    }
    if (o.peer < 0 || o.peer >= c->nranks) return ncclInvalidArgument;
    if (o.kind == 2) {
        const int me = c->rank, to = o.peer;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        // (a message within one piece only fills the mailbox -- what lets a group send and then receive; a longer one
        // waits for the receiver piece by piece: such sends are issued one-way)
        for (size_t off = 0; off < bytes || off == 0; off += BOX) {
            const size_t n = bytes - off < BOX ? bytes - off : BOX;
            if (!wait_for([&] { return s->p_read[me][to] == s->p_posted[me][to]; })) return ncclSystemError;
            if (n && hipMemcpy(s->box[me][to], (const char *)o.send + off, n, hipMemcpyDeviceToHost) != hipSuccess)
                return ncclUnhandledCudaError;
            __sync_synchronize();
            s->p_posted[me][to] = s->p_posted[me][to] + 1;
        }
        return ncclSuccess;
    }
    const int from = o.peer, me = c->rank;
    for (size_t off = 0; off < bytes || off == 0; off += BOX) {
        const size_t n = bytes - off < BOX ? bytes - off : BOX;
        if (!wait_for([&] { return s->p_posted[from][me] > s->p_read[from][me]; })) return ncclSystemError;
        if (n && hipMemcpy((char *)o.recv + off, s->box[from][me], n, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
        __sync_synchronize();
        s->p_read[from][me] = s->p_read[from][me] + 1;
    }
    return ncclSuccess;
}

ncclResult_t submit(const Op &o)
{
    if (g_depth == 0) return run(o);
    if (g_nq >= 32) return ncclInternalError;
    g_q[g_nq++] = o;
    return ncclSuccess;
}
}   // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version)
{
    if (version) *version = 29999;   // "2.99.99": not a release anybody ships
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : r == ncclSystemError ? "rccl_double: a rank did not arrive within 120 s"
                                                                  : "rccl_double: error";
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    unsigned char rnd[8] = {0};
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) {
        (void)!read(fd, rnd, sizeof rnd);
        close(fd);
    }
    std::snprintf(id->internal, sizeof id->internal, "%02x%02x%02x%02x%02x%02x%02x%02x_%d", rnd[0], rnd[1], rnd[2], rnd[3],
                  rnd[4], rnd[5], rnd[6], rnd[7], (int)getpid());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank;
    c->nranks = nranks;
    c->seq = 0;
    id.internal[40] = 0;
    std::snprintf(c->path, sizeof c->path, "/dev/shm/rfgpu_rccl_double_%s", id.internal);
    const int fd = open(c->path, O_RDWR | O_CREAT, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) {
        delete c;
        return ncclSystemError;
    }
    void *m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->shm = static_cast<Shm *>(m);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    munmap(c->shm, sizeof(Shm));
    if (c->rank == 0) unlink(c->path);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    // sends first (they only fill a mailbox), then the rest in order
    for (int pass = 0; pass < 2 && rc == ncclSuccess; ++pass)
        for (int i = 0; i < g_nq && rc == ncclSuccess; ++i)
            if ((g_q[i].kind == 2) == (pass == 0)) rc = run(g_q[i]);
    g_nq = 0;
    return rc;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    return submit(Op{0, sendbuff, recvbuff, sendcount, datatype, -1, reinterpret_cast<Comm *>(comm), stream});
}

ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (root < 0 || root >= c->nranks) return ncclInvalidArgument;
    return submit(Op{1, sendbuff, recvbuff, count, datatype, root, c, stream});
}

ncclResult_t ncclReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root,
                        ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (root < 0 || root >= c->nranks || op != ncclSum) return ncclInvalidArgument;
    return submit(Op{4, sendbuff, recvbuff, count, datatype, root, c, stream});
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream)
{
    return submit(Op{2, sendbuff, nullptr, count, datatype, peer, reinterpret_cast<Comm *>(comm), stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return submit(Op{3, nullptr, recvbuff, count, datatype, peer, reinterpret_cast<Comm *>(comm), stream});
}

}   // extern "C"
