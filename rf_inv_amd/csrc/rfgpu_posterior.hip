// rfgpu_posterior.hip -- posterior accumulation on the device.
//
// Restates the "record sampled model" block of the reference's subroutine mcmc
// (src/pt_mcmc.f90:204-286) for a batch of chains whose current traces live in HBM.
// All of it is integer bookkeeping plus three fp64 running sums; parity is bit-exact:
//   * histogram indices int(x / dbin) + 1 are formed with the reference's operations and
//     no FMA contraction;
//   * the running sums vp_mean / vs_mean / vpvs_mean (and the ocean-layer ASSIGNMENTS of
//     :263-264, which overwrite the running value) are order dependent, so one thread owns
//     one depth bin and walks the batch in chain order -- the reference's order;
//   * integer histograms commute, so they use atomics (namp) or the owning thread (V-z).
// Four small kernels per record call; the amplitude histogram is the only one that touches
// O(n * ntrc * nsmp) data, and it reads the traces where the evaluation left them.
#include "rfgpu_internal.h"
#include <limits.h>

namespace rfgpu {

// Fortran int(x) as compiled for x86-64 (cvttsd2si): truncation toward zero, and the
// "integer indefinite" INT_MIN for NaN or values outside the int32 range.
__device__ __forceinline__ int f_int(double x)
{
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT_MIN;
    return (int)x;
}

__device__ __forceinline__ int clamp_bin(int ibin, int nbin) { return ibin < 1 ? 1 : (ibin > nbin ? nbin : ibin); }

// (1) which batch items are recorded: temp <= 1 + 1e-6 (:204), kept in chain order.
// One block; a block-wide inclusive scan over chunks of blockDim items.
__global__ __launch_bounds__(1024) void post_select_kernel(int n, const double *temps, PostState st)
{
    __shared__ int part[1024];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int start = 0; start < n; start += blockDim.x) {
        const int i = start + threadIdx.x;
        // 1.d0 + 1.0e-6: the default-real literal widened to double (:204)
        const int keep = (i < n) && (!temps || temps[i] <= 1.0 + (double)1.0e-6f);
        part[threadIdx.x] = keep;
        __syncthreads();
        for (int off = 1; off < (int)blockDim.x; off <<= 1) {
            const int v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        if (keep) st.sel[base + part[threadIdx.x] - 1] = i;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) base += part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *st.nsel = base;
        st.nmod[1] = st.nmod[0];     // base index of this batch (nmod before it)
    }
}

// (2) one WAVE per recorded model: counters (:208-231) and the model's depth-profile row (the layer walk of :243-270,
// values only; the accumulation happens in post_accum_kernel).  Lanes own depth bins: each walks the layers with the
// reference's running depth (the same sequence of additions) and keeps the LAST layer that covers its bin -- what the
// reference's in-order overwrites leave.  (Until round 4 one thread per model wrote its row bin by bin: 180 us a call.)
constexpr int PR_WAVES = 4;

__global__ __launch_bounds__(64 * PR_WAVES) void post_rows_kernel(PostConfig c, PostState st, PostBatch b)
{
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * PR_WAVES + (threadIdx.x >> 6);
    if (j >= *st.nsel) return;
    const int i = st.sel[j];
    const long long imod = (long long)st.nmod[1] + j;           // 0-based nmod of this model
    const int k = b.k[i];
    if (lane == 0) {
        if (imod < c.max_models) st.all_likelihood[imod] = b.logl[i];   // :210
        atomicAdd(&st.nk[clamp_bin(k, c.k_max) - 1], 1);                  // :213
    }
    for (int t = lane; t < c.ntrc; t += 64)                           // :216-223
        if (c.sig_mode[t] == 1) {
            const int ibin = f_int((b.sig[(size_t)i * c.ntrc + t] - c.sig_min[t]) / c.dbin_sig[t]) + 1;
            atomicAdd(&st.nsig[(size_t)t * c.nbin_sig + clamp_bin(ibin, c.nbin_sig) - 1], 1);
        }
    for (int il = lane; il < k - 1; il += 64) {                       // :226-229
        const int ibin = f_int((b.z[(size_t)i * (c.k_max - 1) + il] - c.z_min) / c.dbin_z) + 1;
        atomicAdd(&st.nz[clamp_bin(ibin, c.nbin_z) - 1], 1);
    }
    // :243-270 -- which layer covers which depth bin
    const int nl = b.nlay[i];
    const double *A = b.layers + (size_t)i * 4 * b.nlay_pad, *B = A + b.nlay_pad, *H = A + 3 * (size_t)b.nlay_pad;
    double *ra = st.row_a + (size_t)j * c.nbin_z, *rb = st.row_b + (size_t)j * c.nbin_z;
    for (int iz0 = 0; iz0 < c.nbin_z; iz0 += 64) {
        const int iz = iz0 + lane + 1;                               // 1-based depth bin of this lane
        double va = __builtin_nan(""), vb = 0.0;                     // not covered
        double tmpz = 0.0;
        for (int il = 0; il < nl; ++il) {
            const double h = H[il];
            const int iz1 = f_int(tmpz / c.dbin_z) + 1;
            const int iz2 = il < nl - 1 ? f_int((tmpz + h) / c.dbin_z) + 1 : c.nbin_z + 1;
            const int lo = iz1 < 1 ? 1 : iz1, hi = iz2 - 1 > c.nbin_z ? c.nbin_z : iz2 - 1;
            if (iz >= lo && iz <= hi) {
                va = A[il];
                vb = B[il];
            }
            tmpz = tmpz + h;
        }
        if (iz <= c.nbin_z) {
            ra[iz - 1] = va;
            if (va == va) rb[iz - 1] = vb;
        }
    }
}

// (3) one thread per depth bin walks the batch in chain order (:243-270 accumulation part).  The rows of PA_AHEAD
// models are requested together (the loop was a chain of one load's latency per model), the histogram cells take
// atomics without a return value (integer counts commute; nothing waits for them), the three running sums and the
// ocean-layer assignments stay strictly in chain order.
constexpr int PA_AHEAD = 8;

__global__ __launch_bounds__(64) void post_accum_kernel(PostConfig c, PostState st)
{
#pragma clang fp contract(off)
    const int iz = blockIdx.x * blockDim.x + threadIdx.x;   // 0-based
    if (iz >= c.nbin_z) return;
    const int nsel = *st.nsel;
    const long long base = st.nmod[1];
    double vp_mean = st.vp_mean[iz], vs_mean = st.vs_mean[iz], vpvs_mean = st.vpvs_mean[iz];
    for (int j0 = 0; j0 < nsel; j0 += PA_AHEAD) {
        double av[PA_AHEAD], bv[PA_AHEAD];
#pragma unroll
        for (int u = 0; u < PA_AHEAD; ++u) {
            const int j = j0 + u < nsel ? j0 + u : nsel - 1;
            av[u] = st.row_a[(size_t)j * c.nbin_z + iz];
            bv[u] = st.row_b[(size_t)j * c.nbin_z + iz];     // (never read where the bin is not covered)
        }
#pragma unroll
        for (int u = 0; u < PA_AHEAD; ++u) {
            const int j = j0 + u;
            if (j >= nsel) break;
            const double a = av[u];
            if (a != a) continue;
            const double bt = bv[u];
            const int ivp = clamp_bin(f_int((a - c.vp_min) / c.dbin_vp) + 1, c.nbin_vp);
            int ivs = f_int((bt - c.vs_min) / c.dbin_vs) + 1;
            ivs = clamp_bin(ivs, c.nbin_vs);                     // max(1, ivs) :253 (+ upper edge)
            int ivpvs = f_int(((a / bt) - c.vpvs_min) / c.dbin_vpvs) + 1;
            ivpvs = clamp_bin(ivpvs, c.nbin_vpvs);               // :255-256
            atomicAdd(&st.nvpz[(size_t)(ivp - 1) * c.nbin_z + iz], 1);
            atomicAdd(&st.nvsz[(size_t)(ivs - 1) * c.nbin_z + iz], 1);
            atomicAdd(&st.nvpvsz[(size_t)(ivpvs - 1) * c.nbin_z + iz], 1);
            vp_mean = vp_mean + a;
            double vs_row;
            if (bt > 0.0) {
                vpvs_mean = vpvs_mean + a / bt;
                vs_mean = vs_mean + bt;
                vs_row = bt;
            } else {                                             // ocean layer: assignments (:263-264)
                vpvs_mean = c.vpvs_min;
                vs_mean = c.vs_min;
                vs_row = c.vs_min;
            }
            const long long imod = base + j;
            if (imod < c.max_models) {
                st.vp_model[(size_t)imod * c.nbin_z + iz] = a;
                st.vs_model[(size_t)imod * c.nbin_z + iz] = vs_row;
            }
        }
    }
    st.vp_mean[iz] = vp_mean;
    st.vs_mean[iz] = vs_mean;
    st.vpvs_mean[iz] = vpvs_mean;
}

// (4) amplitude histogram of the recorded chains' current traces (:273-285): one block per
// (batch item, trace), threads over the nsmp samples; int32 atomics in L2.
__global__ __launch_bounds__(128) void post_amp_kernel(PostConfig c, PostState st, PostBatch b, WalkerState w)
{
#pragma clang fp contract(off)
    const int j = blockIdx.x / c.ntrc, itrc = blockIdx.x % c.ntrc;
    if (j >= *st.nsel) return;
    const int wk = b.walker_ids[st.sel[j]];
    const double *src = w.rft + (((size_t)w.cur_slot[wk] * w.nslots + wk) * c.ntrc + itrc) * (size_t)w.trace_len;
    int *hist = st.namp + (size_t)itrc * c.nsmp * c.nbin_amp;
    int oor = 0;
    for (int it = threadIdx.x; it < c.nsmp; it += blockDim.x) {
        int ibin = f_int((src[it] - c.amp_min) / c.dbin_amp) + 1;
        if (ibin < 1) {
            ibin = 1;
            ++oor;
        } else if (ibin > c.nbin_amp) {
            ibin = c.nbin_amp;
            ++oor;
        }
        atomicAdd(&hist[(size_t)it * c.nbin_amp + ibin - 1], 1);
    }
    if (oor) atomicAdd((unsigned long long *)st.amp_oor, (unsigned long long)oor);
}

// (5) nmod = nmod + (models recorded)
__global__ void post_finish_kernel(PostState st)
{
    st.nmod[0] += *st.nsel;
}

// vs_model(1, :) = -999.9 marks unused model slots (src/pt_mcmc.f90:419, read by src/mcmc_out.f90:115)
__global__ void post_mark_unused_kernel(double *vs_model, int nbin_z, long long max_models)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < max_models) vs_model[(size_t)i * nbin_z] = -999.9;
}

void launch_post_mark_unused(const PostConfig &c, const PostState &st, hipStream_t s)
{
    if (c.max_models <= 0) return;
    hipLaunchKernelGGL(post_mark_unused_kernel, dim3((unsigned)((c.max_models + 255) / 256)), dim3(256), 0, s,
                       st.vs_model, c.nbin_z, c.max_models);
}

void launch_post_record(const PostConfig &c, const PostState &st, const PostBatch &b, const WalkerState &w,
                        hipStream_t s)
{
    hipLaunchKernelGGL(post_select_kernel, dim3(1), dim3(1024), 0, s, b.n, b.temps, st);
    hipLaunchKernelGGL(post_rows_kernel, dim3((unsigned)((b.n + PR_WAVES - 1) / PR_WAVES)), dim3(64 * PR_WAVES), 0, s, c, st, b);
    hipLaunchKernelGGL(post_accum_kernel, dim3((unsigned)((c.nbin_z + 63) / 64)), dim3(64), 0, s, c, st);
    hipLaunchKernelGGL(post_amp_kernel, dim3((unsigned)(b.n * c.ntrc)), dim3(128), 0, s, c, st, b, w);
    hipLaunchKernelGGL(post_finish_kernel, dim3(1), dim3(1), 0, s, st);
}

} // namespace rfgpu
