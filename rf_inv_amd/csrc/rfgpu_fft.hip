// rfgpu_fft.hip -- the two transforms of the reference's `module fftw` outside the hot path (SURVEY.md 8 rows a12,
// f-4): FFTW's plans `ifft` (dfftw_plan_dft_c2r_1d, src/fftw.f90:44) and `ifft2` (dfftw_plan_dft_r2c_1d, :45) executed
// on the module's own buffers cx / rx, as src/make_syn.f90:91-95,107-111 does to filter its noise series
// (r2c -> times flt -> c2r, once per trace).  Inside the hot path the c2r lives in the trace kernels
// (rfgpu_kernels.hip); these entry points serve the drop-in `module fftw` (rf_inv_amd/fortran/fftw.f90) so that the
// reference's rf_inv.f90 and make_syn.f90 link and run unmodified without FFTW3.
//
// Both are the DEFINITION of the transform, one thread per output, twiddles from an exact table (long double on the
// host, index (j k) mod n kept by integer addition -- no angle is ever formed in floating point), Neumaier-compensated
// sums: an init-time utility, O(n^2 / 2) multiply-adds (n = 65536: 2e9, ~20 ms), as accurate as fp64 allows.
//   c2r (FFTW_BACKWARD, unnormalised):  rx(j) = Re X(0) + 2 sum_{k=1}^{ceil(n/2)-1} Re( X(k) e^{+2 pi i j k / n} )
//                                               [+ (-1)^j Re X(n/2), n even];   Im X(0), Im X(n/2) ignored like FFTW
//   r2c (FFTW_FORWARD):                 X(k) = sum_j rx(j) e^{-2 pi i j k / n},  k = 0 .. n/2
#include "rfgpu_internal.h"
#include "../../include/rfgpu_ext.h"

#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace rfgpu {
int comm_fail(const std::string &msg);   // sets rf_last_error (rfgpu_api.cpp)

// s += v, exactly-rounded running compensation in c (Neumaier)
__device__ __forceinline__ void comp_add(double &s, double &c, double v)
{
    const double t = s + v;
    c += fabs(s) >= fabs(v) ? (s - t) + v : (v - t) + s;
    s = t;
}

__global__ __launch_bounds__(256) void dft_c2r_kernel(int n, const double2 *__restrict__ X, const double2 *__restrict__ tw,
                                                      double *__restrict__ x)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int half = (n + 1) / 2;          // k = 1 .. half - 1 carry a conjugate partner
    double s = 0.0, c = 0.0;
    int idx = 0;
    for (int k = 1; k < half; ++k) {
        idx += j;
        if (idx >= n) idx -= n;
        const double2 w = tw[idx], v = X[k];
        comp_add(s, c, __dmul_rn(v.x, w.x));
        comp_add(s, c, -__dmul_rn(v.y, w.y));
    }
    double r = 2.0 * (s + c) + X[0].x;
    if ((n & 1) == 0) r += (j & 1) ? -X[n / 2].x : X[n / 2].x;
    x[j] = r;
}

__global__ __launch_bounds__(256) void dft_r2c_kernel(int n, const double *__restrict__ x, const double2 *__restrict__ tw,
                                                      double2 *__restrict__ X)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n / 2) return;
    double sr = 0.0, cr = 0.0, si = 0.0, ci = 0.0;
    int idx = 0;
    for (int j = 0; j < n; ++j) {
        const double2 w = tw[idx];
        const double v = x[j];
        comp_add(sr, cr, __dmul_rn(v, w.x));
        comp_add(si, ci, -__dmul_rn(v, w.y));
        idx += k;
        if (idx >= n) idx -= n;
    }
    X[k] = make_double2(sr + cr, si + ci);
}

namespace {
struct Plan {
    double2 *tw = nullptr;   // [n] exp(+2 pi i k / n)
    double2 *cx = nullptr;   // [n / 2 + 1]
    double *rx = nullptr;    // [n]
};
std::mutex g_mu;
std::map<std::pair<int, int>, Plan> g_plans;   // (device, n): tables and device buffers, kept for the life of the process

int plan_for(int n, Plan **out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess)
        return comm_fail("rf_fft: no HIP device available (librfgpu has no CPU fallback)");
    auto it = g_plans.find({dev, n});
    if (it == g_plans.end()) {
        std::vector<double2> tw((size_t)n);
        const long double step = 2.0L * 3.14159265358979323846264338327950288L / (long double)n;
        for (int k = 0; k < n; ++k) {
            // octant symmetry is not needed at table-building cost: cosl / sinl of k step, k step < 2 pi, are good to 1 ulp
            // of long double, far below double's
            tw[(size_t)k] = make_double2((double)cosl(step * k), (double)sinl(step * k));
        }
        Plan p;
        if (hipMalloc((void **)&p.tw, sizeof(double2) * n) != hipSuccess ||
            hipMalloc((void **)&p.cx, sizeof(double2) * (n / 2 + 1)) != hipSuccess ||
            hipMalloc((void **)&p.rx, sizeof(double) * n) != hipSuccess ||
            hipMemcpy(p.tw, tw.data(), sizeof(double2) * n, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (p.tw) (void)hipFree(p.tw);
            if (p.cx) (void)hipFree(p.cx);
            if (p.rx) (void)hipFree(p.rx);
            return comm_fail("rf_fft: device allocation failed");
        }
        it = g_plans.emplace(std::make_pair(dev, n), p).first;
    }
    *out = &it->second;
    return 0;
}
}   // namespace
}   // namespace rfgpu

using namespace rfgpu;

#define FFT_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return comm_fail(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int rf_fft_c2r(int32_t nfft, const double *cx, double *rx)
{
    if (!cx || !rx) return comm_fail("rf_fft_c2r: null argument");
    if (nfft < 2 || nfft > (1 << 20)) return comm_fail("rf_fft_c2r: nfft must be 2 .. 1048576");
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = nullptr;
    if (plan_for(nfft, &p)) return 1;
    FFT_TRY(hipMemcpy(p->cx, cx, sizeof(double2) * (nfft / 2 + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dft_c2r_kernel, dim3((nfft + 255) / 256), dim3(256), 0, 0, nfft, p->cx, p->tw, p->rx);
    FFT_TRY(hipGetLastError());
    FFT_TRY(hipMemcpy(rx, p->rx, sizeof(double) * nfft, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rf_fft_r2c(int32_t nfft, const double *rx, double *cx)
{
    if (!cx || !rx) return comm_fail("rf_fft_r2c: null argument");
    if (nfft < 2 || nfft > (1 << 20)) return comm_fail("rf_fft_r2c: nfft must be 2 .. 1048576");
    std::lock_guard<std::mutex> lock(g_mu);
    Plan *p = nullptr;
    if (plan_for(nfft, &p)) return 1;
    FFT_TRY(hipMemcpy(p->rx, rx, sizeof(double) * nfft, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dft_r2c_kernel, dim3((nfft / 2 + 1 + 255) / 256), dim3(256), 0, 0, nfft, p->rx, p->tw, p->cx);
    FFT_TRY(hipGetLastError());
    FFT_TRY(hipMemcpy(cx, p->cx, sizeof(double2) * (nfft / 2 + 1), hipMemcpyDeviceToHost));
    return 0;
}
