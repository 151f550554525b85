// FP64 vector-FMA peak of the device (datasheet MI355X: 78.6 TFLOP/s).  Not part of the
// product: a calibration point for the roofline in DESIGN.md / bench.py.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_peak tools/fp64_peak.hip && ./fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_kernel(double *out, double a, double b, int iters)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) x[c] = fma(x[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
static void run(int blocks_per_cu)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * blocks_per_cu, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, 0.999999, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * CHAINS * (double)iters * blocks * 256;
        if (rep == 2)
            printf("chains/thread %2d, %d waves/SIMD: %.3f ms  %.1f TFLOP/s fp64\n", CHAINS, blocks_per_cu, ms,
                   flops / ms / 1e9);
    }
    hipFree(out);
}

// The chains above converge to a fixed point: their operands stop toggling, which flatters the power the FMA pipe
// draws.  `chaos_kernel` iterates x <- x * x + c (c = -1.9: bounded, chaotic, every mantissa bit keeps changing) on 8
// chains per lane seeded differently: the same instruction rate with operands that behave like data.
__global__ __launch_bounds__(256) void chaos_kernel(double *out, double c, int iters)
{
    double x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = -1.0 + 1e-3 * (threadIdx.x + 1) + 0.11 * k + 1e-7 * blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = fma(x[k], x[k], c);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// `fp64_peak sustain SECONDS [chaos]`: the 8-chain, 4-waves/SIMD kernel back to back, rate per second of wall time -- does the
// rate hold once the part has been at full FP64 load for a while (tools/clock_under_load.sh samples clock + power beside it)?
static void sustain(double seconds, bool chaos)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flops = 2.0 * 8 * (double)iters * blocks * 256;
    double total_ms = 0;
    for (int window = 0; total_ms < seconds * 1e3; ++window) {
        int n = 0;
        float ms = 0;
        hipEventRecord(e0);
        for (; n < 200; ++n) {
            if (chaos)
                hipLaunchKernelGGL(chaos_kernel, dim3(blocks), dim3(256), 0, 0, out, -1.9, iters);
            else
                hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, 0.999999, 1e-9, iters);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        printf("t=%6.2f s  %.1f TFLOP/s fp64\n", total_ms / 1e3, flops * n / ms / 1e9);
        fflush(stdout);
    }
    hipFree(out);
}

int main(int argc, char **argv)
{
    if (argc > 2 && argv[1][0] == 's') { sustain(atof(argv[2]), argc > 3); return 0; }
    run<1>(1); run<2>(1); run<4>(1); run<8>(1);
    run<1>(2); run<4>(2); run<8>(2);
    run<4>(4); run<8>(4); run<8>(8);
    return 0;
}
