// Do the FP64 matrix pipe (v_mfma_f64_4x4x4_4b_f64 / v_mfma_f64_16x16x4_f64) and the FP64 vector pipe (v_fma_f64) of a
// gfx950 SIMD run CONCURRENTLY -- is the combined rate above the 78.6 TFLOP/s each of them is specified at?
// Answers VERDICT r02 item 6c with numbers.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap tools/mfma_overlap.hip && ./mfma_overlap
//
// Four kernels, all register-resident dependency chains, 256-thread blocks:
//   vec     every wave issues v_fma_f64 only                      (the propagator loop's instruction class)
//   mat     every wave issues v_mfma_f64_4x4x4_4b_f64 only
//   split   waves 0, 1 of a block issue MFMA, waves 2, 3 issue FMA (different waves of one SIMD pair up)
//   mixed   every wave interleaves one MFMA with VPM vector FMAs (one instruction stream)
// Reported: TFLOP/s of each class and their sum.  flops: v_fma_f64 = 64 lanes x 2; v_mfma_f64_4x4x4_4b = 4 blocks x
// 4 x 4 x 4 x 2 = 512; v_mfma_f64_16x16x4 = 16 x 16 x 4 x 2 = 2048.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int VCH = 8;   // independent FMA chains per lane
constexpr int MCH = 4;   // independent MFMA accumulators per lane

// mode: 0 vec, 1 mat (4x4x4), 2 split by wave, 3 mixed in one stream (vpm FMAs per MFMA), 4 mat (16x16x4)
template <int MODE, int VPM>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters)
{
    const int wave = threadIdx.x >> 6;
    double x[VCH];
    double acc[MCH];
    double4_t acc16[2];
#pragma unroll
    for (int c = 0; c < VCH; ++c) x[c] = threadIdx.x * 1e-3 + c;
#pragma unroll
    for (int c = 0; c < MCH; ++c) acc[c] = 0.0;
    acc16[0] = acc16[1] = double4_t{0, 0, 0, 0};
    const bool do_vec = MODE == 0 || (MODE == 2 && wave >= 2) || MODE == 3;
    const bool do_mat = MODE == 1 || (MODE == 2 && wave < 2) || MODE == 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 4) {
#pragma unroll
            for (int c = 0; c < 2; ++c) acc16[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[c], 0, 0, 0);
            continue;
        }
        if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < MCH; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VPM; ++v) x[(c * VPM + v) % VCH] = fma(x[(c * VPM + v) % VCH], a, b);
            }
            continue;
        }
        if (do_mat) {
#pragma unroll
            for (int c = 0; c < MCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
        }
        if (do_vec) {
#pragma unroll
            for (int c = 0; c < VCH; ++c) x[c] = fma(x[c], a, b);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < VCH; ++c) s += x[c];
#pragma unroll
    for (int c = 0; c < MCH; ++c) s += acc[c];
    s += acc16[0][0] + acc16[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int VPM>
static void run(const char *name, int blocks_per_cu)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * blocks_per_cu, iters = 20000;
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, VPM>), dim3(blocks), dim3(256), 0, 0, out, 0.999999, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double waves = (double)blocks * 4;
    double vec_waves = 0, mat_waves = 0, vec_per_iter = VCH, mat_per_iter = MCH, mat_flops = 512;
    if (MODE == 0) vec_waves = waves;
    if (MODE == 1) mat_waves = waves;
    if (MODE == 2) vec_waves = mat_waves = waves / 2;
    if (MODE == 3) { vec_waves = mat_waves = waves; vec_per_iter = MCH * VPM; }
    if (MODE == 4) { mat_waves = waves; mat_per_iter = 2; mat_flops = 2048; }
    const double vf = vec_waves * vec_per_iter * iters * 128.0, mf = mat_waves * mat_per_iter * iters * mat_flops;
    printf("%-28s %d waves/SIMD: %8.3f ms   vector %6.1f TF   matrix %6.1f TF   sum %6.1f TF\n", name, blocks_per_cu, ms,
           vf / ms / 1e9, mf / ms / 1e9, (vf + mf) / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int w : {2, 4, 8}) {
        printf("--- %d waves per SIMD\n", w);
        if (w == 2) { run<0, 0>("vec only", 2); run<1, 0>("mfma 4x4x4 only", 2); run<4, 0>("mfma 16x16x4 only", 2); run<2, 0>("split by wave", 2);
                      run<3, 1>("mixed, 1 fma per mfma", 2); run<3, 2>("mixed, 2 fma per mfma", 2); run<3, 4>("mixed, 4 fma per mfma", 2); }
        if (w == 4) { run<0, 0>("vec only", 4); run<1, 0>("mfma 4x4x4 only", 4); run<4, 0>("mfma 16x16x4 only", 4); run<2, 0>("split by wave", 4);
                      run<3, 1>("mixed, 1 fma per mfma", 4); run<3, 2>("mixed, 2 fma per mfma", 4); run<3, 4>("mixed, 4 fma per mfma", 4); }
        if (w == 8) { run<0, 0>("vec only", 8); run<1, 0>("mfma 4x4x4 only", 8); run<4, 0>("mfma 16x16x4 only", 8); run<2, 0>("split by wave", 8);
                      run<3, 1>("mixed, 1 fma per mfma", 8); run<3, 2>("mixed, 2 fma per mfma", 8); run<3, 4>("mixed, 4 fma per mfma", 8); }
    }
    return 0;
}
