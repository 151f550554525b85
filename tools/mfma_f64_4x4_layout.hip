// Operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found empirically: A one-hot at lane la, B one-hot at lane lb
// -> which lane of D (if any) becomes 1.  One wave per (la, lb) pair.  Prints the inferred maps.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_f64_4x4_layout tools/mfma_f64_4x4_layout.hip && tools/mfma_f64_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int *out)
{
    const int pair = blockIdx.x, la = pair >> 6, lb = pair & 63, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) out[pair] = lane;
}

int main()
{
    int *d_out;
    std::vector<int> h(4096, -1);
    hipMalloc(&d_out, sizeof(int) * 4096);
    hipMemcpy(d_out, h.data(), sizeof(int) * 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d_out);
    hipMemcpy(h.data(), d_out, sizeof(int) * 4096, hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it meets and where the product lands
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d meets B lanes:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) printf(" %d->D%d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
