// mx_rate.hip -- issue-rate probe: how fast does a wave alternate v_mfma_f32_16x16x32_f16 and
// v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 operands), and what clock does the chip hold while
// it does?  Register operands only (no memory traffic): an upper bound for a kernel that
// evaluates the f16x3 correction terms on the MX instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NF16, int NMX>
__global__ __launch_bounds__(256, 1) void rate(float* out, unsigned long long* clk, int iters) {
    v4f acc[44];
    for (int i = 0; i < 44; ++i) acc[i] = v4f{0, 0, 0, 0};
    f16x8 ha, hb;
    v8i a8, b8;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.001f + i); hb[i] = (_Float16)(0.5f + i); a8[i] = 0x38383838 + threadIdx.x; b8[i] = 0x38303830 + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NF16; ++s)
#pragma unroll
            for (int i = 0; i < 44; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < NMX; ++s)
#pragma unroll
            for (int i = 0; i < 44; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 127, 0, 120);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 44; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NF16, int NMX>
void run(const char* name, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 16);
    rate<NF16, NMX><<<256, 256>>>(out, clk, 2); // warm
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); rate<NF16, NMX><<<256, 256>>>(out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / iters, mhz = (double)h[0] / (double)h[1] * 100.0;
    printf("%-28s %d f16 + %d mx slabs of 44: %8.0f cycles/iter (%5.1f per f16-equivalent mfma slot), %6.1f us/iter, clock %4.0f MHz\n",
           name, NF16, NMX, cyc, cyc / (44.0 * (NF16 + 2 * NMX)), ms * 1000 / iters, mhz);
    hipFree(out); hipFree(clk);
}

int main() {
    run<27, 0>("f16x3 today", 400);
    run<9, 5>("f16 main + mx corrections", 400);
    run<0, 14>("mx only", 400);
    run<9, 0>("f16 only (1 term)", 400);
    return 0;
}
