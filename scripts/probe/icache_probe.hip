// Instruction-fetch probe (gfx950): how fast does a wave run straight-line code it has never executed
// (cold in the instruction cache) against the same code run a second time?
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/icache_probe.hip -o scripts/probe/icache_probe
// Layout of the kernel: [flush: 96 KB of v_mov executed once] [t0] [test: N VALU instructions] [t1]
// then the test region again (now resident) [t2].  Reported: cycles per instruction cold / hot, for
// one workgroup of 4 waves (one per SIMD) alone on the chip and for one per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define TEST_BODY(N)                                  \
    asm volatile(".rept " #N "\n"                     \
                 "v_add_f32 %0, %0, %4\n"             \
                 "v_add_f32 %1, %1, %4\n"             \
                 "v_add_f32 %2, %2, %4\n"             \
                 "v_add_f32 %3, %3, %4\n"             \
                 ".endr\n"                            \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(one))

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int rounds) {
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f, one = 1.f;
    unsigned long long t[4] = {0, 0, 0, 0};
    // flush: 24576 x 4-byte instructions = 96 KB, executed once
    asm volatile(".rept 24576\n v_mov_b32 %0, %0\n .endr\n" : "+v"(d));
    for (int r = 0; r < rounds; ++r) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (MODE == 0) TEST_BODY(750);  // 3000 x 4-byte VOP2 = 12 KB
        else {                           // 3000 x 8-byte VOP3 = 24 KB
            asm volatile(".rept 750\n"
                         "v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n"
                         ".endr\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(one));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (r < 4) t[r] = t1 - t0;
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        for (int i = 0; i < 4; ++i) o[i] = t[i];
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

template <int MODE>
static void run(int blocks, const char* what) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, (size_t)blocks * 16 * 8); hipMalloc(&sink, (size_t)blocks * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, sink, 3);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)blocks * 16);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s[3] = {0, 0, 0};
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) for (int i = 0; i < 3; ++i) s[i] += h[((size_t)b * 4 + w) * 4 + i];
    const double n = blocks * 4.0 * 3000.0;
    printf("%-52s cold %.2f  second pass %.2f  third %.2f cycles per instruction\n", what, s[0] / n, s[1] / n, s[2] / n);
    hipFree(out); hipFree(sink);
}

int main() {
    run<0>(1, "4-byte VALU x3000 (12 KB), 1 workgroup");
    run<0>(256, "4-byte VALU x3000 (12 KB), 256 workgroups");
    run<1>(1, "8-byte VALU x3000 (24 KB), 1 workgroup");
    run<1>(256, "8-byte VALU x3000 (24 KB), 256 workgroups");
    return 0;
}
