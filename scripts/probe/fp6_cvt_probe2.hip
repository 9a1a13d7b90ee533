// fp6_cvt_probe2.hip -- the other conversions of the f16m6 epilogue: v_cvt_scalef32_pk32_fp6_f16
// (32 packed f16 -> 32 e2m3), v_cvt_scalef32_pk32_f16_fp6 and _pk32_f32_fp6 (back), and the lane
// exchange ds_swizzle xor 16.  All through early-clobber inline asm (see fp6_cvt_probe.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const _Float16* in, unsigned* out6, _Float16* outh, float* outf, int* sw, float scale) {
    v32h a;
    for (int i = 0; i < 32; ++i) a[i] = in[threadIdx.x * 32 + i];
    v6u r;
    asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(scale));
    for (int i = 0; i < 6; ++i) out6[threadIdx.x * 6 + i] = r[i];
    v32h h;
    asm volatile("v_cvt_scalef32_pk32_f16_fp6 %0, %1, %2" : "=&v"(h) : "v"(r), "v"(scale));
    for (int i = 0; i < 32; ++i) outh[threadIdx.x * 32 + i] = h[i];
    v32f f;
    asm volatile("v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2" : "=&v"(f) : "v"(r), "v"(scale));
    for (int i = 0; i < 32; ++i) outf[threadIdx.x * 32 + i] = f[i];
    sw[threadIdx.x] = __builtin_amdgcn_ds_swizzle((int)threadIdx.x, 0x401F);
}
static float dec(unsigned c) {
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.f : (1.f + m / 8.f) * (float)(1 << (e - 1));
    return s ? -v : v;
}
static unsigned code(const unsigned* w, int j) {
    const int b = 6 * j; unsigned long long x = w[b / 32] | ((unsigned long long)(b / 32 + 1 < 6 ? w[b / 32 + 1] : 0) << 32);
    return (unsigned)(x >> (b % 32)) & 63;
}
int main() {
    std::vector<_Float16> in(64 * 32);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) in[l * 32 + i] = (_Float16)((i < 16 ? (i + 1) / 8.f : -(i - 15) / 8.f) * (l == 1 ? 2.f : 1.f));
    _Float16 *din, *dh; unsigned* d6; float* df; int* dsw;
    hipMalloc(&din, in.size() * 2); hipMalloc(&d6, 64 * 24); hipMalloc(&dh, 64 * 64); hipMalloc(&df, 64 * 128); hipMalloc(&dsw, 256);
    hipMemcpy(din, in.data(), in.size() * 2, hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, d6, dh, df, dsw, 1.0f);
    std::vector<unsigned> o6(64 * 6); std::vector<_Float16> oh(64 * 32); std::vector<float> of(64 * 32); std::vector<int> sw(64);
    hipMemcpy(o6.data(), d6, 64 * 24, hipMemcpyDeviceToHost); hipMemcpy(oh.data(), dh, 64 * 64, hipMemcpyDeviceToHost);
    hipMemcpy(of.data(), df, 64 * 128, hipMemcpyDeviceToHost); hipMemcpy(sw.data(), dsw, 256, hipMemcpyDeviceToHost);
    printf("pk32_fp6_f16 (lane 0: f16 element i = (i+1)/8 for i < 16, -(i-15)/8 after) slot j -> value:\n ");
    for (int j = 0; j < 32; ++j) printf(" %d:%g", j, dec(code(&o6[0], j)));
    printf("\npk32_f16_fp6 of that: element i ->\n ");
    for (int i = 0; i < 32; ++i) printf(" %d:%g", i, (float)oh[i]);
    printf("\npk32_f32_fp6 of that: element i ->\n ");
    for (int i = 0; i < 32; ++i) printf(" %d:%g", i, of[i]);
    k<<<1, 64>>>(din, d6, dh, df, dsw, 4.0f);
    hipMemcpy(o6.data(), d6, 64 * 24, hipMemcpyDeviceToHost); hipMemcpy(oh.data(), dh, 64 * 64, hipMemcpyDeviceToHost); hipMemcpy(of.data(), df, 64 * 128, hipMemcpyDeviceToHost);
    printf("\nscale 4: lane 1 (inputs x2): fp6 slots 0..3 -> %g %g %g %g ; back to f16 (x scale?) %g %g ; to f32 %g %g\n",
           dec(code(&o6[6], 0)), dec(code(&o6[6], 1)), dec(code(&o6[6], 2)), dec(code(&o6[6], 3)), (float)oh[32], (float)oh[33], of[32], of[33]);
    printf("ds_swizzle 0x401F: lane -> value:"); for (int l = 0; l < 64; l += 5) printf(" %d:%d", l, sw[l]); printf("\n");
    return 0;
}
