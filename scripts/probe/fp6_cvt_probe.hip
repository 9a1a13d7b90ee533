// fp6_cvt_probe.hip -- semantics of v_cvt_scalef32_2xpk16_fp6_f32 on gfx950 (the f16m6 epilogue's
// f32 -> e2m3 conversion): where the 2 x 16 inputs land among the 32 six-bit outputs, rounding,
// saturation, what the scale operand does, NaN / infinity.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
__global__ void k(const float* in, unsigned* out, const float* scale) {
    v16f a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    // NOTE: through the builtin, hipcc (ROCm 7.2) may allocate the 6-dword result INSIDE the first source's
    // 16 registers (v[0:5] <- v[2:17], ...): the instruction then overwrites inputs it has not read yet and
    // every value from slot 12 on is garbage (first run of this probe).  Early-clobber inline asm avoids it.
    v6u r;
    const float sc = scale[threadIdx.x];
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(sc));
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}
static float dec(unsigned c) { // e2m3
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.f : (1.f + m / 8.f) * (float)(1 << (e - 1));
    return s ? -v : v;
}
static unsigned code(const unsigned* w, int j) {
    const int b = 6 * j; unsigned long long x = w[b / 32] | ((unsigned long long)(b / 32 + 1 < 6 ? w[b / 32 + 1] : 0) << 32);
    return (unsigned)(x >> (b % 32)) & 63;
}
int main() {
    const int L = 8;
    std::vector<float> in(L * 32), sc(L, 1.f);
    // lane 0: src0[i] = (i+1)/8 (codes 1..16), src1[i] = 2 + i/4 ... distinct values -> ordering
    for (int i = 0; i < 16; ++i) { in[i] = (i + 1) / 8.f; in[16 + i] = -(i + 1) / 8.f; }
    // lane 1: ties and saturation
    const float t[32] = {1.0625f, 1.1875f, 0.0625f, 0.1875f, 2.125f, 2.375f, 4.25f, 4.75f, 7.5f, 7.74f, 7.76f, 8.f, 9.f, 100.f, 1e30f, INFINITY,
                         -1.0625f, -7.76f, -1e30f, -INFINITY, NAN, 0.f, -0.f, 0.06f, 0.07f, 3.9f, 3.95f, 1e-30f, 5.25f, 5.75f, 6.25f, 6.75f};
    for (int i = 0; i < 32; ++i) in[32 + i] = t[i];
    // lanes 2..5: scale semantics on the values 1, 2, 3, 6 in slots 0..3
    for (int l = 2; l < 6; ++l) { for (int i = 0; i < 32; ++i) in[l * 32 + i] = 0; in[l * 32] = 1; in[l * 32 + 1] = 2; in[l * 32 + 2] = 3; in[l * 32 + 3] = 6; }
    for (int i = 0; i < 16; ++i) { in[6 * 32 + i] = 1.f; in[6 * 32 + 16 + i] = 2.f; }
    for (int i = 0; i < 32; ++i) in[7 * 32 + i] = 0.f;
    in[7 * 32 + 6] = 1.f;
    sc[2] = 2.f; sc[3] = 0.5f; sc[4] = 3.f; sc[5] = 0.25f;
    float *din, *dsc; unsigned* dout;
    hipMalloc(&din, in.size() * 4); hipMalloc(&dsc, L * 4); hipMalloc(&dout, L * 24);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsc, sc.data(), L * 4, hipMemcpyHostToDevice);
    k<<<1, L>>>(din, dout, dsc);
    std::vector<unsigned> out(L * 6); hipMemcpy(out.data(), dout, L * 24, hipMemcpyDeviceToHost);
    printf("raw words lane 0:"); for (int i = 0; i < 6; ++i) printf(" %08x", out[i]); printf("\n");
    printf("raw words lane 6 (src0 all 1.0, src1 all 2.0):"); for (int i = 0; i < 6; ++i) printf(" %08x", out[36 + i]); printf("\n");
    printf("raw words lane 7 (src0[i] = 1.0 only at i = 6, rest 0):"); for (int i = 0; i < 6; ++i) printf(" %08x", out[42 + i]); printf("\n");
    printf("ordering (lane 0: src0[i] = (i+1)/8, src1[i] = -(i+1)/8), output slot j -> value:\n ");
    for (int j = 0; j < 32; ++j) printf(" %d:%g", j, dec(code(&out[0], j)));
    printf("\nrounding / saturation (lane 1), input -> output by input index (assuming slot = index):\n");
    for (int j = 0; j < 32; ++j) printf("  %g -> %g (code %02x)\n", t[j], dec(code(&out[6], j)), code(&out[6], j));
    for (int l = 2; l < 6; ++l) printf("scale %g: 1,2,3,6 -> %g %g %g %g\n", sc[l], dec(code(&out[l * 6], 0)), dec(code(&out[l * 6], 1)), dec(code(&out[l * 6], 2)), dec(code(&out[l * 6], 3)));
    return 0;
}
