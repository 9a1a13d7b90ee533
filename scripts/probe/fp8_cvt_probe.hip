// fp8_cvt_probe.hip -- how do gfx950's f32 -> e4m3 conversions treat values beyond the largest
// normal (448)?  Decides whether the kF16m8 encoder needs its explicit +-448 clamps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const float* x, int n, uint32_t* plain, uint32_t* scaled1, uint32_t* scaled12) {
    const int i = threadIdx.x;
    if (i >= n) return;
    plain[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], -x[i], 0, false) & 0xffffu;
#if __has_builtin(__builtin_amdgcn_cvt_scalef32_pk_fp8_f32)
    typedef short v2s __attribute__((ext_vector_type(2)));
    v2s o = {0, 0};
    v2s a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, x[i], -x[i], 1.0f, false);
    v2s b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, x[i], -x[i], 1.0f / 4096.0f, false);
    scaled1[i] = (uint16_t)a[0];
    scaled12[i] = (uint16_t)b[0];
#else
    scaled1[i] = scaled12[i] = 0xdeadu;
#endif
}

int main() {
    std::vector<float> x = {0.f, 1.f, 0.5f, 300.f, 447.f, 448.f, 449.f, 460.f, 463.9f, 464.f, 470.f, 480.f, 500.f, 1000.f, 65000.f, 1e9f,
                            1.f / 4096.f, 0.1f / 4096.f, 0.109375f, 0.06f, 3.0f / 4096.f, 448.f / 4096.f, 500.f / 4096.f, 16.f};
    const int n = (int)x.size();
    float* dx; uint32_t *d0, *d1, *d2;
    hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, n, d0, d1, d2);
    std::vector<uint32_t> p(n), s1(n), s12(n);
    hipMemcpy(p.data(), d0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s12.data(), d2, n * 4, hipMemcpyDeviceToHost);
    printf("bytes: lo = e4m3(x), hi = e4m3(-x).  0x7e = +448 (largest normal), 0x7f = NaN\n");
    for (int i = 0; i < n; ++i)
        printf("x=%-14.8g cvt_pk_fp8 %04x | cvt_scalef32(scale 1) %04x | cvt_scalef32(scale 2^-12) %04x\n", x[i], p[i], s1[i], s12[i]);
    return 0;
}
