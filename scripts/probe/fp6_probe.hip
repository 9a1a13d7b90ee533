// fp6_probe.hip -- v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 (fp6) operands on gfx950:
//   (1) issue rate against the fp8 form and the f16 MFMA (MI355X_MICROARCH.md "Matrix cores": fp6 = the
//       cycles of the bf16 form of the same MxN at 4x the K, i.e. HALF the cycles of the e4m3 form),
//       alone, with mixed formats, and in the f16m8 loop's slab pattern (2 f16 slabs + 1 MX slab);
//   (2) operand layout, found empirically with one-hot data (no ISA document in this image): where the
//       32 six-bit values of a lane sit in its 6 VGPRs, which k they are, how E8M0 block scales apply.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp6_probe fp6_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- (1) rate ------------------------------------------------------------------------------
template <int NF16, int NMX, int FA, int FB>
__global__ __launch_bounds__(256, 1) void rate(float* out, unsigned long long* clk, int iters) {
    v4f acc[44];
    for (int i = 0; i < 44; ++i) acc[i] = v4f{0, 0, 0, 0};
    f16x8 ha, hb;
    v8i a8, b8;
    for (int i = 0; i < 8; ++i) {
        ha[i] = (_Float16)(threadIdx.x * 0.001f + i);
        hb[i] = (_Float16)(0.5f + i);
        a8[i] = 0x28a28a28 + (int)threadIdx.x * 0x01041041; // random-ish bit patterns: every field varies
        b8[i] = 0x2492c924 + i * 0x00820821;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < NF16; ++s)
#pragma unroll
            for (int i = 0; i < 44; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < NMX; ++s)
#pragma unroll
            for (int i = 0; i < 44; ++i)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], FA, FB, 0, 120, 0, 120);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 44; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NF16, int NMX, int FA, int FB>
void run(const char* name, int iters) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 16);
    rate<NF16, NMX, FA, FB><<<256, 256>>>(out, clk, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); rate<NF16, NMX, FA, FB><<<256, 256>>>(out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double cyc = (double)h[0] / iters, mhz = (double)h[0] / (double)h[1] * 100.0;
    const int n = 44 * (NF16 + NMX);
    printf("%-44s %2d f16 + %2d mx slabs of 44: %8.0f cycles/iter = %5.1f cycles per instruction, %7.1f us/iter, clock %4.0f MHz\n",
           name, NF16, NMX, cyc, cyc / n, ms * 1000 / iters, mhz);
    hipFree(out); hipFree(clk);
}

// ---- (2) layout ----------------------------------------------------------------------------
// per lane: 8 dwords for A and B (fp6 uses the first 6), one E8M0 scale byte each.
template <int FA, int FB>
__global__ void probe(const int* A, const int* Bm, const uint8_t* sa, const uint8_t* sb, float* D) {
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = A[lane * 8 + i]; b[i] = Bm[lane * 8 + i]; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FA, FB, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int i = 0; i < 4; ++i) D[lane * 4 + i] = c[i];
}

struct Op { // one operand: 64 lanes x 8 dwords
    std::vector<int> w = std::vector<int>(64 * 8, 0);
    void clear() { std::fill(w.begin(), w.end(), 0); }
    void setBits(int lane, int bit, int nbits, unsigned v) { // little-endian bit string over the lane's dwords
        for (int i = 0; i < nbits; ++i) {
            const int b = bit + i;
            unsigned& d = (unsigned&)w[lane * 8 + b / 32];
            d = (d & ~(1u << (b % 32))) | (((v >> i) & 1u) << (b % 32));
        }
    }
    void fp6(int lane, int j, unsigned v) { setBits(lane, 6 * j, 6, v); }   // value j at bits [6j, 6j+6)
    void fp8(int lane, int j, unsigned v) { setBits(lane, 8 * j, 8, v); }
    void fill6(unsigned v) { for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) fp6(l, j, v); }
    void fill8(unsigned v) { for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) fp8(l, j, v); }
};

int *dA, *dB; uint8_t *dSA, *dSB; float* dD;
std::vector<uint8_t> SA(64, 127), SB(64, 127);
std::vector<float> D(256);

template <int FA, int FB>
void go(const Op& A, const Op& B) {
    hipMemcpy(dA, A.w.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(dB, B.w.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
    probe<FA, FB><<<1, 64>>>(dA, dB, dSA, dSB, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
}
float total() { float s = 0; for (float v : D) s += v; return s; }
// D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
float at(int row, int col) { return D[(16 * (row >> 2) + col) * 4 + (row & 3)]; }

int main() {
    printf("== rate (256 workgroups x 4 waves, register operands) ==\n");
    run<27, 0, 0, 0>("f16 only", 300);
    run<0, 14, 0, 0>("mx e4m3 x e4m3", 300);
    run<0, 14, 2, 2>("mx e2m3 x e2m3", 300);
    run<0, 14, 2, 0>("mx e2m3 (A) x e4m3 (B)", 300);
    run<0, 14, 0, 2>("mx e4m3 (A) x e2m3 (B)", 300);
    run<0, 14, 4, 4>("mx e2m1 x e2m1 (fp4)", 300);
    run<18, 9, 0, 0>("f16m8 pattern: 2 f16 + 1 mx(e4m3)", 300);
    run<18, 9, 2, 2>("f16m6 pattern: 2 f16 + 1 mx(e2m3)", 300);

    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 1024);
    const unsigned one6 = 0x08;  // e2m3 1.0: s 0, e 01, m 000 (bias 1)
    const unsigned one8 = 0x38;  // e4m3 1.0
    printf("\n== layout, e2m3 x e2m3 (value j of a lane at bits [6j, 6j+6) of its first 6 dwords) ==\n");
    Op A, B;
    B.fill6(one6);
    A.fill6(one6);
    go<2, 2>(A, B);
    printf("all ones: D[0][0] = %g (expect 128), sum %g (expect 32768)\n", at(0, 0), total());
    // value decoding: A = v everywhere in row 0's lanes... simpler: A all = code c, B all ones -> D = 128 * value(c)
    printf("e2m3 code -> value (D/128):");
    for (unsigned c = 0; c < 64; ++c) { A.fill6(c); go<2, 2>(A, B); printf(" %02x:%g", c, at(0, 0) / 128.f); }
    printf("\n");
    // row mapping of A(lane, j)
    printf("A one-hot (lane, j) -> rows lit (B all ones):\n");
    for (int la : {0, 1, 15, 16, 17, 32, 48, 63}) for (int j : {0, 1, 5, 15, 16, 31}) {
        A.clear(); A.fp6(la, j, one6); go<2, 2>(A, B);
        int row = -1, rows = 0; for (int r = 0; r < 16; ++r) if (at(r, 0) != 0) { row = r; ++rows; }
        printf("  A lane %2d value %2d -> row %2d (rows lit %d) D = %g\n", la, j, row, rows, row >= 0 ? at(row, 0) : 0.f);
    }
    // k mapping: A one-hot (la, ja) x B one-hot (lb, jb) nonzero iff same k
    printf("k equality, e2m3 x e2m3: A(lane 16g, j) x B(lane 16g', j') ->\n");
    auto kt = [&](int la, int ja, int lb, int jb) { A.clear(); B.clear(); A.fp6(la, ja, one6); B.fp6(lb, jb, one6); go<2, 2>(A, B); return total(); };
    for (int g = 0; g < 4; ++g) for (int j : {0, 7, 16, 31})
        printf("  g %d j %2d: same %g, next group %g, next j %g\n", g, j, kt(16 * g, j, 16 * g, j), kt(16 * g, j, 16 * ((g + 1) & 3), j), kt(16 * g, j, 16 * g, (j + 1) & 31));
    // mixed: A e4m3 byte jb vs B e2m3 value j
    printf("k equality, e4m3 (A) x e2m3 (B): A(lane 16g, byte j) x B(lane 16g, value j') ->\n");
    auto km = [&](int la, int ja, int lb, int jb) { A.clear(); B.clear(); A.fp8(la, ja, one8); B.fp6(lb, jb, one6); go<0, 2>(A, B); return total(); };
    for (int g = 0; g < 4; ++g) for (int j : {0, 7, 16, 31})
        printf("  g %d j %2d: same %g, next group %g, next j %g\n", g, j, km(16 * g, j, 16 * g, j), km(16 * g, j, 16 * ((g + 1) & 3), j), km(16 * g, j, 16 * g, (j + 1) & 31));
    // scales: E8M0 byte per lane = scale of that lane's 32 values (its row/col, its k block)
    A.fill6(one6); B.fill6(one6);
    SA[0] = 128; go<2, 2>(A, B); printf("scaleA lane 0 = 2^1: row0 col0 %g (expect 160 = 96 + 2*32), row 1 %g, row0 col5 %g\n", at(0, 0), at(1, 0), at(0, 5));
    SA[0] = 127; SA[16] = 125; go<2, 2>(A, B); printf("scaleA lane 16 = 2^-2: row0 col0 %g (expect 104 = 96 + 32/4), row 4 %g\n", at(0, 0), at(4, 0));
    SA[16] = 127; SB[3] = 129; go<2, 2>(A, B); printf("scaleB lane 3 = 2^2: row0 col3 %g (expect 224 = 96 + 4*32), row0 col0 %g\n", at(0, 3), at(0, 0));
    SB[3] = 127;
    // subnormals and the top of the range
    A.fill6(0x01); go<2, 2>(A, B); printf("e2m3 0x01 (smallest subnormal): value %g (expect 0.125)\n", at(0, 0) / 128.f);
    A.fill6(0x1f); go<2, 2>(A, B); printf("e2m3 0x1f (largest): value %g (expect 7.5)\n", at(0, 0) / 128.f);
    A.fill6(0x3f); go<2, 2>(A, B); printf("e2m3 0x3f: value %g (expect -7.5)\n", at(0, 0) / 128.f);
    return 0;
}
