// mx_probe.hip -- discovers the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4
// empirically (no ISA document in this image): one-hot fp8 operands, print where they land.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// a, b: per lane 32 bytes (fp8 e4m3).  scale: per lane 1 byte (E8M0) in byte 0 of the scale register.
__global__ void probe(const uint8_t* A, const uint8_t* Bm, const uint8_t* sa, const uint8_t* sb, float* D, int fmtA, int fmtB) {
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = ((const int*)A)[lane * 8 + i];
        b[i] = ((const int*)Bm)[lane * 8 + i];
    }
    v4f c = {0, 0, 0, 0};
    const int scA = sa[lane], scB = sb[lane];
    if (fmtA == 0 && fmtB == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scA, 0, scB);
    for (int i = 0; i < 4; ++i) D[lane * 4 + i] = c[i];
}

int main() {
    const uint8_t one = 0x38; // e4m3 1.0: sign 0, exp 0111, mant 000
    std::vector<uint8_t> A(64 * 32), B(64 * 32), SA(64, 127), SB(64, 127); // E8M0 127 = 2^0
    uint8_t *dA, *dB, *dSA, *dSB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, 64); hipMalloc(&dSB, 64); hipMalloc(&dD, 64 * 4 * 4);
    std::vector<float> D(256);
    auto run = [&]() {
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        hipMemcpy(dSA, SA.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), 64, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dSA, dSB, dD, 0, 0);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    };
    // 1. B all ones: D[i][j] = sum_k A[i][k]; one-hot A(lane la, byte ba) -> which output row lights up (all 16 cols)
    std::fill(B.begin(), B.end(), one);
    printf("A operand: (lane, byte) -> row i   [D col = lane&15, row = 4*(lane>>4)+reg]\n");
    for (int la : {0, 1, 15, 16, 17, 32, 48, 63}) for (int ba : {0, 1, 15, 16, 31}) {
        std::fill(A.begin(), A.end(), 0); A[la * 32 + ba] = one; run();
        int rows = 0, row = -1; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l * 4 + r] != 0) { int rr = 4 * (l >> 4) + r; if (rr != row) { row = rr; ++rows; } }
        printf("  A lane %2d byte %2d -> row %d (distinct rows %d) val %g\n", la, ba, row, rows, D[0 * 4 + 0] + D[16 * 4] + D[32 * 4] + D[48 * 4]);
    }
    // 2. k mapping: A one-hot at (lane la, byte ba), B one-hot at (lane lb, byte bb): D nonzero iff same k
    printf("k index equality: A(lane,byte) vs B(lane,byte) nonzero?\n");
    auto test = [&](int la, int ba, int lb, int bb) {
        std::fill(A.begin(), A.end(), 0); std::fill(B.begin(), B.end(), 0); A[la * 32 + ba] = one; B[lb * 32 + bb] = one; run();
        float s = 0; for (float v : D) s += v; return s; };
    for (int g = 0; g < 4; ++g) for (int ba : {0, 5, 16, 31})
        printf("  A(lane %2d, byte %2d) x B(lane %2d, byte %2d) = %g ; x B(lane %2d, byte %2d) = %g ; x B(lane %2d, byte %2d) = %g\n",
               16 * g, ba, 16 * g, ba, test(16 * g, ba, 16 * g, ba), 16 * ((g + 1) & 3), ba, test(16 * g, ba, 16 * ((g + 1) & 3), ba), 16 * g, (ba + 1) & 31, test(16 * g, ba, 16 * g, (ba + 1) & 31));
    // 3. scale: set scale of lane 0 (A) to 128 (2^1): which outputs double?
    std::fill(A.begin(), A.end(), one); std::fill(B.begin(), B.end(), one); run(); printf("all ones: D[0]=%g (expect 128)\n", D[0]);
    SA[0] = 128; run(); printf("scaleA lane0=2: D[lane0 regs]=%g %g %g %g, D[lane1]=%g, D[lane16]=%g\n", D[0], D[1], D[2], D[3], D[4], D[64]);
    SA[0] = 127; SA[16] = 128; run(); printf("scaleA lane16=2: D[lane0]=%g %g, D[lane16]=%g\n", D[0], D[1], D[64]);
    SA[16] = 127; SB[3] = 129; run(); printf("scaleB lane3=4: D[lane3]=%g D[lane0]=%g D[lane19]=%g\n", D[12], D[0], D[19 * 4]);
    return 0;
}
