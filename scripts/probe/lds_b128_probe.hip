// ds_read_b128 bank-conflict probe (gfx950): cycles per wave-instruction for the row-fragment read
// of the conv tile kernels -- lane = 16 g + li reads 16 bytes at plane(g) + entry(li) * 16 -- with
//   (a) 16 consecutive entries, (b) one hole after li = 5 (a board-row wrap inside the fragment),
//   (c) two holes, (d) 16 entries with distinct residues mod 16 that span 40 entries.
//   hipcc --offload-arch=gfx950 -O3 -w scripts/probe/lds_b128_probe.hip -o scripts/probe/lds_b128_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(unsigned long long* out, unsigned* sink, const int* entryOfLi, int planeBytes, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<u32x4*>(smem)[i % 8192] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    const unsigned char* p = smem + g * planeBytes + entryOfLi[li] * 16;
    u32x4 acc = {0, 0, 0, 0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned addr = (unsigned)(p - smem);
    for (int it = 0; it < iters; ++it) {
        u32x4 v0, v1, v2, v3;
        // sixteen reads at tap-like uniform shifts (immediate offsets), four destinations, nothing consumed
        asm volatile(
            "ds_read_b128 %0, %4 offset:0\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:160\n"
            "ds_read_b128 %0, %4 offset:176\n ds_read_b128 %1, %4 offset:192\n ds_read_b128 %2, %4 offset:320\n ds_read_b128 %3, %4 offset:336\n"
            "ds_read_b128 %0, %4 offset:352\n ds_read_b128 %1, %4 offset:0\n ds_read_b128 %2, %4 offset:16\n ds_read_b128 %3, %4 offset:32\n"
            "ds_read_b128 %0, %4 offset:160\n ds_read_b128 %1, %4 offset:176\n ds_read_b128 %2, %4 offset:192\n ds_read_b128 %3, %4 offset:320\n"
            "s_waitcnt lgkmcnt(0)\n"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
        if (it == iters - 1) acc = v0 ^ v1 ^ v2 ^ v3;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
    const int iters = 200, blocks = 256;
    unsigned long long* out; unsigned* sink; int* ent;
    hipMalloc(&out, blocks * 4 * 8); hipMalloc(&sink, blocks * 256 * 4); hipMalloc(&ent, 16 * 4);
    struct Case { const char* name; int e[16]; int plane; };
    std::vector<Case> cases;
    Case a{"consecutive entries, plane 4096", {}, 4096}, b{"one hole after li=5, plane 4096", {}, 4096}, c{"two holes (after li=3, 12), plane 4096", {}, 4096},
         d{"distinct residues spread over 40 entries, plane 4096", {}, 4096}, e{"consecutive entries, plane 4096+16", {}, 4112}, f{"one hole, plane 3904 (not a multiple of 256)", {}, 3904};
    for (int i = 0; i < 16; ++i) {
        a.e[i] = i; b.e[i] = i + (i > 5); c.e[i] = i + (i > 3) + (i > 12); e.e[i] = i; f.e[i] = i + (i > 5);
        d.e[i] = i + 16 * ((i * 7) % 3); // residue i, blocks 0..2
    }
    cases = {a, b, c, d, e, f};
    for (auto& cs : cases) {
        hipMemcpy(ent, cs.e, 64, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 140 * 1024, 0, out, sink, ent, cs.plane, iters);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("%-56s %.2f cycles per ds_read_b128 per wave (4 waves per CU: x4 = LDS-array cycles if the array is the limit)\n", cs.name, s / h.size() / (iters * 16.0));
    }
    return 0;
}
