#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 2] = r[0]; out[threadIdx.x * 2 + 1] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 7) printf("lane %d: %u %u\n", l, h[2 * l], h[2 * l + 1]);
    return 0;
}
