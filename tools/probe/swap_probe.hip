#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    u32x2 s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y; o[128 + threadIdx.x] = s.x; o[192 + threadIdx.x] = s.y;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"p32.x", "p32.y", "p16.x", "p16.y"};
    for (int q = 0; q < 4; ++q) { printf("%s:", nm[q]); for (int i = 0; i < 64; i += 4) printf(" %u", h[q * 64 + i]); printf("\n"); }
    return 0;
}
