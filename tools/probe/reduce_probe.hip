#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__global__ void k(const float* in, float* o) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
    float u[8], t[4];
    for (int i = 0; i < 8; ++i) {
        const u32x2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[i]), __builtin_bit_cast(unsigned, v[i + 8]), false, false);
        u[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.y);
    }
    for (int i = 0; i < 8; ++i) o[(0 + i) * 64 + threadIdx.x] = u[i];
    for (int i = 0; i < 4; ++i) {
        const u32x2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, u[i]), __builtin_bit_cast(unsigned, u[i + 4]), false, false);
        t[i] = __builtin_bit_cast(float, r.x) + __builtin_bit_cast(float, r.y);
    }
    for (int i = 0; i < 4; ++i) o[(8 + i) * 64 + threadIdx.x] = t[i];
    for (int i = 0; i < 4; ++i) {
        t[i] += dpp_mov<0xB1>(t[i]); t[i] += dpp_mov<0x4E>(t[i]); t[i] += dpp_mov<0x141>(t[i]); t[i] += dpp_mov<0x140>(t[i]);
    }
    for (int i = 0; i < 4; ++i) o[(12 + i) * 64 + threadIdx.x] = t[i];
}
int main() {
    float h[64 * 16]; for (int r = 0; r < 64; ++r) for (int i = 0; i < 16; ++i) h[r * 16 + i] = (i == 0) ? 1.f : (i == 5 ? (float)r : 0.f);
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16 * 64 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    float r[16 * 64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    const char* nm[16] = {"u0","u1","u2","u3","u4","u5","u6","u7","t0","t1","t2","t3","T0","T1","T2","T3"};
    for (int q = 0; q < 16; ++q) { printf("%s:", nm[q]); for (int i = 0; i < 64; i += 8) printf(" %g", r[q * 64 + i]); printf("\n"); }
    return 0;
}
