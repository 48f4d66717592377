// Issue cost of v_pk_fma_f32 against two v_fma_f32 (one wave per SIMD and eight; independent accumulators).
// hipcc --offload-arch=gfx950 -O3 tools/probe/pkfma_probe.hip -o /tmp/pkfma_probe && /tmp/pkfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f2 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, 1.f};
    const f2 va = {a, a * 1.0001f}, vb = {b, b * 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
            else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(va.x), "v"(vb.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(va.y), "v"(vb.y));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 8; waves *= 2)
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = 256 * waves;                       // 4 waves per block: `waves` waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("waves/SIMD %d  %s: %.3f ms  -> %.2f cycles per PAIR of fmas per SIMD at 2.4 GHz\n", waves,
                   mode == 0 ? "v_pk_fma_f32" : "2 x v_fma_f32 ", ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * waves));
        }
    return 0;
}
