// How fast can 256 CUs stream a [2097152 x 768] bf16 matrix (3.2 GB: the folded SegFormerHead's stride-4 map at batch 128) when every
// workgroup owns 256 consecutive rows and reads them as the classifier GEMM does -- a 128-byte column slice of each row per K step,
// the next slice of the same rows a step later -- against slices two or four steps wide?  No arithmetic beyond
// an XOR of the loaded words; DEPTH register sets of loads in flight per thread (as the GEMM's register pipeline).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/probe/stream_pattern_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// WIDTH = 16-byte chunks a row contributes per step (8 = 128 B = one 64-deep bf16 K step, 16 = 256 B, 32 = 512 B).
// 512 threads; a step moves 32 KB per workgroup whatever WIDTH is: rows per step = 2048 / WIDTH.
template <int WIDTH, int DEPTH, bool ONE>
__global__ void __launch_bounds__(512) stream_kernel(const unsigned char* __restrict__ x, uint32_t* __restrict__ out, int rows_per_wg) {
    constexpr int ROW_BYTES = 1536, CHUNKS = 96;
    __shared__ unsigned char one_wg_per_cu[ONE ? 128 * 1024 : 16];        // ONE: as the 256-tile GEMM, whose 136 KB of LDS leave room for one workgroup
    one_wg_per_cu[threadIdx.x & 15] = (unsigned char)threadIdx.x;
    constexpr int RPS = 2048 / WIDTH;                       // rows touched per step
    const unsigned char* base = x + (size_t)blockIdx.x * rows_per_wg * ROW_BYTES;
    const int c = threadIdx.x % WIDTH, r = threadIdx.x / WIDTH;       // (WIDTH <= 512)
    constexpr int LOADS = WIDTH >= 512 ? 1 : 4;             // 16-byte loads per thread per step (512 threads x 4 x 16 B = 32 KB)
    constexpr int RSTRIDE = 512 / WIDTH;                    // rows between a thread's loads
    // step s: row block (s / (CHUNKS / WIDTH)) ... a workgroup walks its rows_per_wg rows in blocks of RPS rows, all column slices of a block
    const int slices = CHUNKS / WIDTH, blocks = rows_per_wg / RPS, steps = slices * blocks;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 st[DEPTH][LOADS];
    auto issue = [&](int s, u32x4 (&dst)[LOADS]) {
        const int sc = s < steps ? s : steps - 1;
        const int blk = sc / slices, sl = sc - blk * slices;
        const unsigned char* p = base + (size_t)(blk * RPS + r) * ROW_BYTES + (sl * WIDTH + c) * 16;
#pragma unroll
        for (int i = 0; i < LOADS; ++i) dst[i] = *reinterpret_cast<const u32x4*>(p + (size_t)i * RSTRIDE * ROW_BYTES);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, st[d]);
    for (int s = 0; s < steps; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i) acc ^= st[d][i];
            issue(s + d + DEPTH, st[d]);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0] + one_wg_per_cu[3];
}

template <int WIDTH, int DEPTH, bool ONE> void run(const unsigned char* x, uint32_t* out, size_t rows, const char* what) {
    const int rows_per_wg = 256;
    const dim3 grid((unsigned)(rows / rows_per_wg));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<WIDTH, DEPTH, ONE>), grid, dim3(512), 0, 0, x, out, rows_per_wg);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<WIDTH, DEPTH, ONE>), grid, dim3(512), 0, 0, x, out, rows_per_wg);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-44s %s depth %d: %7.3f ms  %6.2f TB/s\n", what, ONE ? "one workgroup per CU " : "workgroups as they fit", DEPTH, ms, rows * 1536.0 / ms / 1e9);
}

int main() {
    const size_t rows = 2097152;
    unsigned char* x; uint32_t* out;
    hipMalloc(&x, rows * 1536); hipMalloc(&out, 65536 * 4);
    hipMemset(x, 1, rows * 1536);
    run<8, 2, false>(x, out, rows, "128 B of each of 256 rows per step");
    run<8, 3, false>(x, out, rows, "128 B of each of 256 rows per step");
    run<8, 4, false>(x, out, rows, "128 B of each of 256 rows per step");
    run<16, 2, false>(x, out, rows, "256 B of each of 128 rows per step");
    run<16, 3, false>(x, out, rows, "256 B of each of 128 rows per step");
    run<32, 2, false>(x, out, rows, "512 B of each of 64 rows per step");
    run<32, 3, false>(x, out, rows, "512 B of each of 64 rows per step");
    run<32, 4, false>(x, out, rows, "512 B of each of 64 rows per step");
    run<8, 2, true>(x, out, rows, "128 B of each of 256 rows per step");
    run<8, 3, true>(x, out, rows, "128 B of each of 256 rows per step");
    run<8, 4, true>(x, out, rows, "128 B of each of 256 rows per step");
    run<16, 2, true>(x, out, rows, "256 B of each of 128 rows per step");
    run<16, 3, true>(x, out, rows, "256 B of each of 128 rows per step");
    run<32, 2, true>(x, out, rows, "512 B of each of 64 rows per step");
    run<32, 3, true>(x, out, rows, "512 B of each of 64 rows per step");
    run<32, 4, true>(x, out, rows, "512 B of each of 64 rows per step");
    run<8, 6, true>(x, out, rows, "128 B of each of 256 rows per step");
    run<8, 8, true>(x, out, rows, "128 B of each of 256 rows per step");
    return 0;
}
