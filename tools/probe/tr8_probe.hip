// Probe of ds_read_b64_tr_b8 (gfx950): which LDS bytes does each lane receive?  LDS byte at (row r, col c) of a [16][16] byte tile holds
// (r << 4) | c; lane L of the first 16-lane group supplies the address of (row = L / 2, cols 8 * (L % 2) ...) -- the analogue of the
// b16 form's map (lane 4q + p -> row q, columns 4p .. 4p + 3).  Prints the 8 bytes every lane gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(uint32_t* out, int variant) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 256];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned char)((((i >> 4) & 15) << 4) | (i & 15));     // 64 rows x 16 cols; row id mod 16
    __syncthreads();
    const int L = threadIdx.x & 15, grp = threadIdx.x >> 4;
    int row, col;
    if (variant == 0) { row = L >> 1; col = 8 * (L & 1); }
    else { row = L & 7; col = 8 * (L >> 3); }
    const unsigned char* p = lds + (grp * 16 + row) * 16 + col;      // group g reads rows 16 g ..
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)p);
    out[2 * threadIdx.x] = r[0]; out[2 * threadIdx.x + 1] = r[1];
}
int main() {
    uint32_t* d; hipMalloc(&d, 64 * 8);
    for (int variant = 0; variant < 2; ++variant) {
        k<<<1, 64>>>(d, variant);
        uint32_t h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("variant %d (lane -> %s)\n", variant, variant == 0 ? "row L/2, cols 8(L%2)" : "row L%8, cols 8(L/8)");
        for (int l = 0; l < 16; ++l) {
            printf(" lane %2d:", l);
            for (int b = 0; b < 8; ++b) { unsigned v = (h[2 * l + b / 4] >> (8 * (b % 4))) & 0xff; printf(" r%uc%u", v >> 4, v & 15); }
            printf("\n");
        }
    }
    return 0;
}
