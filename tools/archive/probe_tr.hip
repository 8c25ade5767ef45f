// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds value = index; every lane of a
// 16-lane group g points at 8-byte chunk i of a 4x16 row-major block (rows of `ld` elements).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(short* out, int ld) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const short* p = lds + (4 * g + (i >> 2)) * ld + (i & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
    const int ld = 64;
    probe<<<1, 64>>>(d, ld);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        const int i = l & 15, g = l >> 4;
        printf("lane %2d (i=%2d g=%d):", l, i, g);
        for (int e = 0; e < 4; ++e) {
            const int row = h[l * 4 + e] / ld, col = h[l * 4 + e] % ld;
            printf(" [r%d,c%d]", row, col);
            if (row != 4 * g + e || col != i) ok = 0;   // expected: element e = row 4g+e, column i
        }
        printf("\n");
    }
    printf("TR_SEMANTICS_%s\n", ok ? "AS_EXPECTED" : "DIFFERENT");
    return 0;
}
