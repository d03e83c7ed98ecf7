#include <hip/hip_runtime.h>
#include <cstdio>
// ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive when every lane of a 16-lane group supplies the address
// of 4 contiguous 16-bit elements of a row-major [4][16] block (lane i: row i/4, columns (i%4)*4 .. +4)?
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out) {
    __shared__ short sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, grp = l >> 4;
    const short* p = sm + grp * 64 + (i >> 2) * 16 + (i & 3) * 4;
    s4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = w[j];
}
int main() {
    short* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (h[l * 4 + j] != (l & 15) + j * 16 + (l >> 4) * 64) ok = 0;
    printf("lane l elem j == lds[(l&15) + j*16 + (l>>4)*64]: %s\n", ok ? "yes" : "NO");
    for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
