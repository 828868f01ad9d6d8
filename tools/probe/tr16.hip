// probe: ds_read_b64_tr_b16 semantics on gfx950 (hypothesis: within a 16-lane group, lane i supplies the address of 4 contiguous
// 16-bit elements M[i/4][(i%4)*4 ..+3] of a 4 x 16 block; lane c receives column c: M[0][c], M[1][c], M[2][c], M[3][c])
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint16_t* out)
{
    __shared__ uint16_t lds[64 * 16];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 16; i += 64) lds[i] = (uint16_t)i;          // element value = row*16 + col, rows of 16 elements (32 B)
    __syncthreads();
    const int i = l & 15, r = i >> 2, q = i & 3, grp = l >> 4;
    // group g reads rows 4g .. 4g+3
    const uint32_t addr = (uint32_t)(uintptr_t)lds + (uint32_t)(((grp * 4 + r) * 16 + q * 4) * 2);
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = (uint16_t)(v.x & 0xFFFF); out[l * 4 + 1] = (uint16_t)(v.x >> 16);
    out[l * 4 + 2] = (uint16_t)(v.y & 0xFFFF); out[l * 4 + 3] = (uint16_t)(v.y >> 16);
}
int main()
{
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
