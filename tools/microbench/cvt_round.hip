// Does v_cvt_pk_u8_f32 follow MODE.fp_round?  Sweeps all 2^32 f32 patterns: counts where cvt_pk_u8(x) under round-toward-zero differs
// from cvt_pk_u8(trunc(x)) under round-to-nearest (the `as u8` the rasteriser needs).  hipcc --offload-arch=gfx950 -O2 cvt_round.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sweep(unsigned long long *bad, unsigned *first) {
    unsigned long long n = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        unsigned a, b;
        const float t = __builtin_truncf(x);
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(a) : "v"(t));
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(b) : "v"(x));
        if (a != b) { if (!n) atomicMin(first, (unsigned)i); n++; }
    }
    atomicAdd(bad, n);
}
int main() {
    unsigned long long *bad; unsigned *first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0xff, 4);
    hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long h; unsigned f;
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("patterns where cvt_pk_u8 under RTZ != cvt_pk_u8(trunc) under RNE: %llu (first 0x%08x)\n", h, f);
    return 0;
}
