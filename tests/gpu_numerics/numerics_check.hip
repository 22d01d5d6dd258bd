// tests/gpu_numerics/numerics_check.hip — exhaustive on-device checks of the instruction-level shortcuts the raster
// kernel relies on.  Each check enumerates its whole input domain on the GPU and counts mismatches against the plain
// IEEE / Rust-`as` formulation; the kernel may only use a shortcut whose count is 0.
//
//   A  v_cvt_pk_u8_f32      == `f as u8` (truncate, saturate to 0..255, NaN -> 0) for ALL 2^32 f32 bit patterns
//   B  hoisted-reciprocal divide == IEEE n / d for every integer pair the wall mapper can form:
//      d = bottom_y - top_y in [-65535, 65535], n = y - top_y in [-32767, 49151]   (bitmap_render.rs:256)
//   C  hoisted-reciprocal divide == IEEE n / vy for vy = CFY - y (all multiples of 0.5 with |vy| <= 8192) and EVERY f32
//      numerator bit pattern (32 769 x 2^32 quotients, ~50 s of GPU time); patterns outside the guard band
//      (0 or 2^-64 <= |n| <= 2^64) are skipped exactly as the kernel skips them
//   A2 v_cvt_pk_u8_f32 under round-toward-zero (no v_trunc; the mode is switched around the convert) == `f as u8`, all 2^32 patterns
//   B2/C2 the same two divides WITHOUT the final v_div_fixup (div_prepared_nofix) on the domains the kernel uses that form on:
//      wall d != 0; flat vy != 0
//   D  float floor-modulus helper == i16 reference fix-up for all t in [-32768, 32767], n in [1, 2048] and a sample above
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../doom-rust-renderer_amd/csrc/raster_core.h"

using namespace dg;

__global__ void check_pk_u8(unsigned long long *bad) {
    const uint64_t total = 1ull << 32;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        float f = __uint_as_float((uint32_t)i);
        int32_t ref = f32_as_u8(f);
        uint32_t got = f32_as_u8_pk(f);
        if ((uint32_t)ref != got) atomicAdd(bad, 1ull);
    }
}

__global__ void check_pk_u8_rtz(unsigned long long *bad) {
    const uint64_t total = 1ull << 32;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        float f = __uint_as_float((uint32_t)i);
        int32_t ref = f32_as_u8(f);
        uint32_t got;
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                     : "=v"(got) : "v"(f));
        // the mode must be back to round-to-nearest for what follows: 1 - 2^-25 is a tie that rounds to 1 under RNE but to 1 - 2^-24
        // under round-toward-zero
        float one = __uint_as_float(0x3f800000u), eps = __uint_as_float(0x33000000u);
        asm volatile("" : "+v"(one), "+v"(eps));          // opaque to constant folding: the subtraction runs on the device
        float probe = one - eps;
        if ((uint32_t)ref != got || __float_as_uint(probe) != 0x3f800000u) atomicAdd(bad, 1ull);
    }
}

__global__ void check_div_wall(unsigned long long *bad) {
    // blockIdx.x enumerates d, threads enumerate n
    const int d_i = (int)blockIdx.x - 65535;
    const float d = (float)d_i;
    const float r = prepare_rcp(d);
    for (int n_i = -32767 + (int)threadIdx.x; n_i <= 49151; n_i += blockDim.x) {
        const float n = (float)n_i;
        float ref = n / d;
        float got = div_prepared(n, d, r);
        if (__float_as_uint(ref) != __float_as_uint(got) && !(ref != ref && got != got)) atomicAdd(bad, 1ull);
        if (d_i != 0) {
            float got2 = div_prepared_nofix(n, d, r);
            if (__float_as_uint(ref) != __float_as_uint(got2) && !(ref == 0.0f && got2 == 0.0f)) atomicAdd(bad + 1, 1ull);
        }
    }
}

__global__ void check_div_flat(unsigned long long *bad, unsigned long long *tested, unsigned long long *bad_nofix, int vy_half_lo, int vy_half_hi) {
    // blockIdx.y enumerates vy (in half units), x-dimension enumerates numerator patterns
    const int vh = vy_half_lo + (int)blockIdx.y;
    if (vh > vy_half_hi) return;
    const float d = (float)vh * 0.5f;
    const float r = prepare_rcp(d);
    const uint64_t total = 1ull << 32;
    unsigned long long local_bad = 0, local_n = 0, local_bad2 = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        float n = __uint_as_float((uint32_t)i);
        if (!div_guard_ok(n)) continue;
        float ref = n / d;
        float got = div_prepared(n, d, r);
        local_n++;
        if (__float_as_uint(ref) != __float_as_uint(got) && !(ref != ref && got != got)) local_bad++;
        if (vh != 0) {   // a zero quotient may come out with the other sign (n = -0.0, d > 0): see div_prepared_nofix
            const float got2 = div_prepared_nofix(n, d, r);
            if (__float_as_uint(ref) != __float_as_uint(got2) && !(ref == 0.0f && got2 == 0.0f)) local_bad2++;
        }
    }
    if (local_bad) atomicAdd(bad, local_bad);
    if (local_bad2) atomicAdd(bad_nofix, local_bad2);
    atomicAdd(tested, local_n);
}

__global__ void check_floor_mod(unsigned long long *bad) {
    const int n = (int)blockIdx.x + 1;
    int nn = n <= 2048 ? n : 2048 + (n - 2048) * 15;     // 1..2048 dense, then every 15th up to 32767
    if (nn > 32767) return;
    const int mask = (nn & (nn - 1)) == 0 ? nn - 1 : 0;
    const float rcp = approx_rcp((float)nn);
    for (int t = -32768 + (int)threadIdx.x; t <= 32767; t += blockDim.x) {
        if (floor_mod_fast(t, nn, mask, rcp) != floor_mod_i16(t, nn)) atomicAdd(bad, 1ull);
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 2; } } while (0)

int main() {
    unsigned long long *d_bad, *d_n, h[3];
    CK(hipMalloc(&d_bad, 24));
    d_n = d_bad + 1;
    int fails = 0;

    CK(hipMemset(d_bad, 0, 24));
    hipLaunchKernelGGL(check_pk_u8, dim3(4096), dim3(256), 0, 0, d_bad);
    CK(hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost));
    std::printf("A cvt_pk_u8 vs `as u8`, 2^32 patterns: mismatches %llu\n", h[0]);
    fails += h[0] != 0;

    CK(hipMemset(d_bad, 0, 24));
    hipLaunchKernelGGL(check_pk_u8_rtz, dim3(4096), dim3(256), 0, 0, d_bad);
    CK(hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost));
    std::printf("A2 cvt_pk_u8 under round-toward-zero vs `as u8` (and the mode restored), 2^32 patterns: mismatches %llu\n", h[0]);
    fails += h[0] != 0;

    CK(hipMemset(d_bad, 0, 24));
    hipLaunchKernelGGL(check_div_wall, dim3(131071), dim3(256), 0, 0, d_bad);
    CK(hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost));
    std::printf("B wall ay divide, 131071 x 81919 pairs: mismatches %llu; B2 without v_div_fixup (d != 0): mismatches %llu\n", h[0], h[1]);
    fails += h[0] != 0 || h[1] != 0;

    {
        unsigned long long tot_bad = 0, tot_n = 0, tot_bad2 = 0;
        for (int lo = -16384; lo <= 16384; lo += 512) {                 // 512 vy values per launch keeps every launch ~1 s
            const int hi = lo + 511 > 16384 ? 16384 : lo + 511;
            CK(hipMemset(d_bad, 0, 24));
            hipLaunchKernelGGL(check_div_flat, dim3(2048, (unsigned)(hi - lo + 1)), dim3(256), 0, 0, d_bad, d_n, d_bad + 2, lo, hi);
            CK(hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost));
            tot_bad += h[0]; tot_n += h[1]; tot_bad2 += h[2];
            if ((lo & 4095) == 0) { std::printf("  .. vy/2 up to %d: %llu quotients, %llu mismatches\n", hi, tot_n, tot_bad); std::fflush(stdout); }
        }
        std::printf("C flat divide, every vy in [-8192, 8192] step 0.5 x every f32 numerator: tested %llu mismatches %llu; C2 without v_div_fixup (vy != 0): mismatches %llu\n",
                    tot_n, tot_bad, tot_bad2);
        fails += tot_bad != 0 || tot_bad2 != 0;
    }

    CK(hipMemset(d_bad, 0, 24));
    hipLaunchKernelGGL(check_floor_mod, dim3(4096), dim3(256), 0, 0, d_bad);
    CK(hipMemcpy(h, d_bad, 24, hipMemcpyDeviceToHost));
    std::printf("D floor modulus helper: mismatches %llu\n", h[0]);
    fails += h[0] != 0;

    std::printf(fails ? "NUMERICS FAIL\n" : "NUMERICS OK\n");
    return fails ? 1 : 0;
}
