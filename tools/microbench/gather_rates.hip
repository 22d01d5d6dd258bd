// gather_rates.hip — what one CU of gfx950 (MI355X) sustains for the memory instructions of dg_raster_tiles, by address pattern:
// per-lane byte gathers out of a texture column (consecutive bytes), a 4 KB flat (random within 4 KB), and worse; the palette lookup
// (ds_read_b128 / b96 at a random one of 256 16-byte entries); the read-out's 12-byte-per-lane row-segment stores.
//
// Every CU runs k workgroups of 256 threads (k waves per SIMD); a wave issues REP x 8 instructions, eight in flight before each
// wait.  Printed: core clocks (s_memtime) per wave-instruction per CU = slowest wave's loop time / (instructions per wave x 4 k),
// i.e. how often the CU's texture-address / L1 / LDS path accepts one such wave-instruction when the whole chip does the same.
//
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o gather_rates tools/microbench/gather_rates.hip && ./gather_rates
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REP = 512;

enum Pattern { P_CONSEC, P_MAGNIFIED, P_FLAT4K, P_LINES8, P_STRIDE128, P_RANDOM2M, P_COUNT };
static const char *pattern_name[P_COUNT] = {
    "64 consecutive bytes (a wall column, 1:1)", "32 consecutive bytes, each read twice (a magnified wall column)",
    "random within one 4 KB flat", "random within 8 lines of 128 B", "one byte per 128-B line, 64 lines", "random within 2 MB",
};

__device__ __forceinline__ uint32_t rnd(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ __forceinline__ uint32_t lane_offset(int pattern, int lane, uint32_t salt) {
    switch (pattern) {
    case P_CONSEC: return (salt & 0xfffc0u) + (uint32_t)lane;
    case P_MAGNIFIED: return (salt & 0xfffc0u) + (uint32_t)(lane >> 1);
    case P_FLAT4K: return (salt & 0xff000u) + (rnd(salt + (uint32_t)lane) & 0xfffu);
    case P_LINES8: return (salt & 0xff000u) + (rnd(salt + (uint32_t)lane) & 0x3ffu);
    case P_STRIDE128: return (salt & 0xfe000u) + 128u * (uint32_t)lane;
    default: return rnd(salt + (uint32_t)lane) & 0x1fffffu;
    }
}

template <int BYTES>
__global__ __launch_bounds__(256) void k_gather(unsigned long long *out, const uint8_t *buf, int pattern) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave_salt = rnd(blockIdx.x * 4 + (threadIdx.x >> 6)) & 0x1ff000u;
    uint32_t acc = 0, o[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (lane_offset(pattern, lane, wave_salt + 64u * (uint32_t)j) & 0x1fffffu) & ~(uint32_t)(BYTES - 1);   // addresses fixed outside the timed loop
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
        uint32_t v[8];
        const uint32_t shift = (uint32_t)(i & 15) << 12;      // another 4 KB block every iteration: the same address pattern, other lines
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t a = o[j] + shift;
            if (BYTES == 1) asm volatile("global_load_ubyte %0, %1, %2" : "=v"(v[j]) : "v"(a), "s"(buf) : "memory");
            else asm volatile("global_load_ushort %0, %1, %2" : "=v"(v[j]) : "v"(a), "s"(buf) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
    if (acc == 0x12345678u) out[0] = 0;
}

// Palette lookups: 256 entries of 16 bytes; `distinct` = how many different entries the 64 lanes of one instruction ask for.
template <int WORDS>
__global__ __launch_bounds__(256) void k_palette(unsigned long long *out, const uint8_t *buf, int distinct) {
    __shared__ float4 pal[256];
    pal[threadIdx.x] = make_float4((float)threadIdx.x, 1.0f, 2.0f, 3.0f);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    uint32_t a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t e = rnd((uint32_t)((distinct >= 64 ? lane : lane % distinct) + 64 * j)) & 255u;
        a[j] = (uint32_t)(uintptr_t)pal + 16u * e;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (WORDS == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(a[j]) : "memory");
            else { typedef float f3 __attribute__((ext_vector_type(3))); f3 t; asm volatile("ds_read_b96 %0, %1" : "=v"(t) : "v"(a[j]) : "memory"); v[j] = make_float4(t.x, t.y, t.z, 0.0f); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x + v[j].y + v[j].z;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
    if (acc == 123.456f) out[0] = 0;
}

// The read-out's stores: 12 bytes per lane, 8 adjacent lanes = 96 contiguous bytes of one frame row, 8 rows (3 840 B apart) per instruction.
__global__ __launch_bounds__(256) void k_store12(unsigned long long *out, uint8_t *fb) {
    const int lane = threadIdx.x & 63;
    const size_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t *base = fb + (wave % 4096) * 65536;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(base + (size_t)((lane >> 3) + 8 * ((i * 8 + j) & 1)) * 3840 + (size_t)(lane & 7) * 12 + (size_t)((i * 8 + j) >> 1 & 7) * 96);
            dst[0] = (uint32_t)i; dst[1] = (uint32_t)j; dst[2] = (uint32_t)lane;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[wave * 2] = t0; out[wave * 2 + 1] = t1; }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
    uint8_t *buf; unsigned long long *d_out;
    const size_t buf_bytes = (size_t)4096 * 65536;
    CHECK(hipMalloc(&buf, buf_bytes));
    CHECK(hipMemset(buf, 0x5a, buf_bytes));
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&d_out, (size_t)max_blocks * 8 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h((size_t)max_blocks * 8);
    auto report = [&](const char *what, double per_wave_instr, auto launch) {
        printf("%-72s |", what);
        for (int k : {1, 2, 4, 8}) {
            const int blocks = cus * k;
            for (int rep = 0; rep < 2; rep++) { launch(blocks); CHECK(hipDeviceSynchronize()); }
            CHECK(hipMemcpy(h.data(), d_out, (size_t)blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long worst = 0;
            for (int w = 0; w < blocks * 4; w++) worst = std::max(worst, h[2 * w + 1] - h[2 * w]);
            printf(" %7.1f", (double)worst / (per_wave_instr * 4.0 * k));
        }
        printf("\n");
        fflush(stdout);
    };
    printf("clocks per wave-instruction per CU (4 SIMDs x k waves issuing), slowest wave:%*s |     k=1     k=2     k=4     k=8\n", 0, "");
    for (int p = 0; p < P_COUNT; p++) {
        char name[128];
        snprintf(name, sizeof name, "global_load_ubyte  %s", pattern_name[p]);
        report(name, REP * 8.0, [&](int blocks) { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, d_out, buf, p); });
    }
    for (int p : {P_CONSEC, P_FLAT4K}) {
        char name[128];
        snprintf(name, sizeof name, "global_load_ushort %s (16-bit texels)", pattern_name[p]);
        report(name, REP * 8.0, [&](int blocks) { hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, d_out, buf, p); });
    }
    for (int d : {1, 4, 16, 64, 256}) {
        char name[128];
        snprintf(name, sizeof name, "ds_read_b128 palette entry, %d distinct of 256 entries per instruction%s", d > 64 ? 64 : d, d >= 256 ? " (all lanes random)" : "");
        report(name, REP * 8.0, [&](int blocks) { hipLaunchKernelGGL(k_palette<4>, dim3(blocks), dim3(256), 0, 0, d_out, buf, d); });
    }
    for (int d : {16, 256}) {
        char name[128];
        snprintf(name, sizeof name, "ds_read_b96  palette entry, %d distinct", d > 64 ? 64 : d);
        report(name, REP * 8.0, [&](int blocks) { hipLaunchKernelGGL(k_palette<3>, dim3(blocks), dim3(256), 0, 0, d_out, buf, d); });
    }
    report("global_store_dwordx3, 8 lanes x 12 B contiguous, 8 rows per instruction", REP * 8.0, [&](int blocks) { hipLaunchKernelGGL(k_store12, dim3(blocks), dim3(256), 0, 0, d_out, buf); });
    return 0;
}
