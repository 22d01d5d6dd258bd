// fe_kernels.hip — the device column walk (gfx950): turns the per-seg / per-sprite records of a batch of frames
// (fe_dev.h, built by Walker in parts mode) into the column-major DevRSpan lists dg_raster_tiles walks.
//
//   dg_fe_columns   one lane per (frame, screen column): walks the frame's parts in BSP order with the column's three
//                   occlusion values in registers (the reference keeps them in horizontal_ocl / floor_ver_ocl /
//                   ceiling_ver_ocl, segs.rs:70-74) and appends every wall / visplane / sprite span, in a compact 16-byte
//                   form, to the column's scratch list.  Records are wave-uniform: 64 lanes test 64 records' column
//                   ranges at once, the hits are fetched with one coalesced load and broadcast with v_readlane.  HBM-bound scratch writes are laid
//                   out [slot][column] so the 64 lanes of a wave store 64 adjacent records.
//   dg_fe_gaps      one wave per (frame, part with a sky flat): the 1-pixel sky entries of zero-filled visplane columns.
//   dg_fe_scan      one workgroup per frame: scans the per-column counts into col_off.
//   dg_fe_scatter   ranks every column's spans by draw-order key, resolves their texture-mapping constants (DevRSpan) and
//                   scatters them into the column-major list.
//
// Integer / f32 work only — nothing here is GEMM shaped.
#include <hip/hip_runtime.h>

#include "fe_core.h"
#include "fe_kernels.hpp"

namespace dg {

namespace {

constexpr int FE_COL_THREADS = 256;
constexpr int FE_SCAN_THREADS = 1024;
constexpr int FE_SCATTER_GROUPS = 4;      // slot groups per screen column in dg_fe_scatter

// A record is wave-uniform.  Lanes 0..N-1 fetch one dword each (one coalesced load instead of a chain of scalar-cache
// misses: a frame's records do not fit the 16 KB scalar cache), then every field is broadcast with v_readlane.
template <typename T>
__device__ __forceinline__ uint32_t fetch_words(const T *rec, int lane) {
    constexpr int N = (int)(sizeof(T) / 4);
    static_assert(N <= 64, "record larger than a wave");
    return reinterpret_cast<const uint32_t *>(rec)[lane < N ? lane : 0];
}
template <typename T>
__device__ __forceinline__ T unpack_words(uint32_t v) {
    constexpr int N = (int)(sizeof(T) / 4);
    union { T t; uint32_t w[N]; } u;
#pragma unroll
    for (int k = 0; k < N; k++) u.w[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, k);
    return u.t;
}

__global__ __launch_bounds__(FE_COL_THREADS) void dg_fe_columns(FeParams P) {
    const int f = blockIdx.y;
    const int W = P.k.W;
    const int x = (int)(blockIdx.x * FE_COL_THREADS + threadIdx.x);
    const bool active = x < W;
    const int lane = (int)(threadIdx.x & 63);
    const int wx0 = __builtin_amdgcn_readfirstlane(x - lane), wx1 = wx0 + 63;
    if (wx0 >= W) return;                                                               // whole wave past the right edge
    const FeFrame ff = P.fframes[f];
    const uint32_t n_parts = __builtin_amdgcn_readfirstlane(ff.n_parts), part_base = __builtin_amdgcn_readfirstlane(ff.part_base);
    const uint32_t n_sprites = __builtin_amdgcn_readfirstlane(ff.n_sprites), sprite_base = __builtin_amdgcn_readfirstlane(ff.sprite_base);
    const FePart *parts = P.parts + part_base;
    const uint32_t *bounds = P.bounds + part_base;
    const FeSprite *sprites = P.sprites + sprite_base;

    FeColumn c;
    c.x = x; c.hor = 0; c.fo = P.k.H; c.co = -1; c.nsp = 0; c.nrec = 0; c.ovf = 0;     // Segs::new, segs.rs:97-99

    // Parts in BSP order, 64 at a time: every lane tests one part's column range against the wave's 64 columns, the hits
    // are then processed in order (lowest bit first) with the next hit's record already in flight.
    for (uint32_t base = 0; base < n_parts; base += 64) {
        const uint32_t b = base + (uint32_t)lane < n_parts ? bounds[base + (uint32_t)lane] : 0xffffu;   // sx = 0xffff, ex = 0: never hits
        const int bsx = (int)(b & 0xffffu), bex = (int)(b >> 16);
        uint64_t hit = __ballot(bex >= wx0 && bsx <= wx1);
        uint32_t next = hit ? fetch_words(parts + base + (uint32_t)__builtin_ctzll(hit), lane) : 0u;
        while (hit) {
            const uint32_t pi = base + (uint32_t)__builtin_ctzll(hit);
            hit &= hit - 1;
            const FePart p = unpack_words<FePart>(next);
            if (hit) next = fetch_words(parts + base + (uint32_t)__builtin_ctzll(hit), lane);
            uint32_t ev = 0;
            if (active && x >= p.sx && x <= p.ex) ev = fe_part_column(P, f, p, pi, c);
            if (p.sky_slot >= 0) {                                                      // wave-uniform: all 64 lanes reach the ballots
                const uint64_t bf = __ballot((ev & FE_EV_FADD) != 0), bc = __ballot((ev & FE_EV_CADD) != 0), bl = __ballot((ev & FE_EV_FLUSH) != 0);
                if (lane == 0) {
                    uint64_t *e = P.events + ((size_t)f * FE_MAX_SKY_SLOTS + (size_t)p.sky_slot) * 3 * (size_t)P.w64 + (size_t)(wx0 >> 6);
                    e[0] = bf;
                    e[P.w64] = bc;
                    e[2 * (size_t)P.w64] = bl;
                }
            }
        }
    }
    for (uint32_t base = 0; base < n_sprites; base += 64) {
        int sx0 = 1, sx1 = 0;
        if (base + (uint32_t)lane < n_sprites) { sx0 = sprites[base + (uint32_t)lane].x0; sx1 = sprites[base + (uint32_t)lane].x1; }
        uint64_t hit = __ballot(sx1 > wx0 && sx0 <= wx1 && sx0 < sx1);
        while (hit) {
            const uint32_t si = base + (uint32_t)__builtin_ctzll(hit);
            hit &= hit - 1;
            const FeSprite s = unpack_words<FeSprite>(fetch_words(sprites + si, lane));
            if (active && x >= s.x0 && x < s.x1) fe_sprite_column(P, f, ff, s, si, c);
        }
    }
    if (active) P.cnt[(size_t)f * (size_t)W + (size_t)x] = c.nsp;
    if (c.ovf) atomicOr(&P.flags[f], c.ovf);
}

// One wave per (frame, sky slot): the zero-filled entries of sky visplanes draw one sky pixel at row 0
// (visplanes.rs:61-80 with top = bottom = 0).
__global__ __launch_bounds__(64) void dg_fe_gaps(FeParams P) {
    const int f = blockIdx.y;
    const FeFrame ff = P.fframes[f];
    const uint32_t si = blockIdx.x;
    if (si >= ff.n_sky_slots) return;
    const int W = P.k.W;
    const int lane = (int)threadIdx.x;
    const uint32_t pi = P.sky_parts[ff.sky_base + si];
    const FePart &p = P.parts[ff.part_base + pi];
    const int sx = p.sx, ex = p.ex;
    const uint32_t fl = p.flags;
    uint32_t *cnt = P.cnt + (size_t)f * (size_t)W;
    for (int kind = 0; kind < 2; kind++) {
        if (!(fl & (kind ? FEP_CEIL_SKY : FEP_FLOOR_SKY))) continue;
        const uint64_t *add = fe_event_words(P, f, (int32_t)si, kind), *flush = fe_event_words(P, f, (int32_t)si, 2);
        for (int x = sx + lane; x <= ex; x += 64) {
            if (!fe_gap(add, flush, x, sx, ex)) continue;
            const uint32_t slot = atomicAdd(&cnt[x], 1u);
            if (slot >= P.col_slots) { atomicOr(&P.flags[f], (uint32_t)FE_OVF_SPANS); continue; }
            P.cspans[((size_t)f * P.col_slots + slot) * (size_t)W + (size_t)x] =
                FeU4{FE_KEY_PLANE | (pi << 2) | (uint32_t)kind, 0u, 0u, (uint32_t)SPAN_SKY << FES_KIND_SHIFT};
        }
    }
}

// One workgroup per frame: exclusive scan of the per-column span counts -> col_off.
__global__ __launch_bounds__(FE_SCAN_THREADS) void dg_fe_scan(FeParams P) {
    __shared__ uint32_t wave_sum[FE_SCAN_THREADS / 64];
    __shared__ uint32_t total_s;
    const int f = blockIdx.x;
    const int W = P.k.W;
    const int tid = (int)threadIdx.x;
    const uint32_t *cnt = P.cnt + (size_t)f * (size_t)W;

    // each thread owns a contiguous chunk of columns
    const int chunk = (W + FE_SCAN_THREADS - 1) / FE_SCAN_THREADS;
    const int xa = tid * chunk, xb = min(W, xa + chunk);
    uint32_t mine = 0;
    for (int x = xa; x < xb; x++) mine += min(cnt[x], P.col_slots);
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int w = 0; w < FE_SCAN_THREADS / 64; w++) { const uint32_t v = wave_sum[w]; wave_sum[w] = run; run += v; }
        total_s = run;
    }
    __syncthreads();
    const uint32_t total = total_s;
    const bool fits = total <= P.span_stride;      // a frame that does not fit draws nothing and is redone on the host
    uint32_t off = wave_sum[tid >> 6] + incl - mine;
    uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    if (tid == 0) {
        coff[W] = fits ? total : 0u;
        P.totals[f] = total;
        if (!fits) atomicOr(&P.flags[f], (uint32_t)FE_OVF_FRAME);
    }
    for (int x = xa; x < xb; x++) {
        coff[x] = fits ? off : 0u;
        off += min(cnt[x], P.col_slots);
    }
}

// Every column's spans into draw order and into the raster kernel's form: rank by key among the column's spans
// (distinct keys), resolve the texture-mapping constants (what dg_setup_spans does for host-built lists), scatter into the
// column-major list.  64 adjacent columns x FE_SCATTER_GROUPS slot groups per workgroup: the lanes of a wave read the
// same slot of 64 adjacent columns (coalesced in the [slot][column] scratch layout); the keys are staged in LDS once.
__global__ __launch_bounds__(64 * FE_SCATTER_GROUPS) void dg_fe_scatter(FeParams P) {
    __shared__ uint32_t lkeys[FE_MAX_COL_SLOTS * 64];
    const int f = blockIdx.y;
    const int W = P.k.W;
    const int lx = (int)(threadIdx.x & 63);
    const int x = (int)(blockIdx.x * 64) + lx;
    const uint32_t g = threadIdx.x >> 6;
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    uint32_t off = 0, n = 0;
    if (x < W) { off = coff[x]; n = coff[x + 1] - off; }      // n = 0 for every column of a frame that did not fit
    const FeU4 *src = P.cspans + (size_t)f * P.col_slots * (size_t)W + (size_t)x;
    for (uint32_t i = g; i < n; i += FE_SCATTER_GROUPS) lkeys[i * 64 + (uint32_t)lx] = src[(size_t)i * (size_t)W].x;
    __syncthreads();
    const DevFrame fr = P.frames[f];
    const FeFrame ff = P.fframes[f];
    FeU4 *out = reinterpret_cast<FeU4 *>(P.rspans + fr.span_base);
    for (uint32_t i = g; i < n; i += FE_SCATTER_GROUPS) {
        const FeU4 cs = src[(size_t)i * (size_t)W];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t kj = lkeys[j * 64 + (uint32_t)lx];
            rank += (kj < cs.x || (kj == cs.x && j < i)) ? 1u : 0u;        // the tie-break keeps the scatter a permutation
        }
        const DevRSpan r = fe_resolve(P, fr, ff, x, cs);
        out[2 * (size_t)(off + rank)] = FeU4{r.w[0], r.w[1], r.w[2], r.w[3]};
        out[2 * (size_t)(off + rank) + 1] = FeU4{r.w[4], r.w[5], r.w[6], r.w[7]};
    }
}

}  // namespace

hipError_t launch_fe(const FeParams &P, hipStream_t stream) {
    if (P.n_frames <= 0) return hipSuccess;
    dim3 grid((unsigned)((P.k.W + FE_COL_THREADS - 1) / FE_COL_THREADS), (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_fe_columns, grid, dim3(FE_COL_THREADS), 0, stream, P);
    if (P.max_sky_slots) hipLaunchKernelGGL(dg_fe_gaps, dim3(P.max_sky_slots, (unsigned)P.n_frames), dim3(64), 0, stream, P);
    hipLaunchKernelGGL(dg_fe_scan, dim3((unsigned)P.n_frames), dim3(FE_SCAN_THREADS), 0, stream, P);
    hipLaunchKernelGGL(dg_fe_scatter, dim3((unsigned)((P.k.W + 63) / 64), (unsigned)P.n_frames), dim3(64 * FE_SCATTER_GROUPS), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dg
