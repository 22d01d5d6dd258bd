// fe_kernels.hip — the device column walk (gfx950): turns the per-seg / per-sprite records of a batch of frames
// (fe_dev.h, built by Walker in parts mode) into the column-major DevRSpan lists dg_raster_tiles walks.
//
//   dg_fe_columns   one lane per (frame, screen column): walks the frame's parts in BSP order with the column's three
//                   occlusion values in registers (the reference keeps them in horizontal_ocl / floor_ver_ocl /
//                   ceiling_ver_ocl, segs.rs:70-74), resolves every wall / visplane / sprite span on the spot and appends
//                   it to the column's scratch list.  Part and sprite records are wave-uniform (scalar loads); parts
//                   that miss the wave's 64 columns are skipped by a scalar branch.  HBM-bound scratch writes are laid
//                   out [slot][column] so the 64 lanes of a wave store 64 adjacent records.
//   dg_fe_finalize  one workgroup per frame: adds the 1-pixel sky entries of zero-filled visplane columns (fe_gap),
//                   scans the per-column counts into col_off and scatters every column's spans in draw order.
//
// Integer / f32 work only — nothing here is GEMM shaped.
#include <hip/hip_runtime.h>

#include "fe_core.h"
#include "fe_kernels.hpp"

namespace dg {

namespace {

constexpr int FE_COL_THREADS = 256;
constexpr int FE_FIN_THREADS = 1024;

__global__ __launch_bounds__(FE_COL_THREADS) void dg_fe_columns(FeParams P) {
    const int f = blockIdx.y;
    const int W = P.k.W;
    const int x = (int)(blockIdx.x * FE_COL_THREADS + threadIdx.x);
    const bool active = x < W;
    const int lane = (int)(threadIdx.x & 63);
    const int wx0 = __builtin_amdgcn_readfirstlane(x - lane), wx1 = wx0 + 63;
    const DevFrame fr = P.frames[f];
    const FeFrame ff = P.fframes[f];
    const uint32_t n_parts = __builtin_amdgcn_readfirstlane(ff.n_parts), part_base = __builtin_amdgcn_readfirstlane(ff.part_base);
    const uint32_t n_sprites = __builtin_amdgcn_readfirstlane(ff.n_sprites), sprite_base = __builtin_amdgcn_readfirstlane(ff.sprite_base);

    FeColumn c;
    c.x = x; c.hor = 0; c.fo = P.k.H; c.co = -1; c.nsp = 0; c.nrec = 0; c.ovf = 0;     // Segs::new, segs.rs:97-99

    for (uint32_t pi = 0; pi < n_parts; pi++) {
        const FePart &p = P.parts[part_base + pi];
        const int sx = p.sx, ex = p.ex;
        if (ex < wx0 || sx > wx1) continue;                                             // wave-uniform
        uint32_t ev = 0;
        if (active && x >= sx && x <= ex) ev = fe_part_column(P, f, fr, p, pi, c);
        const int slot = p.sky_slot;
        if (slot >= 0) {                                                                // wave-uniform: all 64 lanes reach the ballots
            const uint64_t bf = __ballot((ev & FE_EV_FADD) != 0), bc = __ballot((ev & FE_EV_CADD) != 0), bl = __ballot((ev & FE_EV_FLUSH) != 0);
            if (lane == 0) {
                uint64_t *e = P.events + ((size_t)f * FE_MAX_SKY_SLOTS + (size_t)slot) * 3 * (size_t)P.w64 + (size_t)(wx0 >> 6);
                e[0] = bf;
                e[P.w64] = bc;
                e[2 * (size_t)P.w64] = bl;
            }
        }
    }
    for (uint32_t si = 0; si < n_sprites; si++) {
        const FeSprite &s = P.sprites[sprite_base + si];
        const int x0 = s.x0, x1 = s.x1;
        if (x1 <= wx0 || x0 > wx1) continue;
        if (active && x >= x0 && x < x1) fe_sprite_column(P, f, ff, s, c);
    }
    if (active) P.cnt[(size_t)f * (size_t)W + (size_t)x] = c.nsp;
    if (c.ovf) atomicOr(&P.flags[f], c.ovf);
}

__global__ __launch_bounds__(FE_FIN_THREADS) void dg_fe_finalize(FeParams P) {
    __shared__ uint32_t wave_sum[FE_FIN_THREADS / 64];
    __shared__ uint32_t total_s;
    const int f = blockIdx.x;
    const int W = P.k.W;
    const int tid = (int)threadIdx.x;
    const DevFrame fr = P.frames[f];
    const FeFrame ff = P.fframes[f];
    uint32_t *cnt = P.cnt + (size_t)f * (size_t)W;

    // 1. zero-filled entries of sky visplanes draw one sky pixel at row 0 (visplanes.rs:61-80 with top = bottom = 0)
    for (uint32_t pi = 0; pi < ff.n_parts; pi++) {
        const FePart &p = P.parts[ff.part_base + pi];
        if (p.sky_slot < 0) continue;
        for (int kind = 0; kind < 2; kind++) {
            if (!(p.flags & (kind ? FEP_CEIL_SKY : FEP_FLOOR_SKY))) continue;
            const uint64_t *add = fe_event_words(P, f, p.sky_slot, kind), *flush = fe_event_words(P, f, p.sky_slot, 2);
            for (int x = p.sx + tid; x <= p.ex; x += FE_FIN_THREADS) {
                if (!fe_gap(add, flush, x, p.sx, p.ex)) continue;
                const uint32_t slot = atomicAdd(&cnt[x], 1u);
                if (slot >= P.col_slots) { atomicOr(&P.flags[f], (uint32_t)FE_OVF_SPANS); continue; }
                const DevRSpan r = resolve_sky_span(fe_span(0, 0, 0, 0, SPAN_SKY, x), P.scene, P.k, fr);
                const size_t i = ((size_t)f * P.col_slots + slot) * (size_t)W + (size_t)x;
                P.keys[i] = FE_KEY_PLANE | (pi << 2) | (uint32_t)kind;
                P.sspans[2 * i] = FeU4{r.w[0], r.w[1], r.w[2], r.w[3]};
                P.sspans[2 * i + 1] = FeU4{r.w[4], r.w[5], r.w[6], r.w[7]};
            }
        }
    }
    __threadfence_block();
    __syncthreads();

    // 2. exclusive scan of the column counts -> col_off (each thread owns a contiguous chunk of columns)
    const int chunk = (W + FE_FIN_THREADS - 1) / FE_FIN_THREADS;
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
        for (int w = 0; w < FE_FIN_THREADS / 64; w++) { const uint32_t v = wave_sum[w]; wave_sum[w] = run; run += v; }
        total_s = run;
    }
    __syncthreads();
    const uint32_t total = total_s;
    const bool fits = total <= P.span_stride;
    if (!fits && tid == 0) atomicOr(&P.flags[f], (uint32_t)FE_OVF_FRAME);
    uint32_t off = wave_sum[tid >> 6] + incl - mine;
    uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    if (tid == 0) { coff[W] = fits ? total : 0u; P.totals[f] = total; }

    // 3. every column's spans in draw order: rank by key (keys of one column are distinct), scatter
    FeU4 *out = reinterpret_cast<FeU4 *>(P.rspans + fr.span_base);
    for (int x = xa; x < xb; x++) {
        const uint32_t n = min(cnt[x], P.col_slots);
        coff[x] = fits ? off : 0u;
        if (fits) {
            for (uint32_t i = 0; i < n; i++) {
                const size_t si = ((size_t)f * P.col_slots + i) * (size_t)W + (size_t)x;
                const uint32_t key = P.keys[si];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < n; j++) {
                    const uint32_t kj = P.keys[((size_t)f * P.col_slots + j) * (size_t)W + (size_t)x];
                    rank += (kj < key || (kj == key && j < i)) ? 1u : 0u;          // the tie-break keeps the scatter a permutation
                }
                out[2 * (size_t)(off + rank)] = P.sspans[2 * si];
                out[2 * (size_t)(off + rank) + 1] = P.sspans[2 * si + 1];
            }
        }
        off += n;
    }
}

}  // namespace

hipError_t launch_fe(const FeParams &P, hipStream_t stream) {
    if (P.n_frames <= 0) return hipSuccess;
    dim3 grid((unsigned)((P.k.W + FE_COL_THREADS - 1) / FE_COL_THREADS), (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_fe_columns, grid, dim3(FE_COL_THREADS), 0, stream, P);
    hipLaunchKernelGGL(dg_fe_finalize, dim3((unsigned)P.n_frames), dim3(FE_FIN_THREADS), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dg
