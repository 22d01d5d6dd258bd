// kernels.hip — gfx950 (MI355X / CDNA4) kernels of the rasteriser.  wave64, no MFMA (nothing here is a
// contraction): the work is per-pixel f32 evaluation + byte gathers, and the output is a stream of RGB24 bytes.
//
// Kernel 1  dg_setup_spans   one lane per span: column-invariant part of the wall/sprite mapper
//                            (bitmap_render.rs:241-251: 3 f32 divides -> texture column + light factor),
//                            sky texture column, floor/ceiling vx; merges the span with its record into one
//                            self-contained 32-byte DevRSpan.
// Kernel 2  dg_raster_tiles  one workgroup (8 wavefronts) per (frame, 64-column x 64-row tile):
//                              * the spans of the tile's 64 columns are ONE contiguous range of the column-major span
//                                array; they are staged in LDS with a single coalesced burst (16 KB), together with the
//                                palette (1 KB) and the tile's column offsets;
//                              * a wavefront owns eight of the tile's columns.  One pre-filter pass with lane = (column,
//                                span slot) finds, for all eight at once, the spans that touch the tile's rows; then the
//                                wave takes one column at a time with lane = row.  Pass 1 walks the column's touching
//                                spans in draw order and records per row the last span covering it (one v_readlane per
//                                span); pass 2 evaluates every row ONCE with its winner's parameters fetched from LDS —
//                                exact last-writer-wins, no overdraw evaluation, one pass per span kind;
//                              * a wall column reads one texture column ([x][y] texel layout => consecutive bytes);
//                              * finished pixels go to an LDS tile [row][col] and leave the CU as fully
//                                coalesced 12-byte-per-lane RGB24 row segments (192 B contiguous per tile row).
//                            Every pixel of the tile is stored (uncovered = 0,0,0) which fuses the reference's
//                            per-frame `Pixels::new()` clear (pixels.rs:10-14) into the one write pass.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (contraction would break bit-exactness; the IEEE
// divide expansion keeps its own internal FMAs, which is what makes it correctly rounded).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "kernels.hpp"
#include "raster_core.h"
#include "strip_core.h"

namespace dg {

constexpr int TILE_W = 64;        // columns per workgroup
constexpr int TILE_H = 64;        // rows per workgroup = lanes per wave
constexpr int TILE_STRIDE = 68;   // dwords per LDS tile row (16-B aligned rows for the b128 read-out)
constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;
constexpr int SPAN_CAP = 512;     // spans of one tile's 64 columns staged in LDS (16 KB); typical tiles hold 100-400

__global__ __launch_bounds__(256) void dg_setup_spans(RasterParams P) {
    const int f = blockIdx.y;
    const DevFrame fr = P.frames[f];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fr.n_spans) return;
    const DevSpan sp = P.spans[fr.span_base + i];
    DevRSpan o;
    if (sp.kind == SPAN_WALL) o = resolve_wall_span(sp, P.walls[fr.wall_base + sp.rec]);
    else if (sp.kind == SPAN_FLAT) o = resolve_flat_span(sp, P.planes[fr.plane_base + sp.rec], P.k);
    else o = resolve_sky_span(sp, P.scene, P.k, fr);
    uint4 *dst = reinterpret_cast<uint4 *>(&P.rspans[fr.span_base + i]);
    dst[0] = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
    dst[1] = make_uint4(o.w[4], o.w[5], o.w[6], o.w[7]);
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// Per-pixel evaluation with per-lane span words (the lanes of one wave may own pixels of different spans).
__device__ __forceinline__ uint32_t eval_wall(const RasterParams &P, const uint32_t *pal, const uint4 a, const uint4 b, int y, bool &opaque) {
    const uint32_t o = wall_texel_offset_staged(a.y, a.z, b.x, b.y, b.z, b.w, y);
    opaque = w0_immediate(a.x) ? P.scene.texel_opq[o] != 0 : true;
    return shade(pal[P.scene.texel_idx[o]], bits_f32(a.w));
}
__device__ __forceinline__ uint32_t eval_flat(const RasterParams &P, const DevFrame &fr, const uint32_t *pal, const uint4 a, const uint4 b,
                                              float vy, float r_vy) {
    float factor;
    const uint32_t o = flat_texel_offset(fr, a.y, a.z, b.x, b.y, b.z, vy, r_vy, factor);
    return shade(pal[P.scene.flats[o]], factor);
}

// One span of the column meets this wave's 64 rows (pass 1).  Every row remembers the LAST opaque span that covers it
// ("winner").  Spans that may be transparent (masked walls, sprites, sky with holes: the immediate flag) are evaluated on
// the spot because whether they overwrite depends on the texel.  w0 and idx are wave-uniform.
__device__ __forceinline__ void span_step(const RasterParams &P, const uint32_t *pal, const uint4 *lsp, uint32_t w0, uint32_t idx, int y, int srow,
                                          uint32_t &color, uint32_t &winner) {
    const bool in = (uint32_t)(y - w0_ctop(w0)) <= (uint32_t)(w0_cbot(w0) - w0_ctop(w0));
    if (!w0_immediate(w0)) {
        if (in) winner = idx;
    } else if (in) {
        const uint4 a = lsp[2 * idx], b = lsp[2 * idx + 1];   // same address in every lane: LDS broadcast
        if (w0_kind(w0) == SPAN_WALL) {
            bool opaque;
            const uint32_t c = eval_wall(P, pal, a, b, y, opaque);
            if (opaque) { color = c; winner = 0xffffffffu; }
        } else {
            const uint32_t o = sky_texel_offset(a.z, srow);
            if (o != 0xffffffffu && P.scene.texel_opq[o]) { color = pal[P.scene.texel_idx[o]]; winner = 0xffffffffu; }
        }
    }
}

// Pass 2: each row fetches its winner's 8 words from LDS and is evaluated once — one pass per span KIND present, with
// per-lane parameters, instead of one pass per span; overdrawn pixels are never evaluated.
__device__ __forceinline__ uint32_t shade_winner(const RasterParams &P, const DevFrame &fr, const uint32_t *pal, const uint4 *lsp, uint32_t winner,
                                                 uint32_t color, int y, float vy, float r_vy, int srow) {
    if (winner != 0xffffffffu) {
        const uint4 a = lsp[2 * winner], b = lsp[2 * winner + 1];
        const uint32_t kind = w0_kind(a.x);
        if (kind == SPAN_FLAT) {
            color = eval_flat(P, fr, pal, a, b, vy, r_vy);
        } else if (kind == SPAN_WALL) {
            bool opaque;
            color = eval_wall(P, pal, a, b, y, opaque);
        } else {                                  // sky bitmap without holes: plain lookup, no lighting
            const uint32_t o = sky_texel_offset(a.z, srow);
            if (o != 0xffffffffu) color = pal[P.scene.texel_idx[o]];
        }
    }
    return color;
}

// One screen column (64 rows of it) for one wavefront, any number of spans (lw0 = word 0 of every span, lsp = all 8 words):
// lane i looks at span i, a ballot picks the spans touching these rows, one v_readlane per such span.
__device__ __forceinline__ uint32_t raster_column(const RasterParams &P, const DevFrame &fr, const uint32_t *pal, const uint32_t *lw0,
                                                  const uint4 *lsp, uint32_t n, int lane, int y, int y0, float vy, float r_vy, int srow) {
    uint32_t color = 0;
    uint32_t winner = 0xffffffffu;            // index (within the column) of the opaque span owning this row
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        uint32_t w0v = 0;
        bool hit = false;
        if (i < n) {
            w0v = lw0[i];
            hit = w0_cbot(w0v) >= y0 && w0_ctop(w0v) <= y0 + (TILE_H - 1);
        }
        unsigned long long m = __ballot(hit);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            span_step(P, pal, lsp, bcast(w0v, j), base + (uint32_t)j, y, srow, color, winner);
        }
    }
    return shade_winner(P, fr, pal, lsp, winner, color, y, vy, r_vy, srow);
}

// The same for a column with at most 8 spans whose row filter was done by the wave-level pre-filter (dg_raster_tiles):
// hm = bit j set when span j touches these rows, its word 0 sits in lane `lane0 + j` of w0f.
__device__ __forceinline__ uint32_t raster_column_small(const RasterParams &P, const DevFrame &fr, const uint32_t *pal, uint32_t hm, uint32_t w0f,
                                                        int lane0, const uint4 *lsp, int y, float vy, float r_vy, int srow) {
    uint32_t color = 0;
    uint32_t winner = 0xffffffffu;
    while (hm) {
        const int j = __builtin_ctz(hm);
        hm &= hm - 1;
        span_step(P, pal, lsp, bcast(w0f, lane0 + j), (uint32_t)j, y, srow, color, winner);
    }
    return shade_winner(P, fr, pal, lsp, winner, color, y, vy, r_vy, srow);
}

struct TileLds {
    uint32_t tile[TILE_H * TILE_STRIDE];
    uint4 lspans[SPAN_CAP * 2];
    uint32_t lw0[SPAN_CAP];
    uint32_t pal[256];
    uint32_t lcoff[TILE_W + 1];
};

// One 64 x 64 tile of frame f: columns x0 .., rows y0 ..
__device__ __forceinline__ void tile_body(const RasterParams &P, TileLds &L, int f, int x0, int y0) {
    const DevFrame fr = P.frames[f];
    const int W = P.k.W, H = P.k.H;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = y0 + lane;
    const float vy = P.k.CFY - (float)y;      // visplanes.rs:109, a per-row constant
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const uint4 *gspans = reinterpret_cast<const uint4 *>(P.rspans + fr.span_base);

    // the three prologue loads are issued together (addresses clamped instead of branching around the loads)
    const uint4 rt = P.row_tab[y < H ? y : H - 1];                    // prepared reciprocal of vy and the sky row (dg_row_table)
    const uint32_t pal_v = P.scene.palette[threadIdx.x & 255];
    const int xc = x0 + (int)(threadIdx.x <= TILE_W ? threadIdx.x : 0);
    const uint32_t coff_v = coff[xc < W ? xc : W];
    const float r_vy = bits_f32(rt.x);
    const int srow = (int)rt.y;
    if (threadIdx.x < 256) L.pal[threadIdx.x] = pal_v;
    if (threadIdx.x <= TILE_W) L.lcoff[threadIdx.x] = coff_v;
    __syncthreads();

    // The spans of adjacent columns are one contiguous range of the column-major span array.  Stage as many whole columns
    // as fit in LDS (normally the whole tile) with one coalesced burst — every load of the workgroup in flight at once, so
    // the dependent chain col_off -> spans is paid once per tile — then rasterise those columns; repeat if needed.
    // Wall spans are put into their per-pixel form on the way (stage_wall_span: texture column start, prepared 1/d).
    int c_lo = 0;
    while (c_lo < TILE_W) {
        const uint32_t t0 = L.lcoff[c_lo];
        int c_hi = TILE_W;
        if (L.lcoff[TILE_W] - t0 > SPAN_CAP) {
            c_hi = c_lo + 1;                  // a single column always fits: the binner caps a column at SPAN_CAP spans
            while (c_hi < TILE_W && L.lcoff[c_hi + 1] - t0 <= SPAN_CAP) c_hi++;
        }
        const uint32_t n_stage = L.lcoff[c_hi] - t0;
        for (uint32_t i = threadIdx.x; i < n_stage; i += THREADS) {
            uint4 a = gspans[2 * ((size_t)t0 + i)], b = gspans[2 * ((size_t)t0 + i) + 1];
            if (w0_kind(a.x) == SPAN_WALL) stage_wall_span(a.y, a.z, b.z, b.w);
            L.lspans[2 * i] = a;
            L.lspans[2 * i + 1] = b;
            L.lw0[i] = a.x;
        }
        __syncthreads();
        // Wave-level pre-filter: this wave owns columns c_lo + wave + 8k (k = 0..7).  Lane (k, slot) = (lane >> 3, lane & 7)
        // tests span `slot` of column k against the tile's rows, so ONE pass filters all eight columns (columns with more
        // than 8 spans take the general path).
        const int fk = lane >> 3, fslot = lane & 7;
        const int fcol = c_lo + wave + WAVES * fk;
        uint32_t f_n0 = 0, f_n = 0, f_w0 = 0;
        bool f_hit = false;
        if (fcol < c_hi) {
            f_n0 = L.lcoff[fcol] - t0;
            f_n = L.lcoff[fcol + 1] - L.lcoff[fcol];
            if ((uint32_t)fslot < f_n && f_n <= 8u) {
                f_w0 = L.lw0[f_n0 + (uint32_t)fslot];
                f_hit = w0_cbot(f_w0) >= y0 && w0_ctop(f_w0) <= y0 + (TILE_H - 1);
            }
        }
        const unsigned long long hitm = __ballot(f_hit);
        int k8 = 0;
        for (int c = c_lo + wave; c < c_hi; c += WAVES, k8 += 8) {
            const uint32_t n0 = bcast(f_n0, k8), n = bcast(f_n, k8);
            uint32_t px;
            if (n > 8u) px = raster_column(P, fr, L.pal, L.lw0 + n0, L.lspans + 2 * n0, n, lane, y, y0, vy, r_vy, srow);
            else px = raster_column_small(P, fr, L.pal, (uint32_t)(hitm >> k8) & 0xffu, f_w0, k8, L.lspans + 2 * n0, y, vy, r_vy, srow);
            L.tile[lane * TILE_STRIDE + c] = px;
        }

        c_lo = c_hi;
        if (c_lo < TILE_W) __syncthreads();   // before the staging area is reused
    }
    __syncthreads();

    // Read-out: groups of 4 pixels (16 B of RGBX in LDS -> 12 B of RGB24 in HBM); 16 groups per tile row, so the lanes of
    // a wave cover four tile rows = 4 x 192 contiguous bytes.
    uint8_t *fb = P.fb + (size_t)f * (size_t)3 * (size_t)W * (size_t)H;
    for (int g = threadIdx.x; g < TILE_H * (TILE_W / 4); g += THREADS) {
        const int row = g >> 4, gc = g & 15;
        const int yy = y0 + row, xx = x0 + 4 * gc;
        if (yy < H && xx < W) {   // W % 4 == 0 (checked at dg_create), so a group never straddles the right edge
            const uint4 p = *reinterpret_cast<const uint4 *>(&L.tile[row * TILE_STRIDE + 4 * gc]);
            const uint32_t o0 = (p.x & 0xffffffu) | (p.y << 24);
            const uint32_t o1 = ((p.y >> 8) & 0xffffu) | (p.z << 16);
            const uint32_t o2 = ((p.z >> 16) & 0xffu) | (p.w << 8);
            uint32_t *dst = reinterpret_cast<uint32_t *>(fb + ((size_t)yy * (size_t)W + (size_t)xx) * 3);
            dst[0] = o0;
            dst[1] = o1;
            dst[2] = o2;
        }
    }
}

// Every tile of every frame (the strip path is off, or a batch is redone because a column exceeded the segment slots).
__global__ __launch_bounds__(THREADS) void dg_raster_tiles(RasterParams P) {
    __shared__ __attribute__((aligned(16))) TileLds L;
    tile_body(P, L, (int)blockIdx.z, (int)blockIdx.x * TILE_W, (int)blockIdx.y * TILE_H);
}

// The tiles dg_resolve_columns listed: those that a possibly-transparent span (masked wall, sprite) touches, where the
// winner of a pixel depends on texels and dg_raster_strips therefore does not go.  A fixed number of persistent workgroups
// pulls tiles off the list, so the launch costs the same whether the list holds 50 tiles or 50 000.  It runs on its own
// stream beside dg_raster_strips: the two kernels stress different parts of a CU (LDS / barriers / ownership walk here,
// VALU there) and fill each other's gaps.
__global__ __launch_bounds__(THREADS) void dg_raster_tile_list(RasterParams P) {
    __shared__ __attribute__((aligned(16))) TileLds L;
    __shared__ uint32_t next_item;
    const uint32_t count = P.tile_counters[0];
    uint32_t item = blockIdx.x;                       // the first tile is free; further ones come off a shared counter that starts at gridDim.x
    while (item < count) {
        const uint32_t t = P.tile_list[item];
        if (threadIdx.x == 0) next_item = gridDim.x + atomicAdd(&P.tile_counters[1], 1u);   // requested now, read after the tile: the round trip is hidden
        tile_body(P, L, (int)(t >> 16), (int)(t & 0xffu) * TILE_W, (int)((t >> 8) & 0xffu) * TILE_H);
        __syncthreads();                              // the tile's read-out is done with L; next_item is visible
        item = next_item;
        __syncthreads();
    }
}

// ---- strip path ---------------------------------------------------------------------------------------------------------

#ifndef DG_STRIPS_MIN_WAVES
#define DG_STRIPS_MIN_WAVES 1
#endif
constexpr int RES_STAGE = 24;     // row-range words of a column staged in LDS by dg_resolve_columns (longer columns read HBM)

__device__ __forceinline__ void resolve_lane(const RasterParams &P, int f, int x, int lane, uint32_t *lw0, uint32_t (*lbands)[8]) {
    const int W = P.k.W, H = P.k.H;
    const DevFrame fr = P.frames[f];
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const uint32_t o = coff[x], n = coff[x + 1] - o;
    const DevRSpan *spans = P.rspans + fr.span_base + o;
    const uint32_t ns = min(n, (uint32_t)RES_STAGE);
    for (uint32_t j = 0; j < ns; j += 4) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) v[q] = spans[min(j + q, ns - 1)].w[0];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) if (j + q < ns) lw0[(j + q) * 64 + (uint32_t)lane] = v[q];
    }
    auto w0_at = [&](uint32_t j) { return j < (uint32_t)RES_STAGE ? lw0[j * 64 + (uint32_t)lane] : spans[j].w[0]; };
    const ResolveResult r = resolve_column(w0_at, spans, n, P.scene, H, P.band_rows, (uint32_t)P.seg_cap,
                                           P.segs + (size_t)f * (size_t)P.seg_cap * (size_t)W + (size_t)x, (size_t)W,
                                           P.band_first + (size_t)f * (size_t)P.n_bands * (size_t)W + (size_t)x, (size_t)W);
    if (r.n_segs == 0xffffffffu) atomicOr(&P.frame_flags[f], 1u);
    // Overlay spans: the bands (= 64-row tiles of this strip) they touch are rendered by dg_raster_tile_list from the draw-ordered
    // spans instead.  Every wave collects its strip's bands in LDS; lane 0 then flags them and appends them to the tile list.
    for (uint32_t j = r.n_base; j < n; j++) {
        const uint32_t w0 = w0_at(j);
        for (int b = w0_ctop(w0) / P.band_rows; b <= w0_cbot(w0) / P.band_rows; b++) atomicOr(&lbands[threadIdx.x >> 6][b >> 5], 1u << (b & 31));
    }
}

// One lane per (frame, screen column): strip_core.h resolve_column.  The row-range words (w0) of the column's spans are
// staged in LDS first (independent loads, all in flight at once), then the scan over boundaries x spans runs out of LDS.
// Real columns hold 2-8 spans.  Negligible next to the raster kernels (320 000 columns per launch against 256 M pixels).
__global__ __launch_bounds__(256) void dg_resolve_columns(RasterParams P) {
    __shared__ uint32_t lw0_all[4][RES_STAGE * 64];
    __shared__ uint32_t lbands[4][8];                 // per wave: bands (up to 256) touched by overlay spans of its strip
    if (threadIdx.x < 32) lbands[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    __syncthreads();
    const int f = blockIdx.y;
    const int W = P.k.W;
    const int lane = threadIdx.x & 63;
    const uint32_t strip = blockIdx.x * 4 + (threadIdx.x >> 6), n_strips = (uint32_t)(W + 63) / 64;
    uint32_t *lw0 = lw0_all[threadIdx.x >> 6];
    const int x = (int)strip * 64 + lane;
    if (x < W) resolve_lane(P, f, x, lane, lw0, lbands);
    __syncthreads();
    // the wave's bands -> flags + tile list: one atomic for the wave, then lane l appends band l (64, 128, 192 + l) at its rank
    if (strip < n_strips) {
        const uint32_t *mw = lbands[threadIdx.x >> 6];
        uint32_t total = 0;
        for (uint32_t g = 0; g < 8; g++) total += (uint32_t)__builtin_popcount(mw[g]);
        if (total) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&P.tile_counters[0], total);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            uint32_t before = 0;                      // bands below this 64-band group
            for (uint32_t g2 = 0; g2 < 4; g2++) {
                const unsigned long long m64 = (unsigned long long)mw[2 * g2] | ((unsigned long long)mw[2 * g2 + 1] << 32);
                if ((m64 >> lane) & 1ull) {
                    const uint32_t b = g2 * 64 + (uint32_t)lane;
                    const uint32_t rank = before + (uint32_t)__builtin_popcountll(m64 & ((1ull << lane) - 1ull));
                    P.band_ovl[((size_t)f * (size_t)P.n_bands + b) * (size_t)n_strips + strip] = 1;
                    P.tile_list[base + rank] = ((uint32_t)f << 16) | (b << 8) | strip;
                }
                before += (uint32_t)__builtin_popcountll(m64);
            }
        }
    }
}

// What a lane of dg_raster_strips keeps about its column's current segment: the DevSeg words it needs per pixel, unpacked
// once when the column enters the segment.  q0..q2 mean different things for walls and flats (never both at once).
enum : uint32_t { CLS_NONE = 0, CLS_FLAT = 1, CLS_WALL = 2, CLS_SKY = 3, CLS_FLAT_SLOW = 4, CLS_WALL_MOD = 5 };
struct SegRegs {
    int32_t end;                  // last row of the segment
    uint32_t cls;                 // CLS_*: kind + which mapper variant is exact for it
    uint32_t base;                // w2: pool offset (flat / bitmap row 0 + tx / sky row 0 + tx)
    float fac;                    // w3: light factor of a wall column; 1 for sky, 0 for "nothing drawn"
    float q0, q1, q2;             // WALL: d, uy1, prepared 1/d          FLAT: wz*vx, gwz, light/255
    int32_t top_y, off_y;         // WALL
    float hf;                     // WALL: h as f32
    int32_t hmask;                // WALL: h - 1 (CLS_WALL: h is a power of two > 1) or h (CLS_WALL_MOD)
    uint32_t wst;                 // WALL: bitmap width = row stride in the pool
};
__device__ __forceinline__ void seg_unpack(const uint4 a, const uint4 b, SegRegs &s) {
    const uint32_t kind = seg_kind(a.x);
    s.end = seg_end(a.x); s.base = a.z; s.fac = bits_f32(a.w);
    s.q0 = bits_f32(a.y); s.q1 = bits_f32(b.x);
    const int32_t h = (int32_t)(b.z & 0xffffu);
    const bool pow2 = h > 1 && (h & (h - 1)) == 0;
    s.top_y = lo_i16(b.y); s.off_y = hi_i16(b.y); s.hf = (float)h; s.wst = b.z >> 16;
    s.hmask = pow2 ? h - 1 : h;
    if (kind == SPAN_WALL) { s.q2 = bits_f32(b.w); s.cls = pow2 ? CLS_WALL : CLS_WALL_MOD; }
    else if (kind == SPAN_FLAT) { s.q2 = bits_f32(b.y); s.cls = (b.z & 1u) ? CLS_FLAT : CLS_FLAT_SLOW; }
    else { s.q2 = 0.0f; s.cls = kind == SPAN_SKY ? CLS_SKY : CLS_NONE; }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(4))) *RowTabPtr;   // "constant" address space: wave-uniform reads become scalar loads

// One wavefront per (frame, 64-column strip, band of rows), lane = column.  Every lane keeps its column's current segment in
// registers (and the next one, prefetched), so a pixel costs its texture mapper and nothing else: no ownership test, no
// per-pixel parameter fetch, no LDS tile.  The row is wave-uniform; its constants (vy, the prepared 1/vy, the sky row) arrive
// by scalar loads.  When all 64 columns are inside floors / ceilings, or all inside walls — which is what most rows of most
// strips look like — the row runs a straight-line mapper without any per-lane kind test (`mode` changes only on rows where
// some column enters its next segment).  Two rows are in flight: the texel of row y + 1 is requested before row y is shaded.
// Finished rows are packed to RGB24 by quads of lanes (one DPP move, one byte permute), parked in LDS and leave four at a
// time, 16 contiguous bytes per lane.  Texels are row-major here (pool), so the 64 adjacent columns of a wall row read a
// handful of cache lines.
// Bands (= 64-row tiles of the strip) that a possibly-transparent span touches are not rendered here but by
// dg_raster_tile_list, from the draw-ordered spans (band_ovl).
__device__ __forceinline__ void strips_body(const RasterParams &P, int band, const float4 *palf, uint32_t *rowbuf) {
    const int f = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int W = P.k.W, H = P.k.H;
    const int x0 = (int)blockIdx.x * 64;
    const int y_lo = band * P.band_rows;
    const int y_hi = min(H, y_lo + P.band_rows) - 1;
    const DevFrame fr = P.frames[f];
    const bool in_w = x0 + lane < W;
    const int x = in_w ? x0 + lane : W - 1;           // lanes past the right edge shadow the last column and store nothing
    const uint32_t s0 = P.band_first[((size_t)f * (size_t)P.n_bands + (size_t)band) * (size_t)W + (size_t)x];
    const uint8_t *segs_f = reinterpret_cast<const uint8_t *>(P.segs + (size_t)f * (size_t)P.seg_cap * (size_t)W);   // wave-uniform
    const uint32_t seg_step = (uint32_t)W * 32u;
    uint32_t seg_at = (s0 * (uint32_t)W + (uint32_t)x) * 32u;       // byte offset of the lane's NEXT segment
    SegRegs S;
    uint4 na, nb;
    {
        const uint4 ca = *reinterpret_cast<const uint4 *>(segs_f + seg_at), cb = *reinterpret_cast<const uint4 *>(segs_f + seg_at + 16);
        seg_unpack(ca, cb, S);
        na = ca; nb = cb;
        seg_at += seg_step;
        if (S.end < H - 1) { na = *reinterpret_cast<const uint4 *>(segs_f + seg_at); nb = *reinterpret_cast<const uint4 *>(segs_f + seg_at + 16); }
    }
    const uint8_t *pool = P.scene.pool;
    const uint32_t sky_w = (uint32_t)P.scene.sky_w;
    const RowTabPtr rows = (RowTabPtr)(uintptr_t)P.row_tab;
    // lane 4q + j (j < 3) stores dword j of its quad's 12 bytes; selector of v_perm_b32 over {next pixel, own pixel}
    const uint32_t perm_sel = (lane & 3) == 0 ? 0x04020100u : (lane & 3) == 1 ? 0x05040201u : 0x06050402u;
    // Rows leave four at a time: lane 4q + j (j < 3) parks dword 3q + j of its row in LDS; then lane L < 48 stores 16
    // contiguous bytes (chunk L % 12 of row L / 12) — one vector-memory instruction per four rows instead of four, which
    // matters because the texture/store address unit takes ~16 clocks per wave instruction whatever its width.
    const uint32_t park_at = (lane & 3) != 3 ? (uint32_t)(lane - (lane >> 2)) : 4u * 48u;   // lanes 4q + 3 have no dword of their own: they write
    const uint32_t park_mul = (lane & 3) != 3 ? 48u : 0u;                                    // one dump slot behind the four rows (never read)
    const int st_row = lane / 12, st_chunk = lane % 12;
    const uint32_t row_bytes = (uint32_t)min(64, W - x0) * 3u;         // of this strip (a multiple of 12: W % 4 == 0)
    const bool st_lane = lane < 48 && (uint32_t)st_chunk * 16u + 16u <= row_bytes;          // the whole 16-byte chunk lies inside the row
    const bool st_part = lane < 48 && !st_lane && (uint32_t)st_chunk * 16u < row_bytes;     // the row ends inside it (last strip of a frame whose width is not a multiple of 64)
    const uint32_t st_off = (uint32_t)st_row * (uint32_t)W * 3u + (uint32_t)st_chunk * 16u;
    uint8_t *rowp = P.fb + (((size_t)f * (size_t)H + (size_t)y_lo) * (size_t)W + (size_t)x0) * 3;
    int parked = 0;                                   // rows in rowbuf

    // Which mapper the next rows run (wave-uniform; recomputed only on rows where a column changes segment).
    enum { MODE_GENERIC = 0, MODE_FLAT = 1, MODE_WALL = 2, MODE_MIXED = 3 };
    auto classify = [&]() {
        if (__builtin_amdgcn_ballot_w64(S.cls == CLS_FLAT) == ~0ull) return (int)MODE_FLAT;
        if (__builtin_amdgcn_ballot_w64(S.cls == CLS_WALL) == ~0ull) return (int)MODE_WALL;
        return __builtin_amdgcn_ballot_w64(S.cls >= CLS_FLAT_SLOW) == 0ull ? (int)MODE_MIXED : (int)MODE_GENERIC;
    };
    int mode = classify();

    // What two rows have in flight between their two halves.  The loads are issued with inline assembly and waited for with
    // an explicit s_waitcnt in pair_b: hipcc's own wait insertion would also wait for the row buffer's STORE before it lets
    // the texels be used, which serialises the rows.
    struct Pair { uint32_t tex0, tex1; float fac0, fac1; };

    // visplanes.rs:108-126 for a floor / ceiling pixel within the divide shortcut's verified domain
    auto flat_px = [&](float vy, float r_vy, float &fac) {
        const float wx = div_prepared(S.q1, vy, r_vy), wy = div_prepared(S.q0, vy, r_vy);
        const float rx = wx * fr.cos_a - wy * fr.sin_a;
        const float ry = wy * fr.cos_a + wx * fr.sin_a;
        const int32_t tx = (f32_as_i16(rx) + fr.pos_x_i16) & 63;
        const int32_t ty = (f32_as_i16(ry) + fr.pos_y_i16) & 63;
        fac = __builtin_fmaf(-(float)f32_as_i16(wx), 1.0f / (16.0f * 256.0f), S.q2);   // strip_core.h seg_flat_offset: exact product
        return S.base + (uint32_t)(ty * 64 + tx);
    };
    // bitmap_render.rs:256-263 for a wall pixel whose bitmap height is a power of two (mask instead of modulus)
    auto wall_px = [&](int y) {
        const float ay = div_prepared((float)(y - S.top_y), S.q0, S.q2);
        const int32_t ty = (f32_as_i16(S.hf + ay * S.q1) + S.off_y) & S.hmask;
        return S.base + (uint32_t)ty * S.wst;
    };
    // columns whose segment ended before row y move to their next one (wave-uniform test: most rows skip all of this)
    auto advance = [&](int y) {
        if (__builtin_amdgcn_ballot_w64(y > S.end) != 0ull) {
            if (y > S.end) {
                seg_unpack(na, nb, S);
                if (S.end < H - 1) {
                    seg_at += seg_step;
                    na = *reinterpret_cast<const uint4 *>(segs_f + seg_at); nb = *reinterpret_cast<const uint4 *>(segs_f + seg_at + 16);
                }
            }
            mode = classify();
        }
    };
    // the texture mapper of row y for lanes of any class -> pool offset of the texel, light factor
    auto mixed_px = [&](int y, const u32x4 rc, float &fac) {
        const float r_vy = bits_f32(rc.x), vy = bits_f32(rc.z);
        uint32_t off = S.base;                        // CLS_NONE: offset 0, factor 0 -> black
        fac = S.fac;
        if (mode == MODE_MIXED) {                     // floors / ceilings, power-of-two walls, sky, nothing: one masked pass each
            if (S.cls == CLS_FLAT) off = flat_px(vy, r_vy, fac);
            if (S.cls == CLS_WALL) off = wall_px(y);
        } else {
            if (S.cls == CLS_FLAT || S.cls == CLS_FLAT_SLOW)
                off = seg_flat_offset(fr, f32_bits(S.q0), S.base, f32_bits(S.q1), f32_bits(S.q2), S.cls == CLS_FLAT ? 1u : 0u, vy, r_vy, fac);
            else if (S.cls == CLS_WALL || S.cls == CLS_WALL_MOD) {
                const int32_t h = (int32_t)S.hf;
                off = S.base + (uint32_t)wall_texel_row(S.q0, S.q2, S.q1, (uint32_t)(uint16_t)S.top_y | ((uint32_t)(uint16_t)S.off_y << 16), h, y) * S.wst;
            }
        }
        if (S.cls == CLS_SKY) {
            const int srow = (int)rc.y;
            if (srow >= 0) off = S.base + (uint32_t)srow * sky_w;
            else { off = 0; fac = 0.0f; }             // row outside the sky bitmap: nothing is drawn
        }
        return off;
    };

    // Rows y0 and y0 + 1, first half: segment changes, then the texture mappers -> the two texel loads are issued and NOT
    // waited for.  When no column changes segment between the two rows (the usual case) both rows run one straight-line
    // mapper on the same per-lane constants, which gives the scheduler two independent dependency chains to interleave.
    auto pair_a = [&](int y0, Pair &R) {
        const int y1 = min(y0 + 1, y_hi);
        const u32x4 rc0 = rows[y0], rc1 = rows[y1];   // scalar loads: prepared 1/vy, sky row, vy (dg_row_table)
        advance(y0);
        uint32_t off0, off1;
        if (__builtin_amdgcn_ballot_w64(y1 > S.end) == 0ull) {
            if (mode == MODE_FLAT) {
                off0 = flat_px(bits_f32(rc0.z), bits_f32(rc0.x), R.fac0);
                off1 = flat_px(bits_f32(rc1.z), bits_f32(rc1.x), R.fac1);
            } else if (mode == MODE_WALL) {
                R.fac0 = R.fac1 = S.fac;
                off0 = wall_px(y0);
                off1 = wall_px(y1);
            } else {
                off0 = mixed_px(y0, rc0, R.fac0);
                off1 = mixed_px(y1, rc1, R.fac1);
            }
        } else {                                      // a segment ends on row y0: the second row runs on the next segment's constants
            off0 = mixed_px(y0, rc0, R.fac0);
            advance(y1);
            off1 = mixed_px(y1, rc1, R.fac1);
        }
#ifdef DG_EXP_NOLOAD
        R.tex0 = off0 & 255u; R.tex1 = off1 & 255u;
#else
        asm volatile("global_load_ubyte %0, %1, %2" : "=v"(R.tex0) : "v"(off0), "s"(pool) : "memory");
        asm volatile("global_load_ubyte %0, %1, %2" : "=v"(R.tex1) : "v"(off1), "s"(pool) : "memory");
#endif
    };
    // `n` parked rows (192 bytes each) -> HBM.
    auto flush_rows = [&](int n) {
#ifdef DG_EXP_NOSTORE
        if (st_lane && st_row < n && rowbuf[0] == 0x12345678u) {
#else
        if (st_lane && st_row < n) {
#endif
            const u32x4 v = *reinterpret_cast<const u32x4 *>(&rowbuf[st_row * 48 + st_chunk * 4]);
            asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(st_off), "v"(v), "s"(rowp) : "memory");
        }
        if (st_part && st_row < n) {
            for (uint32_t i = 0; i < 4 && (uint32_t)st_chunk * 16u + 4u * i < row_bytes; i++)
                *reinterpret_cast<uint32_t *>(rowp + st_off + 4u * i) = rowbuf[st_row * 48 + st_chunk * 4 + (int)i];
        }
        rowp += (size_t)n * (size_t)W * 3;
        parked = 0;
    };
    // palette, lighting (bitmap_render.rs:202-207), RGB24 packing of one pixel per lane -> this lane's dword of the packed row
    auto shade_pack = [&](uint32_t tex, float fac) {
#ifdef DG_EXP_NOPAL
        const float4 c = make_float4((float)tex, (float)(tex >> 1), (float)(tex >> 2), 0.0f);
#else
        const float4 c = palf[tex];
#endif
        uint32_t px;
        const float r = __builtin_truncf(c.x * fac), g = __builtin_truncf(c.y * fac), b = __builtin_truncf(c.z * fac);
        asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(px) : "v"(r));
        asm("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(px) : "v"(g), "v"(px));
        asm("v_cvt_pk_u8_f32 %0, %1, 2, %2" : "=v"(px) : "v"(b), "v"(px));
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)px, (int)px, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
        return __builtin_amdgcn_perm(nx, px, perm_sel);
    };
    // The pair's second half; `keep` of its rows are real (the look-ahead past the band's last row is computed and dropped).
    // The pair's texels must have arrived.  vmcnt counts loads and stores together, but only loads return in issue order
    // among themselves (a store may complete before an older load), so the only operations that may stay in flight are the two
    // loads known to be younger: the next pair's texels.  Anything else issued in between — segment prefetches, the row
    // buffer's store — only makes the wait stricter.
    auto pair_b = [&](Pair &R, int keep) {
        asm volatile("s_waitcnt vmcnt(2)" : "+v"(R.tex0), "+v"(R.tex1) : : "memory");
        const uint32_t out0 = shade_pack(R.tex0, R.fac0), out1 = shade_pack(R.tex1, R.fac1);
        if (keep <= 0) return;                        // wave-uniform
        rowbuf[(uint32_t)parked * park_mul + park_at] = out0;
        if (keep > 1) rowbuf[(uint32_t)(parked + 1) * park_mul + park_at] = out1;
        parked += keep;
        if (parked >= 4) flush_rows(4);
    };
    // Order of the vector-memory operations:  L(p+1) x2  [wait L(p)]  (S)  L(p+2) x2  [wait L(p+1)]  (S) ...
    // The loop has ONE shape for every trip — rows past the band's end are loaded again rather than handled by a peeled tail —
    // so that a texel in flight always sits in the register its load was issued into: a register copy inserted on a loop-exit
    // edge would read the register before the load has landed (tests/test_isa_checks.py looks for such reads in the built ISA).
    Pair A, B;
    pair_a(y_lo, A);
    for (int y = y_lo; y <= y_hi; y += 4) {           // two pairs per trip so that the in-flight texels need no register move
        pair_a(min(y + 2, y_hi), B);
        pair_b(A, min(2, y_hi - y + 1));
        pair_a(min(y + 4, y_hi), A);
        pair_b(B, min(2, y_hi - (y + 2) + 1));
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(A.tex0), "+v"(A.tex1) : : "memory");   // the last, unused look-ahead
    if (parked) flush_rows(parked);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void stage_palette(const RasterParams &P, float4 *palf, int tid, int nthreads) {
    for (int i = tid; i < 256; i += nthreads) {
        const uint32_t c = P.scene.palette[i];
        palf[i] = make_float4((float)(c & 255u), (float)((c >> 8) & 255u), (float)((c >> 16) & 255u), 0.0f);
    }
}

// Four wavefronts = four consecutive bands of one strip per workgroup (they are independent and only share the palette).
__global__ __launch_bounds__(256, DG_STRIPS_MIN_WAVES) void dg_raster_strips(RasterParams P) {
    __shared__ float4 palf[256];                      // palette as f32 triples: shading needs no v_cvt_f32_ubyte
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[4][4 * 48 + 16];   // per wave: four finished rows of the strip (RGB24, 192 B each) + a dump slot
    const int f = blockIdx.z;
    if (P.frame_flags[f] != 0u) return;               // segment slots exceeded: the batch is redone by dg_raster_tiles
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_palette(P, palf, threadIdx.x, 256);
    __syncthreads();
    const int band = (int)blockIdx.y * 4 + wave;
    if (band >= P.n_bands || P.band_ovl[((size_t)f * (size_t)P.n_bands + (size_t)band) * (size_t)gridDim.x + blockIdx.x] != 0) return;   // dg_raster_tile_list
    strips_body(P, band, palf, rowbuf[wave]);
}

// Per-row constants of the flat and sky mappers for one frame size: the prepared reciprocal of vy = CFY - y (visplanes.rs:109)
// and the sky texture row (visplanes.rs:68-72).  Same device code as the per-lane computation it replaces, run once per
// scene upload instead of once per wavefront (~35 VALU instructions of every raster wave).
__global__ void dg_row_table(DevScene scene, DevConsts k, uint4 *row_tab) {
    const int y = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (y >= k.H) return;
    const float vy = k.CFY - (float)y;
    row_tab[y] = make_uint4(f32_bits(prepare_rcp(vy)), (uint32_t)sky_row(scene, k, y), f32_bits(vy), 0u);
}

// Order-independent per-frame checksum (dg_frame_checksums): every dword is mixed with its index, the mixes are summed.
// Pure streaming read: 256 dwords per lane-iteration are coalesced, one 64-bit atomic add per wave.
__global__ __launch_bounds__(256) void dg_checksum(const uint8_t *fb, size_t frame_bytes, unsigned long long *out) {
    const uint32_t *d = reinterpret_cast<const uint32_t *>(fb + (size_t)blockIdx.y * frame_bytes);
    const size_t n = frame_bytes / 4;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long m = ((unsigned long long)d[i] ^ (i * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
        acc += m ^ (m >> 32);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.y], acc);
}

hipError_t launch_checksums(const uint8_t *fb, size_t frame_bytes, int count, unsigned long long *out, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(256, (frame_bytes / 4 + 255) / 256);
    hipLaunchKernelGGL(dg_checksum, dim3(blocks, (unsigned)count), dim3(256), 0, stream, fb, frame_bytes, out);
    return hipGetLastError();
}

hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint4 *row_tab, hipStream_t stream) {
    hipLaunchKernelGGL(dg_row_table, dim3((unsigned)((k.H + 255) / 256)), dim3(256), 0, stream, scene, k, row_tab);
    return hipGetLastError();
}

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream) {
    if (P.n_frames <= 0 || max_spans_per_frame == 0) return hipSuccess;
    dim3 grid((max_spans_per_frame + 255) / 256, (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_setup_spans, grid, dim3(256), 0, stream, P);
    return hipGetLastError();
}

hipError_t launch_raster(const RasterParams &P, hipStream_t stream, hipEvent_t after_resolve, hipStream_t aux, hipEvent_t aux_done) {
    if (P.n_frames <= 0) return hipSuccess;
    const unsigned strips = (unsigned)((P.k.W + TILE_W - 1) / TILE_W);
    if (P.strips) {
        // frame_flags [max_batch], the two tile counters and band_ovl [F][n_bands][strips] are one allocation: one fill clears all
        hipError_t e = hipMemsetAsync(P.frame_flags, 0, (size_t)(P.band_ovl - reinterpret_cast<uint8_t *>(P.frame_flags)) +
                                                           (size_t)P.n_frames * (size_t)P.n_bands * strips, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(dg_resolve_columns, dim3((strips + 3) / 4, (unsigned)P.n_frames), dim3(256), 0, stream, P);
        if (after_resolve) { e = hipEventRecord(after_resolve, stream); if (e != hipSuccess) return e; }
        const bool side = aux && aux_done && after_resolve;
        if (side) {                                   // the tile list beside the strips
            if ((e = hipStreamWaitEvent(aux, after_resolve, 0)) != hipSuccess) return e;
            hipLaunchKernelGGL(dg_raster_tile_list, dim3((unsigned)P.tile_workgroups), dim3(THREADS), 0, aux, P);
            if ((e = hipEventRecord(aux_done, aux)) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(dg_raster_strips, dim3(strips, (unsigned)((P.n_bands + 3) / 4), (unsigned)P.n_frames), dim3(256), 0, stream, P);
        if (side) { if ((e = hipStreamWaitEvent(stream, aux_done, 0)) != hipSuccess) return e; }
        else hipLaunchKernelGGL(dg_raster_tile_list, dim3((unsigned)P.tile_workgroups), dim3(THREADS), 0, stream, P);
        return hipGetLastError();
    }
    dim3 grid(strips, (unsigned)((P.k.H + TILE_H - 1) / TILE_H), (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_raster_tiles, grid, dim3(THREADS), 0, stream, P);
    return hipGetLastError();
}

int strip_band_rows(int H) { (void)H; return TILE_H; }   // a band of dg_raster_strips = one tile row of dg_raster_tile_list

}  // namespace dg
