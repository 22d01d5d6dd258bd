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
    const uint32_t o = wall_texel_offset(a.y, a.z, b.x, b.y, b.z, b.w, y);
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
                                                  const uint4 *lsp, uint32_t n, int lane, int y, int y0, float vy, float r_vy, int srow, uint32_t color) {
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
                                                        int lane0, const uint4 *lsp, int y, float vy, float r_vy, int srow, uint32_t color) {
    uint32_t winner = 0xffffffffu;
    while (hm) {
        const int j = __builtin_ctz(hm);
        hm &= hm - 1;
        span_step(P, pal, lsp, bcast(w0f, lane0 + j), (uint32_t)j, y, srow, color, winner);
    }
    return shade_winner(P, fr, pal, lsp, winner, color, y, vy, r_vy, srow);
}

__global__ __launch_bounds__(THREADS) void dg_raster_tiles(RasterParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_H * TILE_STRIDE];
    __shared__ __attribute__((aligned(16))) uint4 lspans[SPAN_CAP * 2];
    __shared__ uint32_t lw0[SPAN_CAP];
    __shared__ uint32_t pal[256];
    __shared__ uint32_t lcoff[TILE_W + 1];
    __shared__ uint32_t lskip[TILE_W];

    const int f = blockIdx.z;
    const int W = P.k.W, H = P.k.H;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    // Overlay mode: dg_raster_strips has already written every pixel of this frame from the resolved opaque spans; this kernel
    // only applies, in draw order, the spans from each column's first possibly-transparent one on (masked walls, sprites).
    // Tiles that no such span touches leave at once.  A frame whose columns did not fit the segment slots is rendered here
    // from all of its spans, as is everything when the strip path is off.
    const bool overlay = P.strips != 0 && P.frame_flags[f] == 0u;
    if (overlay) {
        const uint32_t bb = P.strip_ovl[(size_t)f * (size_t)gridDim.x + blockIdx.x];
        const int lo = (int)(bb & 0xffffu), hi = (int)(bb >> 16);
        if (lo > hi || hi < y0 || lo > y0 + (TILE_H - 1)) return;
    }
    const DevFrame fr = P.frames[f];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = y0 + lane;
    const float vy = P.k.CFY - (float)y;      // visplanes.rs:109, a per-row constant
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const uint4 *gspans = reinterpret_cast<const uint4 *>(P.rspans + fr.span_base);
    uint8_t *fb = P.fb + (size_t)f * (size_t)3 * (size_t)W * (size_t)H;

    // the prologue loads are issued together (addresses clamped instead of branching around the loads)
    const uint2 rt = P.row_tab[y < H ? y : H - 1];                    // prepared reciprocal of vy and the sky row (dg_row_table)
    const uint32_t pal_v = P.scene.palette[threadIdx.x & 255];
    const int xc = x0 + (int)(threadIdx.x <= TILE_W ? threadIdx.x : 0);
    const uint32_t coff_v = coff[xc < W ? xc : W];
    uint32_t skip_v = 0;
    if (overlay && threadIdx.x < TILE_W) skip_v = P.ov_first[(size_t)f * (size_t)W + (size_t)(xc < W ? xc : W - 1)];
    const float r_vy = bits_f32(rt.x);
    const int srow = (int)rt.y;
    if (threadIdx.x < 256) pal[threadIdx.x] = pal_v;
    if (threadIdx.x <= TILE_W) lcoff[threadIdx.x] = coff_v;
    if (threadIdx.x < TILE_W) lskip[threadIdx.x] = skip_v;
    if (overlay) {
        // read-in: the tile's RGB24 bytes as the strip kernel left them, 12 bytes -> 4 RGBX pixels (the inverse of the read-out)
        for (int g = threadIdx.x; g < TILE_H * (TILE_W / 4); g += THREADS) {
            const int row = g >> 4, gc = g & 15;
            const int yy = y0 + row, xx = x0 + 4 * gc;
            if (yy < H && xx < W) {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(fb + ((size_t)yy * (size_t)W + (size_t)xx) * 3);
                const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
                *reinterpret_cast<uint4 *>(&tile[row * TILE_STRIDE + 4 * gc]) =
                    make_uint4(d0 & 0xffffffu, (d0 >> 24) | ((d1 & 0xffffu) << 8), (d1 >> 16) | ((d2 & 0xffu) << 16), d2 >> 8);
            }
        }
    }
    __syncthreads();

    // The spans of adjacent columns are one contiguous range of the column-major span array.  Stage as many whole columns
    // as fit in LDS (normally the whole tile) with one coalesced burst — every load of the workgroup in flight at once, so
    // the dependent chain col_off -> spans is paid once per tile — then rasterise those columns; repeat if needed.
    int c_lo = 0;
    while (c_lo < TILE_W) {
        const uint32_t t0 = lcoff[c_lo];
        int c_hi = TILE_W;
        if (lcoff[TILE_W] - t0 > SPAN_CAP) {
            c_hi = c_lo + 1;                  // a single column always fits: the binner caps a column at SPAN_CAP spans
            while (c_hi < TILE_W && lcoff[c_hi + 1] - t0 <= SPAN_CAP) c_hi++;
        }
        const uint32_t n_stage = lcoff[c_hi] - t0;
        for (uint32_t i = threadIdx.x; i < 2 * n_stage; i += THREADS) {
            const uint4 v = gspans[2 * (size_t)t0 + i];
            lspans[i] = v;
            if ((i & 1u) == 0) lw0[i >> 1] = v.x;
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
            const uint32_t skip = lskip[fcol];
            f_n0 = lcoff[fcol] - t0 + skip;
            f_n = lcoff[fcol + 1] - lcoff[fcol] - skip;
            if ((uint32_t)fslot < f_n && f_n <= 8u) {
                f_w0 = lw0[f_n0 + (uint32_t)fslot];
                f_hit = w0_cbot(f_w0) >= y0 && w0_ctop(f_w0) <= y0 + (TILE_H - 1);
            }
        }
        const unsigned long long hitm = __ballot(f_hit);
        int k8 = 0;
        for (int c = c_lo + wave; c < c_hi; c += WAVES, k8 += 8) {
            const uint32_t n0 = bcast(f_n0, k8), n = bcast(f_n, k8);
            const uint32_t base = overlay ? tile[lane * TILE_STRIDE + c] : 0u;     // pixels.rs:10-14: a fresh buffer is all zero
            uint32_t px;
            if (n > 8u) px = raster_column(P, fr, pal, lw0 + n0, lspans + 2 * n0, n, lane, y, y0, vy, r_vy, srow, base);
            else px = raster_column_small(P, fr, pal, (uint32_t)(hitm >> k8) & 0xffu, f_w0, k8, lspans + 2 * n0, y, vy, r_vy, srow, base);
            tile[lane * TILE_STRIDE + c] = px;
        }

        c_lo = c_hi;
        if (c_lo < TILE_W) __syncthreads();   // before the staging area is reused
    }
    __syncthreads();

    // Read-out: groups of 4 pixels (16 B of RGBX in LDS -> 12 B of RGB24 in HBM); 16 groups per tile row, so the lanes of
    // a wave cover four tile rows = 4 x 192 contiguous bytes.
    for (int g = threadIdx.x; g < TILE_H * (TILE_W / 4); g += THREADS) {
        const int row = g >> 4, gc = g & 15;
        const int yy = y0 + row, xx = x0 + 4 * gc;
        if (yy < H && xx < W) {   // W % 4 == 0 (checked at dg_create), so a group never straddles the right edge
            const uint4 p = *reinterpret_cast<const uint4 *>(&tile[row * TILE_STRIDE + 4 * gc]);
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

// ---- strip path ---------------------------------------------------------------------------------------------------------

// One lane per (frame, screen column): strip_core.h resolve_column.  The w0 words it scans are 32 bytes apart in the
// column's span list; real columns hold 2-8 spans, so the quadratic scan is a few dozen L1-resident loads.  Negligible next
// to the raster kernels (320 000 columns per launch against 256 M pixels).
__global__ __launch_bounds__(64) void dg_resolve_columns(RasterParams P) {
    const int f = blockIdx.y;
    const int W = P.k.W, H = P.k.H;
    const int lane = threadIdx.x;
    const int x = (int)blockIdx.x * 64 + lane;
    const DevFrame fr = P.frames[f];
    int lo = 0x7fff, hi = -1;
    if (x < W) {
        const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
        const uint32_t o = coff[x], n = coff[x + 1] - o;
        const ResolveResult r = resolve_column(P.rspans + fr.span_base + o, n, P.scene, H, P.band_rows, (uint32_t)P.seg_cap,
                                               P.segs + (size_t)f * (size_t)P.seg_cap * (size_t)W + (size_t)x, (size_t)W,
                                               P.band_first + (size_t)f * (size_t)P.n_bands * (size_t)W + (size_t)x, (size_t)W);
        if (r.n_segs == 0xffffffffu) atomicOr(&P.frame_flags[f], 1u);
        P.ov_first[(size_t)f * (size_t)W + (size_t)x] = (uint16_t)r.n_base;
        lo = r.ov_lo; hi = r.ov_hi;
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    if (lane == 0) P.strip_ovl[(size_t)f * (size_t)gridDim.x + blockIdx.x] = hi < 0 ? 0x0000ffffu : (uint32_t)lo | ((uint32_t)hi << 16);
}

// One wavefront per (frame, 64-column strip, band of rows), lane = column.  Every lane keeps its column's current segment in
// registers (and the next one, prefetched), so a pixel costs its texture mapper and nothing else: no ownership test, no
// per-pixel parameter fetch, no LDS tile.  The row is wave-uniform: vy, its prepared reciprocal and the sky row come from
// scalar loads.  Two rows are in flight: the texel of row y + 1 is requested before row y is shaded and stored.  A row of the
// strip leaves as one 192-byte store: quads of lanes pack their four RGBX pixels into three dwords with one DPP move and one
// byte permute.  Texels are row-major here (pool), so 64 adjacent columns of a wall row read a handful of cache lines.
__global__ __launch_bounds__(64) void dg_raster_strips(RasterParams P) {
    __shared__ uint32_t pal[256];
    const int f = blockIdx.z;
    if (P.frame_flags[f] != 0u) return;               // segment slots exceeded: dg_raster_tiles renders this frame
    const int lane = threadIdx.x;
    const int W = P.k.W, H = P.k.H;
    const int x0 = (int)blockIdx.x * 64;
    const int y_lo = (int)blockIdx.y * P.band_rows;
    const int y_hi = min(H, y_lo + P.band_rows) - 1;
#pragma unroll
    for (int k = 0; k < 4; k++) pal[lane + 64 * k] = P.scene.palette[lane + 64 * k];
    const DevFrame fr = P.frames[f];
    const bool in_w = x0 + lane < W;
    const int x = in_w ? x0 + lane : W - 1;
    const uint32_t s0 = P.band_first[((size_t)f * (size_t)P.n_bands + blockIdx.y) * (size_t)W + (size_t)x];
    const uint4 *sp = reinterpret_cast<const uint4 *>(P.segs + ((size_t)f * (size_t)P.seg_cap + s0) * (size_t)W + (size_t)x);
    const size_t seg_step = (size_t)W * 2;
    uint4 ca = sp[0], cb = sp[1];
    uint4 na = ca, nb = cb;
    sp += seg_step;
    if (seg_end(ca.x) < H - 1) { na = sp[0]; nb = sp[1]; }
    __syncthreads();
    const uint8_t *pool = P.scene.pool;
    const uint32_t sky_w = (uint32_t)P.scene.sky_w;
    // lane 4q + j (j < 3) stores dword j of its quad's 12 bytes; selector of v_perm_b32 over {next pixel, own pixel}
    const uint32_t perm_sel = (lane & 3) == 0 ? 0x04020100u : (lane & 3) == 1 ? 0x05040201u : 0x06050402u;
    const uint32_t st_off = (uint32_t)(lane - (lane >> 2)) * 4u;
    const bool st_on = in_w && (lane & 3) != 3;
    uint8_t *rowp = P.fb + (((size_t)f * (size_t)H + (size_t)y_lo) * (size_t)W + (size_t)x0) * 3;

    // per-row constants of this band: lane r holds those of row y_lo + r (band_rows <= 64), broadcast with v_readlane
    const uint2 rt_l = P.row_tab[min(y_lo + lane, H - 1)];

    // Row y, first half: move to the column's next segment when the row leaves the current one, then the texture mapper of
    // the segment's kind -> pool offset of the texel and the light factor; the texel load is issued and NOT waited for.
    auto row_a = [&](int y, uint32_t &tex, float &fac) {
        if (y > seg_end(ca.x)) {
            ca = na; cb = nb;
            if (seg_end(ca.x) < H - 1) { sp += seg_step; na = sp[0]; nb = sp[1]; }
        }
        const float r_vy = bits_f32((uint32_t)__builtin_amdgcn_readlane((int)rt_l.x, y - y_lo));
        const int srow = __builtin_amdgcn_readlane((int)rt_l.y, y - y_lo);
        const float vy = P.k.CFY - (float)y;
        const uint32_t kind = seg_kind(ca.x);
        uint32_t off = ca.z;                          // SEG_NONE: offset 0, factor 0 -> black
        fac = bits_f32(ca.w);
        if (kind == SPAN_FLAT) off = seg_flat_offset(fr, ca.y, ca.z, cb.x, cb.y, cb.z, vy, r_vy, fac);
        else if (kind == SPAN_WALL) off = seg_wall_offset(ca.y, ca.z, cb.x, cb.y, cb.z, cb.w, y);
        else if (kind == SPAN_SKY) {
            if (srow >= 0) off = ca.z + (uint32_t)srow * sky_w;
            else { off = 0; fac = 0.0f; }             // row outside the sky bitmap: nothing is drawn
        }
        tex = pool[off];
    };
    // Row y, second half: palette, lighting, and the strip's 192 bytes of this row.
    auto row_b = [&](uint32_t tex, float fac) {
        const uint32_t px = shade(pal[tex], fac);
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)px, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
        const uint32_t out = __builtin_amdgcn_perm(nx, px, perm_sel);
        if (st_on) *reinterpret_cast<uint32_t *>(rowp + st_off) = out;
        rowp += (size_t)W * 3;
    };
    uint32_t tex_a, tex_b = 0;
    float fac_a, fac_b = 0.0f;
    row_a(y_lo, tex_a, fac_a);
    int y = y_lo;
    for (; y + 2 <= y_hi; y += 2) {                   // two rows per trip so that the in-flight texel needs no register move
        row_a(y + 1, tex_b, fac_b);
        row_b(tex_a, fac_a);
        row_a(y + 2, tex_a, fac_a);
        row_b(tex_b, fac_b);
    }
    if (y < y_hi) {                                   // rows y, y + 1 left
        row_a(y + 1, tex_b, fac_b);
        row_b(tex_a, fac_a);
        row_b(tex_b, fac_b);
    } else {
        row_b(tex_a, fac_a);
    }
}

// Per-row constants of the flat and sky mappers for one frame size: the prepared reciprocal of vy = CFY - y (visplanes.rs:109)
// and the sky texture row (visplanes.rs:68-72).  Same device code as the per-lane computation it replaces, run once per
// scene upload instead of once per wavefront (~35 VALU instructions of every raster wave).
__global__ void dg_row_table(DevScene scene, DevConsts k, uint2 *row_tab) {
    const int y = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (y >= k.H) return;
    const float vy = k.CFY - (float)y;
    row_tab[y] = make_uint2(f32_bits(prepare_rcp(vy)), (uint32_t)sky_row(scene, k, y));
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

hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint2 *row_tab, hipStream_t stream) {
    hipLaunchKernelGGL(dg_row_table, dim3((unsigned)((k.H + 255) / 256)), dim3(256), 0, stream, scene, k, row_tab);
    return hipGetLastError();
}

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream) {
    if (P.n_frames <= 0 || max_spans_per_frame == 0) return hipSuccess;
    dim3 grid((max_spans_per_frame + 255) / 256, (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_setup_spans, grid, dim3(256), 0, stream, P);
    return hipGetLastError();
}

int strip_band_rows(int H) {
    // 8 bands per 512 rows: 50 rows at H = 800 (16 bands), 25 at 200, 48 at 768, 50 at 1600
    const int groups = (H + 511) / 512;
    return std::max(1, (H + 8 * groups - 1) / (8 * groups));
}

hipError_t launch_raster(const RasterParams &P, hipStream_t stream) {
    if (P.n_frames <= 0) return hipSuccess;
    const unsigned strips = (unsigned)((P.k.W + TILE_W - 1) / TILE_W);
    if (P.strips) {
        hipError_t e = hipMemsetAsync(P.frame_flags, 0, (size_t)P.n_frames * 4, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(dg_resolve_columns, dim3(strips, (unsigned)P.n_frames), dim3(64), 0, stream, P);
        hipLaunchKernelGGL(dg_raster_strips, dim3(strips, (unsigned)P.n_bands, (unsigned)P.n_frames), dim3(64), 0, stream, P);
    }
    dim3 grid(strips, (unsigned)((P.k.H + TILE_H - 1) / TILE_H), (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_raster_tiles, grid, dim3(THREADS), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dg
