// kernels.hip — gfx950 (MI355X / CDNA4) kernels of the rasteriser.  wave64, no MFMA (nothing here is a
// contraction): the work is per-pixel f32 evaluation + byte gathers, and the output is a stream of RGB24 bytes.
//
// Kernel 1  dg_setup_spans   one lane per span: column-invariant part of the wall/sprite mapper
//                            (bitmap_render.rs:241-251: 3 f32 divides -> texture column + light factor),
//                            sky texture column, floor/ceiling vx; merges the span with its record into one
//                            self-contained 32-byte DevRSpan.
// Kernel 2  dg_raster_tiles  one workgroup (8 wavefronts) per (frame, 64-column x 64-row tile), lane = row:
//                              * the spans of the tile's 64 columns are ONE contiguous range of the column-major span array; they
//                                are staged in LDS in their per-pixel form (stage_tile_span) with a single coalesced burst (16 KB),
//                                together with the palette (as f32x4) and the tile's column offsets — one barrier;
//                              * a wavefront owns eight of the tile's columns.  One pre-filter pass with lane = (column, span slot)
//                                finds, for all eight at once, the spans that touch the tile's rows; then the wave takes one column
//                                at a time with lane = row, in two stages that overlap between columns: stage 1 walks the column's
//                                opaque spans in draw order and records per row the last one covering it (three v_readlane per
//                                span), then every row evaluates its owner ONCE — exact last-writer-wins, no overdraw evaluation, no
//                                divergent control flow — and issues its texel gather; stage 2 (after stage 1 of the next column)
//                                shades and lays the possibly-transparent spans (sprites, masked walls) on top;
//                              * a wall column reads one texture column ([x][y] texel layout => consecutive bytes);
//                              * finished pixels go to an LDS tile [col][row] and leave the CU as 12-byte-per-lane RGB24 row
//                                segments.  Every pixel of the tile is stored (uncovered = 0,0,0) which fuses the reference's
//                                per-frame `Pixels::new()` clear (pixels.rs:10-14) into the one write pass.
//            dg_resolve_columns / dg_raster_strips / dg_raster_tile_list   the optional strip path (DOOMGPU_STRIPS=1, strip_core.h)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (contraction would break bit-exactness; the IEEE
// divide expansion keeps its own internal FMAs, which is what makes it correctly rounded).
// Experiment builds (make variant VARIANT=x EXTRA="-D..."; never defined in the product build; results in profiles/r02_*.md):
//   DG_EXP_T_LDSPAD=bytes   pad the tile kernel's LDS (occupancy experiment)      DG_EXP_T_TIMING   per-wave s_memtime phase probe (device printf)
//   DG_EXP_NOLOAD / DG_EXP_NOSTORE / DG_EXP_NOPAL   strip kernel without texel loads / stores / palette reads
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "kernels.hpp"
#include "raster_core.h"
#include "strip_core.h"

namespace dg {

constexpr int TILE_W = 64;        // columns per workgroup
constexpr int TILE_H = 64;        // rows per workgroup = lanes per wave
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

// ---- one screen column x 64 rows for one wavefront (lane = row) ---------------------------------------------------------------
// The column's spans are in draw order; a pixel belongs to the LAST span that writes it.  Three steps:
//   A  opaque spans (no immediate flag): a wave-uniform loop over the spans that touch these rows, three VALU instructions each
//      (row - ctop, unsigned compare with cbot - ctop, select) — `winner` = index of the last opaque span covering the row;
//   B  possibly-transparent spans (masked walls, sprites, a sky bitmap with holes: the immediate flag), in draw order, for the
//      rows they cover and that no LATER opaque span owns (index > winner): texel fetched, written only where it is opaque;
//   C  every row evaluates its winner once: each kind present computes its (texel offset, light factor), then ONE byte gather,
//      palette lookup and shade for all kinds (flats and bitmap texels live in one allocation; sky = factor 1.0, uncovered =
//      factor 0.0 -> exactly 0,0,0).
// No pixel is evaluated twice except under step B's overlays; nothing depends on the order of evaluation but B.

// shade_f with bit 31 of the result set: "an overlay wrote this pixel" (the tile read-out ignores the top byte).
__device__ __forceinline__ uint32_t shade_f_marked(const float4 c, float factor) {
    const float r = c.x * factor, g = c.y * factor, b = c.z * factor;
    uint32_t o;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %1, 0, %4\n\t"
                 "v_cvt_pk_u8_f32 %0, %2, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 2, %0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(o) : "v"(r), "v"(g), "v"(b), "s"(0x80000000u));
    return o;
}

// Both planes of one texel (palette index, opacity) with one round trip: address = scalar base + 32-bit lane offset, waited for here.
__device__ __forceinline__ void gather_u8x2(const uint8_t *base0, const uint8_t *base1, uint32_t o, uint32_t &v0, uint32_t &v1) {
    asm volatile("global_load_ubyte %0, %2, %3\n\tglobal_load_ubyte %1, %2, %4\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1) : "v"(o), "s"(base0), "s"(base1) : "memory");
}

// diminish_color (bitmap_render.rs:202-207) on a palette entry held as three f32 (LDS): three products, rounded to nearest like
// the reference's `as f32 * factor`, then `as u8` = truncate + saturate.  v_cvt_pk_u8_f32 converts in the wave's current f32 rounding
// mode, so under round-toward-zero it IS `as u8` (negative and NaN -> 0, > 255 -> 255; tools/microbench/cvt_round.hip compares it with
// trunc + convert over all 2^32 patterns: no difference) — the mode is switched for exactly these three instructions.
__device__ __forceinline__ uint32_t shade_f(const float4 c, float factor) {
    const float r = c.x * factor, g = c.y * factor, b = c.z * factor;
    uint32_t o;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %2, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 2, %0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(o) : "v"(r), "v"(g), "v"(b));
    return o;
}

// ---- spans as the tile kernel keeps them in LDS (stage_tile_span); record 0 of the staging area is the "nothing" record that an
// unowned row points at (a sky span with factor 0 -> 0,0,0), records 1.. are the tile's spans ------------------------------------
//   WALL  a.x = hm (< 0x4000_0000): h - 1 for a power-of-two bitmap height, else 0x8000 | h;  a.y = d;  a.z = start of the texture
//         column (column-major planes);  a.w = light factor;  b.x = uy1;  b.y = top_y | off_y << 16;  b.z = h as f32, NEGATED when h
//         is not a power of two;  b.w = prepared reciprocal of d
//   FLAT  a.x = 0x4000_0000 | ..;  a.y = wz * vx;  a.z = offset of the flat from texel_idx (flats sit behind the texel plane);
//         b.x = gwz;  b.y = light_level / 255;  b.z = fast-divide-ok << 8
//   SKY   a.x = 0x8000_0000 | .. (negative as i32);  a.z = offset of the sky texture column (0 when the reference would index outside
//         the bitmap);  a.w = 1.0f (0.0f in that case)
struct RowConsts {              // per screen row = per lane, fixed for the tile
    int y;
    float vy, r_vy;             // CFY - y (visplanes.rs:109) and its prepared reciprocal
    uint32_t row_fast;          // 0x100 unless vy == 0 (that row takes the plain divide: x / 0)
    uint32_t sky_row;           // sky texture row, 0 when outside the bitmap
    float sky_fac;              // 1.0f, 0.0f when outside the bitmap
};

__device__ __forceinline__ void stage_tile_span(uint4 &a, uint4 &b, uint32_t flats_rel) {
    const uint32_t kind = w0_kind(a.x);
    if (kind == SPAN_WALL) {
        const uint32_t h = b.z & 0xffffu;
        const bool pot = (h & (h - 1)) == 0;
        stage_wall_span(a.y, a.z, b.z, b.w);
        a.x = pot ? h - 1 : 0x8000u | h;
        b.z = f32_bits(pot ? (float)h : -(float)h);
    } else if (kind == SPAN_FLAT) {
        a.z += flats_rel;
    } else {
        const bool valid = a.z != 0xffffffffu;
        a.z = valid ? a.z : 0u;
        a.w = f32_bits(valid ? 1.0f : 0.0f);
    }
}

// Texel offset of one wall pixel (bitmap_render.rs:256-263) from the staged words.  A wave in which some lane's bitmap height is not
// a power of two takes the general modulus for every lane (it is right for all heights).
// `lanes`: the lanes whose span is a wall (the others compute garbage that is not used).
__device__ __forceinline__ uint32_t wall_offset_tile(const uint4 a, const uint4 b, int y, unsigned long long lanes) {
    const float d = bits_f32(a.y), hs = bits_f32(b.z), hf = __builtin_fabsf(hs);
    const int32_t top_y = lo_i16(b.y), off_y = hi_i16(b.y);
    const float ay = div_prepared_nofix((float)(y - top_y), d, bits_f32(b.w));   // d == 0: uy1 is NaN and so is the sum, whatever ay is
    const int32_t ty = f32_as_i16(hf + ay * bits_f32(b.x));
    if ((__builtin_amdgcn_ballot_w64(hs < 0.0f) & lanes) == 0ull) return a.z + ((uint32_t)(ty + off_y) & a.x);   // the wrap of the i16 add is above the mask
    return a.z + (uint32_t)floor_mod_fast(wrap_i16(ty + off_y), (int32_t)hf, 0, approx_rcp(hf));
}

// Texel offset and light factor of one floor / ceiling pixel (visplanes.rs:108-126); see flat_texel_offset (raster_core.h) for the
// scalar form.  The factor is left unclamped: `as u8` of (colour x negative) is 0, the same as with the reference's
// `if factor < 0.0 { factor = 0.0 }`, and lightf - z / 4096 as one fma is exact because z / 4096 is.
__device__ __forceinline__ uint32_t flat_offset_tile(const DevFrame &f, const uint4 a, const uint4 b, const RowConsts &R, unsigned long long lanes, float &factor) {
    float wx, wy;
    if ((__builtin_amdgcn_ballot_w64((b.z & R.row_fast) == 0u) & lanes) == 0ull) {
        wx = div_prepared_nofix(bits_f32(b.x), R.vy, R.r_vy);
        wy = div_prepared_nofix(bits_f32(a.y), R.vy, R.r_vy);
    } else {                                      // a numerator outside the verified domain or the vy == 0 row somewhere in the wave: plain divides
        wx = bits_f32(b.x) / R.vy;
        wy = bits_f32(a.y) / R.vy;
    }
    const float rx = wx * f.cos_a - wy * f.sin_a;
    const float ry = wy * f.cos_a + wx * f.sin_a;
    const int32_t tx = (f32_as_i16(rx) + f.pos_x_i16) & 63;
    const int32_t ty = (f32_as_i16(ry) + f.pos_y_i16) & 63;
    factor = __builtin_fmaf(-(float)f32_as_i16(wx), 1.0f / 4096.0f, bits_f32(b.y));
    return a.z + (uint32_t)(ty * 64 + tx);
}

// Spans are addressed by their BYTE offset in the staging area (32 x record number): the owner of a row is the largest offset
// among the opaque spans that cover it, 0 = the "nothing" record.
//
// Both span loops below are wave-uniform walks over a 64-bit mask whose bit i says "the span held by lane i is to be looked at".
// Each lane holds its span's first row, row count - 1 and staging offset in three registers, so a step is three v_readlane and
// no scalar unpacking: find-first-bit, clear it, compare, branch are the only scalar instructions.
__device__ __forceinline__ int take_lowest(unsigned long long &m) {
    const int j = __builtin_ctzll(m);
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));
    return j;
}
__device__ __forceinline__ uint32_t owner_loop(unsigned long long m, uint32_t v_lo, uint32_t v_rg, uint32_t v_off, const RowConsts &R, uint32_t winner) {
    while (m) {
        const int j = take_lowest(m);
        winner = ((uint32_t)R.y - bcast(v_lo, j)) <= bcast(v_rg, j) ? bcast(v_off, j) : winner;
    }
    return winner;
}

__device__ __forceinline__ const uint4 *span_at(const uint4 *staged, uint32_t off) {
    return reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(staged) + off);
}

// Possibly-transparent spans in draw order (m_wall: which of them are bitmap columns; the others are sky-with-holes spans).  A
// written pixel carries bit 31 (shade_f_marked); returns the colour so far.  (Fetching two spans' texels per trip without exec
// masking was measured: slower at 1280x800 — most columns meet one such span — and no faster at 320x200.)
__device__ __forceinline__ uint32_t overlay_loop(const RasterParams &P, const float4 *palf, const uint4 *staged, unsigned long long m, unsigned long long m_wall,
                                                 uint32_t v_lo, uint32_t v_rg, uint32_t v_off, const RowConsts &R, uint32_t winner, uint32_t color) {
    while (m) {
        const int j = take_lowest(m);
        const uint32_t off = bcast(v_off, j);
        if (((uint32_t)R.y - bcast(v_lo, j)) <= bcast(v_rg, j) && off > winner) {
            const uint4 a = span_at(staged, off)[0], b = span_at(staged, off)[1];   // same address in every lane: LDS broadcast
            const bool wall = (m_wall >> j) & 1ull;                                   // wave-uniform
            uint32_t o;
            float factor;
            if (wall) {
                o = wall_offset_tile(a, b, R.y, ~0ull);
                factor = bits_f32(a.w);
            } else {                                          // sky bitmap with holes: plain lookup, no lighting (x 1.0 is exact)
                o = a.z + R.sky_row;
                factor = bits_f32(a.w) * R.sky_fac;           // 0.0 where the reference would index outside the bitmap: nothing drawn
            }
            uint32_t tex, opq;
            gather_u8x2(P.scene.texel_idx, P.scene.texel_opq, o, tex, opq);
            const uint32_t c = shade_f_marked(palf[tex], factor);
            color = (opq != 0u && (wall || factor != 0.0f)) ? c : color;
        }
    }
    return color;
}

// Every row evaluates its owner.  The scalar instruction stream is what this kernel is short of (profiles/r02_raster_tiles.md), so
// there is no divergent control flow here: a kind that some row of the wave needs is computed by ALL 64 lanes (on words of another
// kind the arithmetic is garbage but harmless) and each lane then selects — two uniform branches and a few selects instead of
// nested exec-mask regions.  The gather comes after the select, so every address is that of the lane's real owner.
__device__ __forceinline__ uint32_t owner_texel(const DevFrame &fr, const uint4 *staged, uint32_t winner, const RowConsts &R, float &factor_out) {
    const uint4 a = span_at(staged, winner)[0], b = span_at(staged, winner)[1];
    const bool is_wall = a.x < 0x40000000u, is_sky = (int32_t)a.x < 0;
    const unsigned long long m_wall = __builtin_amdgcn_ballot_w64(is_wall), m_sky = __builtin_amdgcn_ballot_w64(is_sky);
    uint32_t o = a.z + R.sky_row;                 // sky bitmap without holes, or nothing: plain lookup, no lighting (x 1.0 is exact)
    float factor = bits_f32(a.w) * R.sky_fac;
    if (~(m_wall | m_sky) != 0ull) {              // some row is owned by a floor / ceiling
        float ff;
        const uint32_t fo = flat_offset_tile(fr, a, b, R, ~(m_wall | m_sky), ff);
        const bool is_flat = !is_wall && !is_sky;
        o = is_flat ? fo : o;
        factor = is_flat ? ff : factor;
    }
    if (m_wall != 0ull) {
        const uint32_t wo = wall_offset_tile(a, b, R.y, m_wall);
        o = is_wall ? wo : o;
        factor = is_wall ? bits_f32(a.w) : factor;
    }
    factor_out = factor;
    return o;
}

// One lane's span -> the three values the loops broadcast.
__device__ __forceinline__ void unpack_span(uint32_t w0, uint32_t off, uint32_t &v_lo, uint32_t &v_rg, uint32_t &v_off) {
    v_lo = w0 & 0x3fffu;
    v_rg = ((w0 >> 16) & 0x3fffu) - v_lo;
    v_off = off;
}

// A column is rendered in two stages so that the texel gather of one column is in flight while the next column's owners are worked
// out (tile_body interleaves them):  stage 1 = the row owners and their (texel offset, light factor);  stage 2 = palette + shade of
// the gathered texel, then the possibly-transparent spans on top.
//
// Stage 1 for a column with any number of spans (lw0 = word 0 of every span of the column, off0 = byte offset of its first span in
// the staging area): lane i looks at span i of each 64-span chunk, ballots pick the spans touching these rows.
__device__ __forceinline__ uint32_t big_column_owner(const uint32_t *lw0, uint32_t off0, uint32_t n, int lane, int y0, const RowConsts &R) {
    uint32_t winner = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t w0v = i < n ? lw0[i] : 0u;
        const bool hit = i < n && w0_cbot(w0v) >= y0 && w0_ctop(w0v) <= y0 + (TILE_H - 1) && !w0_immediate(w0v);
        uint32_t v_lo, v_rg, v_off;
        unpack_span(w0v, off0 + 32u * i, v_lo, v_rg, v_off);
        winner = owner_loop(__ballot(hit), v_lo, v_rg, v_off, R, winner);
    }
    return winner;
}
// Stage 2 for such a column.
__device__ __forceinline__ uint32_t big_column_overlays(const RasterParams &P, const float4 *pal, const uint32_t *lw0, const uint4 *staged, uint32_t off0,
                                                        uint32_t n, int lane, int y0, const RowConsts &R, uint32_t winner, uint32_t base_color) {
    uint32_t color = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t w0v = i < n ? lw0[i] : 0u;
        const bool hit = i < n && w0_cbot(w0v) >= y0 && w0_ctop(w0v) <= y0 + (TILE_H - 1) && w0_immediate(w0v);
        const unsigned long long m = __ballot(hit);
        if (!m) continue;
        uint32_t v_lo, v_rg, v_off;
        unpack_span(w0v, off0 + 32u * i, v_lo, v_rg, v_off);
        color = overlay_loop(P, pal, staged, m, __ballot(hit && w0_kind(w0v) == SPAN_WALL), v_lo, v_rg, v_off, R, winner, color);
    }
    return (int32_t)color < 0 ? color : base_color;
}

constexpr int TILE_TS = 65;       // dwords per tile COLUMN in LDS
constexpr int PACK_ROWS = 8;      // a tile with no more live rows than this is rendered eight columns per wavefront pass

struct TileLds {
    uint32_t tile[TILE_W * TILE_TS];        // [col][row]: conflict-free for lane = row writes
    uint4 lspans[(SPAN_CAP + 1) * 2];       // record 0: "nothing"; records 1 ..: the staged spans (stage_tile_span)
    uint32_t lw0[SPAN_CAP];
    float4 pal[256];                        // r, g, b as f32
    uint32_t lcoff[TILE_W + 1];
#ifdef DG_EXP_T_LDSPAD
    uint32_t pad[DG_EXP_T_LDSPAD / 4];
#endif
};

// One 64 x 64 tile of frame f: columns x0 .., rows y0 ..
__device__ __forceinline__ void tile_body(const RasterParams &P, TileLds &L, int f, int x0, int y0) {
#ifdef DG_EXP_T_TIMING
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define DG_PHASE(k) { const unsigned long long tnow = __builtin_readcyclecounter(); tm[k] += tnow - tprev; tprev = tnow; }
#else
#define DG_PHASE(k)
#endif
    const DevFrame fr = P.frames[f];
    const int W = P.k.W, H = P.k.H;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = y0 + lane;
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const uint4 *gspans = reinterpret_cast<const uint4 *>(P.rspans + fr.span_base);
    const uint32_t flats_rel = (uint32_t)(P.scene.flats - P.scene.texel_idx);   // one allocation: [texel index plane | flats]

    // The spans of adjacent columns are one contiguous range of the column-major span array: [col_off[x0], col_off[x0 + 64]).  The two
    // ends are wave-uniform (scalar loads); when the range fits in LDS — the normal case — every thread fetches its span straight
    // away, together with the palette, the 65 column offsets and the row constants, and ONE barrier publishes all of it.  Wall
    // spans are put into their per-pixel form on the way (stage_wall_tile: texture column start, prepared 1/d, height mask).
    const uint32_t t_first = coff[x0 < W ? x0 : W], t_last = coff[x0 + TILE_W < W ? x0 + TILE_W : W];
    const bool fits = t_last - t_first <= (uint32_t)SPAN_CAP;
    uint4 sa = make_uint4(0u, 0u, 0u, 0u), sb = sa;
    const bool mine = fits && threadIdx.x < t_last - t_first;
    if (mine) {
        sa = gspans[2 * ((size_t)t_first + threadIdx.x)];
        sb = gspans[2 * ((size_t)t_first + threadIdx.x) + 1];
    }
    const uint4 rt = P.row_tab[y < H ? y : H - 1];                    // prepared reciprocal of vy and the sky row (dg_row_table)
    // A tile with at most 8 live rows (the last tile row of a 200-row frame): a wave can take all eight of its columns in ONE pass,
    // lane = (column, row) = (lane >> 3, lane & 7).  Its row constants:
    const bool few_rows = H - y0 <= PACK_ROWS;
    const int yp = y0 + (lane & (PACK_ROWS - 1));
    uint4 rtp = make_uint4(0u, 0u, 0u, 0u);
    if (few_rows) rtp = P.row_tab[yp < H ? yp : H - 1];
    const uint32_t pal_v = P.scene.palette[threadIdx.x & 255];
    const int xc = x0 + (int)(threadIdx.x <= TILE_W ? threadIdx.x : 0);
    const uint32_t coff_v = coff[xc < W ? xc : W];
    auto row_consts = [&](int yy, const uint4 t) {
        RowConsts r;
        r.y = yy;
        r.vy = P.k.CFY - (float)yy;
        r.r_vy = bits_f32(t.x);
        r.row_fast = r.vy != 0.0f ? 0x100u : 0u;
        r.sky_row = (int)t.y < 0 ? 0u : t.y;
        r.sky_fac = (int)t.y < 0 ? 0.0f : 1.0f;
        return r;
    };
    const RowConsts R = row_consts(y, rt);
    if (threadIdx.x < 256) L.pal[threadIdx.x] = make_float4((float)(pal_v & 255u), (float)((pal_v >> 8) & 255u), (float)((pal_v >> 16) & 255u), 0.0f);
    if (threadIdx.x <= TILE_W) L.lcoff[threadIdx.x] = coff_v;
    if (mine) {
        L.lw0[threadIdx.x] = sa.x;
        stage_tile_span(sa, sb, flats_rel);
        L.lspans[2 * threadIdx.x + 2] = sa;
        L.lspans[2 * threadIdx.x + 3] = sb;
    }
    if (threadIdx.x == THREADS - 1) {         // the "nothing" record: a sky span with factor 0
        L.lspans[0] = make_uint4(0x80000000u, 0u, 0u, 0u);
        L.lspans[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    DG_PHASE(0)
    __syncthreads();
    DG_PHASE(1)

    int c_lo = 0;
    while (c_lo < TILE_W) {
        const uint32_t t0 = L.lcoff[c_lo];
        int c_hi = TILE_W;
        if (!fits) {                          // more than SPAN_CAP spans in the tile: as many whole columns at a time as fit
            if (L.lcoff[TILE_W] - t0 > SPAN_CAP) {
                c_hi = c_lo + 1;              // a single column always fits: the binner caps a column at SPAN_CAP spans
                while (c_hi < TILE_W && L.lcoff[c_hi + 1] - t0 <= SPAN_CAP) c_hi++;
            }
            const uint32_t n_stage = L.lcoff[c_hi] - t0;
            for (uint32_t i = threadIdx.x; i < n_stage; i += THREADS) {
                uint4 a = gspans[2 * ((size_t)t0 + i)], b = gspans[2 * ((size_t)t0 + i) + 1];
                L.lw0[i] = a.x;
                stage_tile_span(a, b, flats_rel);
                L.lspans[2 * i + 2] = a;
                L.lspans[2 * i + 3] = b;
            }
            __syncthreads();
        }
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
        uint32_t v_lo, v_rg, v_off;
        unpack_span(f_w0, 32u * (f_n0 + 1u + (uint32_t)fslot), v_lo, v_rg, v_off);
        const unsigned long long hit_op = __ballot(f_hit && !w0_immediate(f_w0)), hit_ov = __ballot(f_hit && w0_immediate(f_w0));
        const unsigned long long hit_ovwall = __ballot(f_hit && w0_immediate(f_w0) && w0_kind(f_w0) == SPAN_WALL);
        const unsigned long long big = __ballot(f_n > 8u);         // all 8 lanes of a column with more than 8 spans
        const int nk = (c_hi - c_lo - wave + WAVES - 1) / WAVES;     // columns of this chunk that are this wave's
        if (few_rows && nk == WAVES && (hit_ov | big) == 0ull) {
            // ---- eight columns x eight rows in one pass.  Lane (k, r) owns row r of the wave's column k; that column's spans sit in the
            // eight lanes of its own group (the pre-filter's layout), so the owner search is eight lane-permutes of one packed word
            // (first row | row count - 1 << 16; a span that does not count can never match).
            const RowConsts Rp = row_consts(yp, rtp);
            const uint32_t mine_word = (f_hit && !w0_immediate(f_w0)) ? (v_lo | (v_rg << 16)) : 0x0000ffffu;
            const uint32_t off_first = v_off - 32u * (uint32_t)fslot;          // staging offset of the column's first span
            uint32_t winner = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~7) + j) * 4, (int)mine_word);
                winner = ((uint32_t)Rp.y - (w & 0xffffu)) <= (w >> 16) ? off_first + 32u * (uint32_t)j : winner;
            }
            float factor;
            const uint32_t o = owner_texel(fr, L.lspans, winner, Rp, factor);
            const uint32_t px = shade_f(L.pal[P.scene.texel_idx[o]], factor);
            L.tile[(c_lo + wave + WAVES * fk) * TILE_TS + fslot] = px;
            c_lo = c_hi;
            if (c_lo < TILE_W) __syncthreads();
            continue;
        }
        // Two columns in flight per wave: stage 1 of a column and its gather are issued before stage 2 of the column before it.
        struct Col { uint32_t tex, winner; float factor; };
        auto stage1 = [&](int k, Col &C) {
            DG_PHASE(2)
            const unsigned long long colmask = 0xffull << (8 * k);
            if (big & colmask) {
                const uint32_t n0 = bcast(f_n0, 8 * k), n = bcast(f_n, 8 * k);
                C.winner = big_column_owner(L.lw0 + n0, 32u * (n0 + 1u), n, lane, y0, R);
            } else {
                C.winner = owner_loop(hit_op & colmask, v_lo, v_rg, v_off, R, 0u);
            }
            const uint32_t o = owner_texel(fr, L.lspans, C.winner, R, C.factor);
            C.tex = P.scene.texel_idx[o];                    // in flight until stage 2
            DG_PHASE(3)
        };
        auto stage2 = [&](int k, const Col &C) {
            const unsigned long long colmask = 0xffull << (8 * k);
            uint32_t px = shade_f(L.pal[C.tex], C.factor);
            DG_PHASE(4)
            if (big & colmask) {
                const uint32_t n0 = bcast(f_n0, 8 * k), n = bcast(f_n, 8 * k);
                px = big_column_overlays(P, L.pal, L.lw0 + n0, L.lspans, 32u * (n0 + 1u), n, lane, y0, R, C.winner, px);
            } else if (hit_ov & colmask) {
                // (issuing the first such span's texel fetch in stage 1, unmasked, was measured: slower — 0.60 against 0.57 ms)
                const uint32_t color = overlay_loop(P, L.pal, L.lspans, hit_ov & colmask, hit_ovwall, v_lo, v_rg, v_off, R, C.winner, 0u);
                px = (int32_t)color < 0 ? color : px;
            }
            L.tile[(c_lo + wave + WAVES * k) * TILE_TS + lane] = px;
            DG_PHASE(5)
        };
        Col A, B;
        if (nk == WAVES) {                                           // a whole tile's worth (the normal case): one loop shape, no conditionals
            stage1(0, A);                                            // A and B alternate so that an in-flight texel never changes register
            for (int k = 1; k < WAVES - 1; k += 2) {
                stage1(k, B);
                stage2(k - 1, A);
                stage1(k + 1, A);
                stage2(k, B);
            }
            stage1(WAVES - 1, B);
            stage2(WAVES - 2, A);
            stage2(WAVES - 1, B);
        } else {
            for (int k = 0; k < nk; k++) {
                stage1(k, A);
                stage2(k, A);
            }
        }

        c_lo = c_hi;
        if (c_lo < TILE_W) __syncthreads();   // before the staging area is reused
    }
    DG_PHASE(2)
    __syncthreads();
    DG_PHASE(6)

    // Read-out: groups of 4 pixels of a row (4 LDS words -> 12 B of RGB24), 16 groups per tile row; the 64 lanes of a wave take
    // 4 rows x 16 groups in an order that is conflict-free in LDS; 8 adjacent lanes write 96 contiguous bytes.
    uint8_t *fb = P.fb + (size_t)f * (size_t)3 * (size_t)W * (size_t)H;
    const int gc = (lane & 7) | ((lane >> 5) << 3), rsub = (lane >> 3) & 3;
#pragma unroll
    for (int pass = 0; pass < TILE_H / (4 * WAVES); pass++) {
        const int row = pass * 4 * WAVES + wave * 4 + rsub;
        const int yy = y0 + row, xx = x0 + 4 * gc;
        if (yy < H && xx < W) {   // W % 4 == 0 (checked at dg_create), so a group never straddles the right edge
            const uint32_t p0 = L.tile[(4 * gc + 0) * TILE_TS + row], p1 = L.tile[(4 * gc + 1) * TILE_TS + row];
            const uint32_t p2 = L.tile[(4 * gc + 2) * TILE_TS + row], p3 = L.tile[(4 * gc + 3) * TILE_TS + row];
            uint32_t *dst = reinterpret_cast<uint32_t *>(fb + ((size_t)yy * (size_t)W + (size_t)xx) * 3);
            dst[0] = (p0 & 0xffffffu) | (p1 << 24);
            dst[1] = ((p1 >> 8) & 0xffffu) | (p2 << 16);
            dst[2] = ((p2 >> 16) & 0xffu) | (p3 << 8);
        }
    }
#ifdef DG_EXP_T_TIMING
    DG_PHASE(7)
    if (lane == 0 && f == 100 && (x0 / TILE_W) % 5 == 2 && (y0 / TILE_H) % 4 == 1)
        printf("[tile %d,%d wave %d] load+stage %llu barrier %llu between %llu stage1 %llu gather+shade %llu overlays+write %llu end-barrier %llu readout %llu\n", x0 / TILE_W,
               y0 / TILE_H, wave, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5], tm[6], tm[7]);
#endif
}

// (launch bounds: 8 waves per SIMD = at most 64 VGPRs; one more register costs a fourth of the resident workgroups, 0.58 -> 0.67 ms)
// Every tile of every frame (the strip path is off, or a batch is redone because a column exceeded the segment slots).
__global__ __launch_bounds__(THREADS, 8) void dg_raster_tiles(RasterParams P) {
    __shared__ __attribute__((aligned(16))) TileLds L;
    tile_body(P, L, (int)blockIdx.z, (int)blockIdx.x * TILE_W, (int)blockIdx.y * TILE_H);
}

// The tiles dg_resolve_columns listed: those that a possibly-transparent span (masked wall, sprite) touches, where the
// winner of a pixel depends on texels and dg_raster_strips therefore does not go.  The list length is only known on the device:
// the grid covers the longest possible list and the surplus workgroups leave at once.  (A persistent-workgroup version that
// pulled tiles off a shared counter was bound by that counter: same-address device-scope atomics cost ~11 ns each.)
__global__ __launch_bounds__(THREADS, 8) void dg_raster_tile_list(RasterParams P) {
    __shared__ __attribute__((aligned(16))) TileLds L;
    if (blockIdx.x >= P.tile_counters[0]) return;
    const uint32_t t = P.tile_list[blockIdx.x];
    tile_body(P, L, (int)(t >> 16), (int)(t & 0xffu) * TILE_W, (int)((t >> 8) & 0xffu) * TILE_H);
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
            hipLaunchKernelGGL(dg_raster_tile_list, dim3(strips * (unsigned)P.n_bands * (unsigned)P.n_frames), dim3(THREADS), 0, aux, P);
            if ((e = hipEventRecord(aux_done, aux)) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(dg_raster_strips, dim3(strips, (unsigned)((P.n_bands + 3) / 4), (unsigned)P.n_frames), dim3(256), 0, stream, P);
        if (side) { if ((e = hipStreamWaitEvent(stream, aux_done, 0)) != hipSuccess) return e; }
        else hipLaunchKernelGGL(dg_raster_tile_list, dim3(strips * (unsigned)P.n_bands * (unsigned)P.n_frames), dim3(THREADS), 0, stream, P);
        return hipGetLastError();
    }
    dim3 grid(strips, (unsigned)((P.k.H + TILE_H - 1) / TILE_H), (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_raster_tiles, grid, dim3(THREADS), 0, stream, P);
    return hipGetLastError();
}

int strip_band_rows(int H) { (void)H; return TILE_H; }   // a band of dg_raster_strips = one tile row of dg_raster_tile_list

}  // namespace dg
