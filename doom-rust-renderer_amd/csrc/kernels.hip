// kernels.hip — gfx950 (MI355X / CDNA4) kernels of the rasteriser.  wave64, no MFMA (nothing here is a
// contraction): the work is per-pixel f32 evaluation + byte gathers, and the output is a stream of RGB24 bytes.
//
// Kernel 1  dg_setup_spans   one lane per span: column-invariant part of the wall/sprite mapper
//                            (bitmap_render.rs:241-251: 3 f32 divides -> texture column + light factor),
//                            sky texture column, floor/ceiling vx; merges the span with its record into one
//                            self-contained 32-byte DevRSpan.
// Kernel 2  dg_raster_tiles  one workgroup (8 wavefronts) per (frame, 64-column x 64-row tile), lane = row:
//                              * the spans of the tile's 64 columns are ONE contiguous range of the column-major span array; they
//                                are copied to LDS (they are in their per-pixel form already) with a single coalesced burst (16 KB),
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
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (contraction would break bit-exactness; the IEEE
// divide expansion keeps its own internal FMAs, which is what makes it correctly rounded).
// Experiment builds: make variant VARIANT=x EXTRA="-D.." builds the same sources under another name (tools/ab_variants.sh compares builds
// on one box); the kernels themselves carry no experiment switches — measured variants live as patches under tools/experiments/.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <hip/hip_ext.h>

#include <algorithm>
#include <type_traits>

#include "kernels.hpp"
#include "raster_core.h"

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
    else if (sp.kind == SPAN_FLAT) o = resolve_flat_span(sp, P.planes[fr.plane_base + sp.rec], P.k, (uint32_t)(P.scene.flats - P.scene.texel_idx));
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
//      rows they cover and that no LATER opaque span owns (index > winner): texel fetched, taken only where it is opaque;
//   C  every row evaluates its winner once: each kind present computes its (texel offset, light factor), then ONE byte gather,
//      palette lookup and shade for all kinds (flats and bitmap texels live in one allocation; sky = factor 1.0, uncovered =
//      factor 0.0 -> exactly 0,0,0).
// No pixel is evaluated twice except under step B's overlays; nothing depends on the order of evaluation but B.

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

// ---- spans as the tile kernel keeps them in LDS: the DevRSpan words as dg_setup_spans / dg_fe_scatter wrote them (lists_dev.h) except
// words 0 and 5 of a wall (the row range lives in lw0); record 0 of the staging area is the "nothing" record that an unowned row
// points at (a sky span with factor 0 -> 0,0,0), records 1.. are the tile's spans ----------------------------------------------
//   WALL  a.x = off_y << 16 | hm:  hm = h - 1 for a power-of-two bitmap height, else 0x8000 | h;  a.y = d;  a.z = start of the texture
//         column (column-major planes);  a.w = light factor;  b.x = uy1;  b.y = top_y AS F32;  b.z = h as f32, NEGATED when h is not a
//         power of two (|b.z| >= 1 is what says "wall");  b.w = prepared reciprocal of d
//   FLAT  a.x = 0x4000_0000 | ..;  a.y = wz * vx;  a.z = offset of the flat from texel_idx (flats sit behind the texel plane);
//         b.x = gwz;  b.y = light_level / 255;  b.z = fast-divide-ok << 8 (as f32: below 1)
//   SKY   a.x = 0x8000_0000 | .. (negative as i32);  a.z = offset of the sky texture column (0 when the reference would index outside
//         the bitmap);  a.w = 1.0f (0.0f in that case);  b.z = 0
//
// Issue cost is what this kernel is bound by, and on gfx950 it has two prices (tools/microbench/issue_rates.hip,
// profiles/r05_issue_model.md): a SIMD issues ONE vector instruction per ~4.25 clocks, plus a second one in the same slot if that one is
// of the simple class — v_fma / v_mul / v_add / v_sub_f32, v_mov, v_add / v_sub_u32, v_and / v_or / v_xor, right shifts — with VGPR,
// inline-constant or literal operands only.  Conversions, compares, selects, left shifts, v_med3, v_readlane, SDWA forms and ANY
// instruction with a scalar-register operand take a slot of their own.  Hence, below: rows and frame constants live in VGPRs as f32
// (a row difference is a v_sub_f32, not an SDWA subtract + convert), records reach the lanes as LDS broadcasts into VGPRs, and the
// plain mappers contain one conversion each where the reference has an `as i16`.
struct RowConsts {              // per screen row = per lane, fixed for the tile (dg_row_table)
    int y;
    float yf;                   // y as f32: (y - top_y) as f32 == yf - top_y as f32 (both integers below 2^15: the difference is exact)
    float vy, r_vy;             // CFY - y (visplanes.rs:109) and its prepared reciprocal; vy == 0 on the horizon row (that row takes the plain divide: x / 0)
    uint32_t sky;               // sky texture row (0 when outside the bitmap) | bits of the sky factor (1.0f, 0.0f when outside): row < 2^16
};
__device__ __forceinline__ uint32_t sky_row_of(const RowConsts &R) { return R.sky & 0xffffu; }
__device__ __forceinline__ float sky_fac_of(const RowConsts &R) { return bits_f32(R.sky & 0xffff0000u); }
struct FrameVgprs {             // frame constants the flat mapper multiplies / adds per pixel, copied to VGPRs once per strip (a scalar operand halves the issue rate)
    float cos_a, sin_a;
    uint32_t pos_pk;            // (pos_x & 63) | (pos_y & 63) << 16
};

// Words 0 and 5 of a staged wall from the record's words 0, 5 and 6; the other kinds keep theirs.
__device__ __forceinline__ void stage_words(uint4 &a, uint4 &b) {
    if (w0_kind(a.x) != SPAN_WALL) return;
    const float hs = bits_f32(b.z);
    const uint32_t h = (uint32_t)__builtin_fabsf(hs);
    a.x = (hs > 0.0f ? h - 1u : 0x8000u | h) | (b.y & 0xffff0000u);
    b.y = f32_bits((float)lo_i16(b.y));
}
__device__ __forceinline__ bool staged_is_wall(const uint4 b) { return __builtin_fabsf(bits_f32(b.z)) >= 1.0f; }

// Texel offset of one wall pixel (bitmap_render.rs:256-263) from the staged words, any bitmap height, any extent.  A wave in which some
// lane's bitmap height is not a power of two takes the general modulus for every lane (it is right for all heights).
// `lanes`: the lanes whose span is a wall (the others compute garbage that is not used).
__device__ __forceinline__ uint32_t wall_offset_tile(const uint4 a, const uint4 b, const RowConsts &R, unsigned long long lanes) {
    const float d = bits_f32(a.y), hs = bits_f32(b.z), hf = __builtin_fabsf(hs);
    const float ay = div_prepared_nofix(R.yf - bits_f32(b.y), d, bits_f32(b.w));   // d == 0: uy1 is NaN and so is the sum, whatever ay is
    const int32_t ty = f32_as_i16(hf + ay * bits_f32(b.x));
    const uint32_t t = (uint32_t)ty + (a.x >> 16);                                 // + off_y: the low 16 bits are the wrapping i16 sum
    if ((__builtin_amdgcn_ballot_w64(hs < 0.0f) & lanes) == 0ull) return a.z + (t & a.x & 0x7fffu);   // the wrap of the i16 add is above the mask
    return a.z + (uint32_t)floor_mod_fast(wrap_i16((int32_t)t), (int32_t)hf, 0, approx_rcp(hf));
}

// Texel offset and light factor of one floor / ceiling pixel (visplanes.rs:108-126); see flat_texel_offset (raster_core.h) for the
// scalar form.  The factor is left unclamped: `as u8` of (colour x negative) is 0, the same as with the reference's
// `if factor < 0.0 { factor = 0.0 }`, and lightf - z / 4096 as one fma is exact because z / 4096 is.
// After the two quotients, the same arithmetic for every flat pixel: the rotation with the frame's cos / sin from VGPRs, ONE
// v_cvt_pk_i16_i32 that does both `as i16` saturations of (rx, ry) and packs them (NaN has become 0 in v_cvt_i32_f32 before), position
// added and `& 63` on both halves at once (both addends are masked first, so the sum of two 6-bit values stays inside its half and the
// low six bits are those of the wrapping i16 sum).  zf = wx as i16 as f32, from the caller.
__device__ __forceinline__ uint32_t flat_tail(const FrameVgprs &T, const uint4 a, const uint4 b, float wx, float wy, float zf, float &factor) {
    const float rx = wx * T.cos_a - wy * T.sin_a;
    const float ry = wy * T.cos_a + wx * T.sin_a;
    const int32_t ix = f32_as_i32(rx), iy = f32_as_i32(ry);
    uint32_t pk;
    asm("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(pk) : "v"(ix), "v"(iy));
    const uint32_t t = ((pk & 0x003f003fu) + T.pos_pk) & 0x003f003fu;     // tx | ty << 16
    factor = __builtin_fmaf(-zf, 1.0f / 4096.0f, bits_f32(b.y));
    return a.z + (((t >> 10) & 0xfc0u) | (t & 0x3fu));
}
__device__ __forceinline__ uint32_t flat_offset_tile(const FrameVgprs &T, const uint4 a, const uint4 b, const RowConsts &R, unsigned long long lanes, float &factor) {
    float wx, wy;
    if ((__builtin_amdgcn_ballot_w64((b.z & 0x100u) == 0u || R.vy == 0.0f) & lanes) == 0ull) {
        wx = div_prepared_nofix(bits_f32(b.x), R.vy, R.r_vy);
        wy = div_prepared_nofix(bits_f32(a.y), R.vy, R.r_vy);
    } else {                                      // a numerator outside the verified domain or the vy == 0 row somewhere in the wave: plain divides
        wx = bits_f32(b.x) / R.vy;
        wy = bits_f32(a.y) / R.vy;
    }
    return flat_tail(T, a, b, wx, wy, (float)f32_as_i16(wx), factor);          // (wx may be NaN here: 0 / 0 on the horizon row)
}

// The same two mappers for a span that is plain (w0_plain) — a, b are its staged words, the same in every lane (an LDS broadcast), and
// nothing has to be voted on.  Written out instruction by instruction: hipcc fuses the shifts, masks and adds below into SDWA /
// three-operand forms (fewer instructions, but each of them takes an issue slot of its own: see the note on the two prices above).
// Plain wall: the bitmap height is a power of two AND h + ay * uy1 stays inside i16 on the span's rows (resolve_wall_span checks both
// ends; the expression is monotonic in y), so `as i16` is the bare conversion: 11 simple instructions + 1 conversion.
__device__ __forceinline__ uint32_t wall_offset_plain(const uint4 a, const uint4 b, const RowConsts &R) {
    uint32_t o, t0, t1;
    asm("v_sub_f32 %1, %3, %8\n\t"            // n = y - top_y                                   (bitmap_render.rs:256)
        "v_mul_f32 %2, %1, %10\n\t"           // q = n * (1 / d)
        "v_fma_f32 %1, -%5, %2, %1\n\t"       // e = n - d * q
        "v_fmac_f32 %2, %1, %10\n\t"          // ay = q + e * (1 / d): the correctly rounded n / d (div_prepared_nofix)
        "v_mul_f32 %2, %2, %7\n\t"            // ay * uy1
        "v_add_f32 %2, %2, %9\n\t"            // h + ay * uy1                                    (:257)
        "v_cvt_i32_f32 %2, %2\n\t"            // as i16 (in range)
        "v_lshrrev_b32 %1, 16, %4\n\t"        // off_y
        "v_add_u32 %2, %2, %1\n\t"            // ty + off_y (the wrap of the i16 add is above the mask)
        "v_and_b32 %1, 0xffff, %4\n\t"        // h - 1
        "v_and_b32 %2, %2, %1\n\t"            // % h                                             (:259-263)
        "v_add_u32 %0, %2, %6"                  // + start of the texture column
        : "=v"(o), "=&v"(t0), "=&v"(t1)
        : "v"(R.yf), "v"(a.x), "v"(a.y), "v"(a.z), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
    return o;
}
// Plain floor / ceiling: numerators inside the prepared divide's domain, vy != 0 and light_level / 255 < 7 (resolve_flat_span), so wx is
// finite, `wx as i16 as f32` is trunc(wx) clamped in f32, and its upper clamp cannot matter: wx >= 32767 makes the factor negative
// (below light - 7.99) and the pixel black either way.  20 simple instructions + 5 of the other kind.
__device__ __forceinline__ uint32_t flat_offset_plain(const FrameVgprs &T, const uint4 a, const uint4 b, const RowConsts &R, float &factor) {
    uint32_t o, t0, t1, t2, t3;
    float fac = bits_f32(b.y);
    asm("v_mul_f32 %1, %6, %10\n\t"           // wx = gwz / vy                                   (visplanes.rs:113)
        "v_fma_f32 %2, -%9, %1, %6\n\t"
        "v_fmac_f32 %1, %2, %10\n\t"
        "v_mul_f32 %2, %7, %10\n\t"           // wy = wz * vx / vy                               (:114)
        "v_fma_f32 %3, -%9, %2, %7\n\t"
        "v_fmac_f32 %2, %3, %10\n\t"
        "v_mul_f32 %3, %1, %11\n\t"           // rx = wx * cos - wy * sin                        (:116)
        "v_mul_f32 %4, %2, %12\n\t"
        "v_sub_f32 %3, %3, %4\n\t"
        "v_mul_f32 %4, %2, %11\n\t"           // ry = wy * cos + wx * sin                        (:117)
        "v_mul_f32 %2, %1, %12\n\t"
        "v_add_f32 %4, %4, %2\n\t"
        "v_cvt_i32_f32 %3, %3\n\t"
        "v_cvt_i32_f32 %4, %4\n\t"
        "v_cvt_pk_i16_i32 %3, %3, %4\n\t"     // rx as i16 | ry as i16 << 16
        "v_trunc_f32 %1, %1\n\t"              // z = wx as i16 as f32                            (:123-124)
        "v_max_f32 %1, 0xc7000000, %1\n\t"
        "v_fmac_f32 %5, 0xb9800000, %1\n\t"   // factor = light / 255 - z / 4096                 (bitmap_render.rs:198-201)
        "v_and_b32 %3, 0x3f003f, %3\n\t"
        "v_add_u32 %3, %3, %13\n\t"           // + position, both halves                         (visplanes.rs:119-120)
        "v_lshrrev_b32 %4, 10, %3\n\t"
        "v_and_b32 %4, 0xfc0, %4\n\t"         // ty * 64
        "v_and_b32 %3, 63, %3\n\t"            // tx
        "v_or_b32 %3, %3, %4\n\t"
        "v_add_u32 %0, %3, %8"                  // + offset of the flat
        : "=v"(o), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(fac)
        : "v"(b.x), "v"(a.y), "v"(a.z), "v"(R.vy), "v"(R.r_vy), "v"(T.cos_a), "v"(T.sin_a), "v"(T.pos_pk));
    factor = fac;
    return o;
}
// Spans are addressed by their BYTE offset in the staging area (32 x record number): the owner of a row is the largest offset
// among the opaque spans that cover it, 0 = the "nothing" record.  The five low bits of such an offset are free and carry what a
// wave-uniform loop wants to branch on without looking at the record: the span's kind (bits 0-1) and its plain flag (bit 2).
constexpr uint32_t OFF_KIND = 3u, OFF_PLAIN = 4u, OFF_ADDR = ~31u;
//
// The span loops below are wave-uniform walks over a 64-bit mask whose bit i says "the span held by lane i is to be looked at".
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
    return reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(staged) + (off & OFF_ADDR));
}
// Both halves of the staged record at LDS address `addr`, the same address in every lane (an LDS broadcast): two 16-byte reads off one
// address register, waited for here (the record's words are what the next instructions need).
__device__ __forceinline__ void lds_record(uint32_t addr, uint4 &a, uint4 &b) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr));   // (no "memory": the compiler would wait for the column's gather in flight first)
}
__device__ __forceinline__ uint32_t lds_address(const void *p) { return (uint32_t)(uintptr_t)p; }   // the low half of a flat LDS address is the LDS offset

// One span's (texel offset, light factor) for all 64 rows from its broadcast record; `flags` = the low bits of its staging offset.
template <bool FLATS>           // (a possibly-transparent span is never a floor / ceiling: its loop leaves that mapper out)
__device__ __forceinline__ uint32_t span_texel(const FrameVgprs &T, const uint4 a, const uint4 b, uint32_t flags, const RowConsts &R, bool tile_vy0, float &factor) {
    const uint32_t kind = flags & OFF_KIND;
    if (kind == SPAN_WALL) {
        factor = bits_f32(a.w);
        return (flags & OFF_PLAIN) ? wall_offset_plain(a, b, R) : wall_offset_tile(a, b, R, ~0ull);
    }
    if (FLATS && kind == SPAN_FLAT) return ((flags & OFF_PLAIN) && !tile_vy0) ? flat_offset_plain(T, a, b, R, factor) : flat_offset_tile(T, a, b, R, ~0ull, factor);
    factor = bits_f32(a.w) * sky_fac_of(R);       // sky: plain lookup, no lighting (x 1.0 is exact); 0.0 where the reference would index outside the bitmap
    return a.z + sky_row_of(R);
}

// Stage 1 of a column that has no sole plain owner: its opaque spans that touch the tile's rows, in draw order, each evaluated with ITS
// mapper on all 64 rows from one broadcast record and selected into the rows it covers — a later span overwrites an earlier one like
// the reference's Pixels::set.  No owner search followed by a per-row record fetch and a vote on kinds: a chunk with a ceiling and a
// wall costs one flat and one wall evaluation of the short kind, and both read their parameters as broadcasts.  Rows nothing covers
// keep texel 0 with factor 0 -> 0,0,0.
__device__ __forceinline__ uint32_t span_loop(const FrameVgprs &T, uint32_t lspans_addr, unsigned long long m, uint32_t v_lo, uint32_t v_rg, uint32_t v_off,
                                              const RowConsts &R, bool tile_vy0, float &factor_out, uint32_t &winner_out) {
    uint32_t o = 0, winner = 0;
    float factor = 0.0f;
    while (m) {
        const int j = take_lowest(m);
        const uint32_t off = bcast(v_off, j);
        uint4 a, b;
        lds_record(lspans_addr + (off & OFF_ADDR), a, b);
        float sf;
        const uint32_t so = span_texel<true>(T, a, b, off, R, tile_vy0, sf);
        const bool in = ((uint32_t)R.y - bcast(v_lo, j)) <= bcast(v_rg, j);
        o = in ? so : o;
        factor = in ? sf : factor;
        winner = in ? off : winner;
    }
    factor_out = factor;
    winner_out = winner;
    return o;
}

// Possibly-transparent spans in draw order (bitmap columns — masked walls, sprites — and sky-with-holes spans): where such a span
// shows — its texel is opaque and no later opaque span owns the row — its (texel, light factor) replace the row's; the pixel is
// shaded once, afterwards.  (Fetching two spans' texels per trip without exec masking was measured: slower at 1280x800 — most columns
// meet one such span — and no faster at 320x200.)
__device__ __forceinline__ void overlay_loop(const RasterParams &P, const FrameVgprs &T, uint32_t lspans_addr, unsigned long long m,
                                             uint32_t v_lo, uint32_t v_rg, uint32_t v_off, const RowConsts &R, uint32_t winner, uint32_t &tex_io, float &factor_io) {
    while (m) {
        const int j = take_lowest(m);
        const uint32_t off = bcast(v_off, j);
        if (((uint32_t)R.y - bcast(v_lo, j)) <= bcast(v_rg, j) && off > winner) {
            uint4 a, b;
            lds_record(lspans_addr + (off & OFF_ADDR), a, b);
            const bool wall = (off & OFF_KIND) == SPAN_WALL;                        // wave-uniform
            float factor;
            const uint32_t o = span_texel<false>(T, a, b, off, R, false, factor);
            uint32_t tex, opq;
            gather_u8x2(P.scene.texel_idx, P.scene.texel_opq, o, tex, opq);
            const bool shows = opq != 0u && (wall || factor != 0.0f);
            tex_io = shows ? tex : tex_io;
            factor_io = shows ? factor : factor_io;
        }
    }
}

// Every row evaluates the owner it was given (columns with more than eight spans, and the eight-rows-per-pass form of a tile with few
// live rows): per-lane records.  There is no divergent control flow here: a kind that some row of the wave needs is computed by ALL 64
// lanes (on words of another kind the arithmetic is garbage but harmless) and each lane then selects — two uniform branches and a few
// selects instead of nested exec-mask regions.  The gather comes after the select, so every address is that of the lane's real owner.
__device__ __forceinline__ uint32_t owner_texel(const FrameVgprs &T, const uint4 *staged, uint32_t winner, const RowConsts &R, float &factor_out) {
    const uint4 a = span_at(staged, winner)[0], b = span_at(staged, winner)[1];
    const bool is_wall = staged_is_wall(b), is_sky = !is_wall && (int32_t)a.x < 0;
    const unsigned long long m_wall = __builtin_amdgcn_ballot_w64(is_wall), m_sky = __builtin_amdgcn_ballot_w64(is_sky);
    uint32_t o = a.z + sky_row_of(R);             // sky bitmap without holes, or nothing: plain lookup, no lighting (x 1.0 is exact)
    float factor = bits_f32(a.w) * sky_fac_of(R);
    if (~(m_wall | m_sky) != 0ull) {              // some row is owned by a floor / ceiling
        float ff;
        const uint32_t fo = flat_offset_tile(T, a, b, R, ~(m_wall | m_sky), ff);
        const bool is_flat = !is_wall && !is_sky;
        o = is_flat ? fo : o;
        factor = is_flat ? ff : factor;
    }
    if (m_wall != 0ull) {
        const uint32_t wo = wall_offset_tile(a, b, R, m_wall);
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
    v_off = off | w0_kind(w0) | (w0_plain(w0) ? OFF_PLAIN : 0u);
}

// A column is rendered in two stages so that the texel gather of one column is in flight while the next column's owners are worked
// out (tile_body interleaves them):  stage 1 = the rows' (texel offset, light factor);  stage 2 = palette + shade of the gathered
// texel, then the possibly-transparent spans on top.
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
__device__ __forceinline__ void big_column_overlays(const RasterParams &P, const FrameVgprs &T, const uint32_t *lw0, uint32_t lspans_addr, uint32_t off0, uint32_t n, int lane, int y0,
                                                    const RowConsts &R, uint32_t winner, uint32_t &tex_io, float &factor_io) {
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const uint32_t w0v = i < n ? lw0[i] : 0u;
        const bool hit = i < n && w0_cbot(w0v) >= y0 && w0_ctop(w0v) <= y0 + (TILE_H - 1) && w0_immediate(w0v);
        const unsigned long long m = __ballot(hit);
        if (!m) continue;
        uint32_t v_lo, v_rg, v_off;
        unpack_span(w0v, off0 + 32u * i, v_lo, v_rg, v_off);
        overlay_loop(P, T, lspans_addr, m, v_lo, v_rg, v_off, R, winner, tex_io, factor_io);
    }
}

constexpr int TILE_TS = 65;       // dwords per tile COLUMN in LDS
constexpr int PACK_ROWS = 8;      // a tile with no more live rows than this is rendered eight columns per wavefront pass

struct TileLds {
    uint32_t tile[TILE_W * TILE_TS];        // [col][row]: conflict-free for lane = row writes
    uint4 lspans[(SPAN_CAP + 1) * 2];       // record 0: "nothing"; records 1 ..: the staged spans
    uint32_t lw0[SPAN_CAP];
    float4 pal[256];                        // r, g, b as f32
    uint32_t lcoff[TILE_W + 1];
};

// The tile rows [ty_begin, ty_end) of one 64-column strip of frame f (columns x0 ..): the strip's spans, the palette and the column
// offsets are staged ONCE, then the 64 x 64 tiles are rendered one after another out of the same staging area.  What a tile costs
// besides its pixels — the load -> LDS -> barrier chain in front, the workgroup launch and drain around it — is paid once per strip
// segment instead of once per tile (profiles/r03_raster_tiles.md: that fixed part was two thirds of the kernel's time).
// W4: the frame width is a multiple of 4 (every BASELINE size, the reference's native 1024): a group of four pixels never straddles the
// right edge and every row starts on a dword — the read-out stores dwords.  Any other width takes byte stores (dg_raster_tiles_anyw).
template <bool W4>
__device__ __forceinline__ void strip_body(const RasterParams &P, TileLds &L, int f, int x0, int ty_begin, int ty_end) {
    const DevFrame fr = P.frames[f];
    const int W = P.k.W, H = P.k.H;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const uint4 *gspans = reinterpret_cast<const uint4 *>(P.rspans + fr.span_base);
    const uint32_t lspans_addr = lds_address(L.lspans), pal_addr = lds_address(L.pal);

    // The spans of adjacent columns are one contiguous range of the column-major span array: [col_off[x0], col_off[x0 + 64]).  The two
    // ends are wave-uniform (scalar loads); when the range fits in LDS — the normal case — every thread fetches its span straight
    // away, together with its palette entry (already f32x4 in HBM) and the 65 column offsets, and ONE barrier publishes all of it.
    // The records are in their per-pixel form already (raster_core.h resolve_*_span): staging is a copy (stage_words: two words of a wall).
    const uint32_t t_first = coff[x0 < W ? x0 : W], t_last = coff[x0 + TILE_W < W ? x0 + TILE_W : W];
    const bool fits = t_last - t_first <= (uint32_t)SPAN_CAP;
    {
        uint4 sa = make_uint4(0u, 0u, 0u, 0u), sb = sa;
        const bool mine = fits && threadIdx.x < t_last - t_first;
        if (mine) {
            sa = gspans[2 * ((size_t)t_first + threadIdx.x)];
            sb = gspans[2 * ((size_t)t_first + threadIdx.x) + 1];
        }
        float4 pal_v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (threadIdx.x < 256) pal_v = reinterpret_cast<const float4 *>(P.scene.palette_f32)[threadIdx.x];
        const int xc = x0 + (int)(threadIdx.x <= TILE_W ? threadIdx.x : 0);
        const uint32_t coff_v = coff[xc < W ? xc : W];
        if (threadIdx.x < 256) L.pal[threadIdx.x] = pal_v;
        if (threadIdx.x <= TILE_W) L.lcoff[threadIdx.x] = coff_v;
        if (mine) {
            L.lw0[threadIdx.x] = sa.x;
            stage_words(sa, sb);
            L.lspans[2 * threadIdx.x + 2] = sa;
            L.lspans[2 * threadIdx.x + 3] = sb;
        }
        if (threadIdx.x == THREADS - 1) {         // the "nothing" record: a sky span with factor 0
            L.lspans[0] = make_uint4(0x80000000u, 0u, 0u, 0u);
            L.lspans[1] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    auto row_consts = [&](int yy, const uint4 t) {
        RowConsts r;
        r.y = yy;
        r.yf = (float)yy;
        r.r_vy = bits_f32(t.x);
        r.vy = bits_f32(t.y);
        r.sky = (t.z & 0xffu) | t.w;
        return r;
    };
    // The frame constants the flat mapper uses per pixel, in VGPRs (the asm keeps the compiler from folding them back into scalar operands)
    FrameVgprs T;
    {
        const uint32_t pk = ((uint32_t)fr.pos_x_i16 & 63u) | (((uint32_t)fr.pos_y_i16 & 63u) << 16);
        asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=&v"(T.cos_a), "=&v"(T.sin_a), "=&v"(T.pos_pk) : "s"(fr.cos_a), "s"(fr.sin_a), "s"(pk));
    }
    uint8_t *fb = P.fb + (size_t)f * (size_t)3 * (size_t)W * (size_t)H;
    __syncthreads();

  for (int ty = ty_begin; ty < ty_end; ty++) {
    const int y0 = ty * TILE_H;
    const int y = y0 + lane;
    const uint4 rt = P.row_tab[y < H ? y : H - 1];                    // dg_row_table (prefetching the next tile's under the barrier: measured, neutral)
    // A tile with at most 8 live rows (the last tile row of a 200-row frame): a wave can take all eight of its columns in ONE pass,
    // lane = (column, row) = (lane >> 3, lane & 7).  Its row constants:
    const bool few_rows = H - y0 <= PACK_ROWS;
    const RowConsts R = row_consts(y, rt);
    const bool tile_vy0 = __builtin_amdgcn_ballot_w64(R.vy == 0.0f) != 0ull;   // the horizon row (vy == 0) is in this tile

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
                stage_words(a, b);
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
        const bool f_op = f_hit && !w0_immediate(f_w0), f_ov = f_hit && w0_immediate(f_w0);
        const unsigned long long hit_op = __ballot(f_op);
        const unsigned long long big = __ballot(f_n > 8u);         // all 8 lanes of a column with more than 8 spans
        // Columns with ONE opaque owner: the last OPAQUE span that touches the tile's rows is plain and covers all of the tile's live
        // rows — whatever lies under it in draw order cannot show.  Such a column (two out of three in the benchmark scene) needs no
        // range test and no select: ucol_* carry one bit per column, at the owning span's lane.
        // Possibly-transparent spans drawn after it (a sprite in front of a wall) are laid on top in stage 2 as usual; those drawn
        // before it are dropped here.
        const uint32_t my_ops = (uint32_t)(hit_op >> (lane & ~7)) & 0xffu;
        const bool f_last = (my_ops >> (fslot + 1)) == 0u;
        const bool f_sole = f_op && f_last && w0_plain(f_w0) && w0_ctop(f_w0) <= y0 && w0_cbot(f_w0) >= (y0 + (TILE_H - 1) < H ? y0 + (TILE_H - 1) : H - 1);
        const unsigned long long ucol_wall = __ballot(f_sole && w0_kind(f_w0) == SPAN_WALL);   // (a bitmap height that is not a power of two, a texture row beyond i16: span_loop)
        const unsigned long long ucol_flat = tile_vy0 ? 0ull : __ballot(f_sole && w0_kind(f_w0) == SPAN_FLAT);   // (the vy == 0 row, like numerators outside the prepared divide's domain: span_loop)
        const unsigned long long ucol = ucol_wall | ucol_flat;
        const uint32_t my_ucol = (uint32_t)(ucol >> (lane & ~7)) & 0xffu;                      // the sole owner of my column, if it has one
        const bool under = my_ucol != 0u && (1u << fslot) < my_ucol;                           // drawn before it: cannot show
        const unsigned long long hit_ov = __ballot(f_ov && !under);
        const unsigned long long walk2 = big | hit_ov;                        // columns whose stage 2 has more to do than shading
        const int nk = (c_hi - c_lo - wave + WAVES - 1) / WAVES;     // columns of this chunk that are this wave's
        if (few_rows && nk == WAVES && walk2 == 0ull) {
            // ---- eight columns x eight rows in one pass.  Lane (k, r) owns row r of the wave's column k; that column's spans sit in the
            // eight lanes of its own group (the pre-filter's layout), so the owner search is eight lane-permutes of one packed word
            // (first row | row count - 1 << 16; a span that does not count can never match).
            const int yp = y0 + (lane & (PACK_ROWS - 1));
            const RowConsts Rp = row_consts(yp, P.row_tab[yp < H ? yp : H - 1]);
            const uint32_t mine_word = (f_hit && !w0_immediate(f_w0)) ? (v_lo | (v_rg << 16)) : 0x0000ffffu;
            const uint32_t off_first = (v_off & OFF_ADDR) - 32u * (uint32_t)fslot;          // staging offset of the column's first span
            uint32_t winner = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~7) + j) * 4, (int)mine_word);
                winner = ((uint32_t)Rp.y - (w & 0xffffu)) <= (w >> 16) ? off_first + 32u * (uint32_t)j : winner;
            }
            float factor;
            const uint32_t o = owner_texel(T, L.lspans, winner, Rp, factor);
            const uint32_t px = shade_f(L.pal[P.scene.texel_idx[o]], factor);
            L.tile[(c_lo + wave + WAVES * fk) * TILE_TS + fslot] = px;
            c_lo = c_hi;
            if (c_lo < TILE_W) __syncthreads();
            continue;
        }
        // Two columns in flight per wave: stage 1 of a column and its gather are issued before stage 2 of the column before it.
        struct Col { uint32_t tex, winner; float factor; };
        uint32_t *const tcol = &L.tile[(c_lo + wave) * TILE_TS + lane];       // this lane's row in the wave's first column; column k is k * 8 columns on
        auto stage1 = [&](int k, Col &C) {
            const unsigned long long colmask = 0xffull << (8 * k);
            const unsigned long long mu = ucol & colmask;
            uint32_t o;
            if (mu) {                                                // one plain owner for all rows of this column
                const uint32_t off = bcast(v_off, __builtin_ctzll(mu));
                uint4 a, b;
                lds_record(lspans_addr + (off & OFF_ADDR), a, b);
                if (ucol_wall & mu) { o = wall_offset_plain(a, b, R); C.factor = bits_f32(a.w); }
                else o = flat_offset_plain(T, a, b, R, C.factor);
                C.winner = off;                                      // (stage 2 lays later possibly-transparent spans on top)
            } else {
                if (big & colmask) {
                    const uint32_t n0 = bcast(f_n0, 8 * k), n = bcast(f_n, 8 * k);
                    C.winner = big_column_owner(L.lw0 + n0, 32u * (n0 + 1u), n, lane, y0, R);
                } else {
                    C.winner = owner_loop(hit_op & colmask, v_lo, v_rg, v_off, R, 0u);
                }
                o = owner_texel(T, L.lspans, C.winner, R, C.factor);
            }
            C.tex = P.scene.texel_idx[o];                            // in flight until stage 2
        };
        auto stage2 = [&](int k, const Col &C) {
            const unsigned long long colmask = 0xffull << (8 * k);
            uint32_t tex = C.tex;
            float factor = C.factor;
            if (walk2 & colmask) {                                   // possibly-transparent spans on top, in draw order
                if (big & colmask) {
                    const uint32_t n0 = bcast(f_n0, 8 * k), n = bcast(f_n, 8 * k);
                    big_column_overlays(P, T, L.lw0 + n0, lspans_addr, 32u * (n0 + 1u), n, lane, y0, R, C.winner, tex, factor);
                } else {
                    overlay_loop(P, T, lspans_addr, hit_ov & colmask, v_lo, v_rg, v_off, R, C.winner, tex, factor);
                }
            }
            float4 c;                                                // palette entry: one 16-byte read (a 12-byte one costs twice the LDS cycles)
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(pal_addr + (tex << 4)));
            tcol[k * WAVES * TILE_TS] = shade_f(c, factor);          // palette x light, `as u8` (bitmap_render.rs:202-207), once per pixel
        };
        Col A, B;
        if (nk == WAVES) {                                           // a whole tile's worth (the normal case): one loop shape, no conditionals
            stage1(0, A);                                            // A and B alternate so that an in-flight texel never changes register
#pragma unroll                                                       // k a literal in every copy: the column masks and lane numbers fold (measured: -2.6 %)
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
    __syncthreads();

    // Read-out: groups of 4 pixels of a row (4 LDS words -> 12 B of RGB24), 16 groups per tile row; the 64 lanes of a wave take
    // 4 rows x 16 groups in an order that is conflict-free in LDS; 8 adjacent lanes write 96 contiguous bytes.
    const int gc = (lane & 7) | ((lane >> 5) << 3), rsub = (lane >> 3) & 3;
#pragma unroll
    for (int pass = 0; pass < TILE_H / (4 * WAVES); pass++) {
        const int row = pass * 4 * WAVES + wave * 4 + rsub;
        const int yy = y0 + row, xx = x0 + 4 * gc;
        if (yy < H && xx < W) {
            const uint32_t p0 = L.tile[(4 * gc + 0) * TILE_TS + row], p1 = L.tile[(4 * gc + 1) * TILE_TS + row];
            const uint32_t p2 = L.tile[(4 * gc + 2) * TILE_TS + row], p3 = L.tile[(4 * gc + 3) * TILE_TS + row];
            if constexpr (W4) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(fb + ((size_t)yy * (size_t)W + (size_t)xx) * 3);
                // (non-temporal: the frame is written once and not read again by this launch — stored plainly its 3 MB per frame pass through
                // the XCD's 4 MB L2 as dirty lines and push out the span records and texture lines the next tiles want: - 4.5 % at 1280x800)
                __builtin_nontemporal_store((p0 & 0xffffffu) | (p1 << 24), dst + 0);
                __builtin_nontemporal_store(((p1 >> 8) & 0xffffu) | (p2 << 16), dst + 1);
                __builtin_nontemporal_store(((p2 >> 16) & 0xffu) | (p3 << 8), dst + 2);
            } else {
                uint8_t *dst = fb + ((size_t)yy * (size_t)W + (size_t)xx) * 3;
                const uint32_t px[4] = {p0, p1, p2, p3};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (xx + j >= W) break;
                    __builtin_nontemporal_store((uint8_t)px[j], dst + 3 * j + 0);
                    __builtin_nontemporal_store((uint8_t)(px[j] >> 8), dst + 3 * j + 1);
                    __builtin_nontemporal_store((uint8_t)(px[j] >> 16), dst + 3 * j + 2);
                }
            }
        }
    }
    if (ty + 1 < ty_end) __syncthreads();     // the LDS tile (and, when the spans did not fit, the staging area) is written again
  }
}

// Which (frame, strip, segment) a workgroup renders.  Dispatch order (blockIdx) deals a frame's workgroups over all eight XCDs — blocks b and
// b + 8 are observed to share one (MI355X_MICROARCH.md: placement is not a contract, so this is for speed only: any bijection renders the
// same frames).  With P.frame_per_xcd the workgroups of one frame go to ONE XCD instead, the frames dealt round robin: every XCD then renders
// whole frames — an even share of the work whatever the strips cost, where dispatch order gives an XCD the SAME few strips of every frame
// (at 1024 columns, 16 strips: two of them) — and neighbouring strips, which sample the same wall textures and flats, find each other's
// lines in that XCD's L2 (a strip's two segments read the same span records: once from HBM instead of twice).  Measured, 1 000-frame batches
// (profiles/r05_raster_tiles.md sections 5g, 5i), with the non-temporal frame stores of the read-out: 1024x768 - 7 %, 800x600 - 4 %,
// 320x200 - 2 %, 1280x800 and 640x400 0 (before those stores, when the written frame still pushed spans and texels out of the L2: - 2.4 %),
// 1920x1080 + 1 %, 2560x1600 + 3 to + 6 % — launch_raster asks for it up to 1.1 M pixels.  Runs of 2 / 4 / 8 consecutive frames per XCD
// + 2 %.  The frames beyond the last multiple of eight keep dispatch order.
__device__ __forceinline__ void raster_block(const RasterParams &P, uint32_t &bx, uint32_t &by, uint32_t &f) {
    bx = blockIdx.x; by = blockIdx.y; f = blockIdx.z;
    if (!P.frame_per_xcd) return;
    const uint32_t gx = gridDim.x, gy = gridDim.y, pf = gx * gy;
    const uint32_t orig = bx + gx * (by + gy * f);
    if (orig >= ((uint32_t)P.n_frames & ~7u) * pf) return;
    const uint32_t xcd = orig & 7u, idx = orig >> 3;                   // the idx-th workgroup of its XCD: workgroup idx % pf of that XCD's frame idx / pf
    // (the two divisions as multiplications by launch_raster's reciprocals — exact for these ranges: a u32 division by a run-time value
    // is two dozen instructions per wave, and there are eight waves per workgroup)
    const uint32_t j = __umulhi(idx, P.xcd_rcp_pf), w = idx - j * pf;
    f = j * 8u + xcd;
    by = __umulhi(w, P.xcd_rcp_gx);
    bx = w - by * gx;
}

// (launch bounds: 8 waves per SIMD = at most 64 VGPRs; one more register costs a fourth of the resident workgroups, 0.58 -> 0.67 ms)
// One workgroup per (frame, 64-column strip, segment of P.tile_rows_per_wg tile rows).
__global__ __launch_bounds__(THREADS, 8) void dg_raster_tiles(RasterParams P) {
    __shared__ __attribute__((aligned(16))) TileLds L;
    const int n_tile_rows = (P.k.H + TILE_H - 1) / TILE_H;
    uint32_t bx, by, f;
    raster_block(P, bx, by, f);
    const int ty_begin = (int)by * P.tile_rows_per_wg;
    strip_body<true>(P, L, (int)f, (int)bx * TILE_W, ty_begin, min(n_tile_rows, ty_begin + P.tile_rows_per_wg));
}
// The same for a frame width that is not a multiple of 4 (constants.rs:3-17 makes any width legal).
__global__ __launch_bounds__(THREADS, 8) void dg_raster_tiles_anyw(RasterParams P) {
    __shared__ __attribute__((aligned(16))) TileLds L;
    const int n_tile_rows = (P.k.H + TILE_H - 1) / TILE_H;
    uint32_t bx, by, f;
    raster_block(P, bx, by, f);
    const int ty_begin = (int)by * P.tile_rows_per_wg;
    strip_body<false>(P, L, (int)f, (int)bx * TILE_W, ty_begin, min(n_tile_rows, ty_begin + P.tile_rows_per_wg));
}

// Per-row constants of the flat and sky mappers for one frame size: vy = CFY - y (visplanes.rs:109), its prepared reciprocal, the
// "prepared divide allowed" bit (not on the vy == 0 row) and the sky texture row (visplanes.rs:68-72; row 0 with factor 0 when it is
// outside the bitmap).  Same device code as the per-lane computation it replaces, run once per scene upload instead of once per tile.
//   x = bits of prepare_rcp(vy)   y = bits of vy   z = sky row | 0x100 when vy != 0   w = bits of the sky factor (1.0f / 0.0f)
__global__ void dg_row_table(DevScene scene, DevConsts k, uint4 *row_tab) {
    const int y = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (y >= k.H) return;
    const float vy = k.CFY - (float)y;
    const int32_t srow = sky_row(scene, k, y);
    row_tab[y] = make_uint4(f32_bits(prepare_rcp(vy)), f32_bits(vy), (srow < 0 ? 0u : (uint32_t)srow) | (vy != 0.0f ? 0x100u : 0u), f32_bits(srow < 0 ? 0.0f : 1.0f));
}

// Order-independent per-frame checksum (dg_frame_checksums): every dword is mixed with its index, the mixes are summed.
// Pure streaming read: 256 dwords per lane-iteration are coalesced, one 64-bit atomic add per wave.
// (A frame whose byte count is not a multiple of 4 — a width that is not — ends in a partial dword, zero-extended, and its frames do not
// start on dwords: bytes are loaded one by one then.)
__global__ __launch_bounds__(256) void dg_checksum(const uint8_t *fb, size_t frame_bytes, unsigned long long *out) {
    const uint8_t *b = fb + (size_t)blockIdx.y * frame_bytes;
    const uint32_t *d = reinterpret_cast<const uint32_t *>(b);
    const bool dwords = (frame_bytes & 3) == 0;
    const size_t n = (frame_bytes + 3) / 4;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v;
        if (dwords) v = d[i];
        else {
            v = 0;
            for (size_t k = 0; k < 4 && 4 * i + k < frame_bytes; k++) v |= (uint32_t)b[4 * i + k] << (8 * k);
        }
        unsigned long long m = ((unsigned long long)v ^ (i * 0x9E3779B97F4A7C15ull)) * 0xBF58476D1CE4E5B9ull;
        acc += m ^ (m >> 32);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.y], acc);
}

hipError_t launch_checksums(const uint8_t *fb, size_t frame_bytes, int count, unsigned long long *out, hipStream_t stream) {
    if (count <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>(256, ((frame_bytes + 3) / 4 + 255) / 256);
    hipLaunchKernelGGL(dg_checksum, dim3(blocks, (unsigned)count), dim3(256), 0, stream, fb, frame_bytes, out);
    return hipGetLastError();
}

hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint4 *row_tab, hipStream_t stream) {
    hipLaunchKernelGGL(dg_row_table, dim3((unsigned)((k.H + 255) / 256)), dim3(256), 0, stream, scene, k, row_tab);
    return hipGetLastError();
}

// Nothing to launch: the events still have to be recorded for whoever waits on them or reads their times.
static hipError_t record_pair(hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
    hipError_t e = hipSuccess;
    if (start) e = hipEventRecord(start, stream);
    if (e == hipSuccess && stop) e = hipEventRecord(stop, stream);
    return e;
}

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
    if (P.n_frames <= 0 || max_spans_per_frame == 0) return record_pair(stream, start, stop);
    dim3 grid((max_spans_per_frame + 255) / 256, (unsigned)P.n_frames);
    hipExtLaunchKernelGGL(dg_setup_spans, grid, dim3(256), 0, stream, start, stop, 0, P);
    return hipGetLastError();
}

// Tile rows one workgroup renders out of one staging pass.  Fewer, longer workgroups pay the staging chain (two scalar loads -> span
// loads -> LDS -> barrier) and the workgroup launch / drain less often, but a launch needs enough workgroups for its tail not to show
// (the chip holds 1 024 of them at a time) and the segments of a strip should be of equal length.  Measured (profiles/r04_raster_tiles.md):
// 1280x800 (13 tile rows) at 1 000 frames per launch 3 / 5 / 7 / 13 rows = 1.83 / 1.78 / 1.75 / 1.81 ms, at 250 frames 0.51 / 0.51 / 0.54 /
// 0.60 ms; 1024x768 (12) at 1 000 frames 3 / 5 / 6 / 7 / 9 = 1.60 / 1.54 / 1.50 / 1.69 / 2.13 ms (7 + 5 and 9 + 3 are unbalanced);
// 2560x1600 (25) at 250 frames 3 / 5 / 9 / 13 = 1.66 / 1.63 / 1.65 / 1.70 ms; 320x200 (4 tile rows) 2 rows win.
// Rule: the fewest equal segments per strip that still give ~30 000 workgroups, never fewer than 3 rows' worth for tall frames.
int raster_tile_rows_per_wg(int W, int H, int n_frames) {
    const int n_tile_rows = (H + TILE_H - 1) / TILE_H;
    if (n_tile_rows < 8) return std::min(2, n_tile_rows);
    const long long strips = (long long)((W + TILE_W - 1) / TILE_W) * (long long)std::max(1, n_frames);
    long long segments = (30000 + strips - 1) / strips;
    segments = std::max(1ll, std::min(segments, (long long)((n_tile_rows + 2) / 3)));
    return (int)((n_tile_rows + segments - 1) / segments);
}

hipError_t launch_raster(const RasterParams &P_in, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
    if (P_in.n_frames <= 0) return record_pair(stream, start, stop);
    RasterParams P = P_in;
    const int n_tile_rows = (P.k.H + TILE_H - 1) / TILE_H;
    if (P.tile_rows_per_wg <= 0) P.tile_rows_per_wg = raster_tile_rows_per_wg(P.k.W, P.k.H, P.n_frames);
    dim3 grid((unsigned)((P.k.W + TILE_W - 1) / TILE_W), (unsigned)((n_tile_rows + P.tile_rows_per_wg - 1) / P.tile_rows_per_wg), (unsigned)P.n_frames);
    P.frame_per_xcd = (size_t)P.k.W * (size_t)P.k.H <= 1100000 ? 1 : 0;               // (raster_block; DOOMGPU_FRAME_PER_XCD=0 / 1 overrides)
    if (const char *e = std::getenv("DOOMGPU_FRAME_PER_XCD")) P.frame_per_xcd = std::atoi(e) != 0 ? 1 : 0;
    {   // n / d = (n * (2^32 / d + 1)) >> 32 for n * d < 2^32: asked of n < workgroups of the launch, d = workgroups per frame / strips per frame
        const uint64_t gx = grid.x, pf = (uint64_t)grid.x * grid.y, total = pf * grid.z;
        if (gx < 2 || total * pf >= (1ull << 32)) P.frame_per_xcd = 0;
        else { P.xcd_rcp_pf = (uint32_t)((1ull << 32) / pf + 1); P.xcd_rcp_gx = (uint32_t)((1ull << 32) / gx + 1); }
    }
    if (P.k.W % 4 == 0) hipExtLaunchKernelGGL(dg_raster_tiles, grid, dim3(THREADS), 0, stream, start, stop, 0, P);
    else hipExtLaunchKernelGGL(dg_raster_tiles_anyw, grid, dim3(THREADS), 0, stream, start, stop, 0, P);
    return hipGetLastError();
}

}  // namespace dg
