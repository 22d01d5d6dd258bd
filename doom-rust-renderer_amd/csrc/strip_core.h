// strip_core.h — bodies of the strip rasteriser (dg_resolve_columns, dg_raster_strips: kernels.hip) as host/device inline
// functions, so that tests/emul can run the same code on the CPU.
//
// The reference's final pixel is "the last Pixels::set wins" over draw calls whose row ranges overlap by design (inclusive
// y ranges share boundary rows, visplanes cover wall edge rows: SURVEY.md Appendix A).  Every draw call whose texels are all
// opaque writes every row of its span, so for the draw-ordered PREFIX of such spans in a screen column the winner of a row
// depends on the row ranges alone.  resolve_column() computes that once per column: sorted, disjoint segments covering rows
// 0 .. H-1, each carrying the texture-mapping constants of its winning span (uncovered rows: SEG_NONE = the zeroed buffer of
// Pixels::new, pixels.rs:10-14).  dg_raster_strips then walks a column top to bottom with the current segment in registers
// and evaluates every pixel exactly once, with no per-pixel ownership test.  The spans from the first possibly-transparent
// one on (masked walls, sprites: whether they write depends on the texel, bitmap_render.rs:265) cannot be resolved like that:
// the 64 x 64 tiles they touch are rendered by dg_raster_tile_list, which replays all spans of its columns in draw order with
// an ownership walk per pixel.  Real frames are mostly made of tiles without such spans.
#pragma once
#include "raster_core.h"

namespace dg {

DG_HD int32_t seg_end(uint32_t w0) { return (int32_t)(w0 & 0x3fffu); }
DG_HD uint32_t seg_kind(uint32_t w0) { return w0 >> 30; }

// The segment a span turns into when it wins rows .. end (the first row is where the previous segment ended).
DG_HD DevSeg seg_from_span(const DevRSpan &sp, int32_t end, const DevScene &sc) {
    DevSeg o;
    const uint32_t kind = w0_kind(sp.w[0]);
    for (int i = 0; i < 8; i++) o.w[i] = 0;
    if (kind == SPAN_WALL) {
        o.w[0] = (uint32_t)end | ((uint32_t)SPAN_WALL << 30);
        o.w[1] = sp.w[1];
        o.w[2] = sp.w[2] + sp.w[7];                                       // row-major: bitmap + tx (+ ty * w per pixel)
        o.w[3] = sp.w[3];
        o.w[4] = sp.w[4];
        o.w[5] = sp.w[5];
        o.w[6] = sp.w[6];
        o.w[7] = f32_bits(prepare_rcp(bits_f32(sp.w[1])));
    } else if (kind == SPAN_FLAT) {
        o.w[0] = (uint32_t)end | ((uint32_t)SPAN_FLAT << 30);
        o.w[1] = sp.w[1];
        o.w[2] = sc.pool_flats + sp.w[2];
        o.w[4] = sp.w[4];
        o.w[5] = sp.w[5];
        o.w[6] = (sp.w[6] >> 8) & 1u;
    } else if (sp.w[7] != 0xffffffffu) {
        o.w[0] = (uint32_t)end | ((uint32_t)SPAN_SKY << 30);
        o.w[2] = sc.sky_texel_off + sp.w[7];
        o.w[3] = f32_bits(1.0f);                                          // draw_sky applies no lighting (visplanes.rs:74-76)
    } else {
        o.w[0] = (uint32_t)end | (SEG_NONE << 30);                        // sky column outside the bitmap: nothing is drawn
    }
    return o;
}
DG_HD DevSeg seg_none(int32_t end) {
    DevSeg o;
    for (int i = 0; i < 8; i++) o.w[i] = 0;
    o.w[0] = (uint32_t)end | (SEG_NONE << 30);
    return o;
}

struct ResolveResult {
    uint32_t n_segs;      // segments written (0xffffffff: more than `cap`, nothing usable was written)
    uint32_t n_base;      // spans [0, n_base) were resolved; spans [n_base, n) are the overlay, still in draw order
    int32_t ov_lo, ov_hi; // rows touched by the overlay spans (ov_lo > ov_hi: none)
};

// One screen column.  spans[0 .. n): the column's DevRSpans in draw order.  Segments go to seg_out[slot * seg_stride], the
// slot of the segment containing row b * band_rows to band_out[b * band_stride] for every band b.
// w0_at(j) returns word 0 (row range, flags) of span j — the kernel keeps those in LDS.
template <typename W0At>
DG_HD ResolveResult resolve_column(W0At w0_at, const DevRSpan *spans, uint32_t n, const DevScene &sc, int32_t H, int32_t band_rows, uint32_t cap,
                                   DevSeg *seg_out, size_t seg_stride, uint8_t *band_out, size_t band_stride) {
    ResolveResult res;
    uint32_t nb = n;
    for (uint32_t j = 0; j < n; j++)
        if (w0_immediate(w0_at(j))) { nb = j; break; }
    res.n_base = nb;
    res.ov_lo = 0x7fff; res.ov_hi = -1;
    for (uint32_t j = nb; j < n; j++) {
        const uint32_t w0 = w0_at(j);
        res.ov_lo = w0_ctop(w0) < res.ov_lo ? w0_ctop(w0) : res.ov_lo;
        res.ov_hi = w0_cbot(w0) > res.ov_hi ? w0_cbot(w0) : res.ov_hi;
    }
    uint32_t nseg = 0;
    int32_t row = 0, pend_owner = -2, pend_start = 0;
    DevRSpan pend_span = {};                          // the pending owner's words, requested when it becomes the owner: by the time
                                                      // its segment is flushed (a boundary scan later) the load has landed
    // One elementary interval per iteration: [row, next boundary).  Its owner is the last span covering `row`; adjacent
    // intervals with the same owner merge.  At most 2 * nb + 1 iterations of nb steps each; nb is 2-8 in real scenes.
    for (;;) {
        int32_t owner = -1, next = H;
        if (row < H) {
            for (uint32_t j = 0; j < nb; j++) {
                const uint32_t w0 = w0_at(j);
                const int32_t t = w0_ctop(w0), b = w0_cbot(w0);
                if (t <= row && row <= b) owner = (int32_t)j;
                if (t > row && t < next) next = t;
                if (b >= row && b + 1 < next) next = b + 1;
            }
        }
        if (owner != pend_owner || row >= H) {
            if (pend_owner != -2) {                                       // flush [pend_start, row - 1]
                if (nseg >= cap) { res.n_segs = 0xffffffffu; return res; }
                seg_out[(size_t)nseg * seg_stride] = pend_owner < 0 ? seg_none(row - 1) : seg_from_span(pend_span, row - 1, sc);
                for (int32_t b = (pend_start + band_rows - 1) / band_rows; b * band_rows < row; b++) band_out[(size_t)b * band_stride] = (uint8_t)nseg;
                nseg++;
            }
            pend_owner = owner; pend_start = row;
            if (owner >= 0) pend_span = spans[owner];
        }
        if (row >= H) break;
        row = next;
    }
    res.n_segs = nseg;
    return res;
}

// ---- per pixel (lane = column: every word is per lane, the row is wave-uniform) ----------------------------------------

// Pool offset of one wall pixel, row-major (bitmap_render.rs:256-263).
DG_HD uint32_t seg_wall_offset(uint32_t w1, uint32_t w2, uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7, int32_t y) {
    const int32_t h = (int32_t)(w6 & 0xffffu), w = (int32_t)(w6 >> 16);
    return w2 + (uint32_t)(wall_texel_row(bits_f32(w1), bits_f32(w7), bits_f32(w4), w5, h, y) * w);
}
// Pool offset and light factor of one floor / ceiling pixel (visplanes.rs:108-126); vy = CFY - y, r_vy = prepare_rcp(vy).
// The factor is NOT clamped at 0 here: shade() converts with saturation, so a negative factor yields 0 like the clamped one.
DG_HD uint32_t seg_flat_offset(const DevFrame &f, uint32_t w1, uint32_t w2, uint32_t w4, uint32_t w5, uint32_t w6, float vy, float r_vy, float &factor) {
    float wx, wy;
    if (w6 & 1u) {
        wx = div_prepared(bits_f32(w4), vy, r_vy);
        wy = div_prepared(bits_f32(w1), vy, r_vy);
    } else {
        wx = bits_f32(w4) / vy;
        wy = bits_f32(w1) / vy;
    }
    const float rx = wx * f.cos_a - wy * f.sin_a;
    const float ry = wy * f.cos_a + wx * f.sin_a;
    const int32_t tx = (f32_as_i16(rx) + f.pos_x_i16) & 63;
    const int32_t ty = (f32_as_i16(ry) + f.pos_y_i16) & 63;
    // diminish_color (bitmap_render.rs:190-201): light/255 - distance * (1/4096).  distance is an i16 and 1/4096 a power of
    // two, so the product is exact and the fused form rounds exactly like the reference's separate multiply and subtract.
    factor = __builtin_fmaf(-(float)f32_as_i16(wx), 1.0f / (16.0f * 256.0f), bits_f32(w5));
    return w2 + (uint32_t)(ty * 64 + tx);
}

}  // namespace dg
