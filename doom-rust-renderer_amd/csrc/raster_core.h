// raster_core.h — the per-column and per-pixel arithmetic of the reference's three texture mappers, as
// host/device inline functions.  The HIP kernels (kernels.hip) are the only product code that calls them;
// tests/emul compiles the same bodies on the CPU to check list generation without a GPU.
//
//   wall / masked wall / sprite column : render_vertical_bitmap_line  src/renderer/bitmap_render.rs:213-276
//   floor / ceiling                    : draw_visplane                src/renderer/visplanes.rs:94-130
//   sky                                : draw_sky                     src/renderer/visplanes.rs:42-80
//   lighting                           : diminish_color               src/renderer/bitmap_render.rs:190-208
//
// Every expression keeps the reference's operand order; nothing may be contracted into an FMA.
#pragma once
#include "lists_dev.h"
#include "rust_num.h"

namespace dg {

// diminish_color's factor for a sector light (already divided by 255) and an i16 distance.
DG_HD float light_factor(float lightf, int32_t distance_i16) {
    float factor = lightf - (float)distance_i16 * (1.0f / (16.0f * 256.0f));
    return factor < 0.0f ? 0.0f : factor;
}

// palette entry (r | g<<8 | b<<16) times factor, each channel `as u8`; returns r | g<<8 | b<<16.
DG_HD uint32_t shade(uint32_t rgbx, float factor) {
    int32_t r = f32_as_u8((float)(rgbx & 255u) * factor);
    int32_t g = f32_as_u8((float)((rgbx >> 8) & 255u) * factor);
    int32_t b = f32_as_u8((float)((rgbx >> 16) & 255u) * factor);
    return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
}

// Column-invariant part of render_vertical_bitmap_line (bitmap_render.rs:241-251): texture column + light factor.
DG_HD DevSpanAux wall_column_setup(const DevWallRec &r, int32_t x) {
    float ax = (float)(x - r.start_x) / r.dxf;
    float oma = 1.0f - ax;
    float den = oma * r.C + ax * r.D;
    int32_t tx = f32_as_i16((oma * r.A + ax * r.B) / den);
    tx = wrap_i16(tx + r.off_x);
    tx = floor_mod_i16(tx, r.w);
    int32_t z = f32_as_i16((oma + ax) / den);
    DevSpanAux a;
    a.texcol = r.texel_off + (uint32_t)tx * (uint32_t)r.h;
    a.factor = light_factor(r.lightf, z);
    return a;
}

// Texture row of one wall pixel (bitmap_render.rs:256-263).
DG_HD int32_t wall_texture_row(const DevWallRec &r, int32_t top_y, int32_t bot_y, int32_t y) {
    float ay = (float)(y - top_y) / (float)(bot_y - top_y);
    int32_t ty = f32_as_i16((float)r.h + (1.0f - ay) * 0.0f + ay * r.uy1);
    ty = wrap_i16(ty + r.off_y);
    return floor_mod_i16(ty, r.h);
}

// One wall / masked / sprite pixel.  Returns false for a transparent texel (`None`, bitmap_render.rs:265).
DG_HD bool wall_pixel(const DevScene &sc, const DevWallRec &r, const DevSpanAux &a, int32_t top_y, int32_t bot_y, int32_t y, uint32_t &rgb) {
    uint32_t o = a.texcol + (uint32_t)wall_texture_row(r, top_y, bot_y, y);
    if (r.has_holes && !sc.texel_opq[o]) return false;
    rgb = shade(sc.palette[sc.texel_idx[o]], a.factor);
    return true;
}

// Per-column part of draw_visplane: vx (visplanes.rs:108).
DG_HD float flat_column_vx(const DevConsts &k, int32_t x) { return (k.CFX - (float)x) / k.ARC; }

// One floor / ceiling pixel (visplanes.rs:108-128).
DG_HD uint32_t flat_pixel(const DevScene &sc, const DevConsts &k, const DevFrame &f, const DevPlaneRec &p, float vx, int32_t y) {
    float vy = k.CFY - (float)y;
    float wx = p.gwz / vy;
    float wy = p.wz * vx / vy;
    float rx = wx * f.cos_a - wy * f.sin_a;
    float ry = wy * f.cos_a + wx * f.sin_a;
    int32_t tx = wrap_i16(f32_as_i16(rx) + f.pos_x_i16) & 63;
    int32_t ty = wrap_i16(f32_as_i16(ry) + f.pos_y_i16) & 63;
    uint32_t idx = sc.flats[p.flat_off + (uint32_t)(ty * 64 + tx)];
    return shade(sc.palette[idx], light_factor(p.lightf, f32_as_i16(wx)));
}

// Per-column part of draw_sky: texture column (visplanes.rs:65-66).  Returns the texel offset of the column,
// or 0xffffffff when the reference would index outside the sky bitmap.
DG_HD uint32_t sky_column_setup(const DevScene &sc, const DevConsts &k, const DevFrame &f, int32_t x) {
    int32_t tx = f32_as_i16((float)x * 256.0f / (float)k.W);
    tx = wrap_i16(tx + f.sky_tx_offset) % 256;
    if (tx < 0 || tx >= sc.sky_w) return 0xffffffffu;
    return sc.sky_texel_off + (uint32_t)tx * (uint32_t)sc.sky_h;
}

// One sky pixel (visplanes.rs:68-77): no lighting; transparent texels are skipped.
DG_HD bool sky_pixel(const DevScene &sc, const DevConsts &k, uint32_t texcol, int32_t y, uint32_t &rgb) {
    int32_t ty = f32_as_i16((float)y * 128.0f * 2.0f / (float)k.H);
    if (ty < 0) ty = wrap_i16(ty + 128);
    ty %= 128;
    if (texcol == 0xffffffffu || ty < 0 || ty >= sc.sky_h) return false;
    uint32_t o = texcol + (uint32_t)ty;
    if (!sc.texel_opq[o]) return false;
    rgb = sc.palette[sc.texel_idx[o]];
    return true;
}

}  // namespace dg
