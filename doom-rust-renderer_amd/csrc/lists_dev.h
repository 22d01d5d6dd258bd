// lists_dev.h — device-side list formats (plain structs shared by the host binner and the HIP kernels).
//
// Layout in HBM for one submission of F frames (all arrays are slabs sized at dg_create for max_batch):
//   DevFrame   frames[F]                 per-frame constants + bases into the arrays below
//   uint32_t   col_off[F][W + 1]         column-major index: spans of column x of frame f are
//                                        spans[frames[f].span_base + col_off[f][x] .. + col_off[f][x+1])
//   DevSpan    spans[]                   16 B each, per column in DRAW ORDER (later span overwrites earlier)
//   DevRSpan   rspans[]                  32 B each, written by the setup kernel: the self-contained record the raster kernel walks
//   DevWallRec walls[]                   48 B per drawn BitmapRender (per-record constants of render_vertical_bitmap_line)
//   DevPlaneRec planes[]                 16 B per drawn visplane
// plus the immutable scene: palette (256 x RGBX u32), texel index plane + opacity plane (u8, column-major
// per bitmap: off + x*h + y), flats (4096 B each, [y][x]).
#pragma once
#include <stdint.h>

namespace dg {

enum : uint8_t { SPAN_WALL = 0, SPAN_FLAT = 1, SPAN_SKY = 2 };

struct DevSpan {              // one vertical run of one draw call in one screen column
    int16_t ctop, cbot;       // rows to write, inclusive, already clamped to [0, H-1]
    int16_t top_y, bot_y;     // WALL: unclipped column extent (BitmapColumn.top_y / bottom_y)
    uint16_t rec;             // index into the frame's walls[] (WALL) or planes[] (FLAT)
    uint8_t kind, pad0;
    int16_t x, pad1;
};
static_assert(sizeof(DevSpan) == 16, "DevSpan must be 16 bytes");

// The column-major, draw-ordered list the rasteriser starts from: one self-contained 32-byte record per span, written by
// the setup kernel (host lists) or dg_fe_scatter (device column walk) from DevSpan + its wall/plane record.
//   word   WALL (bitmap_render.rs:241-263)                    FLAT (visplanes.rs:103-126)           SKY (visplanes.rs:65-72)
//   w0     ctop | plain << 14 | imm << 15 | cbot << 16 | kind << 30   same                            same      (imm: may be transparent; plain: raster_core.h pack_w0)
//   w1     d = (bottom_y - top_y) as f32                      wz * vx (f32)                          -
//   w2     start of the texture column (texel_off + tx * h)   offset of the 64x64 flat from the      start of the sky texture column
//                                                             texel index plane (flats follow it)     (0 when tx is outside the bitmap)
//   w3     light factor (f32, clamped >= 0)                   -                                      1.0f (0.0f when tx is outside)
//   w4     uy1 = top_height - bottom_height (NaN if d == 0)   gwz = GCFX * wz (f32)                  -
//   w5     top_y | off_y << 16                                light_level / 255 (f32)                -
//   w6     h as f32, NEGATED when h is not a power of two     fast-divide-ok << 8                    -
//   w7     prepared reciprocal of d (raster_core.h)           -                                      -
// This is the per-pixel form: a tile of dg_raster_tiles stages the record as it is (only the height mask of a wall is derived).
struct DevRSpan { uint32_t w[8]; };
static_assert(sizeof(DevRSpan) == 32, "DevRSpan must be 32 bytes");

struct DevWallRec {           // bitmap_render.rs:233-251 hoisted per record
    float A, B, C, D;         // ux0/uz0 (= 0.0/uz0), ux1/uz1 (= len/uz1), 1.0/uz0, 1.0/uz1
    float uy1;                // top_height - bottom_height
    float lightf;             // light_level as f32 / 255.0
    float dxf;                // (end_x - start_x) as f32
    int32_t start_x;
    uint32_t texel_off;
    int16_t w, h;
    int16_t off_x;            // clipped_line.start_offset as i16 + offset_x (wrapping)
    int16_t off_y;
    uint32_t has_holes;
};
static_assert(sizeof(DevWallRec) == 48, "DevWallRec must be 48 bytes");

struct DevPlaneRec {          // visplanes.rs:103-126 hoisted per visplane
    float wz;                 // height as f32 - player.floor_height - 41.0
    float gwz;                // GAME_CAMERA_FOCUS_X * wz
    float lightf;             // light_level as f32 / 255.0
    uint32_t flat_off;        // byte offset of the 64x64 flat
};
static_assert(sizeof(DevPlaneRec) == 16, "DevPlaneRec must be 16 bytes");

struct DevFrame {
    float cos_a, sin_a;       // rotate(player.angle)
    int32_t pos_x_i16, pos_y_i16;   // player.position.{x,y} as i16
    int32_t sky_tx_offset;    // draw_sky's tx_offset after the negative fix-up (visplanes.rs:54-58)
    uint32_t span_base, n_spans;
    uint32_t wall_base, plane_base;
    uint32_t pad[3];
};
static_assert(sizeof(DevFrame) == 48, "DevFrame must be 48 bytes");

struct DevScene {             // immutable, uploaded once per map
    const uint32_t *palette;  // 256 x (r | g<<8 | b<<16)
    const float *palette_f32; // 256 x (r, g, b, 0) as f32: what a tile of dg_raster_tiles copies to LDS
    const uint8_t *texel_idx; // column-major per bitmap (off + x*h + y): dg_raster_tiles, lane = row
    const uint8_t *texel_opq;
    const uint8_t *flats;
    uint32_t sky_texel_off;   // sky bitmap (256 x 128 expected)
    int32_t sky_w, sky_h;
    uint32_t sky_has_holes;   // any transparent texel in the sky bitmap (then sky spans are evaluated in draw order)
};

struct DevConsts {            // src/renderer/constants.rs, as f32 bit patterns computed on the host
    float ARC, GCFX, CFX, CFY;
    int32_t W, H;
};

}  // namespace dg
