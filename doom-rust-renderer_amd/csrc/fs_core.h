// fs_core.h — the per-seg and per-sprite half of the reference's front end as host/device inline functions: what
// Segs::process_seg (src/renderer/segs.rs:353-590) and Segs::process_sidedef (:121-200) do BEFORE the column loop, clip_to_viewport and
// the projection (src/renderer/misc.rs:13-161, src/geometry.rs:56-86, src/map/vertexes.rs:20-34) and the per-object part of
// draw_map_objects (src/renderer/map_objects.rs:34-129).  Two callers execute these bodies:
//   * frontend.cpp  Walker (host, parts mode): one thread per frame walks the BSP in order (DG_FE_DEVICE);
//   * fs_kernels.hip (GPU, DG_FE_DEVICE_SEGS): one lane per (frame, seg) / (frame, map object), order restored afterwards —
// and tests/emul runs the GPU's bodies on the CPU against the host walker, record by record.
// Arithmetic contract as everywhere: IEEE f32 in the reference's operand order, no contraction, Rust `as` casts through rust_num.h.
#pragma once
#include "fe_dev.h"
#include "rust_num.h"

namespace dg {

struct V2 { float x, y; };
struct Seg2 { V2 a, b; };

DG_HD V2 v2_sub(V2 p, V2 q) { return V2{p.x - q.x, p.y - q.y}; }
DG_HD V2 v2_rot(V2 v, float c, float s) { return V2{v.x * c - v.y * s, v.y * c + v.x * s}; }      // vertexes.rs:20-25
DG_HD bool left_of(V2 v, const Seg2 &l) {                                                             // vertexes.rs:27-34
    V2 p = v2_sub(v, l.a), d = v2_sub(l.b, l.a);
    return p.x * d.y - p.y * d.x <= 0.0f;
}
DG_HD float v2_dist(V2 p, V2 q) { float dx = p.x - q.x, dy = p.y - q.y; return __builtin_sqrtf(dx * dx + dy * dy); }
DG_HD float fs_fmin(float a, float b) { return __builtin_fminf(a, b); }
DG_HD float fs_fmax(float a, float b) { return __builtin_fmaxf(a, b); }

// Line::intersection, src/geometry.rs:56-82
DG_HD bool line_intersect(const Seg2 &m, const Seg2 &n, V2 &out) {
    float x1 = m.a.x, y1 = m.a.y, x2 = m.b.x, y2 = m.b.y, x3 = n.a.x, y3 = n.a.y, x4 = n.b.x, y4 = n.b.y;
    float quot = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
    if (__builtin_fabsf(quot) < 0.001f) return false;
    float inv = 1.0f / quot;
    float c12 = x1 * y2 - y1 * x2, c34 = x3 * y4 - y3 * x4;
    out.x = inv * (c12 * (x3 - x4) - (x1 - x2) * c34);
    out.y = inv * (c12 * (y3 - y4) - (y1 - y2) * c34);
    return true;
}

struct Clipped { Seg2 line; float start_offset; };

// clip_to_viewport, src/renderer/misc.rs:13-115
DG_HD bool clip_to_viewport(const Seg2 &line, Clipped &out) {
    const Seg2 left{{0.0f, 0.0f}, {1.0f, 1.0f}}, right{{0.0f, 0.0f}, {1.0f, -1.0f}};
    bool s_out_l = left_of(line.a, left), e_out_l = left_of(line.b, left);
    bool s_out_r = !left_of(line.a, right), e_out_r = !left_of(line.b, right);
    bool s_in = line.a.x > 0.0f && !s_out_l && !s_out_r;
    bool e_in = line.b.x > 0.0f && !e_out_l && !e_out_r;
    if (s_in && e_in) { out.line = line; out.start_offset = 0.0f; return true; }
    V2 li{0.0f, 0.0f}, ri{0.0f, 0.0f};
    bool l_hit = line_intersect(line, left, li) && li.x >= 0.0f;
    bool r_hit = line_intersect(line, right, ri) && ri.x >= 0.0f;
    if (!s_in && !e_in && !l_hit && !r_hit) return false;
    if (!s_in && !e_in && (l_hit != r_hit)) return false;
    if ((r_hit && s_out_r && e_out_r) || (l_hit && s_out_l && e_out_l)) return false;
    V2 s = line.a, e = line.b;
    float so = 0.0f;
    if (l_hit) {
        if (s_out_l) { so = v2_dist(li, s); s = li; }
        if (e_out_l) e = li;
    }
    if (r_hit) {
        if (s_out_r) s = ri;
        if (e_out_r) e = ri;
    }
    out.line = Seg2{s, e};
    out.start_offset = so;
    return true;
}

struct ScreenLine { int32_t sx, sy, ex, ey; };

// perspective_transform + make_sidedef_non_vertical_line, src/renderer/misc.rs:130-161.  K: FrameConsts (host) or DevConsts (device).
template <typename K> DG_HD ScreenLine project(const K &k, const Seg2 &l, float height) {
    float tsx = k.GCFX * l.a.y / l.a.x, tsy = k.GCFX * height / l.a.x;
    float tex = k.GCFX * l.b.y / l.b.x, tey = k.GCFX * height / l.b.x;
    tsx *= k.ARC;
    tex *= k.ARC;
    ScreenLine o;
    o.sx = f32_as_i32(k.CFX - tsx);
    o.sy = f32_as_i32(k.CFY - tsy);
    o.ex = f32_as_i32(k.CFX - tex);
    o.ey = f32_as_i32(k.CFY - tey);
    o.sx = o.sx < k.W - 1 ? o.sx : k.W - 1;
    o.ex = o.ex < k.W - 1 ? o.ex : k.W - 1;
    return o;
}

// BitmapRender::is_behind_vertex (bitmap_render.rs:137-165) for a record's clipped line
DG_HD bool fs_behind(const Seg2 &line, V2 v) {
    const float min_x = fs_fmin(line.a.x, line.b.x), max_x = fs_fmax(line.a.x, line.b.x);
    if (min_x > v.x) return true;
    if (max_x > v.x && !left_of(v, line)) return true;
    return false;
}

// What the texture mapper needs of a bitmap (scene.hpp BitmapInfo; the GPU holds a table of these)
struct FsBitmap { uint32_t texel_off; int16_t w, h; int16_t top_offset; uint16_t has_holes; };
static_assert(sizeof(FsBitmap) == 12, "FsBitmap layout");

// Per-record constants of render_vertical_bitmap_line (bitmap_render.rs:233-251)
DG_HD DevWallRec fs_wall_rec(const FsBitmap &bi, float lsx, float lsy, float lex, float ley, float start_offset, int32_t start_x, int32_t end_x,
                             float bottom_height, float top_height, int16_t offset_x, int16_t offset_y, int16_t light_level) {
    DevWallRec d;
    float dx = lsx - lex, dy = lsy - ley;
    float len = __builtin_sqrtf(dx * dx + dy * dy);             // Line::length, geometry.rs:84-86
    float uz0 = lsx, uz1 = lex;
    d.A = 0.0f / uz0;
    d.B = len / uz1;
    d.C = 1.0f / uz0;
    d.D = 1.0f / uz1;
    d.uy1 = top_height - bottom_height;
    d.lightf = (float)light_level / 255.0f;
    d.dxf = (float)(end_x - start_x);
    d.start_x = start_x;
    d.texel_off = bi.texel_off;
    d.w = bi.w; d.h = bi.h;
    d.off_x = (int16_t)wrap_i16(f32_as_i16(start_offset) + offset_x);
    d.off_y = offset_y;
    d.has_holes = bi.has_holes;
    return d;
}

// ---- one seg -----------------------------------------------------------------------------------------------------------------------
// Everything process_seg reads about a seg, flattened once per scene (Scene::fs_segs; the same table is uploaded to the GPU).
struct FsSeg {
    float v1x, v1y, v2x, v2y;
    int32_t front_sector, back_sector;      // front_sector < 0: the seg's side has no sidedef (segs.rs:358-362: skipped); back_sector < 0: one-sided
    float sd_xoff, sd_yoff;                 // front sidedef offsets
    int32_t tex_mid, tex_low, tex_up;       // bitmap id, TEX_NONE (-1) for "-", TEX_UNKNOWN (-2) when Textures::get would panic
    int16_t seg_offset;
    uint16_t ld_flags;                      // linedef flags: 4 two-sided, 8 upper unpegged, 16 lower unpegged (linedefs.rs:10-18)
};
static_assert(sizeof(FsSeg) == 48, "FsSeg layout");
struct FsSector {                           // the immutable part of a sector (its light level is game state)
    int16_t floor_h, ceil_h;
    int32_t floor_flat, ceil_flat;          // flat id (>= 0) when not animated, FLAT_MISSING (-2) when the lump does not exist
    int32_t floor_anim, ceil_anim;          // index into the animation lists, or -1
    uint32_t ceil_tex_sky;                  // sector.ceiling_texture.contains("SKY") (segs.rs:463-469)
};
static_assert(sizeof(FsSector) == 24, "FsSector layout");
struct FsAnim { int32_t n; int32_t flat[4]; };

// Flats::get_animated at this frame's timestamp (flats.rs:103-111)
DG_HD int32_t fs_resolve_flat(int32_t flat, int32_t anim, const FsAnim *anims, float timestamp) {
    if (anim < 0) return flat;
    const FsAnim &a = anims[anim];
    float t = timestamp * 3.0f;
    unsigned long long cyc = !(t > 0.0f) ? 0ull : (t >= 18446744073709551616.0f ? ~0ull : (unsigned long long)t);
    return a.flat[cyc % (unsigned long long)a.n];
}

enum : int32_t {                             // outcome of fs_seg / fs_part
    FS_OK = 0,
    FS_SKIP = 1,                             // nothing to record (not an error)
    FS_FAIL_CLIP_X = 2,                      // "Clipped line x < -0.01" (segs.rs:431-436)
    FS_FAIL_FLAT = 3,                        // Flat::new unwrap (flats.rs:117)
    FS_FAIL_TEXTURE = 4,                     // Textures::get panics (textures.rs:158)
    FS_FAIL_VERTICAL = 5,                    // "Wall start not vertical" (segs.rs:140-145)
    FS_FAIL_LINE_X = 6,                      // "Invalid line x" (segs.rs:103-111)
    FS_FAIL_PARTS = 7,                       // only the host list path can judge (zero-sized bitmap)
};

struct FsCall {                              // one process_sidedef call of a seg (segs.rs:493-588)
    float bottom_height, top_height;
    int32_t offset_y, tex;
    uint32_t flags;                          // FEP_ONLY_OCCL | FEP_LOWER | FEP_UPPER | FEP_DRAW_CEILING | FEP_TWO_SIDED_MID
};
struct FsSegOut {                            // process_seg up to its process_sidedef calls
    Clipped cl;
    int16_t seg_offset, floor_h, ceil_h, light;
    int32_t floor_flat, ceil_flat;
    float sd_xoff, sd_yoff;
    int32_t n_calls;                         // slots to look at: 1 (one-sided) or 4; call_mask says which of them hold a call.  The slots
    uint32_t call_mask;                      // are fixed — only-occlusion, two-sided middle, lower, upper (the reference's order) — so that
    FsCall call[4];                          // every index into call[] is a constant (an indexed array would live in scratch memory on the GPU)
};
// The call in slot i (a constant after unrolling).
DG_HD FsCall fs_call(const FsSegOut &o, uint32_t i) { return i == 0u ? o.call[0] : i == 1u ? o.call[1] : i == 2u ? o.call[2] : o.call[3]; }

// Segs::process_seg, segs.rs:353-590, for a viewer at ppos looking along (cos_na, sin_na) = (cos, sin)(-angle).  light: the front
// sector's CURRENT light level.  Returns FS_OK with 1 .. 5 calls, FS_SKIP, or a failure.
template <typename K>
DG_HD int32_t fs_seg(const K &k, const FsSeg &sg, const FsSector *sectors, const FsAnim *anims, V2 ppos, float cos_na, float sin_na, float player_height,
                     float timestamp, int16_t light, FsSegOut &o) {
    if (sg.front_sector < 0) return FS_SKIP;
    // (the clip comes first here: it needs the seg's two vertices only and turns away most segs of a map; the sector heights the
    // reference reads before it — segs.rs:364-402 — have no side effects and are read below, by the segs that are left)
    V2 a = v2_rot(v2_sub(V2{sg.v1x, sg.v1y}, ppos), cos_na, sin_na);
    V2 b = v2_rot(v2_sub(V2{sg.v2x, sg.v2y}, ppos), cos_na, sin_na);
    if (!clip_to_viewport(Seg2{a, b}, o.cl)) return FS_SKIP;
    if (o.cl.line.a.x < -0.01f) return FS_FAIL_CLIP_X;

    const FsSector &fs = sectors[sg.front_sector];
    const FsSector *bs = sg.back_sector >= 0 ? &sectors[sg.back_sector] : nullptr;
    float floor_height = (float)fs.floor_h, ceiling_height = (float)fs.ceil_h;
    bool has_pb = false, has_pt = false;
    float pb_h = 0.0f, pt_h = 0.0f;
    if (bs) {
        if (bs->floor_h > fs.floor_h) { has_pb = true; pb_h = (float)bs->floor_h; }
        if (bs->ceil_h < fs.ceil_h) { has_pt = true; pt_h = (float)bs->ceil_h; }
    }
    const bool two_sided = (sg.ld_flags & 4) != 0, top_unpegged = (sg.ld_flags & 8) != 0, bottom_unpegged = (sg.ld_flags & 16) != 0;

    ScreenLine fl = project(k, o.cl.line, floor_height - player_height);
    if (fl.sx > fl.ex) return FS_SKIP;                            // back face

    o.floor_flat = fs_resolve_flat(fs.floor_flat, fs.floor_anim, anims, timestamp);
    o.ceil_flat = fs_resolve_flat(fs.ceil_flat, fs.ceil_anim, anims, timestamp);
    if (o.floor_flat < 0 || o.ceil_flat < 0) return FS_FAIL_FLAT;

    bool draw_ceiling = true;
    if (bs && fs.ceil_tex_sky && bs->ceil_tex_sky) {              // sky hack, segs.rs:463-477
        has_pt = false;
        ceiling_height = fs_fmin((float)bs->ceil_h, ceiling_height);
        draw_ceiling = false;
    }
    o.seg_offset = sg.seg_offset; o.floor_h = fs.floor_h; o.ceil_h = fs.ceil_h; o.light = light;
    o.sd_xoff = sg.sd_xoff; o.sd_yoff = sg.sd_yoff;
    const uint32_t dc = draw_ceiling ? FEP_DRAW_CEILING : 0u;
    if (!two_sided) {
        const int32_t oy = bottom_unpegged ? f32_as_i32(floor_height - ceiling_height) : 0;
        o.call[0] = FsCall{floor_height - player_height, ceiling_height - player_height, oy, sg.tex_mid, dc};
        o.n_calls = 1; o.call_mask = 1u;
        return FS_OK;
    }
    o.call[0] = FsCall{floor_height - player_height, ceiling_height - player_height, 0, sg.tex_mid, dc | FEP_ONLY_OCCL};
    const float mid_floor = has_pb ? pb_h : floor_height, mid_ceil = has_pt ? pt_h : ceiling_height;
    o.call[1] = FsCall{mid_floor - player_height, mid_ceil - player_height, 0, sg.tex_mid, dc | FEP_TWO_SIDED_MID};
    o.call_mask = 3u;
    if (has_pb) {
        const int32_t oy = bottom_unpegged ? f32_as_i32(ceiling_height - pb_h) : 0;
        o.call[2] = FsCall{floor_height - player_height, pb_h - player_height, oy, sg.tex_low, dc | FEP_LOWER};
        o.call_mask |= 4u;
    }
    if (has_pt) {
        const int32_t oy = top_unpegged ? 0 : f32_as_i32(pt_h - ceiling_height);
        o.call[3] = FsCall{pt_h - player_height, ceiling_height - player_height, oy, sg.tex_up, dc | FEP_UPPER};
        o.call_mask |= 8u;
    }
    o.n_calls = 4;
    return FS_OK;
}

// The x half of make_sidedef_non_vertical_line (misc.rs:138-161): the screen columns of a clipped line do not depend on the height.
template <typename K> DG_HD void project_x(const K &k, const Seg2 &l, int32_t &sx, int32_t &ex) {
    float tsx = k.GCFX * l.a.y / l.a.x;
    float tex = k.GCFX * l.b.y / l.b.x;
    tsx *= k.ARC;
    tex *= k.ARC;
    sx = f32_as_i32(k.CFX - tsx);
    ex = f32_as_i32(k.CFX - tex);
    sx = sx < k.W - 1 ? sx : k.W - 1;
    ex = ex < k.W - 1 ? ex : k.W - 1;
}
// The tests of Segs::process_sidedef (segs.rs:121-157) that decide whether a call reaches its column loop, and what the hidden-part
// culling needs of it: columns [sx, ex] and the FEP_* flags.  (Both edges of a part project to the same columns — the x of
// perspective_transform does not see the height — so the reference's "Wall start not vertical" panic cannot fire; fs_part keeps the test.)
template <typename K>
DG_HD int32_t fs_part_head(const K &k, const FsSegOut &s, const FsCall &c, const FsBitmap *bitmaps, const uint8_t *flat_sky, int32_t &sx, int32_t &ex, uint32_t &flags) {
    project_x(k, s.cl.line, sx, ex);
    if (c.tex == -2) return FS_FAIL_TEXTURE;                      // TEX_UNKNOWN
    if (wrap_i16(sx) == wrap_i16(ex)) return FS_SKIP;
    if (sx < 0 || sx >= k.W || ex < 0 || ex >= k.W) return FS_FAIL_LINE_X;
    if (c.tex >= 0 && (bitmaps[c.tex].w <= 0 || bitmaps[c.tex].h <= 0)) return FS_FAIL_PARTS;
    const bool fsky = flat_sky[s.floor_flat] != 0, csky = flat_sky[s.ceil_flat] != 0;
    flags = c.flags | (c.tex >= 0 ? FEP_HAS_BITMAP : 0u) | (fsky ? FEP_FLOOR_SKY : 0u) | (csky ? FEP_CEIL_SKY : 0u);
    return FS_OK;
}

// Segs::process_sidedef (segs.rs:121-200) up to its column loop, for one call of a seg: the FePart the device column walk consumes,
// without the two fields that depend on what came before in BSP order (sky_slot, seq: left -1 / 0).  FS_SKIP: a zero-width part.
// view_floor_height: player.floor_height (visplanes.rs:112).  flat_sky[flat]: the flat's name contains "SKY".
template <typename K>
DG_HD int32_t fs_part(const K &k, const FsSegOut &s, const FsCall &c, const FsBitmap *bitmaps, const uint8_t *flat_sky, float view_floor_height, FePart &p) {
    const Seg2 &line = s.cl.line;
    const ScreenLine bot = project(k, line, c.bottom_height);
    const ScreenLine top = project(k, line, c.top_height);
    if (c.tex == -2) return FS_FAIL_TEXTURE;                      // TEX_UNKNOWN
    if (bot.sx != top.sx || bot.ex != top.ex) return FS_FAIL_VERTICAL;
    if (wrap_i16(bot.sx) == wrap_i16(bot.ex) || wrap_i16(top.sx) == wrap_i16(top.ex)) return FS_SKIP;
    if (bot.sx < 0 || bot.sx >= k.W || bot.ex < 0 || bot.ex >= k.W) return FS_FAIL_LINE_X;
    const float bottom_delta = ((float)bot.sy - (float)bot.ey) / ((float)bot.sx - (float)bot.ex);
    const float top_delta = ((float)top.sy - (float)top.ey) / ((float)top.sx - (float)top.ex);
    if (c.tex >= 0 && (bitmaps[c.tex].w <= 0 || bitmaps[c.tex].h <= 0)) return FS_FAIL_PARTS;
    const int16_t offset_x = (int16_t)wrap_i16(f32_as_i16(s.sd_xoff) + s.seg_offset);
    const int16_t offset_y = (int16_t)wrap_i16(f32_as_i16(s.sd_yoff) + wrap_i16(c.offset_y));
    p.sx = bot.sx; p.ex = bot.ex;
    p.bsy = (float)bot.sy; p.bsx = (float)bot.sx; p.bdelta = bottom_delta;
    p.tsy = (float)top.sy; p.tsx = (float)top.sx; p.tdelta = top_delta;
    const bool fsky = flat_sky[s.floor_flat] != 0, csky = flat_sky[s.ceil_flat] != 0;
    p.flags = c.flags | (c.tex >= 0 ? FEP_HAS_BITMAP : 0u) | (fsky ? FEP_FLOOR_SKY : 0u) | (csky ? FEP_CEIL_SKY : 0u);
    p.sky_slot = -1;
    p.seq = 0;
    p.pad = 0;
    if (c.tex >= 0) {
        p.wall = fs_wall_rec(bitmaps[c.tex], line.a.x, line.a.y, line.b.x, line.b.y, s.cl.start_offset, bot.sx, bot.ex, c.bottom_height, c.top_height,
                             offset_x, offset_y, s.light);
    } else {
        p.wall = DevWallRec{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0, 0u, 0, 0, 0, 0, 0u};
    }
    const float lightf = (float)s.light / 255.0f;
    p.floor_plane.wz = (float)s.floor_h - view_floor_height - 41.0f;      // visplanes.rs:112
    p.floor_plane.gwz = k.GCFX * p.floor_plane.wz;
    p.floor_plane.lightf = lightf;
    p.floor_plane.flat_off = (uint32_t)s.floor_flat * 4096u;
    p.ceil_plane.wz = (float)s.ceil_h - view_floor_height - 41.0f;
    p.ceil_plane.gwz = k.GCFX * p.ceil_plane.wz;
    p.ceil_plane.lightf = lightf;
    p.ceil_plane.flat_off = (uint32_t)s.ceil_flat * 4096u;
    return FS_OK;
}
// Does a part with these flags open / extend visplanes (segs.rs:263-318), and may it therefore need a sky event slot?
DG_HD bool fs_part_wants_sky_slot(uint32_t flags) {
    const bool full_height = !(flags & (FEP_LOWER | FEP_UPPER | FEP_ONLY_OCCL));
    const bool planes_here = !(flags & FEP_TWO_SIDED_MID) && (full_height || (flags & FEP_ONLY_OCCL));
    return planes_here && ((flags & FEP_FLOOR_SKY) || ((flags & FEP_CEIL_SKY) && (flags & FEP_DRAW_CEILING)));
}
// A full-height solid part: every column it spans becomes horizontally occluded (segs.rs:341-344)
DG_HD bool fs_part_is_solid(uint32_t flags) { return !(flags & (FEP_LOWER | FEP_UPPER | FEP_ONLY_OCCL | FEP_TWO_SIDED_MID)); }

// ---- one map object ----------------------------------------------------------------------------------------------------------------
struct FsMobj { float x, y, angle; int32_t sector; };            // the immutable part (position never changes in the reference); sector < 0: outside the map
static_assert(sizeof(FsMobj) == 16, "FsMobj layout");
struct FsSpriteFrame { int32_t rotate; int32_t bitmap[8]; };     // sprites.rs:20-23

enum : int32_t { FS_FAIL_ROTATION = 8, FS_FAIL_MOBJ_CLIP_X = 9, FS_FAIL_MOBJ_COLUMN = 10 };

struct FsSpriteOut { FeSprite sp; Seg2 line; V2 centre; int32_t sort_key; };
// draw_map_objects (map_objects.rs:34-129, 168-209) for one object whose state shows (sprite_frame >= 0): FS_OK with the FeSprite
// (seq / behind_off left 0), FS_SKIP, or a failure.  light: the object's sector's current light level.
template <typename K>
DG_HD int32_t fs_mobj(const K &k, const FsMobj &m, const FsSpriteFrame &sf, const FsBitmap *bitmaps, const FsSector *sectors, V2 ppos, float view_angle,
                      float cos_na, float sin_na, float player_height, int32_t full_bright, int16_t sector_light, FsSpriteOut &o) {
    const float kPi = 3.14159265358979323846f;
    float angle = view_angle - m.angle - kPi;
    angle += kPi / 16.0f;
    angle = __builtin_fmodf(angle, 2.0f * kPi);
    if (angle < 0.0f) angle += 2.0f * kPi;
    angle = __builtin_fmodf(angle, 2.0f * kPi);
    const int rotation = f32_as_u8(angle * 8.0f / (2.0f * kPi));
    if (rotation > 7) return FS_FAIL_ROTATION;
    const int bitmap = sf.rotate ? sf.bitmap[rotation] : sf.bitmap[0];
    const FsBitmap &bi = bitmaps[bitmap];

    const V2 vpv = v2_rot(v2_sub(V2{m.x, m.y}, ppos), cos_na, sin_na);
    const int16_t width = bi.w;
    const V2 a = v2_sub(vpv, V2{0.0f, (float)(int16_t)(-width) / 2.0f});
    const V2 b = v2_sub(vpv, V2{0.0f, (float)width / 2.0f});
    Clipped cl;
    if (!clip_to_viewport(Seg2{a, b}, cl)) return FS_SKIP;
    if (cl.line.a.x < -0.01f) return FS_FAIL_MOBJ_CLIP_X;
    if (m.sector < 0) return FS_SKIP;                             // "Thing is outside map"
    const FsSector &sec = sectors[m.sector];
    const int16_t light = full_bright ? (int16_t)255 : sector_light;

    const int16_t bh = bi.h;
    float bottom_height = (float)sec.floor_h - player_height;
    float top_height = (float)sec.floor_h + (float)bh - 1.0f - player_height;
    bottom_height += (float)bi.top_offset - (float)bh;
    top_height += (float)bi.top_offset - (float)bh;
    const ScreenLine bot = project(k, cl.line, bottom_height);
    const ScreenLine top = project(k, cl.line, top_height);
    const int x0 = wrap_i16(bot.sx), x1 = wrap_i16(bot.ex);       // columns [x0, x1)
    if (x0 < x1 && (x0 < 0 || x1 > k.W)) return FS_FAIL_MOBJ_COLUMN;
    if (bi.w <= 0 || bi.h <= 0) return FS_FAIL_PARTS;
    FeSprite &sp = o.sp;
    sp.x0 = x0; sp.x1 = x1;
    sp.bsy = (float)bot.sy; sp.bsx = (float)bot.sx;
    sp.bdelta = ((float)bot.sy - (float)bot.ey) / ((float)bot.sx - (float)bot.ex);
    sp.tsy = (float)top.sy; sp.tsx = (float)top.sx;
    sp.tdelta = ((float)top.sy - (float)top.ey) / ((float)top.sx - (float)top.ex);
    sp.seq = 0; sp.behind_off = 0; sp.pad[0] = sp.pad[1] = 0;
    sp.wall = fs_wall_rec(bi, cl.line.a.x, cl.line.a.y, cl.line.b.x, cl.line.b.y, cl.start_offset, bot.sx, bot.ex, bottom_height, top_height, 0, 0, light);
    o.line = cl.line;
    o.centre = vpv;
    o.sort_key = f32_as_i16(cl.line.a.x);                         // bitmap_render.rs:168-174
    return FS_OK;
}

}  // namespace dg
