// frontend.cpp — host list generation: BSP walk, seg classification, projection, per-column clip arrays,
// visplane assembly, sprite projection/clipping/sorting.  No pixel work happens here; see frontend.hpp.
//
// Arithmetic contract: IEEE f32, no contraction (-ffp-contract=off), Rust `as` casts via rust_num.h.
#include "frontend.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "fs_core.h"
#include "rust_num.h"

namespace dg {

namespace {

const float kPi = 3.14159265358979323846f;
const float kEye = 41.0f;  // PLAYER_EYE_HEIGHT, src/renderer/constants.rs:3

// (V2, Seg2, clip_to_viewport, project, ... : fs_core.h — shared with the GPU's per-seg kernels)
inline V2 sub(V2 p, V2 q) { return v2_sub(p, q); }
inline V2 rot(V2 v, float c, float s) { return v2_rot(v, c, s); }

enum : uint8_t { ST_SOLID, ST_TWOSIDED, ST_DRAWN, ST_MAPOBJECT };   // bitmap_render.rs:11-17

}  // namespace

struct FrameArena::Rec {                                             // bitmap_render.rs:29-45
    Seg2 line;
    float start_offset, bottom_height, top_height;
    float min_x, max_x;            // of line.a.x / line.b.x, for is_behind_vertex
    int32_t start_x, end_x, bitmap;
    uint32_t first_col, n_cols;
    int32_t out_index;             // index in arena.renders once emitted, else -1
    int16_t light, offset_x, offset_y;
    uint8_t state, ext_bottom, ext_top, draw_ceiling;
    int16_t sort_key;              // map objects: line.start.x as i16 (bitmap_render.rs:168-174)
};

FrameArena::FrameArena() : recs(new std::vector<Rec>()) {}
FrameArena::~FrameArena() { delete recs; }

FrameConsts make_consts(int W, int H) {
    FrameConsts k;
    k.ARC = 200.0f / 240.0f;
    k.GSW = (float)W / k.ARC;
    k.GCFX = k.GSW / 2.0f;
    k.CFX = (float)W / 2.0f;
    k.CFY = (float)H / 2.0f;
    k.W = W;
    k.H = H;
    return k;
}

DevWallRec make_wall_rec(const BitmapInfo &bi, float lsx, float lsy, float lex, float ley, float start_offset, int32_t start_x, int32_t end_x,
                         float bottom_height, float top_height, int16_t offset_x, int16_t offset_y, int16_t light_level) {
    const FsBitmap fb{bi.texel_off, (int16_t)bi.w, (int16_t)bi.h, bi.top_offset, (uint16_t)bi.has_holes};
    return fs_wall_rec(fb, lsx, lsy, lex, ley, start_offset, start_x, end_x, bottom_height, top_height, offset_x, offset_y, light_level);
}

void fill_view_trig(dg_view &v) {
    if (v.trig_valid) return;
    v.cos_a = cosf(v.angle);
    v.sin_a = sinf(v.angle);
    v.cos_na = cosf(-v.angle);
    v.sin_na = sinf(-v.angle);
    v.trig_valid = 1;
}

namespace {

using Rec = FrameArena::Rec;

struct PlaneBuilder {          // one open visplane of SidedefVisPlanes (sidedef_visplanes.rs:7-17)
    bool used = false;
    int16_t left = -1, right = -1;
    uint32_t first = 0;        // entry index in its pool
};

// Union of closed column intervals, kept sorted and merged (adjacent intervals fuse): Doom's "solidsegs", here only a
// conservative summary of horizontal_ocl used to drop records that cannot matter.
struct ColumnIntervals {
    std::vector<std::pair<int, int>> iv;
    void clear() { iv.clear(); }
    bool covers(int lo, int hi) const {
        for (const auto &r : iv) {
            if (r.first > lo) return false;
            if (r.second >= lo) return r.second >= hi;
        }
        return false;
    }
    void add(int lo, int hi) {
        size_t i = 0;
        while (i < iv.size() && iv[i].second + 1 < lo) i++;
        size_t j = i;
        while (j < iv.size() && iv[j].first <= hi + 1) { lo = std::min(lo, iv[j].first); hi = std::max(hi, iv[j].second); j++; }
        iv.erase(iv.begin() + (ptrdiff_t)i, iv.begin() + (ptrdiff_t)j);
        iv.insert(iv.begin() + (ptrdiff_t)i, std::make_pair(lo, hi));
    }
};

struct Walker {
    const Scene &sc;
    const FrameConsts k;
    const dg_view &view;
    FrameArena &A;
    std::vector<Rec> &recs;
    std::string &err;
    V2 ppos;
    float player_height;
    int status = DG_OK;
    bool parts_mode = false;      // record FePart / FeSprite instead of walking columns
    ColumnIntervals solid_cols;   // columns spanned by full-height solid parts so far
    uint32_t n_floor_planes_marker = 0;
    // visplanes carry a pool tag in the top bit of first_entry until finalisation
    static constexpr uint32_t kCeilPool = 0x80000000u;

    const dg_view_state *state = nullptr;   // this view's game-state snapshot (light levels, mobj states) on top of the scene's

    Walker(const Scene &s, int W, int H, const dg_view &v, FrameArena &a, std::string &e, const dg_view_state *st = nullptr)
        : sc(s), k(make_consts(W, H)), view(v), A(a), recs(*a.recs), err(e) {
        ppos = V2{v.x, v.y};
        player_height = v.floor_height + kEye;
        apply_state(st);
    }
    ~Walker() {                                                      // the arena's overlay tables go back to "no override"
        if (!state) return;
        for (uint32_t i = 0; i < state->n_lights; i++)
            if ((size_t)state->lights[i].sector < A.light_ov.size()) A.light_ov[(size_t)state->lights[i].sector] = kNoOverride;
        for (uint32_t i = 0; i < state->n_mobjs; i++)
            if ((size_t)state->mobjs[i].mobj < A.mobj_ov.size()) A.mobj_ov[(size_t)state->mobjs[i].mobj] = kNoOverride;
    }
    static constexpr int32_t kNoOverride = INT32_MIN;

    // sector.light_level / map object state as the reference's thinkers would have left them before this frame
    // (src/lights.rs:47-259, src/map_objects.rs:63-121): the view's snapshot entry if there is one, else the scene's value.
    void apply_state(const dg_view_state *st) {
        if (!st || (st->n_lights == 0 && st->n_mobjs == 0)) return;
        if (A.light_ov.size() != sc.sectors.size()) A.light_ov.assign(sc.sectors.size(), kNoOverride);
        if (A.mobj_ov.size() != sc.mobjs.size()) A.mobj_ov.assign(sc.mobjs.size(), kNoOverride);
        for (uint32_t i = 0; i < st->n_lights; i++) {
            const dg_sector_light &l = st->lights[i];
            if (l.sector < 0 || (size_t)l.sector >= sc.sectors.size()) { status = DG_ERR_INVALID; err = "view state: sector index out of range"; continue; }
            A.light_ov[(size_t)l.sector] = (int32_t)l.light_level;
        }
        for (uint32_t i = 0; i < st->n_mobjs; i++) {
            const dg_mobj_state &m = st->mobjs[i];
            if (m.mobj < 0 || (size_t)m.mobj >= sc.mobjs.size() || m.sprite_frame >= (int32_t)sc.sprite_frames.size()) {
                status = DG_ERR_INVALID; err = "view state: map object or sprite frame index out of range"; continue;
            }
            A.mobj_ov[(size_t)m.mobj] = (m.sprite_frame < 0 ? -1 : m.sprite_frame) * 2 + (m.full_bright ? 1 : 0);   // -2 / -1: S_NULL
        }
        state = st;
    }
    int16_t sector_light(int sector) const {
        if (state && A.light_ov[(size_t)sector] != kNoOverride) return (int16_t)A.light_ov[(size_t)sector];
        return sc.sectors[(size_t)sector].light;
    }
    void mobj_state(size_t i, int32_t &sprite_frame, int32_t &full_bright) const {
        const MapObjectRec &m = sc.mobjs[i];
        sprite_frame = m.sprite_frame; full_bright = m.full_bright;
        if (state && A.mobj_ov[i] != kNoOverride) {
            const int32_t v = A.mobj_ov[i];
            sprite_frame = v < 0 ? -1 : v >> 1; full_bright = v < 0 ? 0 : v & 1;
        }
    }

    int fail(const std::string &m) {
        if (status == DG_OK) { status = DG_ERR_RENDER; err = m; }
        return status;
    }
    int fail_parts(const std::string &m) {                          // parts mode cannot tell: let the host list path decide
        if (status == DG_OK) { status = kPartsUnsupported; err = m; }
        return status;
    }

    void occlude(int x) {                                           // segs.rs:113-117
        A.hor_ocl[(size_t)x] = 1;
        A.floor_ocl[(size_t)x] = (int16_t)((int16_t)k.H / 2);
        A.ceil_ocl[(size_t)x] = (int16_t)((int16_t)k.H / 2);
    }

    void plane_add(PlaneBuilder &p, std::vector<int16_t> &pool, int16_t x, int16_t top, int16_t bottom) {
        if (!p.used) {                                              // sidedef_visplanes.rs:60-83
            p.left = x;
            p.first = (uint32_t)(pool.size() / 2);
            p.used = true;
        } else {
            for (int g = p.right + 1; g < x; g++) { pool.push_back(0); pool.push_back(0); }  // Visplane::new zero fill
        }
        p.right = x;
        pool.push_back(top);
        pool.push_back(bottom);
    }
    void plane_flush(PlaneBuilder &p, bool ceil, int flat, int16_t height, int16_t light) {
        if (!p.used) return;
        dg_visplane v;
        v.flat = flat; v.height = height; v.light_level = light; v.left = p.left; v.right = p.right;
        v.first_entry = p.first | (ceil ? kCeilPool : 0u);
        A.visplanes.push_back(v);
        p.used = false;
    }

    struct Side {                                                   // SideDefDetails, segs.rs:42-51
        const Clipped *cl;
        int16_t offset_x, floor_h, ceil_h, light;
        int floor_flat, ceil_flat;
    };
    struct Flags { bool only_occlusions, lower, upper, draw_ceiling, two_sided_mid; };   // segs.rs:53-59

    void emit_draw(Rec &r) {
        if (r.out_index < 0) {
            dg_bitmap_render o;
            std::memset(&o, 0, sizeof o);
            o.bitmap = r.bitmap; o.light_level = r.light; o.offset_x = r.offset_x; o.offset_y = r.offset_y;
            o.line_start_x = r.line.a.x; o.line_start_y = r.line.a.y; o.line_end_x = r.line.b.x; o.line_end_y = r.line.b.y;
            o.start_offset = r.start_offset; o.start_x = r.start_x; o.end_x = r.end_x;
            o.bottom_height = r.bottom_height; o.top_height = r.top_height;
            o.first_column = r.first_col; o.n_columns = r.n_cols;
            r.out_index = (int32_t)A.renders.size();
            A.renders.push_back(o);
        }
        A.order.push_back(dg_draw_cmd{0u, (uint32_t)r.out_index});
    }

    // Segs::process_sidedef, segs.rs:121-350
    void process_sidedef(const FsSegOut &so, const FsCall &call) {
        const int W = k.W, H = k.H;
        const Clipped *cl = &so.cl;
        const Side s{cl, so.seg_offset, so.floor_h, so.ceil_h, so.light, so.floor_flat, so.ceil_flat};
        const float bottom_height = call.bottom_height, top_height = call.top_height;
        const int32_t offset_y = call.offset_y;
        const int tex = call.tex;
        const Flags f{(call.flags & FEP_ONLY_OCCL) != 0, (call.flags & FEP_LOWER) != 0, (call.flags & FEP_UPPER) != 0, (call.flags & FEP_DRAW_CEILING) != 0,
                      (call.flags & FEP_TWO_SIDED_MID) != 0};
        if (parts_mode) {                                            // the per-column half runs on the GPU: record the part
            FePart p;
            std::memset(&p, 0, sizeof p);
            const int32_t st = fs_part(k, so, call, sc.fs_bitmaps.data(), sc.flat_sky.data(), view.floor_height, p);
            if (st == FS_SKIP) return;
            if (st == FS_FAIL_PARTS) { fail_parts(fs_message(st)); return; }
            if (st != FS_OK) { fail(fs_message(st)); return; }
            // Columns that an earlier full-height solid part spans are horizontally occluded whatever that part's own
            // visibility was (segs.rs:341-344 runs for every column of the part).  A part lying entirely inside them can
            // neither draw, clip, add a visplane entry nor occlude anything new (segs.rs:211,337-341): it is dropped here.
            if (solid_cols.covers(p.sx, p.ex)) return;
            if (fs_part_is_solid(p.flags) && p.sx <= p.ex) solid_cols.add(p.sx, p.ex);
            if (fs_part_wants_sky_slot(p.flags)) {
                p.sky_slot = (int32_t)A.n_sky_slots++;
                A.sky_parts.push_back((uint32_t)A.parts.size());
            }
            if (A.parts.size() >= 65535) { fail_parts("more than 65535 wall records in a frame"); return; }
            A.parts.push_back(p);
            Rec r;
            std::memset(&r, 0, sizeof r);
            r.line = cl->line;
            r.min_x = std::fmin(r.line.a.x, r.line.b.x); r.max_x = std::fmax(r.line.a.x, r.line.b.x);
            r.state = f.two_sided_mid ? ST_TWOSIDED : ST_SOLID;
            r.out_index = -1;
            recs.push_back(r);
            return;
        }
        ScreenLine bot = project(k, s.cl->line, bottom_height);
        ScreenLine top = project(k, s.cl->line, top_height);
        if (tex == TEX_UNKNOWN) { fail("Unknown texture (Textures::get panics, textures.rs:158)"); return; }
        if (bot.sx != top.sx || bot.ex != top.ex) { fail("Wall start not vertical (segs.rs:140-145)"); return; }
        if (wrap_i16(bot.sx) == wrap_i16(bot.ex) || wrap_i16(top.sx) == wrap_i16(top.ex)) return;
        if (bot.sx < 0 || bot.sx >= W || bot.ex < 0 || bot.ex >= W) { fail("Invalid line x (segs.rs:103-111)"); return; }

        float bottom_delta = ((float)bot.sy - (float)bot.ey) / ((float)bot.sx - (float)bot.ex);
        float top_delta = ((float)top.sy - (float)top.ey) / ((float)top.sx - (float)top.ex);

        PlaneBuilder pb, pt;
        bool full_height = !f.lower && !f.upper && !f.only_occlusions;

        Rec r;
        r.line = s.cl->line;
        r.start_offset = s.cl->start_offset;
        r.bottom_height = bottom_height; r.top_height = top_height;
        r.min_x = std::fmin(r.line.a.x, r.line.b.x); r.max_x = std::fmax(r.line.a.x, r.line.b.x);
        r.start_x = bot.sx; r.end_x = bot.ex; r.bitmap = tex;
        r.first_col = (uint32_t)A.columns.size(); r.n_cols = 0; r.out_index = -1;
        r.light = s.light;
        r.offset_x = (int16_t)wrap_i16(f32_as_i16(so.sd_xoff) + s.offset_x);
        r.offset_y = (int16_t)wrap_i16(f32_as_i16(so.sd_yoff) + wrap_i16(offset_y));
        r.state = f.two_sided_mid ? ST_TWOSIDED : ST_SOLID;
        r.ext_bottom = f.lower || (!f.two_sided_mid && full_height);
        r.ext_top = f.upper || (!f.two_sided_mid && full_height);
        r.draw_ceiling = f.draw_ceiling;
        r.sort_key = 0;

        const bool planes_here = !f.two_sided_mid && (full_height || f.only_occlusions);
        // Columns that an earlier full-height solid part spans are horizontally occluded whatever that part's own
        // visibility was (segs.rs:341-344 runs for every column of the part).  A part lying entirely inside them can
        // neither draw, clip, add a visplane entry nor occlude anything new (segs.rs:211,337-341): it is dropped here.
        if (solid_cols.covers(bot.sx, bot.ex)) return;
        if (!f.two_sided_mid && full_height && bot.sx <= bot.ex) solid_cols.add(bot.sx, bot.ex);
        const int16_t hm1 = (int16_t)(H - 1);
        for (int x = bot.sx; x <= bot.ex; x++) {
            if (!A.hor_ocl[(size_t)x]) {
                int16_t bottom_y = (int16_t)f32_as_i16((float)bot.sy + ((float)x - (float)bot.sx) * bottom_delta);
                int16_t top_y = (int16_t)f32_as_i16((float)top.sy + ((float)x - (float)top.sx) * top_delta);
                int16_t fo = A.floor_ocl[(size_t)x], co = A.ceil_ocl[(size_t)x];
                int16_t cb = std::min(hm1, std::min(fo, bottom_y));
                int16_t ct = std::max((int16_t)0, std::max(co, top_y));
                bool vis = cb >= ct;
                if (vis) {
                    A.columns.push_back(dg_bitmap_column{(int16_t)x, ct, cb, bottom_y, top_y});
                    r.n_cols++;
                }
                if (planes_here && vis) {
                    bool added = false;
                    if (cb < fo && cb != hm1) { plane_add(pb, A.floor_tb, (int16_t)x, cb, fo); added = true; }
                    if (f.draw_ceiling && ct > co && ct != -1) { plane_add(pt, A.ceil_tb, (int16_t)x, co, ct); added = true; }
                    if (!added) {
                        plane_flush(pb, false, s.floor_flat, s.floor_h, s.light);
                        plane_flush(pt, true, s.ceil_flat, s.ceil_h, s.light);
                    }
                } else if (planes_here && !vis && fo > co) {        // occluded wall, open vertical gap (segs.rs:293-318)
                    if (bottom_y <= co) { plane_add(pb, A.floor_tb, (int16_t)x, co, fo); occlude(x); }
                    if (f.draw_ceiling && top_y >= fo) { plane_add(pt, A.ceil_tb, (int16_t)x, co, fo); occlude(x); }
                }
                if (!f.two_sided_mid && vis) {
                    if (f.only_occlusions) {
                        A.floor_ocl[(size_t)x] = cb;
                        if (f.draw_ceiling) A.ceil_ocl[(size_t)x] = ct;
                    }
                    if (f.lower) A.floor_ocl[(size_t)x] = ct;
                    if (f.upper) A.ceil_ocl[(size_t)x] = cb;
                }
            } else {
                plane_flush(pb, false, s.floor_flat, s.floor_h, s.light);
                plane_flush(pt, true, s.ceil_flat, s.ceil_h, s.light);
            }
            if (!f.two_sided_mid && full_height) occlude(x);
        }
        plane_flush(pb, false, s.floor_flat, s.floor_h, s.light);
        plane_flush(pt, true, s.ceil_flat, s.ceil_h, s.light);

        recs.push_back(r);
        // inline draw of solid / upper / lower parts (segs.rs:231-258)
        if (!f.two_sided_mid && !f.only_occlusions && tex >= 0 && r.n_cols > 0) emit_draw(recs.back());
    }

    static const char *fs_message(int32_t code) {
        switch (code) {
            case FS_FAIL_CLIP_X: return "Clipped line x < -0.01 (segs.rs:431-436)";
            case FS_FAIL_FLAT: return "Could not find flat lump (Flat::new unwrap, flats.rs:117)";
            case FS_FAIL_TEXTURE: return "Unknown texture (Textures::get panics, textures.rs:158)";
            case FS_FAIL_VERTICAL: return "Wall start not vertical (segs.rs:140-145)";
            case FS_FAIL_LINE_X: return "Invalid line x (segs.rs:103-111)";
            case FS_FAIL_ROTATION: return "Invalid rotation (sprites.rs:106-108)";
            case FS_FAIL_MOBJ_CLIP_X: return "Clipped line x < -0.01 (map_objects.rs:92-97)";
            case FS_FAIL_MOBJ_COLUMN: return "map object column out of range (index panic)";
            default: return "zero-sized bitmap";
        }
    }

    // Segs::process_seg, segs.rs:353-590: fs_seg (fs_core.h) classifies the seg and lists its process_sidedef calls
    void process_seg(size_t seg_index) {
        const FsSeg &sg = sc.fs_segs[seg_index];
        FsSegOut so;
        const int16_t light = sg.front_sector >= 0 ? sector_light(sg.front_sector) : (int16_t)0;
        const int32_t st = fs_seg(k, sg, sc.fs_sectors.data(), sc.fs_anims.data(), ppos, view.cos_na, view.sin_na, player_height, view.timestamp, light, so);
        if (st == FS_SKIP) return;
        if (st != FS_OK) { fail(fs_message(st)); return; }
        for (int i = 0; i < so.n_calls && !status; i++)
            if ((so.call_mask >> i) & 1u) process_sidedef(so, so.call[i]);
    }

    // Renderer::render_node, mod.rs:69-104 — iterative, front child first, no culling (the reference has none)
    // Can any seg inside the box (map coordinates) still matter?  No if the box lies behind the viewer or
    // outside the 90-degree frustum (clip_to_viewport rejects every seg in it), or if the screen columns it can project to
    // are all spanned by earlier full-height solid parts (every part of every seg in it would be dropped by
    // solid_cols.covers).  The column range is widened by two columns against f32 rounding of the per-seg projection;
    // boxes that straddle the viewer's depth-zero plane are always walked.
    //
    // Why two columns are enough (W <= 16384; tests/test_host_logic.py::test_subtree_cull_at_16384_columns_next_to_walls is the stress
    // case).  Past the `xmin < 1` test all four corners have view depth x >= 1.  In exact arithmetic every seg endpoint e inside the
    // box maps to a point p of the convex hull of the transformed corners, x(p) >= 1 there, and t = y / x — a ratio of affine
    // functions, monotone along any segment — takes its extremes over the hull at corners: t(p) in [tmin, tmax].  clip_to_viewport
    // (misc.rs:13-115) replaces an endpoint only by a point of the same seg on y = x or y = -x, so every endpoint that reaches
    // make_sidedef_non_vertical_line has t in [max(tmin, -1), min(tmax, 1)]: the interval projected below.  In f32: corner and
    // endpoint go through the same six operations of rot(sub(.)), each with relative error u = 2^-24, so |dx|, |dy| <= 4.3 u r with
    // r = |e - pos|; a kept endpoint has |y| <= x, i.e. r <= 1.42 x, which bounds its error in t by (|dy| + |t| |dx|) / x <= 12.3 u;
    // an intersection computed by Line::intersection (geometry.rs:56-82) adds a few more operations — 32 u = 1.9e-6 is generous.
    // sx = trunc(CFX - ARC * (GCFX * y / x)) (misc.rs:147-158) then moves by at most K * 1.9e-6 = 0.016 columns (K = ARC * GCFX <=
    // 8192) plus three roundings of values below 16384 (<= 0.003 columns): under 0.02 columns between the corner-derived bounds and
    // any endpoint's sx, before truncation.  floor() of the bounds minus / plus 2 therefore leaves a margin of more than 1.9 columns.
    bool box_matters(const float *bb) const {
        if (bb[0] > bb[2]) return false;                             // no segs below this child
        float tmin = 3.0e38f, tmax = -3.0e38f, xmin = 3.0e38f, xmax = -3.0e38f;
        for (int c = 0; c < 4; c++) {
            V2 v = rot(sub(V2{bb[(c & 1) ? 2 : 0], bb[(c & 2) ? 3 : 1]}, ppos), view.cos_na, view.sin_na);
            xmin = std::fmin(xmin, v.x); xmax = std::fmax(xmax, v.x);
            if (v.x > 0.0f) { float t = v.y / v.x; tmin = std::fmin(tmin, t); tmax = std::fmax(tmax, t); }
        }
        if (xmax < -1.0f) return false;                              // wholly behind the viewer
        if (xmin < 1.0f) return true;                                // around or next to the viewer: walk it
        if (tmin > 1.001f || tmax < -1.001f) return false;           // wholly outside the frustum (|y| <= x)
        const float K = k.ARC * k.GCFX;                              // sx = CFX - K * (y / x), misc.rs:147-158
        float lo = k.CFX - K * std::fmin(tmax, 1.0f), hi = k.CFX - K * std::fmax(tmin, -1.0f);
        int ilo = std::max(0, (int)std::floor(lo) - 2), ihi = std::min(k.W - 1, (int)std::floor(hi) + 2);
        return !solid_cols.covers(ilo, ihi);
    }

    void walk_bsp() {
        int16_t stack[256];
        const float *box[256];
        int sp = 0;
        const bool cull = !sc.may_panic;                             // a seg whose lookup would panic must be reached
        stack[sp] = (int16_t)(sc.nodes.size() - 1);
        box[sp++] = nullptr;
        while (sp > 0 && !status) {
            --sp;
            int16_t c = stack[sp];
            if (cull && box[sp] && !box_matters(box[sp])) continue;
            if (c & (int16_t)0x8000) {
                const SubSectorRec &ss = sc.subsectors[(size_t)(c & 0x7fff)];
                for (int i = 0; i < ss.count && !status; i++) process_seg((size_t)(ss.first + i));
                continue;
            }
            const NodeRec &n = sc.nodes[(size_t)c];
            V2 v1{n.x, n.y}, v2{n.x + n.dx, n.y + n.dy};
            bool is_left = left_of(ppos, Seg2{v1, v2});
            int16_t front = is_left ? n.lchild : n.rchild, back = is_left ? n.rchild : n.lchild;
            if (sp + 2 > 256) { fail("BSP deeper than 256"); return; }
            stack[sp] = back;
            box[sp++] = n.bb[is_left ? 0 : 1];
            stack[sp] = front;
            box[sp++] = n.bb[is_left ? 1 : 0];
        }
    }

    static bool behind(const Rec &r, V2 v) {                         // is_behind_vertex, bitmap_render.rs:137-165
        if (r.min_x > v.x) return true;
        if (r.max_x > v.x && !left_of(v, r.line)) return true;
        return false;
    }

    // draw_map_objects, renderer/map_objects.rs:19-241
    void map_objects() {
        const int W = k.W, H = k.H;
        const size_t n_wall_recs = recs.size();
        if (parts_mode) A.behind_words = (uint32_t)((n_wall_recs + 31) / 32);
        std::vector<uint32_t> mo;   // indices of map-object records in recs
        for (size_t mi = 0; mi < sc.mobjs.size(); mi++) {
            const MapObjectRec &m = sc.mobjs[mi];
            int32_t m_sprite_frame, m_full_bright;
            mobj_state(mi, m_sprite_frame, m_full_bright);
            if (m_sprite_frame < 0) continue;                         // S_NULL
            if (parts_mode) {                                         // the per-column half runs on the GPU: record the sprite (fs_core.h)
                FsSpriteOut so;
                std::memset(&so.sp, 0, sizeof so.sp);
                const FsMobj &fm = sc.fs_mobjs[mi];
                const int32_t st = fs_mobj(k, fm, sc.sprite_frames_fs()[(size_t)m_sprite_frame], sc.fs_bitmaps.data(), sc.fs_sectors.data(), ppos, view.angle,
                                           view.cos_na, view.sin_na, player_height, m_full_bright, fm.sector >= 0 ? sector_light(fm.sector) : (int16_t)0, so);
                if (st == FS_SKIP) continue;
                if (st == FS_FAIL_PARTS) { fail_parts(fs_message(st)); return; }
                if (st != FS_OK) { fail(fs_message(st)); return; }
                Rec r;
                std::memset(&r, 0, sizeof r);
                r.line = so.line;
                r.min_x = std::fmin(r.line.a.x, r.line.b.x); r.max_x = std::fmax(r.line.a.x, r.line.b.x);
                r.first_col = (uint32_t)A.sprites.size(); r.out_index = -1;   // first_col: index into A.sprites
                r.state = ST_MAPOBJECT;
                r.sort_key = (int16_t)so.sort_key;
                // which wall records clip this sprite (the ones NOT behind its centre, map_objects.rs:138-140)
                const size_t row = A.behind.size();
                A.behind.resize(row + A.behind_words, 0u);
                for (size_t ri = 0; ri < n_wall_recs; ri++)
                    if (behind(recs[ri], so.centre)) A.behind[row + (ri >> 5)] |= 1u << (ri & 31);
                so.sp.behind_off = (uint32_t)row;
                A.sprites.push_back(so.sp);
                mo.push_back((uint32_t)recs.size());
                recs.push_back(r);
                continue;
            }
            float angle = view.angle - m.angle - kPi;
            angle += kPi / 16.0f;
            angle = std::fmod(angle, 2.0f * kPi);
            if (angle < 0.0f) angle += 2.0f * kPi;
            angle = std::fmod(angle, 2.0f * kPi);
            int rotation = f32_as_u8(angle * 8.0f / (2.0f * kPi));
            if (rotation > 7) { fail("Invalid rotation (sprites.rs:106-108)"); return; }
            const SpriteFrameRec &sf = sc.sprite_frames[(size_t)m_sprite_frame];
            int bitmap = sf.rotate ? sf.bitmap[rotation] : sf.bitmap[0];
            const BitmapInfo &bi = sc.bitmaps[(size_t)bitmap];

            V2 vpv = rot(sub(V2{m.x, m.y}, ppos), view.cos_na, view.sin_na);
            int16_t width = (int16_t)bi.w;
            V2 a = sub(vpv, V2{0.0f, (float)(int16_t)(-width) / 2.0f});
            V2 b = sub(vpv, V2{0.0f, (float)width / 2.0f});
            Clipped cl;
            if (!clip_to_viewport(Seg2{a, b}, cl)) continue;
            if (cl.line.a.x < -0.01f) { fail("Clipped line x < -0.01 (map_objects.rs:92-97)"); return; }
            if (m.sector < 0) continue;                               // "Thing is outside map"
            const SectorRec &sec = sc.sectors[(size_t)m.sector];
            int16_t light = m_full_bright ? (int16_t)255 : sector_light(m.sector);

            int16_t bh = (int16_t)bi.h;
            float bottom_height = (float)sec.floor_h - player_height;
            float top_height = (float)sec.floor_h + (float)bh - 1.0f - player_height;
            bottom_height += (float)bi.top_offset - (float)bh;
            top_height += (float)bi.top_offset - (float)bh;
            ScreenLine bot = project(k, cl.line, bottom_height);
            ScreenLine top = project(k, cl.line, top_height);

            int x0 = wrap_i16(bot.sx), x1 = wrap_i16(bot.ex);        // columns [x0, x1)
            if (x0 < x1 && (x0 < 0 || x1 > W)) { fail("map object column out of range (index panic)"); return; }
            for (int x = x0; x < x1; x++) { A.top_clip[(size_t)x] = -1; A.bottom_clip[(size_t)x] = (int16_t)H; }
            if (x0 < x1) {
                for (size_t ri = 0; ri < n_wall_recs; ri++) {         // :135-166 (only x in [x0,x1) is ever read back)
                    const Rec &r = recs[ri];
                    if (r.n_cols == 0 || behind(r, vpv)) continue;
                    bool solid = r.state == ST_SOLID;
                    if (solid && !r.ext_bottom && !r.ext_top) continue;
                    const dg_bitmap_column *c0 = &A.columns[r.first_col], *c1 = c0 + r.n_cols;
                    const dg_bitmap_column *c = std::lower_bound(c0, c1, x0, [](const dg_bitmap_column &q, int v) { return q.x < v; });
                    for (; c < c1 && c->x < x1; ++c) {
                        size_t x = (size_t)c->x;
                        if (solid) {
                            if (r.ext_bottom) A.bottom_clip[x] = std::min(A.bottom_clip[x], c->clipped_top_y);
                            if (r.ext_top) A.top_clip[x] = std::max(A.top_clip[x], c->clipped_bottom_y);
                        } else {                                       // ST_TWOSIDED (nothing is ST_DRAWN yet)
                            if (r.draw_ceiling) A.top_clip[x] = std::max(A.top_clip[x], c->top_y);
                            A.bottom_clip[x] = std::min(A.bottom_clip[x], c->bottom_y);
                        }
                    }
                }
            }
            Rec r;
            r.line = cl.line; r.start_offset = cl.start_offset;
            r.bottom_height = bottom_height; r.top_height = top_height;
            r.min_x = std::fmin(r.line.a.x, r.line.b.x); r.max_x = std::fmax(r.line.a.x, r.line.b.x);
            r.start_x = bot.sx; r.end_x = bot.ex; r.bitmap = bitmap;
            r.first_col = (uint32_t)A.columns.size(); r.n_cols = 0; r.out_index = -1;
            r.light = light; r.offset_x = 0; r.offset_y = 0;
            r.state = ST_MAPOBJECT; r.ext_bottom = r.ext_top = r.draw_ceiling = 0;
            r.sort_key = (int16_t)f32_as_i16(cl.line.a.x);
            float bottom_delta = ((float)bot.sy - (float)bot.ey) / ((float)bot.sx - (float)bot.ex);
            float top_delta = ((float)top.sy - (float)top.ey) / ((float)top.sx - (float)top.ex);
            for (int x = x0; x < x1; x++) {
                int16_t bottom_y = (int16_t)f32_as_i16((float)bot.sy + ((float)x - (float)bot.sx) * bottom_delta);
                int16_t top_y = (int16_t)f32_as_i16((float)top.sy + ((float)x - (float)top.sx) * top_delta);
                int16_t ct = std::max((int16_t)0, std::max(top_y, A.top_clip[(size_t)x]));
                int16_t cb = std::min((int16_t)(H - 1), std::min(bottom_y, A.bottom_clip[(size_t)x]));
                A.columns.push_back(dg_bitmap_column{(int16_t)x, ct, cb, bottom_y, top_y});
                r.n_cols++;
            }
            mo.push_back((uint32_t)recs.size());
            recs.push_back(r);
        }
        // sort() is stable ascending on the key, then reverse()  (map_objects.rs:216-217)
        std::stable_sort(mo.begin(), mo.end(), [&](uint32_t p, uint32_t q) { return recs[p].sort_key < recs[q].sort_key; });
        std::reverse(mo.begin(), mo.end());

        if (parts_mode) {                                            // same interleave, recorded as sequence numbers
            uint32_t seq = 0;
            for (uint32_t mi : mo) {
                const Rec &m = recs[mi];
                V2 v{(m.line.a.x + m.line.b.x) / 2.0f, (m.line.a.y + m.line.b.y) / 2.0f};
                for (size_t ri = n_wall_recs; ri-- > 0;) {
                    Rec &r = recs[ri];
                    if (r.state != ST_TWOSIDED || !behind(r, v)) continue;
                    A.parts[ri].seq = seq++;
                    r.state = ST_DRAWN;
                }
                A.sprites[m.first_col].seq = seq++;
            }
            for (size_t ri = n_wall_recs; ri-- > 0;) {
                Rec &r = recs[ri];
                if (r.state != ST_TWOSIDED) continue;
                A.parts[ri].seq = seq++;
                r.state = ST_DRAWN;
            }
            return;
        }
        // `segs` was reversed before this call (mod.rs:124): iterate wall records back to front
        for (uint32_t mi : mo) {
            const Rec &m = recs[mi];
            V2 v{(m.line.a.x + m.line.b.x) / 2.0f, (m.line.a.y + m.line.b.y) / 2.0f};
            for (size_t ri = n_wall_recs; ri-- > 0;) {
                Rec &r = recs[ri];
                if (r.state != ST_TWOSIDED || !behind(r, v)) continue;
                if (r.bitmap >= 0 && r.n_cols > 0) emit_draw(r);
                r.state = ST_DRAWN;
            }
            if (recs[mi].n_cols > 0) emit_draw(recs[mi]);
        }
        for (size_t ri = n_wall_recs; ri-- > 0;) {                   // draw_remaining_segs, segs.rs:593-597
            Rec &r = recs[ri];
            if (r.state != ST_TWOSIDED) continue;
            if (r.bitmap >= 0 && r.n_cols > 0) emit_draw(r);
            r.state = ST_DRAWN;
        }
    }
};

}  // namespace

int build_frame_lists(const Scene &sc, int W, int H, const dg_view &view, FrameArena &A, dg_frame_lists &out, std::string &err, const dg_view_state *state) {
    if (W <= 0 || H <= 0 || W > 16384 || H > 16384) { err = "bad frame size"; return DG_ERR_INVALID; }
    A.renders.clear(); A.columns.clear(); A.visplanes.clear(); A.plane_tb.clear(); A.order.clear();
    A.recs->clear(); A.floor_tb.clear(); A.ceil_tb.clear();
    A.hor_ocl.assign((size_t)W, 0);                                  // Segs::new, segs.rs:97-99
    A.floor_ocl.assign((size_t)W, (int16_t)H);
    A.ceil_ocl.assign((size_t)W, (int16_t)-1);
    A.top_clip.resize((size_t)W);
    A.bottom_clip.resize((size_t)W);

    Walker wk(sc, W, H, view, A, err, state);
    if (wk.status) return wk.status;
    wk.walk_bsp();
    if (wk.status) return wk.status;
    // visplanes are drawn after all inline walls, in push order (mod.rs:122)
    const uint32_t n_floor_entries = (uint32_t)(A.floor_tb.size() / 2);
    A.plane_tb.reserve(A.floor_tb.size() + A.ceil_tb.size());
    A.plane_tb.insert(A.plane_tb.end(), A.floor_tb.begin(), A.floor_tb.end());
    A.plane_tb.insert(A.plane_tb.end(), A.ceil_tb.begin(), A.ceil_tb.end());
    for (uint32_t i = 0; i < A.visplanes.size(); i++) {
        dg_visplane &v = A.visplanes[i];
        if (v.first_entry & Walker::kCeilPool) v.first_entry = (v.first_entry & ~Walker::kCeilPool) + n_floor_entries;
        A.order.push_back(dg_draw_cmd{1u, i});
    }
    wk.map_objects();
    if (wk.status) return wk.status;

    out.view = view;
    out.renders = A.renders.data(); out.n_renders = (uint32_t)A.renders.size();
    out.columns = A.columns.data(); out.n_columns = (uint32_t)A.columns.size();
    out.visplanes = A.visplanes.data(); out.n_visplanes = (uint32_t)A.visplanes.size();
    out.plane_tb = A.plane_tb.data(); out.n_plane_tb = (uint32_t)A.plane_tb.size();
    out.order = A.order.data(); out.n_order = (uint32_t)A.order.size();
    return DG_OK;
}

namespace {
// Counting sort of records into FE_BIN_W-column strips, keeping their order (lo/hi: first / last column, inclusive).
template <typename T, typename Range>
void bin_by_columns(const std::vector<T> &recs, int W, Range range, std::vector<uint32_t> &off, std::vector<uint16_t> &idx) {
    const int nb = (W + FE_BIN_W - 1) / FE_BIN_W;
    off.assign((size_t)nb + 1, 0u);
    for (const T &r : recs) {
        int lo, hi;
        if (!range(r, lo, hi)) continue;
        for (int b = lo / FE_BIN_W; b <= hi / FE_BIN_W; b++) off[(size_t)b + 1]++;
    }
    for (int b = 0; b < nb; b++) off[(size_t)b + 1] += off[(size_t)b];
    idx.resize(off[(size_t)nb]);
    std::vector<uint32_t> cur(off.begin(), off.end() - 1);
    for (size_t i = 0; i < recs.size(); i++) {
        int lo, hi;
        if (!range(recs[i], lo, hi)) continue;
        for (int b = lo / FE_BIN_W; b <= hi / FE_BIN_W; b++) idx[cur[(size_t)b]++] = (uint16_t)i;
    }
}
}  // namespace

int build_frame_parts(const Scene &sc, int W, int H, const dg_view &view, FrameArena &A, std::string &err, const dg_view_state *state) {
    if (W <= 0 || H <= 0 || W > 16384 || H > 16384) { err = "bad frame size"; return DG_ERR_INVALID; }
    A.parts.clear(); A.sprites.clear(); A.behind.clear(); A.sky_parts.clear(); A.behind_words = 0; A.n_sky_slots = 0;
    A.recs->clear();
    Walker wk(sc, W, H, view, A, err, state);
    if (wk.status) return wk.status;
    wk.parts_mode = true;
    wk.walk_bsp();
    if (wk.status) return wk.status;
    wk.map_objects();
    if (wk.status) return wk.status;
    if (A.sprites.size() > 65535) { err = "more than 65535 sprites in a frame"; return kPartsUnsupported; }
    bin_by_columns(A.parts, W, [](const FePart &p, int &lo, int &hi) { lo = p.sx; hi = p.ex; return true; }, A.bin_off, A.bin_parts);
    bin_by_columns(A.sprites, W, [](const FeSprite &s, int &lo, int &hi) { lo = s.x0; hi = s.x1 - 1; return s.x0 < s.x1; }, A.sbin_off, A.sbin_sprites);
    return DG_OK;
}

}  // namespace dg
