// binner.cpp — turns the reference-shaped, draw-ordered lists into what the raster kernel walks:
// for every screen column, the vertical spans that touch it, in draw order.  The last span covering a pixel
// with an opaque texel owns it — exactly the reference's "later Pixels::set wins" (SURVEY.md Appendix A).
// Also hoists the per-record constants of the texture mappers (same f32 operations the reference performs
// per column / per pixel, done once).
#include "binner.hpp"

#include <cmath>
#include <cstring>

#include "rust_num.h"

namespace dg {

namespace {
const float kPi = 3.14159265358979323846f;
const uint32_t kMaxSpansPerColumn = 512;   // = SPAN_CAP of dg_raster_tiles (kernels.hip)
}

DevFrame make_frame_header(const dg_view &v) {
    DevFrame h;
    std::memset(&h, 0, sizeof h);
    h.cos_a = v.cos_a;
    h.sin_a = v.sin_a;
    h.pos_x_i16 = f32_as_i16(v.x);
    h.pos_y_i16 = f32_as_i16(v.y);
    // draw_sky's tx_offset, visplanes.rs:51-58
    int32_t t = wrap_i16(f32_as_i16(-256.0f * v.angle / (kPi / 2.0f)) + 256);
    if (t < 0) t = wrap_i16(t + wrap_i16(256 * wrap_i16(1 - t / 256)));
    h.sky_tx_offset = t;
    return h;
}

int bin_frame(const Scene &sc, const FrameConsts &k, const dg_frame_lists &fl, BinnedFrame &out, std::string &err) {
    const int W = k.W, H = k.H;
    out.events.clear(); out.walls.clear(); out.planes.clear(); out.covered_pixels = 0;
    const dg_view &v = fl.view;

    out.hdr = make_frame_header(v);
    const float wz_base = v.floor_height;

    for (uint32_t oi = 0; oi < fl.n_order; oi++) {
        const dg_draw_cmd &cmd = fl.order[oi];
        if (cmd.kind == 0) {
            if (cmd.index >= fl.n_renders) { err = "draw order references a missing render record"; return DG_ERR_INVALID; }
            const dg_bitmap_render &r = fl.renders[cmd.index];
            if (r.bitmap < 0 || (size_t)r.bitmap >= sc.bitmaps.size()) { err = "render record has an invalid bitmap id"; return DG_ERR_INVALID; }
            if ((uint64_t)r.first_column + r.n_columns > fl.n_columns) { err = "render record column range out of bounds"; return DG_ERR_INVALID; }
            const BitmapInfo &bi = sc.bitmaps[(size_t)r.bitmap];
            if (bi.w <= 0 || bi.h <= 0) { err = "zero-sized bitmap (reference divides by zero)"; return DG_ERR_RENDER; }
            if (out.walls.size() >= 65535) { err = "more than 65535 drawn records in a frame"; return DG_ERR_CAPACITY; }
            const DevWallRec d = make_wall_rec(bi, r.line_start_x, r.line_start_y, r.line_end_x, r.line_end_y, r.start_offset, r.start_x, r.end_x,
                                               r.bottom_height, r.top_height, r.offset_x, r.offset_y, r.light_level);
            uint16_t rec = (uint16_t)out.walls.size();
            out.walls.push_back(d);
            const dg_bitmap_column *c = fl.columns + r.first_column;
            for (uint32_t i = 0; i < r.n_columns; i++, c++) {
                if (c->x < 0 || c->x >= W) continue;                    // Pixels::set drops x >= W (pixels.rs:23)
                int ct = c->clipped_top_y < 0 ? 0 : c->clipped_top_y;
                int cb = c->clipped_bottom_y > H - 1 ? H - 1 : c->clipped_bottom_y;
                if (ct > cb) continue;                                  // empty y range (sprites clipped away)
                DevSpan s;
                s.ctop = (int16_t)ct; s.cbot = (int16_t)cb; s.top_y = c->top_y; s.bot_y = c->bottom_y;
                s.rec = rec; s.kind = SPAN_WALL; s.pad0 = 0; s.x = c->x; s.pad1 = 0;
                out.events.push_back(s);
                out.covered_pixels += (uint64_t)(cb - ct + 1);
            }
        } else if (cmd.kind == 1) {
            if (cmd.index >= fl.n_visplanes) { err = "draw order references a missing visplane"; return DG_ERR_INVALID; }
            const dg_visplane &p = fl.visplanes[cmd.index];
            if (p.flat < 0 || (size_t)p.flat >= sc.flat_names.size()) { err = "visplane has an invalid flat id"; return DG_ERR_INVALID; }
            if (p.left < 0 || p.right >= W || p.right < p.left) { err = "visplane x range out of bounds"; return DG_ERR_INVALID; }
            uint64_t n_ent = (uint64_t)(p.right - p.left + 1);
            if (((uint64_t)p.first_entry + n_ent) * 2 > fl.n_plane_tb) { err = "visplane top/bottom range out of bounds"; return DG_ERR_INVALID; }
            const bool sky = sc.flat_sky[(size_t)p.flat] != 0;
            uint16_t rec = 0;
            if (!sky) {
                if (out.planes.size() >= 65535) { err = "more than 65535 visplanes in a frame"; return DG_ERR_CAPACITY; }
                DevPlaneRec d;
                d.wz = (float)p.height - wz_base - 41.0f;               // visplanes.rs:112
                d.gwz = k.GCFX * d.wz;                                  // visplanes.rs:113 (GCFX * wz) / vy
                d.lightf = (float)p.light_level / 255.0f;
                d.flat_off = (uint32_t)p.flat * 4096u;
                rec = (uint16_t)out.planes.size();
                out.planes.push_back(d);
            } else {
                const BitmapInfo &sb = sc.bitmaps[(size_t)sc.sky_bitmap];
                if (sb.w < 256 || sb.h < 128) { err = "sky texture smaller than 256x128 (reference index panic, visplanes.rs:74)"; return DG_ERR_RENDER; }
            }
            const int16_t *tb = fl.plane_tb + (size_t)p.first_entry * 2;
            for (int x = p.left; x <= p.right; x++, tb += 2) {
                int top = tb[0] < 0 ? 0 : tb[0];                        // visplanes.rs:61-62 / 95-96
                int bot = tb[1] > H - 1 ? H - 1 : tb[1];
                if (sky) {
                    if (top > bot) continue;
                } else if (wrap_i16(bot - top) <= 1) {                  // visplanes.rs:99-101
                    continue;
                }
                DevSpan s;
                s.ctop = (int16_t)top; s.cbot = (int16_t)bot; s.top_y = 0; s.bot_y = 0;
                s.rec = rec; s.kind = sky ? SPAN_SKY : SPAN_FLAT; s.pad0 = 0; s.x = (int16_t)x; s.pad1 = 0;
                out.events.push_back(s);
                out.covered_pixels += (uint64_t)(bot - top + 1);
            }
        } else {
            err = "unknown draw command kind";
            return DG_ERR_INVALID;
        }
    }

    // stable counting sort of the events by column
    out.col_off.assign((size_t)W + 1, 0);
    for (const DevSpan &s : out.events) out.col_off[(size_t)s.x + 1]++;
    for (int x = 0; x < W; x++) {
        if (out.col_off[(size_t)x + 1] > kMaxSpansPerColumn) { err = "more than 512 spans in one screen column (raster kernel staging limit)"; return DG_ERR_CAPACITY; }
        out.col_off[(size_t)x + 1] += out.col_off[(size_t)x];
    }
    out.cursor.assign(out.col_off.begin(), out.col_off.end() - 1);
    out.spans.resize(out.events.size());
    for (const DevSpan &s : out.events) out.spans[out.cursor[(size_t)s.x]++] = s;
    out.hdr.n_spans = (uint32_t)out.spans.size();
    return DG_OK;
}

}  // namespace dg
