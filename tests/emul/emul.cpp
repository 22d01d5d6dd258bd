// tests/emul/emul.cpp — TEST HARNESS ONLY (never part of libdoomgpu, never loaded by the package).
//
// Compiles the product's host list generation (scene.cpp, frontend.cpp, binner.cpp) together with the
// kernel *bodies* (raster_core.h) for the CPU, and replays the column-major span lists the way the raster
// kernel does (per column, spans in order, later span overwrites).  It lets the CPU-only test tier check the
// host logic and the list format against the oracle in a container without a GPU.  The GPU tier repeats the
// same comparison through the real HIP kernels and the C-ABI.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>

#include "../../doom-rust-renderer_amd/csrc/binner.hpp"
#include "../../doom-rust-renderer_amd/csrc/fe_core.h"
#include "../../doom-rust-renderer_amd/csrc/frontend.hpp"
#include "../../doom-rust-renderer_amd/csrc/raster_core.h"
#include "../../doom-rust-renderer_amd/csrc/scene.hpp"

using namespace dg;

static std::string g_err;

// What dg_raster_tiles does for every column: spans in order, every row of the span (lane = row), later span overwrites.
static void raster_spans(const DevScene &ds, const DevConsts &k, const DevFrame &hdr, const uint32_t *pal, const uint32_t *col_off,
                         const DevRSpan *rs, int W, int H, uint8_t *rgb) {
    std::memset(rgb, 0, (size_t)3 * W * H);
    for (int x = 0; x < W; x++) {
        for (uint32_t i = col_off[x]; i < col_off[x + 1]; i++) {
            const uint32_t *w = rs[i].w;
            for (int y = w0_ctop(w[0]); y <= w0_cbot(w[0]); y++) {
                uint32_t c = 0;
                bool wr = false;
                const uint32_t kind = w0_kind(w[0]);
                if (kind == SPAN_WALL) {
                    uint32_t o = wall_texel_offset(w[1], w[2], w[4], w[5], w[6], w[7], y);
                    if (!w0_immediate(w[0]) || ds.texel_opq[o]) { c = shade(pal[ds.texel_idx[o]], bits_f32(w[3])); wr = true; }
                } else if (kind == SPAN_FLAT) {
                    float factor;
                    const float vy = k.CFY - (float)y;
                    uint32_t o = flat_texel_offset(hdr, w[1], w[2], w[4], w[5], w[6], vy, prepare_rcp(vy), factor);
                    c = shade(pal[ds.texel_idx[o]], factor); wr = true;
                } else {
                    uint32_t o = sky_texel_offset(w[2], w[3], sky_row(ds, k, y));
                    if (o != 0xffffffffu && ds.texel_opq[o]) { c = pal[ds.texel_idx[o]]; wr = true; }
                }
                if (wr) {
                    uint8_t *p = rgb + 3 * ((size_t)y * W + x);
                    p[0] = c & 255; p[1] = (c >> 8) & 255; p[2] = (c >> 16) & 255;
                }
            }
        }
    }
}

// ---- statistics only (tools/tile_stats.py): which (screen column, 64-row tile) chunks have ONE owner — what the sole-owner fast path of
// dg_raster_tiles finds with its wave ballots (kernels.hip, ucol_*) ---------------------------------------------------------------
//   CH_EMPTY    no span touches the chunk's rows: the pixels are the frame's clear colour
//   CH_WALL /   the last span (in draw order) that touches the rows is opaque, plain (pack_w0) and covers every live row of the chunk:
//   CH_FLAT /   it owns all of them whatever lies underneath — no owner search, no per-row parameters, no kind vote; the low bits are
//   CH_SKY      that span's index in the frame's span array
//   CH_GENERIC  anything else (several owners, possibly-transparent spans on top, long modulus, the horizon row of a floor, a tile
//               with at most `pack_rows` live rows): the draw-ordered walk
enum : uint32_t { CH_EMPTY = 0, CH_WALL = 1, CH_FLAT = 2, CH_SKY = 3, CH_GENERIC = 4, CH_CLASS_SHIFT = 28, CH_INDEX_MASK = (1u << 28) - 1u };

// spans [c0, c1) of the column (w0_at(i) = word 0 of span i), rows y0 .. y_last of a frame of height H; vy0_row: the row with
// CFY - y == 0 (or -1), which a floor / ceiling may only cross with the plain divide (visplanes.rs:113-114).
template <typename W0At>
static uint32_t classify_chunk(W0At w0_at, uint32_t c0, uint32_t c1, int32_t y0, int32_t y_last, int32_t H, int32_t vy0_row, int32_t pack_rows) {
    uint32_t last = 0xffffffffu, last_w0 = 0;
    for (uint32_t i = c0; i < c1; i++) {
        const uint32_t w0 = w0_at(i);
        if (w0_cbot(w0) >= y0 && w0_ctop(w0) <= y_last) { last = i; last_w0 = w0; }
    }
    if (last == 0xffffffffu) return CH_EMPTY << CH_CLASS_SHIFT;
    const uint32_t kind = w0_kind(last_w0);
    const bool covers = w0_ctop(last_w0) <= y0 && w0_cbot(last_w0) >= y_last && !w0_immediate(last_w0);
    (void)vy0_row;
    uint32_t cls = CH_GENERIC;
    if (covers && kind == SPAN_WALL && w0_plain(last_w0)) cls = CH_WALL;
    else if (covers && kind == SPAN_FLAT && w0_plain(last_w0)) cls = CH_FLAT;         // (the horizon row is handled in place: x * +inf)
    if (H - y0 <= pack_rows || last >= (1u << 27)) cls = CH_GENERIC;      // (few live rows: the packed pass is cheaper; the raster kernel shifts the index left by 5 in 32 bits)
    return (cls << CH_CLASS_SHIFT) | (last & CH_INDEX_MASK);
}

// What the sole-owner fast path of dg_raster_tiles does: every (column, 64-row tile) chunk that is not CH_GENERIC is
// rendered from its ONE owner span (or left black), and must equal the draw-order replay.  Returns false (and says why) otherwise;
// counts[class] += chunks.
static bool check_chunk_descriptors(const DevScene &ds, const DevConsts &k, const DevFrame &hdr, const uint32_t *pal, const uint32_t *col_off,
                                    const DevRSpan *rs, int W, int H, const uint8_t *rgb, uint64_t *counts) {
    const int vy0_row = H % 2 == 0 ? H / 2 : -1;
    for (int x = 0; x < W; x++)
        for (int y0 = 0; y0 < H; y0 += 64) {
            const int y_last = std::min(H, y0 + 64) - 1;
            const uint32_t d = classify_chunk([&](uint32_t i) { return rs[i].w[0]; }, col_off[x], col_off[x + 1], y0, y_last, H, vy0_row, 8);
            const uint32_t cls = d >> CH_CLASS_SHIFT;
            if (counts) counts[cls]++;
            if (cls == CH_GENERIC) continue;
            const uint32_t *w = rs[d & CH_INDEX_MASK].w;
            for (int y = y0; y <= y_last; y++) {
                uint32_t c = 0;
                if (cls == CH_WALL) {
                    const uint32_t h = (uint32_t)bits_f32(w[6]);
                    const int32_t top_y = lo_i16(w[5]), off_y = hi_i16(w[5]);
                    const float ay = (float)(y - top_y) / bits_f32(w[1]);
                    const int32_t ty = f32_as_i16(bits_f32(w[6]) + ay * bits_f32(w[4]));
                    c = shade(pal[ds.texel_idx[w[2] + ((uint32_t)(ty + off_y) & (h - 1u))]], bits_f32(w[3]));
                } else if (cls == CH_FLAT) {
                    float factor;
                    const float vy = k.CFY - (float)y;
                    const uint32_t o = flat_texel_offset(hdr, w[1], w[2], w[4], w[5], 0x100u, vy, 0.0f, factor);
                    c = shade(pal[ds.texel_idx[o]], factor);
                } else if (cls == CH_SKY) {
                    const int32_t row = sky_row(ds, k, y);
                    const float fac = bits_f32(w[3]) * (row < 0 ? 0.0f : 1.0f);
                    c = shade(pal[ds.texel_idx[w[2] + (row < 0 ? 0u : (uint32_t)row)]], fac);
                }
                const uint8_t *p = rgb + 3 * ((size_t)y * W + x);
                if (p[0] != (c & 255) || p[1] != ((c >> 8) & 255) || p[2] != ((c >> 16) & 255)) {
                    g_err = "chunk descriptor class " + std::to_string(cls) + " at column " + std::to_string(x) + " row " + std::to_string(y) + " differs from the draw-order replay";
                    return false;
                }
            }
        }
    return true;
}

extern "C" {

const char *emul_last_error() { return g_err.c_str(); }

void *emul_load(const uint8_t *wad, size_t len, const char *map_name) {
    return load_scene_from_wad(wad, len, map_name, g_err);
}
void emul_free(void *s) { delete (Scene *)s; }
int emul_set_sector_light(void *s, int sector, int16_t light) { ((Scene *)s)->sectors[(size_t)sector].light = light; return 0; }
int emul_set_mobj_state(void *s, int mobj, const char *sprite, uint8_t frame, int full_bright) {
    Scene *sc = (Scene *)s;
    MapObjectRec &m = sc->mobjs[(size_t)mobj];
    if (!sprite) { m.sprite_frame = -1; return 0; }
    int sf = sc->find_or_add_sprite_frame(sprite, frame, g_err);
    if (sf < 0) return -1;
    m.sprite_frame = sf; m.full_bright = full_bright;
    return 0;
}

int emul_sprite_frame(void *s, const char *sprite, uint8_t frame) { return ((Scene *)s)->find_or_add_sprite_frame(sprite, frame, g_err); }

static int emul_render_impl(void *scene, int W, int H, const dg_view *view_in, const dg_view_state *state, uint8_t *rgb, uint64_t *stats);
// stats[0..3] = spans, walls, planes, covered pixels
int emul_render(void *scene, int W, int H, const dg_view *view_in, uint8_t *rgb, uint64_t *stats) { return emul_render_impl(scene, W, H, view_in, nullptr, rgb, stats); }
// the same with a per-view game-state snapshot (include/doomgpu.h dg_view_state)
int emul_render_state(void *scene, int W, int H, const dg_view *view_in, const dg_sector_light *lights, uint32_t n_lights, const dg_mobj_state *mobjs,
                      uint32_t n_mobjs, uint8_t *rgb) {
    const dg_view_state st{lights, n_lights, mobjs, n_mobjs};
    return emul_render_impl(scene, W, H, view_in, &st, rgb, nullptr);
}
static int emul_render_impl(void *scene, int W, int H, const dg_view *view_in, const dg_view_state *state, uint8_t *rgb, uint64_t *stats) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    static thread_local BinnedFrame bf;
    dg_frame_lists fl;
    int rc = build_frame_lists(sc, W, H, view, arena, fl, g_err, state);
    if (rc) return rc;
    FrameConsts fk = make_consts(W, H);
    rc = bin_frame(sc, fk, fl, bf, g_err);
    if (rc) return rc;

    std::vector<uint32_t> pal(256);
    for (int i = 0; i < 256; i++) pal[i] = sc.palette[3 * i] | (sc.palette[3 * i + 1] << 8) | (sc.palette[3 * i + 2] << 16);
    // [texel index plane | flats] in one buffer, like the device scene: a flat span's offset is relative to the texel plane
    std::vector<uint8_t> texels(((sc.texel_idx.size() + 255) & ~(size_t)255) + sc.flat_pool.size());
    std::copy(sc.texel_idx.begin(), sc.texel_idx.end(), texels.begin());
    std::copy(sc.flat_pool.begin(), sc.flat_pool.end(), texels.begin() + (long)((sc.texel_idx.size() + 255) & ~(size_t)255));
    DevScene ds{};
    ds.palette = pal.data(); ds.texel_idx = texels.data(); ds.texel_opq = sc.texel_opq.data(); ds.flats = texels.data() + ((sc.texel_idx.size() + 255) & ~(size_t)255);
    const BitmapInfo &sky = sc.bitmaps[(size_t)sc.sky_bitmap];
    ds.sky_texel_off = sky.texel_off; ds.sky_w = sky.w; ds.sky_h = sky.h; ds.sky_has_holes = sky.has_holes;
    DevConsts k{fk.ARC, fk.GCFX, fk.CFX, fk.CFY, W, H};

    std::vector<DevRSpan> rs(bf.spans.size());
    for (size_t i = 0; i < bf.spans.size(); i++) {                       // what dg_setup_spans does (one lane per span)
        const DevSpan &sp = bf.spans[i];
        rs[i] = sp.kind == SPAN_WALL ? resolve_wall_span(sp, bf.walls[sp.rec])
              : sp.kind == SPAN_FLAT ? resolve_flat_span(sp, bf.planes[sp.rec], k, (uint32_t)(ds.flats - ds.texel_idx))
                                     : resolve_sky_span(sp, ds, k, bf.hdr);
    }
    raster_spans(ds, k, bf.hdr, pal.data(), bf.col_off.data(), rs.data(), W, H, rgb);
    if (!check_chunk_descriptors(ds, k, bf.hdr, pal.data(), bf.col_off.data(), rs.data(), W, H, rgb, nullptr)) return DG_ERR_INVALID;
    if (stats) { stats[0] = bf.spans.size(); stats[1] = bf.walls.size(); stats[2] = bf.planes.size(); stats[3] = bf.covered_pixels; }
    return 0;
}
// The device column walk (fe_core.h bodies, the loops of fe_kernels.hip restated serially) on the CPU, compared span for
// span with the host list path.  stats[0..3] = spans, parts, sprites, overflow flags; stats[4] = 1 when the column-major
// DevRSpan list is byte-identical to the host path's; stats[5] = zero-filled sky visplane entries found by fe_gap.
int emul_render_fe(void *scene, int W, int H, const dg_view *view_in, uint8_t *rgb, uint64_t *stats) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena, arena2;
    static thread_local BinnedFrame bf;
    FrameConsts fk = make_consts(W, H);
    // host path (the comparison target)
    dg_frame_lists fl;
    int rc = build_frame_lists(sc, W, H, view, arena2, fl, g_err);
    if (rc) return rc;
    rc = bin_frame(sc, fk, fl, bf, g_err);
    if (rc) return rc;
    rc = build_frame_parts(sc, W, H, view, arena, g_err);
    if (rc) return rc;
    if (arena.n_sky_slots > FE_MAX_SKY_SLOTS) { g_err = "too many sky slots"; return DG_ERR_CAPACITY; }

    std::vector<uint32_t> pal(256);
    for (int i = 0; i < 256; i++) pal[i] = sc.palette[3 * i] | (sc.palette[3 * i + 1] << 8) | (sc.palette[3 * i + 2] << 16);
    // [texel index plane | flats] in one buffer, like the device scene: a flat span's offset is relative to the texel plane
    std::vector<uint8_t> texels(((sc.texel_idx.size() + 255) & ~(size_t)255) + sc.flat_pool.size());
    std::copy(sc.texel_idx.begin(), sc.texel_idx.end(), texels.begin());
    std::copy(sc.flat_pool.begin(), sc.flat_pool.end(), texels.begin() + (long)((sc.texel_idx.size() + 255) & ~(size_t)255));
    DevScene ds{};
    ds.palette = pal.data(); ds.texel_idx = texels.data(); ds.texel_opq = sc.texel_opq.data(); ds.flats = texels.data() + ((sc.texel_idx.size() + 255) & ~(size_t)255);
    const BitmapInfo &sky = sc.bitmaps[(size_t)sc.sky_bitmap];
    ds.sky_texel_off = sky.texel_off; ds.sky_w = sky.w; ds.sky_h = sky.h; ds.sky_has_holes = sky.has_holes;
    DevConsts k{fk.ARC, fk.GCFX, fk.CFX, fk.CFY, W, H};

    DevFrame fr = bf.hdr;            // view constants (same helper as the product: bin_frame fills them)
    fr.span_base = 0;
    FeFrame ff{0, (uint32_t)arena.parts.size(), 0, (uint32_t)arena.sprites.size(), 0, arena.behind_words, arena.n_sky_slots, 0, 0, 0, {0, 0}};
    const uint32_t w64 = (uint32_t)((W + 63) / 64);
    std::vector<uint32_t> cnt((size_t)W), col_off((size_t)W + 1), flags(1, 0);
    std::vector<FeU4> cspans((size_t)FE_DEFAULT_COL_SLOTS * W);
    std::vector<FeColRec> recs((size_t)FE_DEFAULT_COL_SLOTS * W);
    const size_t ev_kind = (size_t)arena.n_sky_slots * w64;               // zeroed like the product: no add, flushed
    std::vector<uint64_t> events(3 * ev_kind + 1, 0);
    std::vector<DevRSpan> rspans((size_t)W * FE_DEFAULT_COL_SLOTS);
    FeParams P{};
    P.scene = ds; P.k = k; P.frames = &fr; P.fframes = &ff; P.parts = arena.parts.data(); P.sprites = arena.sprites.data();
    P.behind = arena.behind.data(); P.bin_off = arena.bin_off.data(); P.bin_parts = arena.bin_parts.data();
    P.sbin_off = arena.sbin_off.data(); P.sbin_sprites = arena.sbin_sprites.data(); P.cspans = cspans.data(); P.recs = recs.data(); P.cnt = cnt.data();
    P.events = events.data(); P.flags = flags.data(); P.totals = nullptr; P.col_off = col_off.data(); P.rspans = rspans.data();
    P.n_frames = 1; P.max_sky_slots = arena.n_sky_slots; P.gap_waves = 0; P.span_stride = (uint32_t)rspans.size(); P.w64 = w64; P.col_slots = FE_DEFAULT_COL_SLOTS;

    // dg_fe_columns, one "lane" at a time (each bin's near records in their own little arrays, like a wave's LDS)
    std::vector<uint32_t> near_cand((size_t)FE_NEAR_RECS * FE_BIN_W);
    std::vector<uint16_t> near_part((size_t)FE_NEAR_RECS * FE_BIN_W);
    const FeRecStore st{near_cand.data(), near_part.data()};
    for (int x = 0; x < W; x++) {
        FeColumn c = fe_column_start(P, 0, x);
        const uint32_t bin = (uint32_t)x / FE_BIN_W;                       // the wave that owns this column walks its bin's lists
        for (uint32_t bi = arena.bin_off[bin]; bi < arena.bin_off[bin + 1]; bi++) {
            const uint32_t pi = arena.bin_parts[bi];
            const FePart &p = P.parts[pi];
            if (x < p.sx || x > p.ex) continue;
            uint32_t ev = fe_part_column(P, 0, p, pi, c, st);
            if (p.sky_slot >= 0) {
                const uint64_t bit = 1ull << (x & 63);
                for (int kind = 0; kind < 3; kind++) {                 // kind 2 stores "walked and not flushed"
                    uint64_t &w = fe_event_words(P, 0, p.sky_slot, kind)[x >> 6];
                    const bool on = kind < 2 ? (ev & (1u << kind)) != 0 : !(ev & FE_EV_FLUSH);
                    w = on ? (w | bit) : (w & ~bit);
                }
            }
            if (c.hor) break;      // the rest of the bin only yields flush events = the preset (dg_fe_columns stops per wave)
        }
        for (uint32_t bi = arena.sbin_off[bin]; bi < arena.sbin_off[bin + 1]; bi++) {
            const uint32_t si = arena.sbin_sprites[bi];
            const FeSprite &s = P.sprites[si];
            if (s.behind_off != si * ff.behind_words) { g_err = "behind row of sprite i is not row i"; return DG_ERR_INVALID; }   // dg_fe_columns stages rows by index
            if (x >= s.x0 && x < s.x1) fe_sprite_column(P, 0, ff, s, si, c, st, fe_behind_row(P, ff, s));
        }
        cnt[(size_t)x] = c.nsp;
        flags[0] |= c.ovf;
    }
    // dg_fe_gaps, dg_fe_scan, dg_fe_scatter
    uint64_t n_gaps = 0;
    if (arena.sky_parts.size() != arena.n_sky_slots) { g_err = "sky_parts / n_sky_slots mismatch"; return DG_ERR_INVALID; }
    for (uint32_t si = 0; si < ff.n_sky_slots; si++) {            // dg_fe_gaps: one wave per sky slot
        const uint32_t pi = arena.sky_parts[si];
        const FePart &p = P.parts[pi];
        if (p.sky_slot != (int32_t)si) { g_err = "sky slot table does not point back at its part"; return DG_ERR_INVALID; }
        for (int kind = 0; kind < 2; kind++) {
            if (!(p.flags & (kind ? FEP_CEIL_SKY : FEP_FLOOR_SKY))) continue;
            const uint64_t *add = fe_event_words(P, 0, p.sky_slot, kind), *open = fe_event_words(P, 0, p.sky_slot, 2);
            for (int x = p.sx; x <= p.ex; x++) {
                if (!fe_gap(add, open, x, p.sx, p.ex)) continue;
                n_gaps++;
                const uint32_t slot = cnt[(size_t)x]++;
                if (slot >= FE_DEFAULT_COL_SLOTS) { flags[0] |= FE_OVF_SPANS; continue; }
                cspans[(size_t)slot * W + (size_t)x] = FeU4{FE_KEY_PLANE | (pi << 2) | (uint32_t)kind, 0u, 0u, (uint32_t)SPAN_SKY << FES_KIND_SHIFT};
            }
        }
    }
    uint32_t off = 0;
    for (int x = 0; x < W; x++) {
        const uint32_t n = std::min<uint32_t>(cnt[(size_t)x], FE_DEFAULT_COL_SLOTS);
        col_off[(size_t)x] = off;
        for (uint32_t i = 0; i < n; i++) {
            const FeU4 cs = cspans[(size_t)i * W + x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t kj = cspans[(size_t)j * W + x].x;
                rank += (kj < cs.x || (kj == cs.x && j < i)) ? 1u : 0u;
            }
            rspans[off + rank] = fe_resolve(P, fr, ff, x, cs);
        }
        off += n;
    }
    col_off[(size_t)W] = off;

    // span-for-span comparison with the host path
    bool same = off == bf.spans.size() && !flags[0];
    for (int x = 0; same && x <= W; x++) same = col_off[(size_t)x] == bf.col_off[(size_t)x];
    for (size_t i = 0; same && i < bf.spans.size(); i++) {
        const DevSpan &sp = bf.spans[i];
        DevRSpan r = sp.kind == SPAN_WALL ? resolve_wall_span(sp, bf.walls[sp.rec])
                   : sp.kind == SPAN_FLAT ? resolve_flat_span(sp, bf.planes[sp.rec], k, (uint32_t)(ds.flats - ds.texel_idx))
                                          : resolve_sky_span(sp, ds, k, bf.hdr);
        same = std::memcmp(&r, &rspans[i], sizeof r) == 0;
    }
    raster_spans(ds, k, fr, pal.data(), col_off.data(), rspans.data(), W, H, rgb);
    if (!check_chunk_descriptors(ds, k, fr, pal.data(), col_off.data(), rspans.data(), W, H, rgb, nullptr)) return DG_ERR_INVALID;
    if (stats) { stats[0] = off; stats[1] = ff.n_parts; stats[2] = ff.n_sprites; stats[3] = flags[0]; stats[4] = same ? 1 : 0; stats[5] = n_gaps; }
    return 0;
}
// Host-side cost of one frame (seconds, single thread): mode 0 = build_frame_lists + bin_frame (DG_FE_HOST), 1 = build_frame_parts (DG_FE_DEVICE).
double emul_time_front_end(void *scene, int W, int H, const dg_view *views, int n, int iters, int mode) {
    const Scene &sc = *(const Scene *)scene;
    static thread_local FrameArena arena;
    static thread_local BinnedFrame bf;
    FrameConsts fk = make_consts(W, H);
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; it++)
        for (int i = 0; i < n; i++) {
            dg_view v = views[i];
            fill_view_trig(v);
            if (mode == 0) {
                dg_frame_lists fl;
                if (build_frame_lists(sc, W, H, v, arena, fl, err) || bin_frame(sc, fk, fl, bf, err)) return -1.0;
            } else if (build_frame_parts(sc, W, H, v, arena, err)) return -1.0;
        }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / ((double)iters * n);
}

// Workload statistics of dg_raster_tiles for one frame (tuning aid, tools/tile_stats.py): what the kernel's loops will meet.
// out[0] chunks (column x 64-row tile)   [1] opaque span hits (owner-loop trips)   [2] overlay span hits (overlay-loop trips)
// [3] overlay evaluations (some row covered and not owned by a later opaque span)   [4] chunks whose owners are of one kind only
// [5] chunks with flat AND wall owners   [6] chunks with exactly one owner span for all live rows   [7] staged spans summed over tiles
// [8] staged spans that touch their tile's rows   [9] pixels owned by flats   [10] by walls   [11] by sky   [12] by nothing
// [13] tiles   [14] tiles with an overlay hit   [15] columns with more than 8 spans (summed over tiles)
// [16] flat chunks (>= 1 flat-owned row)   [17] flat chunks whose flat rows all carry the (gwz, light) of the same rows 8 columns to the left
int emul_tile_stats(void *scene, int W, int H, const dg_view *view_in, uint64_t *out) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    static thread_local BinnedFrame bf;
    dg_frame_lists fl;
    int rc = build_frame_lists(sc, W, H, view, arena, fl, g_err, nullptr);
    if (rc) return rc;
    FrameConsts fk = make_consts(W, H);
    rc = bin_frame(sc, fk, fl, bf, g_err);
    if (rc) return rc;
    for (int i = 0; i < 32; i++) out[i] = 0;
    const int TH = 64, TW = 64;
    std::vector<int32_t> owner((size_t)H), prev_key((size_t)W * (size_t)H * 2, 0);
    std::vector<uint32_t> keyg((size_t)W * (size_t)H, 0), keyl((size_t)W * (size_t)H, 0);
    std::vector<uint8_t> isflat((size_t)W * (size_t)H, 0);
    for (int x = 0; x < W; x++) {
        const uint32_t c0 = bf.col_off[(size_t)x], c1 = bf.col_off[(size_t)x + 1];
        for (int y = 0; y < H; y++) owner[(size_t)y] = -1;
        for (uint32_t i = c0; i < c1; i++) {
            const DevSpan &sp = bf.spans[i];
            const bool imm = sp.kind == SPAN_WALL ? bf.walls[sp.rec].has_holes != 0 : (sp.kind == SPAN_SKY && sc.bitmaps[(size_t)sc.sky_bitmap].has_holes);
            if (!imm) for (int y = sp.ctop; y <= sp.cbot; y++) owner[(size_t)y] = (int32_t)i;
        }
        for (int y = 0; y < H; y++) {
            const int32_t o = owner[(size_t)y];
            if (o < 0) { out[12]++; continue; }
            const DevSpan &sp = bf.spans[(size_t)o];
            out[sp.kind == SPAN_FLAT ? 9 : sp.kind == SPAN_WALL ? 10 : 11]++;
            if (sp.kind == SPAN_FLAT) {
                isflat[(size_t)y * W + x] = 1;
                const DevPlaneRec &pr = bf.planes[sp.rec];
                std::memcpy(&keyg[(size_t)y * W + x], &pr.gwz, 4);
                std::memcpy(&keyl[(size_t)y * W + x], &pr.lightf, 4);
            }
        }
        for (int y0 = 0; y0 < H; y0 += TH) {
            const int y1 = std::min(H, y0 + TH) - 1;
            out[0]++;
            bool kinds[4] = {false, false, false, false};
            int32_t first_owner = owner[(size_t)y0];
            bool single = true;
            for (int y = y0; y <= y1; y++) {
                const int32_t o = owner[(size_t)y];
                single &= o == first_owner;
                kinds[o < 0 ? 3 : bf.spans[(size_t)o].kind] = true;
            }
            const int nk = (int)kinds[0] + (int)kinds[1] + (int)kinds[2];
            if (nk <= 1) out[4]++;
            if (kinds[SPAN_FLAT] && kinds[SPAN_WALL]) out[5]++;
            if (single && first_owner >= 0) out[6]++;
            if (kinds[SPAN_FLAT]) {
                out[16]++;
                bool same = x >= 8;
                for (int y = y0; same && y <= y1; y++)
                    if (isflat[(size_t)y * W + x]) same = isflat[(size_t)y * W + x - 8] && keyg[(size_t)y * W + x] == keyg[(size_t)y * W + x - 8] && keyl[(size_t)y * W + x] == keyl[(size_t)y * W + x - 8];
                if (same) out[17]++;
            }
            if (c1 - c0 > 8) out[15]++;
            {   // [24] chunks with exactly 2 distinct opaque owners over the live rows, no uncovered row, no overlay hit, both plain;
                // [25] the same with a visible overlay; [26] >= 3 owners; [27] single owner but an overlay on top; [28] some row uncovered
                int owners[64]; int no = 0; bool uncovered = false;
                for (int y = y0; y <= y1; y++) {
                    const int32_t o = owner[(size_t)y];
                    if (o < 0) { uncovered = true; continue; }
                    bool seen = false;
                    for (int q = 0; q < no; q++) seen |= owners[q] == o;
                    if (!seen && no < 64) owners[no++] = o;
                }
                bool ov = false;
                for (uint32_t i = c0; i < c1; i++) {
                    const DevSpan &sp = bf.spans[i];
                    if (sp.cbot < y0 || sp.ctop > y1) continue;
                    const bool imm = sp.kind == SPAN_WALL ? bf.walls[sp.rec].has_holes != 0 : (sp.kind == SPAN_SKY && sc.bitmaps[(size_t)sc.sky_bitmap].has_holes);
                    if (!imm) continue;
                    for (int y = std::max(y0, (int)sp.ctop); y <= std::min(y1, (int)sp.cbot); y++) ov |= owner[(size_t)y] < (int32_t)i;
                }
                if (uncovered) out[28]++;
                else if (no == 2 && !ov) out[24]++;
                else if (no == 2 && ov) out[25]++;
                else if (no >= 3) out[26]++;
                else if (no == 1 && ov) out[27]++;
            }
            for (uint32_t i = c0; i < c1; i++) {
                const DevSpan &sp = bf.spans[i];
                if (sp.cbot < y0 || sp.ctop > y1) continue;
                const bool imm = sp.kind == SPAN_WALL ? bf.walls[sp.rec].has_holes != 0 : (sp.kind == SPAN_SKY && sc.bitmaps[(size_t)sc.sky_bitmap].has_holes);
                if (!imm) { out[1]++; continue; }
                out[2]++;
                bool any = false;
                for (int y = std::max(y0, (int)sp.ctop); y <= std::min(y1, (int)sp.cbot); y++) any |= owner[(size_t)y] < (int32_t)i;
                if (any) out[3]++;
            }
        }
    }
    {   // [18 + class] chunks per descriptor class; [23] tiles without a CH_GENERIC chunk
        std::vector<DevRSpan> rs(bf.spans.size());
        for (size_t i = 0; i < bf.spans.size(); i++) rs[i].w[0] = pack_w0(bf.spans[i].ctop, bf.spans[i].cbot, bf.spans[i].kind,
            bf.spans[i].kind == SPAN_WALL ? bf.walls[bf.spans[i].rec].has_holes != 0 : (bf.spans[i].kind == SPAN_SKY && sc.bitmaps[(size_t)sc.sky_bitmap].has_holes),
            bf.spans[i].kind == SPAN_WALL ? (bf.walls[bf.spans[i].rec].h & (bf.walls[bf.spans[i].rec].h - 1)) == 0 : bf.spans[i].kind == SPAN_FLAT);
        for (int x0 = 0; x0 < W; x0 += TW)
            for (int y0 = 0; y0 < H; y0 += TH) {
                bool any_gen = false;
                for (int x = x0; x < std::min(W, x0 + TW); x++) {
                    const uint32_t d = classify_chunk([&](uint32_t i) { return rs[i].w[0]; }, bf.col_off[(size_t)x], bf.col_off[(size_t)x + 1], y0, std::min(H, y0 + TH) - 1, H, H % 2 == 0 ? H / 2 : -1, 8);
                    out[18 + (d >> CH_CLASS_SHIFT)]++;
                    any_gen |= (d >> CH_CLASS_SHIFT) == CH_GENERIC;
                }
                if (!any_gen) out[23]++;
            }
    }
    for (int x0 = 0; x0 < W; x0 += TW)
        for (int y0 = 0; y0 < H; y0 += TH) {
            const int x1 = std::min(W, x0 + TW), y1 = std::min(H, y0 + TH) - 1;
            out[13]++;
            bool ov = false;
            for (uint32_t i = bf.col_off[(size_t)x0]; i < bf.col_off[(size_t)x1]; i++) {
                const DevSpan &sp = bf.spans[i];
                out[7]++;
                if (sp.cbot < y0 || sp.ctop > y1) continue;
                out[8]++;
                ov |= sp.kind == SPAN_WALL && bf.walls[sp.rec].has_holes != 0;
            }
            if (ov) out[14]++;
        }
    return 0;
}

// Debug aid: the resolved spans of one screen column (8 words each) and their count.
int emul_column_spans(void *scene, int W, int H, const dg_view *view_in, int x, uint32_t *out, int cap) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    static thread_local BinnedFrame bf;
    dg_frame_lists fl;
    if (build_frame_lists(sc, W, H, view, arena, fl, g_err, nullptr)) return -1;
    FrameConsts fk = make_consts(W, H);
    if (bin_frame(sc, fk, fl, bf, g_err)) return -1;
    DevConsts k{fk.ARC, fk.GCFX, fk.CFX, fk.CFY, W, H};
    DevScene ds{};
    const BitmapInfo &sky = sc.bitmaps[(size_t)sc.sky_bitmap];
    ds.sky_texel_off = sky.texel_off; ds.sky_w = sky.w; ds.sky_h = sky.h; ds.sky_has_holes = sky.has_holes;
    int n = 0;
    for (uint32_t i = bf.col_off[(size_t)x]; i < bf.col_off[(size_t)x + 1] && n < cap; i++, n++) {
        const DevSpan &sp = bf.spans[i];
        DevRSpan r = sp.kind == SPAN_WALL ? resolve_wall_span(sp, bf.walls[sp.rec]) : sp.kind == SPAN_FLAT ? resolve_flat_span(sp, bf.planes[sp.rec], k, 0u) : resolve_sky_span(sp, ds, k, bf.hdr);
        std::memcpy(out + 8 * n, r.w, 32);
    }
    return n;
}

// What the host ships to the device column walk for one frame (FePart records, fe_dev.h), for tests/np_front_end.py: 12 dwords per
// part = sx, ex, bits of bsy, bsx, bdelta, tsy, tsx, tdelta, flags, seq, has-columns hint (unused), reserved.  Returns the part count.
int emul_frame_parts(void *scene, int W, int H, const dg_view *view_in, uint32_t *out, int cap) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    if (build_frame_parts(sc, W, H, view, arena, g_err)) return -1;
    int n = 0;
    for (const FePart &p : arena.parts) {
        if (n >= cap) return -2;
        uint32_t *o = out + 12 * n++;
        o[0] = (uint32_t)p.sx; o[1] = (uint32_t)p.ex;
        o[2] = f32_bits(p.bsy); o[3] = f32_bits(p.bsx); o[4] = f32_bits(p.bdelta);
        o[5] = f32_bits(p.tsy); o[6] = f32_bits(p.tsx); o[7] = f32_bits(p.tdelta);
        o[8] = p.flags; o[9] = p.seq; o[10] = 0; o[11] = 0;
    }
    return n;
}

// The host list builder's output for one frame, flattened for Python (the layout of include/doomgpu.h dg_frame_lists):
//   renders: 4 ints each = start_x, end_x, first_column, n_columns;  columns: 5 int16 each;  visplanes: 4 ints = left, right, first_entry, flat;
//   plane_tb: int16 pairs;  order: 2 uints.  counts[5] = how many of each.  Returns 0, or -2 when a buffer is too small.
int emul_frame_lists(void *scene, int W, int H, const dg_view *view_in, int32_t *renders, int16_t *columns, int32_t *visplanes, int16_t *plane_tb, uint32_t *order,
                     uint32_t *counts, uint32_t cap) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    dg_frame_lists fl;
    if (build_frame_lists(sc, W, H, view, arena, fl, g_err)) return -1;
    if (fl.n_renders > cap || fl.n_columns > cap || fl.n_visplanes > cap || fl.n_plane_tb > 2 * cap || fl.n_order > cap) return -2;
    for (uint32_t i = 0; i < fl.n_renders; i++) { renders[4 * i] = fl.renders[i].start_x; renders[4 * i + 1] = fl.renders[i].end_x; renders[4 * i + 2] = (int32_t)fl.renders[i].first_column; renders[4 * i + 3] = (int32_t)fl.renders[i].n_columns; }
    for (uint32_t i = 0; i < fl.n_columns; i++) { const dg_bitmap_column &c = fl.columns[i]; columns[5 * i] = c.x; columns[5 * i + 1] = c.clipped_top_y; columns[5 * i + 2] = c.clipped_bottom_y; columns[5 * i + 3] = c.bottom_y; columns[5 * i + 4] = c.top_y; }
    for (uint32_t i = 0; i < fl.n_visplanes; i++) { visplanes[4 * i] = fl.visplanes[i].left; visplanes[4 * i + 1] = fl.visplanes[i].right; visplanes[4 * i + 2] = (int32_t)fl.visplanes[i].first_entry; visplanes[4 * i + 3] = fl.visplanes[i].flat; }
    for (uint32_t i = 0; i < fl.n_plane_tb; i++) plane_tb[i] = fl.plane_tb[i];
    for (uint32_t i = 0; i < fl.n_order; i++) { order[2 * i] = fl.order[i].kind; order[2 * i + 1] = fl.order[i].index; }
    counts[0] = fl.n_renders; counts[1] = fl.n_columns; counts[2] = fl.n_visplanes; counts[3] = fl.n_plane_tb; counts[4] = fl.n_order;
    return 0;
}
}

// ---- the device seg walk (fs_frame.h) on the CPU ------------------------------------------------------------------------------------
// Runs the bodies of dg_fs_segs / dg_fs_frame for one frame — the "lanes" of every phase one after another, a barrier
// between phases — and compares what they produce with the host walker's parts mode (build_frame_parts), record by record:
// every FePart byte for byte, every FeSprite (but its behind_off: the row stride differs), the behind bits, the sky slot table, both
// column-bin tables.  Returns 0 and stats = [parts, sprites, sky slots, flags, capacities exceeded, candidates], 1 when the host walker
// itself refuses the frame (then the device walk must have flagged it), 2 when the device walk gave the frame up because it exceeds a
// capacity (stats[4]: 1 parts, 2 candidates, 4 sprites, 8 sky parts, 16 part bins, 32 sprite bins), 3 when it gave it up for no such
// reason, < 0 with emul_last_error on a mismatch.
static bool g_no_cl_rows = false;          // emul_fs_no_cl_rows(1): as if the ctx could not allocate the global candidate rows (FS_CL_CAP is the limit then)
extern "C" void emul_fs_no_cl_rows(int on) { g_no_cl_rows = on != 0; }
extern "C" int emul_fs_frame(void *scene, int W, int H, const dg_view *view_in, uint64_t *stats) {
    const Scene &sc = *(const Scene *)scene;
    if (!sc.fs_ok || W > FS_MAX_W) { g_err = "scene / frame size not eligible for the device seg walk"; return -100; }
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    const int host_rc = build_frame_parts(sc, W, H, view, arena, g_err);
    FrameConsts fk = make_consts(W, H);
    const uint32_t nb = (uint32_t)(W + FE_BIN_W - 1) / FE_BIN_W;
    std::vector<int16_t> lights(sc.sectors.size());
    for (size_t i = 0; i < sc.sectors.size(); i++) lights[i] = sc.sectors[i].light;
    std::vector<int32_t> mstate(sc.mobjs.size());
    for (size_t i = 0; i < sc.mobjs.size(); i++) mstate[i] = sc.mobjs[i].sprite_frame < 0 ? -1 : sc.mobjs[i].sprite_frame * 2 + (sc.mobjs[i].full_bright ? 1 : 0);
    std::vector<uint32_t> flags(1, 0);
    std::vector<uint2> lite(sc.segs.size() * FS_CALLS + 1, uint2{0xdeadbeefu, 0xdeadbeefu});      // (stale entries: only what the occupancy row marks may be read)
    std::vector<FeFrame> ffr(1);
    std::vector<FePart> parts(FS_PART_CAP);
    std::vector<FeSprite> sprites(FS_SPRITE_CAP);
    std::vector<uint32_t> behind((size_t)FS_SPRITE_CAP * FS_BEHIND_WORDS, 0), sky_parts(FS_SKY_CAP), bin_off(nb + 1), sbin_off(nb + 1);
    std::vector<uint16_t> bin_parts(FS_BIN_CAP), sbin_sprites(FS_SBIN_CAP);
    FsParams P{};
    P.k = DevConsts{fk.ARC, fk.GCFX, fk.CFX, fk.CFY, W, H};
    P.segs = sc.fs_segs.data(); P.seg_leaf = sc.fs_seg_leaf.data(); P.leaf_first = sc.fs_leaf_first.data();
    P.sectors = sc.fs_sectors.data(); P.anims = sc.fs_anims.data(); P.bitmaps = sc.fs_bitmaps.data(); P.flat_sky = sc.flat_sky.data();
    P.mobjs = sc.fs_mobjs.data(); P.sframes = sc.sprite_frames_fs();
    P.anc_off = sc.fs_anc_off.data(); P.anc = sc.fs_anc.data();
    P.n_segs = (uint32_t)sc.segs.size(); P.n_leaves = (uint32_t)sc.subsectors.size(); P.n_mobjs = (uint32_t)sc.mobjs.size();
    P.sprite_stride = std::min<uint32_t>(FS_SPRITE_CAP, std::max<uint32_t>(32u, (P.n_mobjs + 31u) / 32u * 32u));        // (context.cpp: upload_fs_scene)
    P.sbin_stride = std::min<uint32_t>(FS_SBIN_CAP, P.sprite_stride * nb);
    P.sector_light = lights.data(); P.mobj_state = mstate.data();
    P.views = &view; P.n_frames = 1;
    std::vector<uint32_t> occ(fs_occ_words(P.n_segs), 0);
    const uint32_t cl_row_cap = (P.n_segs * FS_CALLS + 31u) / 32u * 32u;                                  // (context.cpp: upload_fs_scene)
    std::vector<uint32_t> cl_rows(cl_row_cap, 0xdeadbeefu), keep_rows(cl_row_cap / 32, 0xdeadbeefu);
    P.cl_rows = cl_rows.data(); P.keep_rows = keep_rows.data(); P.cl_row_cap = g_no_cl_rows ? 0u : cl_row_cap;
    P.lite = lite.data(); P.occ = occ.data(); P.flags = flags.data();
    P.fframes = ffr.data(); P.parts = parts.data(); P.sprites = sprites.data(); P.behind = behind.data(); P.sky_parts = sky_parts.data();
    P.bin_off = bin_off.data(); P.bin_parts = bin_parts.data(); P.sbin_off = sbin_off.data(); P.sbin_sprites = sbin_sprites.data();

    for (uint32_t s = 0; s < P.n_segs; s++) if (P.seg_leaf[s] != 0xffffu) fs_seg_lane(P, 0, s);             // dg_fs_segs
    uint32_t n_cand = 0;                                                                                   // process_sidedef calls that reach their column loop
    for (uint32_t w : occ) n_cand += (uint32_t)__builtin_popcount(w);
    static thread_local FsShared S;                                                                        // dg_fs_frame
    static thread_local FsSpriteTmp T[FS_LANES];
#define LANES(body) for (int lane = 0; lane < FS_LANES; lane++) { body; }
    fs_ph_init(S);
    LANES(fs_ph_cand_count(P, S, 0, lane))
    LANES(fs_ph_block_sums(S, lane))
    LANES(fs_ph_cand_stage(P, S, 0, lane))
    LANES(fs_ph_first_clear(P, S, lane))
    LANES(fs_ph_solids(P, S, 0, lane))
    LANES(fs_ph_keep(P, S, 0, lane))
    LANES(fs_ph_kept_count(P, S, 0, lane))
    LANES(fs_ph_block_sums(S, lane))
    LANES(fs_ph_kept_place(P, S, 0, lane))
    LANES(fs_ph_emit(P, S, 0, lane))
    for (uint32_t base = 0; base < P.n_mobjs; base += FS_LANES) {
        const uint32_t n_before = S.n_sprites;
        LANES(fs_ph_mobj(P, S, 0, base, lane, T[lane]))
        LANES(fs_ph_block_sums(S, lane))
        LANES(fs_ph_mobj_emit(P, S, 0, lane, T[lane], n_before))
    }
    LANES(fs_ph_behind(P, S, 0, lane))
    LANES(fs_ph_sprite_order(S, lane))
    LANES(fs_ph_masked_when(S, lane))
    LANES(fs_ph_seq(P, S, 0, lane))
    LANES(fs_ph_bin_clear(P, S, lane))
    LANES(fs_ph_bin_mark(P, S, lane))
    LANES(fs_ph_bin_count(P, S, lane))
    fs_ph_bin_prefix(P, S, 0);
    LANES(fs_ph_bin_fill(P, S, 0, lane))
    LANES(fs_ph_clean(P, 0, lane))
    fs_ph_header(P, S, 0);
#undef LANES
    for (uint32_t w : occ) if (w) { g_err = "dg_fs_frame left an occupancy row dirty"; return -3; }
    const FeFrame &ff = ffr[0];
    if (stats) { stats[0] = ff.n_parts; stats[1] = ff.n_sprites; stats[2] = ff.n_sky_slots; stats[3] = flags[0]; stats[4] = 0; stats[5] = n_cand; }
    if (host_rc) {                                    // the host walker refuses the frame: the device walk must have given it up too
        if (!(flags[0] & FE_OVF_SEGS)) { g_err = "host walker fails (" + g_err + ") but the device seg walk did not flag the frame"; return -1; }
        return 1;
    }
    if (flags[0] & FE_OVF_SEGS) {                     // given up on the device although the host walker completes the frame: only a NAMED capacity may do that
        const uint32_t why = (arena.parts.size() > FS_PART_CAP ? 1u : 0u) | (n_cand > std::max(FS_CL_CAP, P.cl_row_cap) ? 2u : 0u) | (arena.sprites.size() > P.sprite_stride ? 4u : 0u) |
                             (arena.n_sky_slots > FS_SKY_CAP ? 8u : 0u) | (arena.bin_off[nb] > FS_BIN_CAP ? 16u : 0u) | (arena.sbin_off[nb] > P.sbin_stride ? 32u : 0u);
        if (stats) stats[4] = why;
        if (!why) { g_err = "the device seg walk gave up a frame that exceeds none of its capacities and that the host walker completes"; return 3; }
        return 2;
    }
    auto bad = [&](const std::string &m) { g_err = m; return -2; };
    if (ff.n_parts != arena.parts.size()) return bad("part count " + std::to_string(ff.n_parts) + " != host " + std::to_string(arena.parts.size()));
    for (uint32_t i = 0; i < ff.n_parts; i++)
        if (std::memcmp(&parts[i], &arena.parts[i], sizeof(FePart)) != 0) return bad("part " + std::to_string(i) + " differs from the host walker's");
    if (ff.n_sprites != arena.sprites.size()) return bad("sprite count " + std::to_string(ff.n_sprites) + " != host " + std::to_string(arena.sprites.size()));
    for (uint32_t i = 0; i < ff.n_sprites; i++) {
        FeSprite a = sprites[i], b = arena.sprites[i];
        if (a.behind_off != i * FS_BEHIND_WORDS) return bad("sprite behind_off");
        a.behind_off = b.behind_off = 0;
        if (std::memcmp(&a, &b, sizeof a) != 0) return bad("sprite " + std::to_string(i) + " differs from the host walker's");
        for (uint32_t p = 0; p < ff.n_parts; p++) {
            const bool dv = (behind[(size_t)i * FS_BEHIND_WORDS + (p >> 5)] >> (p & 31)) & 1u;
            const bool hv = (arena.behind[arena.sprites[i].behind_off + (p >> 5)] >> (p & 31)) & 1u;
            if (dv != hv) return bad("behind bit of sprite " + std::to_string(i) + ", part " + std::to_string(p));
        }
    }
    if (ff.n_sky_slots != arena.n_sky_slots) return bad("sky slot count");
    for (uint32_t i = 0; i < ff.n_sky_slots; i++)
        if (sky_parts[i] != arena.sky_parts[i]) return bad("sky slot " + std::to_string(i));
    for (uint32_t b = 0; b <= nb; b++)
        if (bin_off[b] != arena.bin_off[b] || sbin_off[b] != arena.sbin_off[b]) return bad("bin offsets of bin " + std::to_string(b));
    for (uint32_t i = 0; i < bin_off[nb]; i++) if (bin_parts[i] != arena.bin_parts[i]) return bad("part bin entry " + std::to_string(i));
    for (uint32_t i = 0; i < sbin_off[nb]; i++) if (sbin_sprites[i] != arena.sbin_sprites[i]) return bad("sprite bin entry " + std::to_string(i));
    return 0;
}
