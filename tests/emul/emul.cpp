// tests/emul/emul.cpp — TEST HARNESS ONLY (never part of libdoomgpu, never loaded by the package).
//
// Compiles the product's host list generation (scene.cpp, frontend.cpp, binner.cpp) together with the
// kernel *bodies* (raster_core.h) for the CPU, and replays the column-major span lists the way the raster
// kernel does (per column, spans in order, later span overwrites).  It lets the CPU-only test tier check the
// host logic and the list format against the oracle in a container without a GPU.  The GPU tier repeats the
// same comparison through the real HIP kernels and the C-ABI.
#include <cstring>
#include <string>

#include "../../doom-rust-renderer_amd/csrc/binner.hpp"
#include "../../doom-rust-renderer_amd/csrc/frontend.hpp"
#include "../../doom-rust-renderer_amd/csrc/raster_core.h"
#include "../../doom-rust-renderer_amd/csrc/scene.hpp"

using namespace dg;

static std::string g_err;

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

// stats[0..3] = spans, walls, planes, covered pixels
int emul_render(void *scene, int W, int H, const dg_view *view_in, uint8_t *rgb, uint64_t *stats) {
    const Scene &sc = *(const Scene *)scene;
    dg_view view = *view_in;
    fill_view_trig(view);
    static thread_local FrameArena arena;
    static thread_local BinnedFrame bf;
    dg_frame_lists fl;
    int rc = build_frame_lists(sc, W, H, view, arena, fl, g_err);
    if (rc) return rc;
    FrameConsts fk = make_consts(W, H);
    rc = bin_frame(sc, fk, fl, bf, g_err);
    if (rc) return rc;

    std::vector<uint32_t> pal(256);
    for (int i = 0; i < 256; i++) pal[i] = sc.palette[3 * i] | (sc.palette[3 * i + 1] << 8) | (sc.palette[3 * i + 2] << 16);
    DevScene ds;
    ds.palette = pal.data(); ds.texel_idx = sc.texel_idx.data(); ds.texel_opq = sc.texel_opq.data(); ds.flats = sc.flat_pool.data();
    const BitmapInfo &sky = sc.bitmaps[(size_t)sc.sky_bitmap];
    ds.sky_texel_off = sky.texel_off; ds.sky_w = sky.w; ds.sky_h = sky.h; ds.sky_has_holes = sky.has_holes;
    DevConsts k{fk.ARC, fk.GCFX, fk.CFX, fk.CFY, W, H};

    std::memset(rgb, 0, (size_t)3 * W * H);
    for (int x = 0; x < W; x++) {
        for (uint32_t i = bf.col_off[x]; i < bf.col_off[x + 1]; i++) {
            const DevSpan &s = bf.spans[i];
            // what dg_setup_spans does (one lane per span) ...
            DevRSpan r = s.kind == SPAN_WALL ? resolve_wall_span(s, bf.walls[s.rec])
                       : s.kind == SPAN_FLAT ? resolve_flat_span(s, bf.planes[s.rec], k)
                                             : resolve_sky_span(s, ds, k, bf.hdr);
            const uint32_t *w = r.w;
            // ... and what dg_raster_tiles does for every row of the span (lane = row)
            for (int y = w0_ctop(w[0]); y <= w0_cbot(w[0]); y++) {
                uint32_t c = 0;
                bool wr = false;
                const uint32_t kind = w0_kind(w[0]);
                if (kind == SPAN_WALL) {
                    uint32_t o = wall_texel_offset(w[1], w[2], w[4], w[5], w[6], w[7], y);
                    if (!(w[6] & 0x100u) || ds.texel_opq[o]) { c = shade(pal[ds.texel_idx[o]], bits_f32(w[3])); wr = true; }
                } else if (kind == SPAN_FLAT) {
                    float factor;
                    const float vy = k.CFY - (float)y;
                    uint32_t o = flat_texel_offset(bf.hdr, w[1], w[2], w[4], w[5], w[6], vy, prepare_rcp(vy), factor);
                    c = shade(pal[ds.flats[o]], factor); wr = true;
                } else {
                    uint32_t o = sky_texel_offset(w[2], sky_row(ds, k, y));
                    if (o != 0xffffffffu && ds.texel_opq[o]) { c = pal[ds.texel_idx[o]]; wr = true; }
                }
                if (wr) {
                    uint8_t *p = rgb + 3 * ((size_t)y * W + x);
                    p[0] = c & 255; p[1] = (c >> 8) & 255; p[2] = (c >> 16) & 255;
                }
            }
        }
    }
    if (stats) { stats[0] = bf.spans.size(); stats[1] = bf.walls.size(); stats[2] = bf.planes.size(); stats[3] = bf.covered_pixels; }
    return 0;
}
}
