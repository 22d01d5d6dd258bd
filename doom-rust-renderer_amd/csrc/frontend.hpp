// frontend.hpp — per-frame list generation on the host.
//
// The reference draws solid walls inline while it walks the BSP and records a replay list for masked walls
// and sprites (src/renderer/segs.rs:185-200,231-260,349).  Here the same walk records *everything* — no
// pixel is touched on the host — and emits the lists in the order the reference would have drawn them:
//   1. solid / upper / lower wall records, BSP front-to-back           (src/renderer/mod.rs:61-104)
//   2. visplanes in push order                                          (src/renderer/mod.rs:106-116)
//   3. sprites far->near, each preceded by the masked walls behind it  (src/renderer/map_objects.rs:216-240)
//   4. remaining masked walls                                           (src/renderer/segs.rs:593-597)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/doomgpu.h"
#include "fe_dev.h"
#include "scene.hpp"

namespace dg {

struct FrameConsts {   // src/renderer/constants.rs:3-17, evaluated in f32 exactly like the const items
    float ARC, GSW, GCFX, CFX, CFY;
    int W, H;
};
FrameConsts make_consts(int W, int H);

// Reusable per-thread storage: one build() call fills it, the dg_frame_lists view points into it.
struct FrameArena {
    std::vector<dg_bitmap_render> renders;
    std::vector<dg_bitmap_column> columns;
    std::vector<dg_visplane> visplanes;
    std::vector<int16_t> plane_tb;
    std::vector<dg_draw_cmd> order;
    // parts mode (device column walk): per-seg / per-sprite records instead of finished columns
    std::vector<FePart> parts;
    std::vector<FeSprite> sprites;
    std::vector<uint32_t> sky_parts;    // per sky slot: index of the part
    std::vector<uint32_t> bin_off, sbin_off;         // column bins (fe_dev.h): n_bins + 1 offsets each
    std::vector<uint16_t> bin_parts, sbin_sprites;
    std::vector<uint32_t> behind;       // n_sprites rows of behind_words bits: wall record p is behind sprite s
    uint32_t behind_words = 0, n_sky_slots = 0;
    // scratch
    struct Rec;
    std::vector<Rec> *recs = nullptr;   // opaque (defined in frontend.cpp)
    std::vector<int16_t> floor_tb, ceil_tb;
    std::vector<uint8_t> hor_ocl;
    std::vector<int16_t> floor_ocl, ceil_ocl, top_clip, bottom_clip;
    std::vector<int32_t> light_ov, mobj_ov;     // per sector / per map object: this view's snapshot value or "no override" (Walker)
    FrameArena();
    ~FrameArena();
    FrameArena(const FrameArena &) = delete;
    FrameArena &operator=(const FrameArena &) = delete;
};

// Fills `arena` and `out` (pointers into arena).  Returns DG_OK or DG_ERR_RENDER with `err` set where the
// reference would panic.  `view` must have its trig fields filled.
// `state` (optional): the view's game-state snapshot (include/doomgpu.h dg_view_state).
int build_frame_lists(const Scene &sc, int W, int H, const dg_view &view, FrameArena &arena, dg_frame_lists &out, std::string &err,
                      const dg_view_state *state = nullptr);

void fill_view_trig(dg_view &v);

// Parts mode: the per-seg and per-sprite half only (BSP order, transform, clip, projection, pegging, sprite sorting and the
// sprite / masked-wall draw sequence); fills arena.parts / sprites / behind.  The per-column half runs on the GPU (frontend.hip).
// Returns DG_OK, DG_ERR_RENDER (the reference would panic before any column is walked) or kPartsUnsupported (a case only the
// host list path can judge: the caller redoes the frame with build_frame_lists).
constexpr int kPartsUnsupported = 1000;
int build_frame_parts(const Scene &sc, int W, int H, const dg_view &view, FrameArena &arena, std::string &err, const dg_view_state *state = nullptr);

// Per-record constants of render_vertical_bitmap_line (bitmap_render.rs:233-251), shared by the binner and the parts builder.
DevWallRec make_wall_rec(const BitmapInfo &bi, float lsx, float lsy, float lex, float ley, float start_offset, int32_t start_x, int32_t end_x,
                         float bottom_height, float top_height, int16_t offset_x, int16_t offset_y, int16_t light_level);

}  // namespace dg
