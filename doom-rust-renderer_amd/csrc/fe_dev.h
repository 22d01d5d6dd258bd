// fe_dev.h — formats of the device column walk ("device front end"): the per-column half of the reference's
// Segs::process_sidedef (src/renderer/segs.rs:202-345), the visplane column entries of SidedefVisPlanes
// (src/renderer/sidedef_visplanes.rs) and the per-column half of draw_map_objects (src/renderer/map_objects.rs:130-209)
// run on the GPU with one lane per screen column.  The host keeps the per-seg half (BSP order, view transform, frustum
// clip, projection, texture pegging — segs.rs:353-590 up to the column loop) and the per-sprite half (projection, sorting,
// the sprite/masked-wall interleave), which are tiny and independent of the frame width, and ships them as FePart /
// FeSprite records instead of finished span lists.
#pragma once
#include <stdint.h>

#include "lists_dev.h"

namespace dg {

enum : uint32_t {
    FEP_ONLY_OCCL = 1u << 0,     // Flags.only_occlusions
    FEP_LOWER = 1u << 1,         // Flags.is_lower_wall
    FEP_UPPER = 1u << 2,         // Flags.is_upper_wall
    FEP_DRAW_CEILING = 1u << 3,  // Flags.draw_ceiling
    FEP_TWO_SIDED_MID = 1u << 4, // Flags.is_two_sided_middle_wall
    FEP_HAS_BITMAP = 1u << 5,    // texture != "-"
    FEP_FLOOR_SKY = 1u << 6,     // floor flat name contains "SKY"
    FEP_CEIL_SKY = 1u << 7,      // ceiling flat name contains "SKY"
};

// One process_sidedef call that reached its column loop, in BSP order (index = position in Segs.segs).
struct FePart {
    int32_t sx, ex;               // bottom.start.x / bottom.end.x (== top's)
    float bsy, bsx, bdelta;       // bottom.start.y as f32, bottom.start.x as f32, bottom_delta  (segs.rs:158-161,205-209)
    float tsy, tsx, tdelta;       // same for the top edge
    uint32_t flags;
    int32_t sky_slot;             // index into the per-frame event bit arrays when a sky flat is involved, else -1
    uint32_t seq;                 // two-sided middle walls: position in the phase-3/4 draw sequence
    uint32_t pad;
    DevWallRec wall;              // constants of render_vertical_bitmap_line for this record (valid with FEP_HAS_BITMAP)
    DevPlaneRec floor_plane;      // constants of draw_visplane for the floor / ceiling plane of this call
    DevPlaneRec ceil_plane;
};
static_assert(sizeof(FePart) == 48 + 48 + 32, "FePart layout");

// One visible map object (src/renderer/map_objects.rs:168-209), any order; `seq` is its place in the draw sequence.
struct FeSprite {
    int32_t x0, x1;               // columns [x0, x1)
    float bsy, bsx, bdelta;
    float tsy, tsx, tdelta;
    uint32_t seq;
    uint32_t behind_off;          // word offset of this sprite's "wall record is behind me" bit row (is_behind_vertex)
    uint32_t pad[2];
    DevWallRec wall;
};
static_assert(sizeof(FeSprite) == 48 + 48, "FeSprite layout");

struct FeFrame {
    uint32_t part_base, n_parts;
    uint32_t sprite_base, n_sprites;
    uint32_t behind_base;         // uint32 words
    uint32_t behind_words;        // words per sprite row = ceil(n_parts / 32)
    uint32_t n_sky_slots;
    uint32_t sky_base;            // into the batch's sky_parts array
    uint32_t bin_base;            // into the batch's bin_parts array (entries)
    uint32_t sbin_base;           // into the batch's sbin_sprites array
    uint32_t pad[2];
};
static_assert(sizeof(FeFrame) == 48, "FeFrame layout");

// Column bins: the frame is cut into FE_BIN_W-column strips (one wavefront of dg_fe_columns each); for every strip the
// host lists, in BSP order, the parts whose column range touches it, and likewise the sprites.  off[] arrays are
// [frame][n_bins + 1], relative to the frame's bin_base / sbin_base.
constexpr int FE_BIN_W = 64;

// Per-column scratch written by the walk and read back by the same lane when it clips the sprites: what one wall record
// contributes to draw_map_objects' clip arrays at this screen column (map_objects.rs:141-163) if it is not behind the
// sprite.  Solid records clip with their clipped extent, two-sided ones with their unclipped extent (BitmapColumn,
// bitmap_render.rs:19-25); a side that does not clip is stored as the neutral element of max / min.
struct FeColRec {
    uint16_t part;
    uint16_t pad;
    int16_t top_cand;             // top_clip    = max(top_clip, top_cand)
    int16_t bottom_cand;          // bottom_clip = min(bottom_clip, bottom_cand)
};
static_assert(sizeof(FeColRec) == 8, "FeColRec layout");

enum : uint32_t { FE_EV_FADD = 1, FE_EV_CADD = 2, FE_EV_FLUSH = 4 };          // SidedefVisPlanes events of one column of one part
enum : uint32_t { FE_OVF_SPANS = 1, FE_OVF_RECS = 2, FE_OVF_FRAME = 4 };      // per-frame overflow flags (the batch is redone on the host)

constexpr uint32_t FE_DEFAULT_COL_SLOTS = 48;  // spans / wall-record columns a screen column may hold before its frame is flagged as overflowing
constexpr uint32_t FE_MAX_COL_SLOTS = 128;     // dg_fe_scatter stages 64 columns x this many keys in LDS (32 KB)
constexpr uint32_t FE_KEY_WALL = 1u << 30;  // sort key = phase << 30 | major << 2 | minor
constexpr uint32_t FE_KEY_PLANE = 2u << 30;
constexpr uint32_t FE_KEY_LATE = 3u << 30;
constexpr uint32_t FE_MAX_SKY_SLOTS = 512;   // parts per frame that may produce sky visplanes

}  // namespace dg
