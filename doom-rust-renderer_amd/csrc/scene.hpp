// scene.hpp — immutable world data for the rasteriser, flattened for cache-friendly host walks and for HBM upload.
//
// Host-side equivalent of what the reference keeps in Rc graphs: Map (src/map/mod.rs:34-44), Palette
// (src/graphics/palette.rs), Textures/Pictures (src/graphics/textures.rs, pictures.rs), Flats (flats.rs),
// Sprites (sprites.rs) and the spawn-state view of MapObjects (src/map_objects.rs:25-50).  Everything the
// reference looks up by String per seg per frame (textures.rs:155-158, flats.rs:92-111) is resolved to an
// integer id once, here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "fs_frame.h"

namespace dg {

struct BitmapInfo {          // reference Bitmap (src/graphics/bitmap.rs:11-15): [y][x] Option<u8>
    int32_t w = 0, h = 0;
    uint32_t texel_off = 0;  // into Scene::texel_idx / texel_opq, column-major: off + x*h + y
    uint32_t has_holes = 0;  // any None texel
    int16_t top_offset = 0;  // pictures only (pictures.rs:26)
    int16_t left_offset = 0;
};

struct SectorRec {
    int16_t floor_h, ceil_h, light;
    int16_t pad;
    int32_t floor_flat, ceil_flat;  // flat id when not animated (>= 0), FLAT_MISSING if the lump does not exist
    int32_t floor_anim, ceil_anim;  // index into Scene::anim (or -1)
    uint8_t floor_sky, ceil_sky;    // name contains "SKY"
    uint8_t ceil_tex_sky;           // sector.ceiling_texture.contains("SKY") (segs.rs:463-469) — same string as ceil_sky
    uint8_t pad2;
};
struct SidedefRec {
    float xoff, yoff;
    int32_t upper, lower, middle;   // bitmap id, TEX_NONE for "-", TEX_UNKNOWN if Textures::get would panic
    int32_t sector;
};
struct LinedefRec { int32_t v1, v2; int16_t flags; int16_t pad; int32_t front, back; };
struct SegRec { int32_t v1, v2, linedef; int16_t offset; uint8_t direction; uint8_t pad; };
struct SubSectorRec { int32_t first, count; };
struct NodeRec {
    float x, y, dx, dy;
    int16_t rchild, lchild;
    float bb[2][4];   // [0] = right child, [1] = left child: min x, min y, max x, max y over the seg vertices of the subtree
                      // (computed at load from the segs themselves, not the NODES lump's boxes, which the reference ignores)
};
struct SpriteFrameRec { int32_t rotate; int32_t bitmap[8]; };     // sprites.rs:20-23 (= FsSpriteFrame, fs_core.h)
static_assert(sizeof(SpriteFrameRec) == sizeof(FsSpriteFrame), "SpriteFrameRec / FsSpriteFrame");
struct MapObjectRec {                                            // map_objects.rs:11-17, renderer-visible part
    float x, y, angle;
    int32_t sprite_frame;   // index into Scene::sprite_frames, -1 = state S_NULL (skipped: renderer/map_objects.rs:37)
    int32_t full_bright;
    int32_t sector;         // get_sector_from_vertex(position) — position is immutable in the reference
};
struct AnimList { int32_t n; int32_t flat[4]; };

enum : int32_t { TEX_NONE = -1, TEX_UNKNOWN = -2, FLAT_MISSING = -2 };

struct Scene {
    std::vector<uint8_t> wad;
    std::string map_name;
    // map
    std::vector<float> vx, vy;
    std::vector<SectorRec> sectors;
    std::vector<SidedefRec> sidedefs;
    std::vector<LinedefRec> linedefs;
    std::vector<SegRec> segs;
    std::vector<SubSectorRec> subsectors;
    std::vector<NodeRec> nodes;
    std::vector<MapObjectRec> mobjs;
    float start_x = 0, start_y = 0, start_angle = 0;
    bool has_start = false;
    bool may_panic = false;                         // some sidedef texture / sector flat lookup would panic in the reference when reached
    // graphics
    uint8_t palette[768];
    std::vector<BitmapInfo> bitmaps;
    std::vector<std::string> bitmap_names;          // "T:<texture>" / "P:<picture>[:M]"
    std::vector<uint8_t> texel_idx, texel_opq;      // column-major planes
    std::vector<std::string> flat_names;            // requested names, verbatim
    std::vector<uint8_t> flat_sky;                  // per flat: name contains "SKY"
    std::vector<uint8_t> flat_pool;                 // 4096 B each, [y][x]
    std::vector<AnimList> anim;
    std::vector<SpriteFrameRec> sprite_frames;
    std::vector<std::string> sprite_frame_keys;     // "SPRT<frame>"
    int32_t sky_bitmap = TEX_UNKNOWN;
    uint64_t revision = 0;                          // bumped by the mutable-state setters
    // The per-seg / per-sprite inputs of fs_core.h, flattened (rebuild_fs_tables: at load and whenever bitmaps or sprite frames are added):
    // what the host walker reads per seg and what dg_upload_scene copies to the GPU for DG_FE_DEVICE_SEGS.
    std::vector<FsSeg> fs_segs;                     // one per seg
    std::vector<uint16_t> fs_seg_leaf;              // subsector of every seg
    std::vector<FsSector> fs_sectors;
    std::vector<FsAnim> fs_anims;
    std::vector<FsBitmap> fs_bitmaps;
    std::vector<FsMobj> fs_mobjs;
    // BSP tables of the device seg walk (fs_frame.h): per node its partition and the seg counts of its subtrees; per leaf its ancestors,
    // root first (node | lies-in-the-LEFT-subtree << 31).  fs_ok: the map is a proper tree whose leaves partition the segs they
    // reference — otherwise (and for maps in which a texture / flat lookup would panic) only the host walker is used.
    std::vector<FsAnc> fs_anc;
    std::vector<uint32_t> fs_anc_off, fs_leaf_first;
    bool fs_ok = false;
    const FsSpriteFrame *sprite_frames_fs() const { return reinterpret_cast<const FsSpriteFrame *>(sprite_frames.data()); }
    void rebuild_fs_tables();

    // lookups used by the C-ABI
    int texture_id(const std::string &name) const;                               // Textures::get
    int flat_id(const std::string &name, float timestamp) const;                 // Flats::get_animated; sky => -(id+1)... see .cpp
    int sprite_bitmap_id(const std::string &sprite, uint8_t frame, uint8_t rotation) const;
    int sector_from_vertex(float x, float y) const;                              // renderer/bsp.rs:9-44
    int find_or_add_sprite_frame(const std::string &sprite, uint8_t frame, std::string &err);
};

// Returns nullptr and fills err on any condition where the reference's loaders panic.
Scene *load_scene_from_wad(const uint8_t *wad, size_t len, const char *map_name, std::string &err);

}  // namespace dg
