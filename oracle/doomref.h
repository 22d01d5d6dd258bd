/* doomref.h — ORACLE (test infrastructure only; never linked into the product library).
 *
 * Plain-C restatement of freewilll/doom-rust-renderer's `src/renderer` hot path plus the
 * loaders it needs.  Every function in doomref.c cites the reference file:line it follows.
 * PARITY UNPINNED: the reference ships no tests/golden vectors and cannot be built here
 * (no rustc/cargo/SDL2/WAD), so this oracle is pinned only by its own KATs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef DOOMREF_H
#define DOOMREF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dr_scene dr_scene;

/* One viewpoint = the reference's `Player` (src/game.rs:40-45) + Renderer::new's timestamp
 * (src/renderer/mod.rs:47).  cos/sin of +angle and -angle are recorded inputs so libm is out of
 * the parity loop (SURVEY.md §0 consequence 2).  If trig_valid == 0 they are filled from
 * cosf/sinf(angle) by dr_render. */
typedef struct dr_view {
    float x, y, angle, floor_height;
    float cos_a, sin_a, cos_na, sin_na;
    float timestamp;
    int32_t trig_valid;
} dr_view;

/* WadFile::new + Map::new + Palette/Flats/Textures/Sprites/MapObjects::new (src/game.rs:142-167).
 * Returns NULL on any condition where the reference would panic; message via dr_last_error(). */
dr_scene *dr_load(const uint8_t *wad, size_t len, const char *map_name);
void dr_free(dr_scene *s);
const char *dr_last_error(void);

/* Player1Start thing -> x, y, angle (src/game.rs:151-156). Returns 0 on success. */
int dr_player_start(const dr_scene *s, float *x, float *y, float *angle);
/* get_sector_from_vertex(...).floor_height (src/renderer/bsp.rs:9-44, src/game.rs:386-388).
 * Returns 0 and writes *h if a sector was found, 1 otherwise (caller keeps the old height). */
int dr_floor_height_at(const dr_scene *s, float x, float y, float *h);

/* Game-state inputs the renderer reads and the reference mutates between frames: sector.light_level
 * (src/lights.rs) and a map object's current State (src/map_objects.rs:63-121).  sprite == NULL means StateId::S_NULL. */
int dr_sector_count(const dr_scene *s);
int dr_set_sector_light(dr_scene *s, int sector, int16_t light_level);
int dr_mobj_count(const dr_scene *s);
int dr_set_mobj_state(dr_scene *s, int mobj, const char *sprite, uint8_t frame, int full_bright);

/* Pixels::new() + Renderer::new(..).render() (src/game.rs:505-519) at a runtime W x H.
 * rgb must hold 3*W*H bytes; it is zeroed first (fresh Vec, src/renderer/pixels.rs:10-14).
 * flags bit0: evaluate cosf/sinf per floor/ceiling pixel exactly where the source does
 * (visplanes.rs:117 -> vertexes.rs:20-25) instead of using the per-frame constants
 * (value-identical when the constants came from the same libm; CPU-baseline variant only).
 * Returns 0, or -1 where the reference would panic. */
int dr_render(dr_scene *s, int W, int H, const dr_view *view, uint8_t *rgb, int flags);

/* List-level replay for KATs: the caller's BitmapRender / BitmapColumn / Visplane records (bitmap_render.rs:19-45,
 * visplanes.rs:17-26) in draw order through the oracle's own pixel functions.  order = n_order pairs (kind, index): kind 0 = replay
 * render `index` (all its columns), kind 1 = draw visplane `index`.  plane_tb = (top, bottom) pairs for [left, right]. */
typedef struct dr_list_column { int32_t x, clipped_top_y, clipped_bottom_y, bottom_y, top_y; } dr_list_column;
typedef struct dr_list_render {
    const char *texture;                 /* Textures::get name */
    int16_t light_level, offset_x, offset_y, reserved;
    float line_start_x, line_start_y, line_end_x, line_end_y, start_offset;
    int32_t start_x, end_x;
    float bottom_height, top_height;
    uint32_t first_column, n_columns;
} dr_list_render;
typedef struct dr_list_visplane { const char *flat; int16_t height, light_level, left, right; uint32_t first_entry; } dr_list_visplane;
int dr_draw_lists(dr_scene *s, int W, int H, const dr_view *view, const dr_list_render *renders, int n_renders, const dr_list_column *columns,
                  const dr_list_visplane *visplanes, int n_visplanes, const int16_t *plane_tb, const uint32_t *order, int n_order, uint8_t *rgb);

/* Scalar entry points for KATs. */
void dr_diminish_color(const uint8_t rgb_in[3], int16_t light_level, int16_t distance, uint8_t rgb_out[3]);
int16_t dr_f32_as_i16(float f);
int32_t dr_f32_as_i32(float f);
uint8_t dr_f32_as_u8(float f);
void dr_constants(int W, int H, float out5[5]); /* ARC, GSW, GCFX, CFX, CFY */

/* Frame statistics of the last dr_render call (for sizing tests/bench; not part of parity). */
typedef struct dr_stats {
    int32_t n_records, n_columns, n_visplanes, n_mobj_records;
    int64_t wall_pixels, flat_pixels, sky_pixels, masked_pixels, mobj_pixels; /* pixel WRITES incl. overdraw */
} dr_stats;
void dr_last_stats(dr_stats *out);

#ifdef __cplusplus
}
#endif
#endif
