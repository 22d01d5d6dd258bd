/* doomgpu.h — C-ABI of libdoomgpu: an MI355X (gfx950) column/span rasteriser that is a drop-in for the
 * `src/renderer` hot path of freewilll/doom-rust-renderer (reference paths below are relative to that repo).
 *
 * What it replaces.  In the reference, `Game::render` (src/game.rs:491-534) does, once per frame:
 *     let mut pixels = Pixels::new();                                  // src/renderer/pixels.rs:10-14
 *     Renderer::new(&mut pixels, &map, &map_objects, &mut textures, &mut sprites, sky_texture,
 *                   &mut flats, &palette, &player, timestamp).render(); // src/renderer/mod.rs:37-58,118-136
 *     buffer.copy_from_slice(pixels.pixels.as_ref());                  // RGB24, src/game.rs:521-525
 * This library produces byte-identical `pixels.pixels` (3*W*H bytes, R,G,B, row-major) for a batch of
 * viewpoints at once, with every per-pixel evaluation done by hand-written HIP kernels.
 *
 * Two entry levels:
 *   dg_render_views  — full path: the library walks the BSP / builds the seg, visplane and sprite lists
 *                      itself (host C++, multi-threaded over frames) and rasterises them on the GPU.
 *   dg_draw_lists    — list path: the caller (e.g. the Rust host, keeping its own src/renderer/segs.rs walk)
 *                      hands over the recorded BitmapRender / Visplane lists in draw order; the library only
 *                      rasterises.  Record layouts mirror src/renderer/bitmap_render.rs:19-45 and
 *                      src/renderer/visplanes.rs:17-26.
 *
 * Conventions: every function returns 0 on success or a negative dg_status; nothing unwinds across the
 * boundary (the reference panics instead: e.g. src/renderer/segs.rs:103-111,140-145,431-436).  One dg_ctx per
 * GPU; a ctx is not thread-safe, different ctxs are independent.  There is no CPU fallback: dg_create fails
 * when no gfx950 device is present.
 */
#ifndef DOOMGPU_H
#define DOOMGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum dg_status {
    DG_OK = 0,
    DG_ERR_INVALID = -1,     /* bad argument */
    DG_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
    DG_ERR_HIP = -3,         /* HIP runtime error, see dg_last_error */
    DG_ERR_WAD = -4,         /* WAD/map parse error (reference: panic in src/wad.rs, src/map/, src/graphics/) */
    DG_ERR_RENDER = -5,      /* a condition on which the reference renderer panics */
    DG_ERR_CAPACITY = -6     /* batch / list larger than the ctx was created for */
} dg_status;

typedef struct dg_scene dg_scene; /* host-side immutable world: Map + Palette + Textures + Flats + Sprites + MapObjects */
typedef struct dg_ctx dg_ctx;     /* one GPU: device copy of a scene, staging rings, streams */

/* ---- scene (reference: Game::new minus SDL, src/game.rs:142-167) ------------------------------------------ */
/* WadFile::new + Map::new + Palette/Flats/Textures/Sprites/MapObjects::new + get_sky_texture.
 * `wad` is copied.  map_name as given to `--map` (src/main.rs:33-35), e.g. "e1m1". */
int dg_scene_load_wad(const uint8_t *wad, size_t len, const char *map_name, dg_scene **out);
void dg_scene_free(dg_scene *s);
/* Player1Start (src/game.rs:151-156). */
int dg_scene_player_start(const dg_scene *s, float *x, float *y, float *angle);
/* get_sector_from_vertex(..).floor_height (src/renderer/bsp.rs:9-44, src/game.rs:386-388).
 * Returns DG_OK and writes *floor_height, or 1 if the point is in no sector (value untouched). */
int dg_scene_floor_height_at(const dg_scene *s, float x, float y, float *floor_height);
/* Per-frame game-state snapshot hooks (the reference mutates these between frames: src/lights.rs,
 * src/map_objects.rs:63-121).  Not needed for frame-0 parity. */
int dg_scene_sector_count(const dg_scene *s);
int dg_scene_set_sector_light(dg_scene *s, int sector, int16_t light_level);
int dg_scene_mobj_count(const dg_scene *s);
/* state: sprite name (4 chars), frame (0 = 'A'), full_bright, or sprite == NULL for StateId::S_NULL (not drawn). */
int dg_scene_set_mobj_state(dg_scene *s, int mobj, const char *sprite, uint8_t frame, int full_bright);
/* A state change may decode sprite bitmaps the GPU copy does not hold yet: submissions then fail with DG_ERR_INVALID until
 * dg_upload_scene is called again (light levels never need a re-upload). */

/* ---- viewpoint (reference: `Player`, src/game.rs:40-45, + Renderer::new's timestamp) ------------------------ */
typedef struct dg_view {
    float x, y;          /* player.position */
    float angle;         /* player.angle (radians) */
    float floor_height;  /* player.floor_height */
    float cos_a, sin_a;  /* f32::cos/sin(angle)   as the host libm returns them (src/map/vertexes.rs:20-25) */
    float cos_na, sin_na;/* f32::cos/sin(-angle) */
    float timestamp;     /* clock.timestamp: selects the animated-flat frame (src/graphics/flats.rs:103-111) */
    int32_t trig_valid;  /* 0: the library fills the four trig fields with cosf/sinf */
} dg_view;

/* Game state of ONE view on top of the scene's: what the reference's thinkers changed before that frame was drawn — sector
 * light levels (LightFlash / StrobeFlash / GlowingLight / FireFlicker, src/lights.rs:47-259) and map-object states
 * (src/map_objects.rs:63-121).  Entries override the scene's value for that view only, so the frames of a recorded play-through
 * can travel in one batch.  sprite_frame: dg_scene_sprite_frame(), or -1 for StateId::S_NULL (not drawn). */
typedef struct dg_sector_light { int32_t sector; int32_t light_level; } dg_sector_light;
typedef struct dg_mobj_state { int32_t mobj; int32_t sprite_frame; int32_t full_bright; int32_t reserved; } dg_mobj_state;
typedef struct dg_view_state {
    const dg_sector_light *lights; uint32_t n_lights;
    const dg_mobj_state *mobjs;    uint32_t n_mobjs;
} dg_view_state;
/* Handle of (sprite, frame) for dg_mobj_state (Sprites::get_picture's first two arguments, src/graphics/sprites.rs:99-117);
 * negative on error.  Like dg_scene_set_mobj_state it may decode new bitmaps: call it before dg_upload_scene. */
int dg_scene_sprite_frame(dg_scene *s, const char *sprite, uint8_t frame);

/* ---- context ------------------------------------------------------------------------------------------------ */
typedef struct dg_config {
    int32_t device;        /* HIP device ordinal */
    int32_t width, height; /* frame size (the reference's SCREEN_WIDTH/HEIGHT, src/game.rs:28-29); any width (a multiple of 4 takes the faster read-out) */
    int32_t max_batch;     /* frames per submission */
    int32_t slots;         /* in-flight submissions (>= 1); each owns a framebuffer slab of max_batch frames */
    int32_t host_threads;  /* list-generation threads for dg_render_views / dg_submit_views (0 = the process's CPU share: affinity mask and cgroup CPU quota, capped at 16).
                            * The default assumes ONE ctx per container quota: a process tree with several contexts (one rank per GPU) passes each its
                            * part of the quota, as bench.py does (default_host_threads) */
    int32_t front_end;     /* DG_FE_*: where the per-column half of Segs::process_sidedef / draw_map_objects runs */
} dg_config;

/* dg_config.front_end: how much of the front end (mod.rs:61-104, segs.rs:121-590, sidedef_visplanes.rs, map_objects.rs) runs on the GPU.
 *   DG_FE_HOST         everything on the host: it walks every screen column and ships finished span lists (what dg_draw_lists consumes)
 *   DG_FE_DEVICE       the host does the per-seg half (BSP order, transform, clip, projection: segs.rs:353-590,121-200) and ships per-seg /
 *                      per-sprite records; one GPU lane per screen column does segs.rs:202-345, sidedef_visplanes.rs and
 *                      map_objects.rs:130-209.  A frame that exceeds a device-side capacity is redone through DG_FE_HOST transparently
 *                      (same pixels either way).
 *   DG_FE_DEVICE_SEGS  the per-seg half runs on the GPU too (one lane per seg / map object, one wavefront per frame for what depends on
 *                      the BSP order): the host ships 88 bytes per view and nothing else (with per-view game state,
 *                      dg_submit_views_state: plus one copy of the sector lights and map-object states per view).  Frames it cannot
 *                      judge (a reference panic, a per-frame capacity: 256 parts after culling, 512 map objects in view, 2 560 columns)
 *                      and maps in which a texture / flat lookup would panic fall back to DG_FE_DEVICE / DG_FE_HOST transparently.
 *                      Costs device memory per ctx: max_batch x segs x 41 B of candidate rows + occupancy bits, plus max_batch x segs x 21 B for maps with
 *                      more than 307 segs (candidate lists longer than shared memory holds); when that allocation fails the ctx simply
 *                      keeps the host's per-seg half (DG_FE_DEVICE).
 *   DG_FE_AUTO         DG_FE_DEVICE or DG_FE_DEVICE_SEGS per batch, whichever is the faster way for it: the GPU takes the per-seg half when
 *                      nothing is in flight (the host's time would be exposed), until a seg-walk batch has been timed, or when the host has been measured to be the slower
 *                      side (few host threads, small frames); batches of fewer than 64 views always use the host walker.  The pixels
 *                      are the same whichever is picked; dg_timing.front_end says which one it was.  The choice rests on wall-clock
 *                      measurements (host time per view, kernel time of finished batches): which front end — and therefore which
 *                      instruction stream, and which fallback counters — a given batch gets is NOT reproducible from run to run.
 *                      Ask for DG_FE_DEVICE or DG_FE_DEVICE_SEGS where that matters. */
enum { DG_FE_AUTO = 0, DG_FE_HOST = 1, DG_FE_DEVICE = 2, DG_FE_DEVICE_SEGS = 3 };

int dg_create(const dg_config *cfg, dg_ctx **out);
void dg_destroy(dg_ctx *ctx);
/* Number of host threads the ctx uses for list generation (after the default / cap has been applied). */
int dg_ctx_host_threads(const dg_ctx *ctx);
/* Submissions in which a device-side capacity was exceeded: the column scratch of the device column walk (the frames concerned
 * are redone through the host list path, see dg_ctx_redone_frames).  Same pixels either way; a workload that keeps hitting it
 * should raise DOOMGPU_FE_COLUMN_SLOTS.
 * Error reporting differs from the reference in one documented way: BSP subtrees that cannot contribute to the frame are not
 * walked, so a panic the reference would raise while processing a seg in such a subtree
 * (segs.rs:140-145,431-436) is not reported as DG_ERR_RENDER; missing texture / flat lookups disable the skipping for the map. */
int dg_ctx_fallbacks(const dg_ctx *ctx, uint64_t *front_end);
/* Frames that were redone one at a time through the host list path because THEY overflowed a capacity of the device column walk
 * (the other frames of their batch were kept); a frame that does not fit the single-frame scratch either makes the whole batch go
 * through DG_FE_HOST, which dg_ctx_fallbacks counts like every overflow event. */
int dg_ctx_redone_frames(const dg_ctx *ctx, uint64_t *frames);
/* Copy palette, texel planes, flats to HBM (immutable per map). The scene must outlive the ctx's use of it. */
int dg_upload_scene(dg_ctx *ctx, const dg_scene *scene);

/* ---- full path ---------------------------------------------------------------------------------------------- */
/* Synchronous: render n views; if rgb24_out != NULL copy n*3*W*H bytes to host memory.  Uses slot 0. */
int dg_render_views(dg_ctx *ctx, const dg_view *views, int n, uint8_t *rgb24_out);
/* Asynchronous: build lists on the host (blocking), then enqueue the H2D copy on the slot's stream and the kernels behind it on the
 * ctx's kernel stream (all slots' kernels run there, in submission order). */
int dg_submit_views(dg_ctx *ctx, int slot, const dg_view *views, int n);
/* The same with a game-state snapshot per view (states[i] for views[i]; states == NULL: none). */
int dg_submit_views_state(dg_ctx *ctx, int slot, const dg_view *views, const dg_view_state *states, int n);
int dg_render_views_state(dg_ctx *ctx, const dg_view *views, const dg_view_state *states, int n, uint8_t *rgb24_out);
int dg_wait(dg_ctx *ctx, int slot);
/* Device address of the slot's framebuffer slab (frame i at + i*3*W*H). Valid until the slot is re-submitted. */
int dg_slot_framebuffer(dg_ctx *ctx, int slot, void **device_ptr);
/* Page-locked host memory for dg_readback / dg_render_views targets (a pageable buffer works too, at a lower PCIe rate). */
void *dg_alloc_host(size_t bytes);
void dg_free_host(void *p);
/* D2H copy of frames [first, first+count) of a completed slot. */
int dg_readback(dg_ctx *ctx, int slot, int first, int count, uint8_t *rgb24_out);
/* The same without waiting: the copy is queued behind the slot's kernels on the slot's own copy stream, so it overlaps the
 * kernels of the NEXT submission on another slot (the reference's caller consumes `pixels.pixels` on the host every frame,
 * src/game.rs:521-525).  rgb24_out should be page-locked (dg_alloc_host) and is complete after dg_wait(slot); every call that renders
 * into the slot again (dg_submit_views*, dg_render_views*, dg_prepare_views, dg_replay_slot, dg_draw_lists) and dg_upload_scene
 * complete it first, so the copy never sees a half-overwritten frame.  One readback in flight per slot. */
int dg_readback_async(dg_ctx *ctx, int slot, int first, int count, uint8_t *rgb24_out);
/* Frame sink without the PCIe copy: one 64-bit checksum per frame of a finished slot, computed on the GPU over the frame's
 * RGB24 bytes taken as little-endian dwords d[0 .. ceil(3*W*H/4)) (a last partial dword is zero-extended):
 *     sum over i of  m ^ (m >> 32),   m = (d[i] ^ (i * 0x9E3779B97F4A7C15)) * 0xBF58476D1CE4E5B9    (all mod 2^64)
 * so a host that holds reference frames (e.g. `pixels.pixels` dumps of the reference, src/game.rs:521-525) can compare
 * thousands of large frames by 8 bytes each.  Waits for the slot like dg_readback. */
int dg_frame_checksums(dg_ctx *ctx, int slot, int first, int count, uint64_t *out);

/* Pre-built list path used by benchmarks that want the raster kernels alone: build + upload lists for n views
 * into the slot (untimed), then dg_replay_slot re-runs only the device work (setup + raster kernels). */
int dg_prepare_views(dg_ctx *ctx, int slot, const dg_view *views, int n);
int dg_replay_slot(dg_ctx *ctx, int slot);

/* ---- list path ---------------------------------------------------------------------------------------------- */
/* BitmapColumn, src/renderer/bitmap_render.rs:19-25 (all five values originate as i16: segs.rs:205-220,260) */
typedef struct dg_bitmap_column {
    int16_t x, clipped_top_y, clipped_bottom_y, bottom_y, top_y;
} dg_bitmap_column;

/* BitmapRender, src/renderer/bitmap_render.rs:29-45, restricted to what render_vertical_bitmap_line reads */
typedef struct dg_bitmap_render {
    int32_t bitmap;              /* dg_scene bitmap id (dg_scene_texture_id / dg_scene_sprite_bitmap_id) */
    int16_t light_level;
    int16_t offset_x, offset_y;
    int16_t reserved;
    float line_start_x, line_start_y, line_end_x, line_end_y; /* clipped_line.line */
    float start_offset;                                        /* clipped_line.start_offset */
    int32_t start_x, end_x;
    float bottom_height, top_height;
    uint32_t first_column, n_columns; /* range in the columns array */
} dg_bitmap_render;

/* Visplane, src/renderer/visplanes.rs:17-26; top/bottom stored only for [left, right] */
typedef struct dg_visplane {
    int32_t flat;        /* dg_scene flat id (dg_scene_flat_id); negative = sky (flat name contains "SKY") */
    int16_t height, light_level, left, right;
    uint32_t first_entry; /* index of (top[left], bottom[left]) in the plane_tb array; entries are (top, bottom) pairs */
} dg_visplane;

/* One draw call of the reference, in the order Renderer::render issues them. */
typedef struct dg_draw_cmd {
    uint32_t kind;  /* 0 = replay a dg_bitmap_render (all its columns), 1 = draw_visplane */
    uint32_t index;
} dg_draw_cmd;

typedef struct dg_frame_lists {
    dg_view view;
    const dg_bitmap_render *renders; uint32_t n_renders;
    const dg_bitmap_column *columns; uint32_t n_columns;
    const dg_visplane *visplanes;    uint32_t n_visplanes;
    const int16_t *plane_tb;         uint32_t n_plane_tb; /* int16 count (2 per entry) */
    const dg_draw_cmd *order;        uint32_t n_order;
} dg_frame_lists;

int dg_scene_texture_id(const dg_scene *s, const char *name);                 /* Textures::get (textures.rs:154-179); <0 unknown */
int dg_scene_flat_id(const dg_scene *s, const char *name, float timestamp);   /* Flats::get_animated (flats.rs:103-111); sky => negative */
int dg_scene_sprite_bitmap_id(const dg_scene *s, const char *sprite, uint8_t frame, uint8_t rotation); /* Sprites::get_picture */
int dg_scene_bitmap_size(const dg_scene *s, int bitmap, int *w, int *h);

/* Rasterise caller-built lists for n frames into the slot (synchronous, like dg_render_views). */
int dg_draw_lists(dg_ctx *ctx, int slot, const dg_frame_lists *frames, int n, uint8_t *rgb24_out);

/* The library's own list builder, exposed so a host can inspect / compare lists (and so tests can check the
 * host logic without a GPU).  The returned pointers live in an internal per-thread arena and stay valid
 * until the next dg_build_lists call on the same thread. */
int dg_build_lists(const dg_scene *s, int width, int height, const dg_view *view, dg_frame_lists *out);

/* ---- misc ----------------------------------------------------------------------------------------------------- */
const char *dg_last_error(void); /* thread-local message of the last failing call */
/* "doomgpu <release> (gfx950; ABI <n>)".  The ABI number changes whenever a struct in this header changes size or a function its
 * arguments: ABI 3 (round 3) dropped dg_timing.strips_ms and the third argument of dg_ctx_fallbacks; ABI 4 changes no signature
 * (it marks the library in which dg_version started to carry the number).  A caller built against another ABI must not call on. */
const char *dg_version(void);

/* Timing of the last dg_replay_slot / submit on a slot (ms), from HIP events attached to the kernel dispatches themselves on the ctx's
 * kernel stream: setup_ms = start of the first front-end kernel .. end of the last, raster_ms = the raster launch, total_ms = both. */
typedef struct dg_timing {
    float setup_ms, raster_ms, total_ms;
    float host_ms;            /* host list generation + binning + packing of that submission (wall clock) */
    uint64_t n_spans, n_frames, covered_pixels;
    uint64_t n_walls, n_planes, list_bytes; /* drawn records / visplanes, bytes of lists copied to HBM */
    int32_t front_end;        /* DG_FE_HOST, DG_FE_DEVICE or DG_FE_DEVICE_SEGS: what that submission actually used; with DG_FE_DEVICE setup_ms is
                                 the column walk (dg_fe_columns, dg_fe_gaps, dg_fe_scan, dg_fe_scatter), n_walls = wall records,
                                 n_planes = sprites, covered_pixels is not tracked (0) */
} dg_timing;
int dg_slot_timing(dg_ctx *ctx, int slot, dg_timing *out);

#ifdef __cplusplus
}
#endif
#endif
