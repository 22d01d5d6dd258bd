// kernels.hpp — launch interface between the context (host) and kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include "lists_dev.h"

namespace dg {

struct RasterParams {
    DevScene scene;
    DevConsts k;
    const DevFrame *frames;      // [n_frames]
    const uint32_t *col_off;     // [n_frames][W + 1]
    const DevSpan *spans;
    DevRSpan *rspans;            // written by dg_setup_spans, walked by dg_raster_tiles
    const DevWallRec *walls;
    const DevPlaneRec *planes;
    uint8_t *fb;                 // n_frames x 3*W*H, RGB24
    const uint4 *row_tab;        // [H] per screen row: x = bits of prepare_rcp(CFY - y), y = sky texture row (or -1), z = bits of CFY - y; dg_row_table
    int32_t n_frames;
    // strip path (strip_core.h): written by dg_resolve_columns, read by dg_raster_strips and dg_raster_tile_list
    DevSeg *segs;                // [n_frames][seg_cap][W]
    uint8_t *band_first;         // [n_frames][n_bands][W] slot of the segment that contains the first row of a band
    uint32_t *frame_flags;       // [max_batch] != 0: a column needed more than seg_cap segments; the batch is redone with strips = 0
    uint32_t *tile_counters;     // [0] tiles on the list, [1] spare
    uint8_t *band_ovl;           // [n_frames][n_bands][ceil(W / 64)] != 0: an overlay span (strip_core.h) touches that band of that 64-column
                                 // strip -> rendered by dg_raster_tile_list (these three directly follow each other: one fill clears them)
    uint32_t *tile_list;         // frame << 16 | band << 8 | strip
    int32_t seg_cap, band_rows, n_bands;
    int32_t strips;              // 0: dg_raster_tiles alone renders everything (no resolve, no strips)
};

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream);
// dg_resolve_columns + dg_raster_strips + dg_raster_tile_list when P.strips, else dg_raster_tiles.
// after_resolve (optional) is recorded between dg_resolve_columns and the pixel kernels.
// With aux / aux_done (a second stream and an event), dg_raster_tile_list runs on aux beside dg_raster_strips; `stream` waits for it.
hipError_t launch_raster(const RasterParams &P, hipStream_t stream, hipEvent_t after_resolve = nullptr, hipStream_t aux = nullptr, hipEvent_t aux_done = nullptr);
// Rows per band of dg_raster_strips for a frame height (one wavefront renders 64 columns x band_rows rows).
int strip_band_rows(int H);
// out[k] = checksum (include/doomgpu.h: dg_frame_checksums) of frame k of `count` consecutive frames of `frame_bytes` bytes at fb; out must be zeroed.
hipError_t launch_checksums(const uint8_t *fb, size_t frame_bytes, int count, unsigned long long *out, hipStream_t stream);
// Fills row_tab[0 .. H) for the given scene / frame size (once per dg_upload_scene).
hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint4 *row_tab, hipStream_t stream);

}  // namespace dg
