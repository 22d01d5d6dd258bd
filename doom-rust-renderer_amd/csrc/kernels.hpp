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
    // strip path (strip_core.h): written by dg_resolve_columns, read by dg_raster_strips and dg_raster_tiles (overlay mode)
    DevSeg *segs;                // [n_frames][seg_cap][W]
    uint8_t *band_first;         // [n_frames][n_bands][W] slot of the segment that contains the first row of a band
    uint16_t *ov_first;          // [n_frames][W] index (within the column's draw-ordered spans) of the first overlay span
    DevSeg *ov_inline;           // [n_frames][OV_INLINE_MAX][W] the column's inline overlay spans in draw order (strip_core.h overlay_record)
    uint8_t *ov_cnt;             // [n_frames][W] how many of them (0: none, or the column goes through dg_overlay_strips)
    uint32_t *frame_flags;       // [n_frames] != 0: a column needed more than seg_cap segments; the batch is redone with strips = 0
    uint8_t *band_ovl;           // [n_frames][n_bands][ceil(W / 64)] != 0: an overlay span touches that band of that 64-column strip
    uint8_t *band_inl;           // same shape: a column of that strip has an INLINE overlay span (strip_core.h) in that band -> dg_raster_strips_ov
                                 // (both directly behind frame_flags [max_batch]: one fill clears all three)
    int32_t seg_cap, band_rows, n_bands;
    int32_t strips;              // 0: dg_raster_tiles alone renders everything (no resolve, no strips)
};

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream);
// dg_resolve_columns + dg_raster_strips + dg_overlay_strips when P.strips, else dg_raster_tiles.
hipError_t launch_raster(const RasterParams &P, hipStream_t stream);
// Rows per band of dg_raster_strips for a frame height (one wavefront renders 64 columns x band_rows rows).
int strip_band_rows(int H);
// out[k] = checksum (include/doomgpu.h: dg_frame_checksums) of frame k of `count` consecutive frames of `frame_bytes` bytes at fb; out must be zeroed.
hipError_t launch_checksums(const uint8_t *fb, size_t frame_bytes, int count, unsigned long long *out, hipStream_t stream);
// Fills row_tab[0 .. H) for the given scene / frame size (once per dg_upload_scene).
hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint4 *row_tab, hipStream_t stream);

}  // namespace dg
