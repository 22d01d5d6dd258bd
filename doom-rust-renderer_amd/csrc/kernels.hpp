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
};

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream);
// dg_raster_tiles over every (frame, 64 x 64 tile).
hipError_t launch_raster(const RasterParams &P, hipStream_t stream);
// out[k] = checksum (include/doomgpu.h: dg_frame_checksums) of frame k of `count` consecutive frames of `frame_bytes` bytes at fb; out must be zeroed.
hipError_t launch_checksums(const uint8_t *fb, size_t frame_bytes, int count, unsigned long long *out, hipStream_t stream);
// Fills row_tab[0 .. H) for the given scene / frame size (once per dg_upload_scene).
hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint4 *row_tab, hipStream_t stream);

}  // namespace dg
