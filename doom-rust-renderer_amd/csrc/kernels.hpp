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
    const uint4 *row_tab;        // [H] per screen row: prepared 1 / vy, vy = CFY - y, sky row | fast-divide bit, sky factor (dg_row_table, kernels.hip)
    int32_t n_frames;
    int32_t frame_per_xcd;       // the workgroups of a frame on one XCD (kernels.hip: raster_block); set by launch_raster, like the two
    uint32_t xcd_rcp_pf, xcd_rcp_gx;   // reciprocals it divides by (workgroups per frame, strips per frame)
    int32_t tile_rows_per_wg;    // tile rows one workgroup of dg_raster_tiles renders out of one staging pass; <= 0: launch_raster picks
};

// (start / stop: optional timing events attached to the dispatch itself — hipExtLaunchKernelGGL — instead of recorded around it: an event
// record is a packet of its own in the stream and costs ~5 us between two kernels)
hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// dg_raster_tiles over every (frame, 64 x 64 tile).
hipError_t launch_raster(const RasterParams &P, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);
// out[k] = checksum (include/doomgpu.h: dg_frame_checksums) of frame k of `count` consecutive frames of `frame_bytes` bytes at fb; out must be zeroed.
hipError_t launch_checksums(const uint8_t *fb, size_t frame_bytes, int count, unsigned long long *out, hipStream_t stream);
// Fills row_tab[0 .. H) for the given scene / frame size (once per dg_upload_scene).
hipError_t launch_row_table(const DevScene &scene, const DevConsts &k, uint4 *row_tab, hipStream_t stream);

}  // namespace dg
