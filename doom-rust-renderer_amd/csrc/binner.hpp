// binner.hpp — draw-ordered record lists (dg_frame_lists) -> column-major device spans (lists_dev.h).
#pragma once
#include <string>
#include <vector>

#include "../../include/doomgpu.h"
#include "frontend.hpp"
#include "lists_dev.h"
#include "scene.hpp"

namespace dg {

struct BinnedFrame {
    DevFrame hdr;                       // bases are filled by the batch packer
    std::vector<DevSpan> spans;         // column-major, draw order inside a column
    std::vector<uint32_t> col_off;      // W + 1
    std::vector<DevWallRec> walls;
    std::vector<DevPlaneRec> planes;
    uint64_t covered_pixels = 0;        // sum of span heights (pixel evaluations incl. overdraw)
    // scratch
    std::vector<DevSpan> events;
    std::vector<uint32_t> cursor;
};

// Per-frame view constants of the texture mappers (rotation, position as i16, draw_sky's tx_offset); bases left 0.
DevFrame make_frame_header(const dg_view &v);

// Returns DG_OK / DG_ERR_INVALID (malformed caller lists) / DG_ERR_RENDER (reference would panic).
int bin_frame(const Scene &sc, const FrameConsts &k, const dg_frame_lists &fl, BinnedFrame &out, std::string &err);

}  // namespace dg
