// fs_kernels.hpp — launch interface of the device seg walk (fs_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "fs_frame.h"

namespace dg {

// dg_fs_segs, dg_fs_frame on `stream`.  P.flags must be zeroed (in stream order) before the launch; P.occ must be zero as well —
// dg_fs_frame leaves it so (the context zeroes it at upload and after a launch that failed half way).
// start: attached to the first kernel's dispatch.
hipError_t launch_fs(const FsParams &P, hipStream_t stream, hipEvent_t start = nullptr);

}  // namespace dg
