// fe_kernels.hpp — launch interface of the device column walk (fe_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "fe_core.h"

namespace dg {

// dg_fe_columns + dg_fe_finalize on `stream`.  P.flags must be zeroed (in stream order) before the launch.
// start: attached to the first kernel's dispatch, stop: to the last one's (see kernels.hpp)
hipError_t launch_fe(const FeParams &P, hipStream_t stream, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);

}  // namespace dg
