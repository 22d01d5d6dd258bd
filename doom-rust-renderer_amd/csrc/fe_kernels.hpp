// fe_kernels.hpp — launch interface of the device column walk (fe_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "fe_core.h"

namespace dg {

// dg_fe_columns + dg_fe_finalize on `stream`.  P.flags must be zeroed (in stream order) before the launch.
hipError_t launch_fe(const FeParams &P, hipStream_t stream);

}  // namespace dg
