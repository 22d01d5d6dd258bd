// fs_kernels.hip — the device seg walk (DG_FE_DEVICE_SEGS): BSP visit order (inside dg_fs_segs: fs_leaf_base), per-seg processing, hidden-part culling, map objects, draw
// sequence and column bins on the GPU.  Bodies: fs_core.h (arithmetic shared with the host walker) and fs_frame.h (the per-frame
// phases, also run by tests/emul on the CPU).  Integer / f32 work with short dependent chains; nothing here is a contraction (no MFMA).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "fs_kernels.hpp"

namespace dg {

namespace {

__global__ __launch_bounds__(64) void dg_fs_segs(FsParams P) {
    const uint32_t si = blockIdx.x * 64u + threadIdx.x;
    if (si < P.n_segs) fs_seg_lane(P, (int)blockIdx.y, si);
}

// One workgroup (four wavefronts) per frame.  What the phases cost is their dependent loads, which is why the serial ones (lane 0)
// read shared memory only and meet nothing but the few candidates that survived the parallel tests.
__global__ __launch_bounds__(FS_LANES) void dg_fs_frame(FsParams P) {
    __shared__ FsShared S;
    const int f = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (lane == 0) fs_ph_init(S);
    __syncthreads();
    fs_ph_cand_count(P, S, f, lane);
    __syncthreads();
    fs_ph_block_sums(S, lane);
    __syncthreads();
    fs_ph_cand_stage(P, S, f, lane);
    fs_ph_first_clear(P, S, lane);
    __syncthreads();
    fs_ph_solids(P, S, f, lane);
    __syncthreads();
    fs_ph_keep(P, S, f, lane);
    __syncthreads();
    fs_ph_kept_count(P, S, f, lane);
    __syncthreads();
    fs_ph_block_sums(S, lane);
    __syncthreads();
    fs_ph_kept_place(P, S, f, lane);
    __syncthreads();
    fs_ph_emit(P, S, f, lane);
    __syncthreads();
    for (uint32_t base = 0; base < P.n_mobjs; base += FS_LANES) {
        FsSpriteTmp T;
        const uint32_t n_before = S.n_sprites;
        fs_ph_mobj(P, S, f, base, lane, T);
        __syncthreads();
        fs_ph_block_sums(S, lane);
        __syncthreads();
        fs_ph_mobj_emit(P, S, f, lane, T, n_before);
        __syncthreads();
    }
    fs_ph_behind(P, S, f, lane);
    fs_ph_sprite_order(S, lane);
    __syncthreads();
    fs_ph_masked_when(S, lane);
    __syncthreads();
    fs_ph_seq(P, S, f, lane);
    fs_ph_bin_clear(P, S, lane);
    __syncthreads();
    fs_ph_bin_mark(P, S, lane);
    __syncthreads();
    fs_ph_bin_count(P, S, lane);
    __syncthreads();
    if (lane == 0) fs_ph_bin_prefix(P, S, f);
    __syncthreads();
    fs_ph_bin_fill(P, S, f, lane);
    __syncthreads();
    fs_ph_clean(P, f, lane);
    if (lane == 0) fs_ph_header(P, S, f);
}

}  // namespace

hipError_t launch_fs(const FsParams &P, hipStream_t stream, hipEvent_t start) {
    if (P.n_frames <= 0) return start ? hipEventRecord(start, stream) : hipSuccess;
    if (P.n_segs == 0) return hipErrorInvalidValue;                     // (upload_fs_scene keeps such a scene off the seg walk)
    hipExtLaunchKernelGGL(dg_fs_segs, dim3((P.n_segs + 63u) / 64u, (unsigned)P.n_frames), dim3(64), 0, stream, start, nullptr, 0, P);
    if (const hipError_t e = hipGetLastError(); e != hipSuccess) return e;   // each launch checked: a later success would hide it
    hipLaunchKernelGGL(dg_fs_frame, dim3((unsigned)P.n_frames), dim3(FS_LANES), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dg
