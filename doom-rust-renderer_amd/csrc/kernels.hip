// kernels.hip — gfx950 (MI355X / CDNA4) kernels of the rasteriser.  wave64, no MFMA (nothing here is a
// contraction): the work is per-pixel f32 evaluation + byte gathers, and the output is a stream of RGB24 bytes.
//
// Kernel 1  dg_setup_spans   one lane per span: column-invariant part of the wall/sprite mapper
//                            (bitmap_render.rs:241-251: 3 f32 divides -> texture column + light factor),
//                            sky texture column, floor/ceiling vx.  Writes DevSpanAux (8 B/span).
// Kernel 2  dg_raster_tiles  one workgroup per (frame, 128-column x 64-row tile); each of its 4 wavefronts
//                            takes one screen column at a time with lane = row, so that
//                              * the span list of the column is wave-uniform (one coalesced 1 KB load of up to 64
//                                spans, then ballot + readlane; spans are applied in draw order so the last
//                                writer wins exactly as in the reference),
//                              * a wall column reads one texture column ([x][y] texel layout => consecutive bytes),
//                              * finished pixels go to an LDS tile [row][col] and leave the CU as fully
//                                coalesced 12-byte-per-lane RGB24 row segments (384 B contiguous per tile row).
//                            Every pixel of the tile is stored (uncovered = 0,0,0) which fuses the reference's
//                            per-frame `Pixels::new()` clear (pixels.rs:10-14) into the one write pass.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (contraction would break bit-exactness; the IEEE
// divide expansion keeps its own internal FMAs, which is what makes it correctly rounded).
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "raster_core.h"

namespace dg {

constexpr int TILE_W = 128;       // columns per workgroup
constexpr int TILE_H = 64;        // rows per workgroup = lanes per wave
constexpr int TILE_STRIDE = 132;  // dwords per LDS tile row: 16-B aligned rows, breaks the power-of-two stride
constexpr int WAVES = 4;

__global__ __launch_bounds__(256) void dg_setup_spans(RasterParams P) {
    const int f = blockIdx.y;
    const DevFrame fr = P.frames[f];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fr.n_spans) return;
    const DevSpan sp = P.spans[fr.span_base + i];
    DevSpanAux a;
    if (sp.kind == SPAN_WALL) {
        a = wall_column_setup(P.walls[fr.wall_base + sp.rec], sp.x);
    } else if (sp.kind == SPAN_FLAT) {
        a.texcol = 0;
        a.factor = flat_column_vx(P.k, sp.x);
    } else {
        a.texcol = sky_column_setup(P.scene, P.k, fr, sp.x);
        a.factor = 0.0f;
    }
    P.aux[fr.span_base + i] = a;
}

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

__global__ __launch_bounds__(256) void dg_raster_tiles(RasterParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_H * TILE_STRIDE];

    const int f = blockIdx.z;
    const DevFrame fr = P.frames[f];
    const int W = P.k.W, H = P.k.H;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = y0 + lane;
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const DevSpan *spans = P.spans + fr.span_base;
    const DevSpanAux *aux = P.aux + fr.span_base;
    const DevWallRec *walls = P.walls + fr.wall_base;
    const DevPlaneRec *planes = P.planes + fr.plane_base;

    for (int c = wave; c < TILE_W; c += WAVES) {
        const int x = x0 + c;
        uint32_t color = 0;
        if (x < W) {
            const uint32_t n0 = coff[x], n1 = coff[x + 1];
            for (uint32_t base = n0; base < n1; base += 64) {
                const uint32_t i = base + (uint32_t)lane;
                uint4 raw = make_uint4(0, 0, 0, 0);
                bool hit = false;
                if (i < n1) {
                    raw = *reinterpret_cast<const uint4 *>(&spans[i]);
                    const int ctop = (int)(int16_t)(raw.x & 0xffffu), cbot = (int)(int16_t)(raw.x >> 16);
                    hit = cbot >= y0 && ctop <= y0 + (TILE_H - 1);
                }
                unsigned long long m = __ballot(hit);
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t w0 = readlane_u32(raw.x, j), w1 = readlane_u32(raw.y, j), w2 = readlane_u32(raw.z, j);
                    const int ctop = (int)(int16_t)(w0 & 0xffffu), cbot = (int)(int16_t)(w0 >> 16);
                    const int top_y = (int)(int16_t)(w1 & 0xffffu), bot_y = (int)(int16_t)(w1 >> 16);
                    const uint32_t rec = w2 & 0xffffu, kind = (w2 >> 16) & 0xffu;
                    const DevSpanAux a = aux[base + (uint32_t)j];
                    if (y >= ctop && y <= cbot) {
                        if (kind == SPAN_WALL) {
                            uint32_t rgb;
                            if (wall_pixel(P.scene, walls[rec], a, top_y, bot_y, y, rgb)) color = rgb;
                        } else if (kind == SPAN_FLAT) {
                            color = flat_pixel(P.scene, P.k, fr, planes[rec], a.factor, y);
                        } else {
                            uint32_t rgb;
                            if (sky_pixel(P.scene, P.k, a.texcol, y, rgb)) color = rgb;
                        }
                    }
                }
            }
        }
        tile[lane * TILE_STRIDE + c] = color;
    }
    __syncthreads();

    // Read-out: groups of 4 pixels (16 B of RGBX in LDS -> 12 B of RGB24 in HBM); 32 groups per tile row, lanes
    // of a wave cover two full tile rows = 2 x 384 contiguous bytes.
    uint8_t *fb = P.fb + (size_t)f * (size_t)3 * (size_t)W * (size_t)H;
    for (int g = threadIdx.x; g < TILE_H * (TILE_W / 4); g += 256) {
        const int row = g >> 5, gc = g & 31;
        const int yy = y0 + row, xx = x0 + 4 * gc;
        if (yy < H && xx < W) {   // W % 4 == 0 (checked at dg_create), so a group never straddles the right edge
            const uint4 p = *reinterpret_cast<const uint4 *>(&tile[row * TILE_STRIDE + 4 * gc]);
            const uint32_t o0 = (p.x & 0xffffffu) | (p.y << 24);
            const uint32_t o1 = ((p.y >> 8) & 0xffffu) | (p.z << 16);
            const uint32_t o2 = ((p.z >> 16) & 0xffu) | (p.w << 8);
            uint32_t *dst = reinterpret_cast<uint32_t *>(fb + ((size_t)yy * (size_t)W + (size_t)xx) * 3);
            dst[0] = o0;
            dst[1] = o1;
            dst[2] = o2;
        }
    }
}

hipError_t launch_setup(const RasterParams &P, uint32_t max_spans_per_frame, hipStream_t stream) {
    if (P.n_frames <= 0 || max_spans_per_frame == 0) return hipSuccess;
    dim3 grid((max_spans_per_frame + 255) / 256, (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_setup_spans, grid, dim3(256), 0, stream, P);
    return hipGetLastError();
}

hipError_t launch_raster(const RasterParams &P, hipStream_t stream) {
    if (P.n_frames <= 0) return hipSuccess;
    dim3 grid((unsigned)((P.k.W + TILE_W - 1) / TILE_W), (unsigned)((P.k.H + TILE_H - 1) / TILE_H), (unsigned)P.n_frames);
    hipLaunchKernelGGL(dg_raster_tiles, grid, dim3(256), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dg
