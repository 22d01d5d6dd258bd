// kernels.hip — gfx950 (MI355X / CDNA4) kernels of the rasteriser.  wave64, no MFMA (nothing here is a
// contraction): the work is per-pixel f32 evaluation + byte gathers, and the output is a stream of RGB24 bytes.
//
// Kernel 1  dg_setup_spans   one lane per span: column-invariant part of the wall/sprite mapper
//                            (bitmap_render.rs:241-251: 3 f32 divides -> texture column + light factor),
//                            sky texture column, floor/ceiling vx; merges the span with its record into one
//                            self-contained 32-byte DevRSpan.
// Kernel 2  dg_raster_tiles  one workgroup (8 wavefronts) per (frame, 128-column x 64-row tile); a wavefront
//                            takes one screen column at a time with lane = row, so that
//                              * the span list of the column is wave-uniform: ONE coalesced load brings in up to 64
//                                spans (lane i = span i, 2 x 16 B), a ballot picks the ones touching these 64 rows
//                                and v_readlane broadcasts their 8 words — no dependent record loads; spans are
//                                applied in draw order so the last writer wins exactly as in the reference,
//                              * a wall column reads one texture column ([x][y] texel layout => consecutive bytes),
//                              * the palette lives in LDS (1 KB),
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

constexpr int TILE_W = 64;        // columns per workgroup
constexpr int TILE_H = 64;        // rows per workgroup = lanes per wave
constexpr int TILE_STRIDE = 68;   // dwords per LDS tile row (16-B aligned rows for the b128 read-out)
constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;
constexpr int SPAN_CAP = 512;     // spans of one tile's 64 columns staged in LDS (16 KB); typical tiles hold 100-400

__global__ __launch_bounds__(256) void dg_setup_spans(RasterParams P) {
    const int f = blockIdx.y;
    const DevFrame fr = P.frames[f];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= fr.n_spans) return;
    const DevSpan sp = P.spans[fr.span_base + i];
    DevRSpan o;
    if (sp.kind == SPAN_WALL) o = resolve_wall_span(sp, P.walls[fr.wall_base + sp.rec]);
    else if (sp.kind == SPAN_FLAT) o = resolve_flat_span(sp, P.planes[fr.plane_base + sp.rec], P.k);
    else o = resolve_sky_span(sp, P.scene, P.k, fr);
    uint4 *dst = reinterpret_cast<uint4 *>(&P.rspans[fr.span_base + i]);
    dst[0] = make_uint4(o.w[0], o.w[1], o.w[2], o.w[3]);
    dst[1] = make_uint4(o.w[4], o.w[5], o.w[6], o.w[7]);
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

// One screen column (64 rows of it) for one wavefront: walk the column's spans in draw order, keep the last opaque
// writer per row.  `src` points at the column's first DevRSpan — in LDS when the tile's spans were staged, else in HBM.
template <typename SpanPtr>
__device__ __forceinline__ uint32_t raster_column(const RasterParams &P, const DevFrame &fr, const uint32_t *pal, SpanPtr src, uint32_t n,
                                                  int lane, int y, int y0, float vy, float r_vy) {
    const uint8_t *texel_idx = P.scene.texel_idx, *texel_opq = P.scene.texel_opq, *flats = P.scene.flats;
    uint32_t color = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        uint4 ra = make_uint4(0, 0, 0, 0), rb = make_uint4(0, 0, 0, 0);
        bool hit = false;
        if (i < n) {
            ra = src[2 * i];
            rb = src[2 * i + 1];
            hit = hi_i16(ra.x) >= y0 && lo_i16(ra.x) <= y0 + (TILE_H - 1);
        }
        unsigned long long m = __ballot(hit);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const uint32_t w0 = bcast(ra.x, j), w1 = bcast(ra.y, j), w2 = bcast(ra.z, j), w3 = bcast(ra.w, j);
            const uint32_t w4 = bcast(rb.x, j), w5 = bcast(rb.y, j), w6 = bcast(rb.z, j), w7 = bcast(rb.w, j);
            if (y >= lo_i16(w0) && y <= hi_i16(w0)) {
                const uint32_t kind = w6 & 0xffu;
                if (kind == SPAN_WALL) {
                    const uint32_t o = wall_texel_offset(w1, w2, w4, w5, w6, w7, y);
                    const bool opaque = (w6 & 0x100u) ? texel_opq[o] != 0 : true;
                    if (opaque) color = shade(pal[texel_idx[o]], bits_f32(w3));
                } else if (kind == SPAN_FLAT) {
                    float factor;
                    const uint32_t o = flat_texel_offset(fr, w1, w2, w4, w5, w6, vy, r_vy, factor);
                    color = shade(pal[flats[o]], factor);
                } else {
                    const uint32_t o = sky_texel_offset(P.scene, P.k, w2, y);
                    if (o != 0xffffffffu && texel_opq[o]) color = pal[texel_idx[o]];
                }
            }
        }
    }
    return color;
}

__global__ __launch_bounds__(THREADS) void dg_raster_tiles(RasterParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_H * TILE_STRIDE];
    __shared__ __attribute__((aligned(16))) uint4 lspans[SPAN_CAP * 2];
    __shared__ uint32_t pal[256];
    __shared__ uint32_t lcoff[TILE_W + 1];

    const int f = blockIdx.z;
    const DevFrame fr = P.frames[f];
    const int W = P.k.W, H = P.k.H;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = y0 + lane;
    const float vy = P.k.CFY - (float)y;      // visplanes.rs:109, a per-row constant
    const float r_vy = prepare_rcp(vy);
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    const uint4 *gspans = reinterpret_cast<const uint4 *>(P.rspans + fr.span_base);

    // Stage: palette, the tile's 65 column offsets, then the tile's spans.  The spans of 64 adjacent columns are one
    // contiguous range of the column-major span array, so this is a single fully coalesced burst with every load of the
    // workgroup in flight at once — the dependent chain col_off -> spans is paid once per tile instead of once per column.
    if (threadIdx.x < 256) pal[threadIdx.x] = P.scene.palette[threadIdx.x];
    if (threadIdx.x <= TILE_W) {
        const int xc = x0 + (int)threadIdx.x;
        lcoff[threadIdx.x] = coff[xc < W ? xc : W];
    }
    __syncthreads();
    const uint32_t t0 = lcoff[0], n_tile = lcoff[TILE_W] - t0;
    const bool staged = n_tile <= SPAN_CAP;
    if (staged) {
        for (uint32_t i = threadIdx.x; i < 2 * n_tile; i += THREADS) lspans[i] = gspans[2 * t0 + i];
        __syncthreads();
    }

    for (int c = wave; c < TILE_W; c += WAVES) {
        const uint32_t n0 = lcoff[c], n = lcoff[c + 1] - n0;
        uint32_t color;
        if (staged) color = raster_column(P, fr, pal, &lspans[2 * (n0 - t0)], n, lane, y, y0, vy, r_vy);
        else color = raster_column(P, fr, pal, gspans + 2 * (size_t)n0, n, lane, y, y0, vy, r_vy);
        tile[lane * TILE_STRIDE + c] = color;
    }
    __syncthreads();

    // Read-out: groups of 4 pixels (16 B of RGBX in LDS -> 12 B of RGB24 in HBM); 16 groups per tile row, so the lanes of
    // a wave cover four tile rows = 4 x 192 contiguous bytes.
    uint8_t *fb = P.fb + (size_t)f * (size_t)3 * (size_t)W * (size_t)H;
    for (int g = threadIdx.x; g < TILE_H * (TILE_W / 4); g += THREADS) {
        const int row = g >> 4, gc = g & 15;
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
    hipLaunchKernelGGL(dg_raster_tiles, grid, dim3(THREADS), 0, stream, P);
    return hipGetLastError();
}

}  // namespace dg
