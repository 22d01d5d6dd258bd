// fe_kernels.hip — the device column walk (gfx950): turns the per-seg / per-sprite records of a batch of frames
// (fe_dev.h, built by Walker in parts mode) into the column-major DevRSpan lists dg_raster_tiles walks.
//
//   dg_fe_columns   one lane per (frame, screen column): walks the frame's parts in BSP order with the column's three
//                   occlusion values in registers (the reference keeps them in horizontal_ocl / floor_ver_ocl /
//                   ceiling_ver_ocl, segs.rs:70-74) and appends every wall / visplane / sprite span, in a compact 16-byte
//                   form, to the column's scratch list.  A wavefront owns one 64-column bin whose record list the host
//                   prepared; records are wave-uniform, staged through LDS 16 at a time (eight loads in flight) and
//                   broadcast field by field with v_readlane.  A wave whose 64 columns are all horizontally occluded stops
//                   walking.  Scratch writes are laid out [slot][column]: the 64 lanes of a wave store 64 adjacent records.
//   dg_fe_gaps      one wave per (frame, part with a sky flat): the 1-pixel sky entries of zero-filled visplane columns.
//   dg_fe_scan      one workgroup per frame: scans the per-column counts into col_off.
//   dg_fe_scatter   ranks every column's spans by draw-order key, resolves their texture-mapping constants (DevRSpan) and
//                   scatters them into the column-major list.
//
// Integer / f32 work only — nothing here is GEMM shaped.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "fe_core.h"
#include "fs_frame.h"
#include "fe_kernels.hpp"

namespace dg {

namespace {

constexpr int FE_COL_THREADS = 256;
constexpr int FE_SCAN_THREADS = 256;       // small workgroups find a free CU slot while the previous raster kernel is still running
constexpr int FE_SCATTER_GROUPS = 4;      // slot groups per screen column in dg_fe_scatter

// A record is wave-uniform: it is staged as dwords across lanes (one coalesced load instead of a chain of scalar-cache
// misses: a frame's records do not fit the 16 KB scalar cache) and every field is then broadcast with v_readlane.  The column walk
// reads only the first FE_HEAD_WORDS dwords of a FePart / FeSprite (what follows are the texture-mapping constants dg_fe_scatter
// resolves spans with), so only those travel: sixteen lanes per record, four records per load instruction.
constexpr uint32_t FE_HEAD_WORDS = 12;
static_assert(offsetof(FePart, wall) == FE_HEAD_WORDS * 4 && offsetof(FeSprite, wall) == FE_HEAD_WORDS * 4, "head of a record");

template <typename T>
__device__ __forceinline__ T unpack_head(uint32_t v) {
    union U { T t; uint32_t w[sizeof(T) / 4]; __device__ U() {} } u;
#pragma unroll
    for (uint32_t k = 0; k < FE_HEAD_WORDS; k++) u.w[k] = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k);
    return u.t;       // (the tail is never read)
}

// The heads of 4 * LOADS records, list positions g .. of the m indices held one per lane in `my`, into lrec[record * STRIDE + word]:
// lane (r, j) = (lane >> 4, lane & 15) fetches word j of records g + r, g + 4 + r, ...; all loads are issued before any is used.
template <typename T, uint32_t STRIDE, uint32_t LOADS>
__device__ __forceinline__ void stage_heads(const T *recs, int my, uint32_t g, uint32_t m, uint32_t *lrec, int lane) {
    const uint32_t j = (uint32_t)(lane & 15), r = (uint32_t)(lane >> 4);
    uint32_t v[LOADS];
#pragma unroll
    for (uint32_t k = 0; k < LOADS; k++) {
        const uint32_t idx = (uint32_t)__shfl(my, (int)min(g + 4 * k + r, m - 1));
        v[k] = reinterpret_cast<const uint32_t *>(recs + idx)[min(j, FE_HEAD_WORDS - 1)];
    }
    if (j < FE_HEAD_WORDS) {
#pragma unroll
        for (uint32_t k = 0; k < LOADS; k++) lrec[(4 * k + r) * STRIDE + j] = v[k];
    }
}


constexpr uint32_t FE_NEAR_BEHIND = 8;     // a behind-bit row of at most this many words (256 parts) is staged in LDS with its sprite's record
constexpr uint32_t FE_PART_ROUND = 32, FE_SPRITE_ROUND = 16;                  // records per staging round
constexpr uint32_t FE_SPRITE_STRIDE = FE_HEAD_WORDS + FE_NEAR_BEHIND;         // a staged sprite: its head, then its behind-bit row

struct ColumnsLds {                         // per wavefront: 4 992 bytes, so that eight workgroups share a CU
    uint32_t rec[FE_PART_ROUND * FE_HEAD_WORDS];       // the staged heads (sprites: FE_SPRITE_ROUND x FE_SPRITE_STRIDE words)
    uint32_t near_cand[FE_NEAR_RECS * 64];  // FeRecStore
    uint16_t near_part[FE_NEAR_RECS * 64];
};
static_assert(FE_SPRITE_ROUND * FE_SPRITE_STRIDE <= FE_PART_ROUND * FE_HEAD_WORDS, "sprite round fits the staging area");
static_assert(sizeof(ColumnsLds) * (FE_COL_THREADS / 64) * 8 <= 160 * 1024, "eight workgroups per CU");

__global__ __launch_bounds__(FE_COL_THREADS) void dg_fe_columns(FeParams P) {
    __shared__ ColumnsLds lds_all[FE_COL_THREADS / 64];
    const int W = P.k.W;
    const uint32_t groups = (uint32_t)(W + FE_COL_THREADS - 1) / FE_COL_THREADS;
    uint32_t item = blockIdx.x;                                                         // (frame, 256-column group): fe_core.h
    if (P.order) {
        uint32_t at = blockIdx.x;
        if (P.order_cnt) {                                                              // class lists: skip the classes in front of this workgroup
            uint32_t k = 0, i = blockIdx.x;
            for (; k + 1 < FS_ORDER_CLASSES && i >= P.order_cnt[k]; k++) i -= P.order_cnt[k];
            at = k * gridDim.x + i;
        }
        item = P.order[at];
    }
    const int f = (int)(item / groups);
    const int x = (int)((item % groups) * FE_COL_THREADS + threadIdx.x);
    const bool active = x < W;
    const int lane = (int)(threadIdx.x & 63);
    const int wx0 = __builtin_amdgcn_readfirstlane(x - lane);
    if (wx0 >= W) return;                                                               // whole wave past the right edge
    const uint32_t bin = (uint32_t)wx0 / FE_BIN_W;                                      // this wave's column bin
    ColumnsLds &L = lds_all[threadIdx.x >> 6];
    uint32_t *lrec = L.rec;                                                             // this wave's staging area
    const FeRecStore st{L.near_cand, L.near_part};
    const FeFrame ff = P.fframes[f];
    const uint32_t *boff = P.bin_off + (size_t)f * (P.w64 + 1) + bin, *sboff = P.sbin_off + (size_t)f * (P.w64 + 1) + bin;
    const uint32_t b0 = __builtin_amdgcn_readfirstlane(boff[0]), b1 = __builtin_amdgcn_readfirstlane(boff[1]);
    const uint32_t s0 = __builtin_amdgcn_readfirstlane(sboff[0]), s1 = __builtin_amdgcn_readfirstlane(sboff[1]);
    const uint32_t part_base = __builtin_amdgcn_readfirstlane(ff.part_base), sprite_base = __builtin_amdgcn_readfirstlane(ff.sprite_base);
    {
        FeColumn c = fe_column_start(P, f, x);

        // parts in BSP order: the ones whose column range touches the bin (listed by the host / by dg_fs_frame); 64 list entries per
        // load, 32 heads per staging round, then one part after the other with its head broadcast into scalar registers
        {
            const uint16_t *list = P.bin_parts + ff.bin_base + b0;
            const FePart *recs = P.parts + part_base;
            const uint32_t n = b1 - b0;
            bool open = true;                                                           // some column of the walk is not horizontally occluded yet
            for (uint32_t g64 = 0; open && g64 < n; g64 += 64) {
                const uint32_t m = min(64u, n - g64);
                const int my = (uint32_t)lane < m ? (int)list[g64 + (uint32_t)lane] : 0;
                for (uint32_t g = 0; open && g < m; g += FE_PART_ROUND) {
                    stage_heads<FePart, FE_HEAD_WORDS, FE_PART_ROUND / 4>(recs, my, g, m, lrec, lane);
                    const uint32_t nh = min(FE_PART_ROUND, m - g);
                    for (uint32_t h = 0; open && h < nh; h++) {
                        const uint32_t pi = (uint32_t)__builtin_amdgcn_readlane(my, (int)(g + h));
                        const FePart p = unpack_head<FePart>(lrec[h * FE_HEAD_WORDS + min((uint32_t)(lane & 15), FE_HEAD_WORDS - 1)]);
                        uint32_t ev = 0;
                        const bool walked = active && x >= p.sx && x <= p.ex;
                        if (walked) ev = fe_part_column(P, f, p, pi, c, st);
                        if (p.sky_slot >= 0) {                                          // wave-uniform: all 64 lanes reach the ballots
                            const uint64_t bf = __ballot((ev & FE_EV_FADD) != 0), bc = __ballot((ev & FE_EV_CADD) != 0), bl = __ballot(walked && !(ev & FE_EV_FLUSH));
                            if (lane == 0) {
                                uint64_t *wf = fe_event_words(P, f, p.sky_slot, 0) + bin, *wc = fe_event_words(P, f, p.sky_slot, 1) + bin, *wl = fe_event_words(P, f, p.sky_slot, 2) + bin;
                                *wf = bf; *wc = bc; *wl = bl;
                            }
                        }
                        // Once every column of the walk is horizontally occluded nothing behind can draw, clip or add a visplane entry
                        // (segs.rs:211,337-341): the rest of the bin only yields flush events, which is what the event words are preset to.
                        open = __ballot(active && !c.hor) != 0;
                    }
                }
            }
        }
        // then the sprites (their clip arrays need the finished wall-record columns of this screen column).  The behind-bit rows of the
        // staged sprites travel with their heads (row of sprite i = words [i * behind_words, ..) of the frame's array, fe_dev.h), so that
        // clipping a sprite column is LDS reads only.
        {
            const uint16_t *list = P.sbin_sprites + ff.sbin_base + s0;
            const FeSprite *recs = P.sprites + sprite_base;
            const uint32_t n = s1 - s0, bw = __builtin_amdgcn_readfirstlane(ff.behind_words);
            const bool near_rows = bw <= FE_NEAR_BEHIND;
            for (uint32_t g64 = 0; g64 < n; g64 += 64) {
                const uint32_t m = min(64u, n - g64);
                const int my = (uint32_t)lane < m ? (int)list[g64 + (uint32_t)lane] : 0;
                for (uint32_t g = 0; g < m; g += FE_SPRITE_ROUND) {
                    uint32_t r0 = 0, r1 = 0;                                            // lane (h, j) = (lane >> 2, lane & 3): words 2j, 2j + 1 of sprite h's row
                    if (near_rows) {
                        const uint32_t sidx = (uint32_t)__shfl(my, (int)min(g + (uint32_t)(lane >> 2), m - 1));
                        const uint32_t *brow = P.behind + ff.behind_base + (size_t)sidx * bw;
                        const uint32_t w = 2u * (uint32_t)(lane & 3);
                        if (w < bw) r0 = brow[w];
                        if (w + 1 < bw) r1 = brow[w + 1];
                    }
                    stage_heads<FeSprite, FE_SPRITE_STRIDE, FE_SPRITE_ROUND / 4>(recs, my, g, m, lrec, lane);
                    if (near_rows) {
                        uint32_t *to = lrec + (uint32_t)(lane >> 2) * FE_SPRITE_STRIDE + FE_HEAD_WORDS + 2u * (uint32_t)(lane & 3);
                        to[0] = r0; to[1] = r1;
                    }
                    const uint32_t nh = min(FE_SPRITE_ROUND, m - g);
                    for (uint32_t h = 0; h < nh; h++) {
                        const uint32_t si = (uint32_t)__builtin_amdgcn_readlane(my, (int)(g + h));
                        const FeSprite s = unpack_head<FeSprite>(lrec[h * FE_SPRITE_STRIDE + min((uint32_t)(lane & 15), FE_HEAD_WORDS - 1)]);
                        if (!(active && x >= s.x0 && x < s.x1)) continue;
                        if (near_rows) {
                            const uint32_t *brow = lrec + h * FE_SPRITE_STRIDE + FE_HEAD_WORDS;
                            fe_sprite_column(P, f, ff, s, si, c, st, [brow](uint32_t w) { return brow[w]; });
                        } else {
                            fe_sprite_column(P, f, ff, s, si, c, st, fe_behind_row(P, ff, s));
                        }
                    }
                }
            }
        }
        if (active) P.cnt[(size_t)f * (size_t)W + (size_t)x] = c.nsp;
        if (active && c.ovf) atomicOr(&P.flags[f], c.ovf);
    }
}

// One wave per (frame, sky part): the zero-filled entries of sky visplanes draw one sky pixel at row 0
// (visplanes.rs:61-80 with top = bottom = 0).  (Folding this into dg_fe_scan was measured: 48 us instead of 10 + 5 us,
// a frame's sky parts then queue behind each other in one workgroup.)
__global__ __launch_bounds__(64) void dg_fe_gaps(FeParams P) {
    const int f = blockIdx.y;
    const FeFrame ff = P.fframes[f];
    const int W = P.k.W;
    const int lane = (int)threadIdx.x;
    // (the grid's x extent is the batch's largest sky slot count when the host knows it; the device seg walk only knows a bound, and
    // launches FE_GAP_WAVES waves per frame that stride over the frame's slots)
    for (uint32_t si = blockIdx.x; si < ff.n_sky_slots; si += gridDim.x) {
    const uint32_t pi = P.sky_parts[ff.sky_base + si];
    const FePart &p = P.parts[ff.part_base + pi];
    const int sx = p.sx, ex = p.ex;
    const uint32_t fl = p.flags;
    uint32_t *cnt = P.cnt + (size_t)f * (size_t)W;
    for (int kind = 0; kind < 2; kind++) {
        if (!(fl & (kind ? FEP_CEIL_SKY : FEP_FLOOR_SKY))) continue;
        const uint64_t *add = fe_event_words(P, f, (int32_t)si, kind), *open = fe_event_words(P, f, (int32_t)si, 2);
        for (int x = sx + lane; x <= ex; x += 64) {
            if (!fe_gap(add, open, x, sx, ex)) continue;
            const uint32_t slot = atomicAdd(&cnt[x], 1u);
            if (slot >= P.col_slots) { atomicOr(&P.flags[f], (uint32_t)FE_OVF_SPANS); continue; }
            P.cspans[((size_t)f * P.col_slots + slot) * (size_t)W + (size_t)x] =
                FeU4{FE_KEY_PLANE | (pi << 2) | (uint32_t)kind, 0u, 0u, (uint32_t)SPAN_SKY << FES_KIND_SHIFT};
        }
    }
    }
}

// One workgroup per frame: exclusive scan of the per-column span counts -> col_off.
__global__ __launch_bounds__(FE_SCAN_THREADS) void dg_fe_scan(FeParams P) {
    __shared__ uint32_t wave_sum[FE_SCAN_THREADS / 64];
    __shared__ uint32_t total_s;
    const int f = blockIdx.x;
    const int W = P.k.W;
    const int tid = (int)threadIdx.x;
    const uint32_t *cnt = P.cnt + (size_t)f * (size_t)W;

    // each thread owns a contiguous chunk of columns
    const int chunk = (W + FE_SCAN_THREADS - 1) / FE_SCAN_THREADS;
    const int xa = tid * chunk, xb = min(W, xa + chunk);
    uint32_t mine = 0;
    for (int x = xa; x < xb; x++) mine += min(cnt[x], P.col_slots);
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += up;
    }
    if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int w = 0; w < FE_SCAN_THREADS / 64; w++) { const uint32_t v = wave_sum[w]; wave_sum[w] = run; run += v; }
        total_s = run;
    }
    __syncthreads();
    const uint32_t total = total_s;
    const bool fits = total <= P.span_stride;      // a frame that does not fit draws nothing and is redone on the host
    uint32_t off = wave_sum[tid >> 6] + incl - mine;
    uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    if (tid == 0) {
        coff[W] = fits ? total : 0u;
        P.totals[f] = total;
        // the frame's overflow word goes to the host here, with a plain store: dg_fe_columns and dg_fe_gaps have finished (same stream)
        P.host_flags[f] = P.flags[f] | (fits ? 0u : (uint32_t)FE_OVF_FRAME);
        P.flags[f] = 0u;
    }
    for (int x = xa; x < xb; x++) {
        coff[x] = fits ? off : 0u;
        off += min(cnt[x], P.col_slots);
    }
    // The walk's per-batch state goes back to zero here, after its last reader: the frame's flag word (above), its event words (read by
    // dg_fe_gaps) and the launch-order counters (read by dg_fe_columns) — the next batch of this slot then needs no fill kernel in front of
    // it (5 us plus a kernel boundary per batch; context.cpp clears everything once, and again after a failed enqueue).
    // (only the rows of the frame's own sky slots can hold bits: dg_fe_columns writes a part's row at its sky_slot < n_sky_slots; the rows of
    // the slots up to the batch's bound stay as they were cleared — a third of the kernel's time when the bound is the seg walk's 64)
    const size_t ev_words = (size_t)min(P.fframes[f].n_sky_slots, P.max_sky_slots) * (size_t)P.w64;
    for (int kind = 0; kind < 3; kind++) {
        uint64_t *ev = fe_event_words(P, f, 0, kind);
        for (size_t i = (size_t)tid; i < ev_words; i += FE_SCAN_THREADS) ev[i] = 0ull;
    }
    if (f == 0 && P.order_cnt && tid < (int)FS_ORDER_CLASSES) const_cast<uint32_t *>(P.order_cnt)[tid] = 0u;
}

// Every column's spans into draw order and into the raster kernel's form: rank by key among the column's spans
// (distinct keys), resolve the texture-mapping constants (what dg_setup_spans does for host-built lists), scatter into the
// column-major list.  64 adjacent columns x FE_SCATTER_GROUPS slot groups per workgroup: the lanes of a wave read the
// same slot of 64 adjacent columns (coalesced in the [slot][column] scratch layout); the keys are staged in LDS once.
//
// The spans of the workgroup's 64 columns are ONE contiguous range of the output, but a lane's record lands n_x * 32 bytes from its
// neighbour's: stored straight from the lanes, every record is its own pair of partial-line writes (9 M of the 14.7 M requests the L2
// sees per 1 000 frames, profiles/r03_column_walk.md).  So the records are assembled in LDS at their final position and the workgroup
// then streams the range out, 16 consecutive bytes per lane — full lines.  A range of more than FE_SCATTER_STAGE records is stored directly.
constexpr uint32_t FE_SCATTER_STAGE = 384;

__global__ __launch_bounds__(64 * FE_SCATTER_GROUPS) void dg_fe_scatter(FeParams P) {
    extern __shared__ uint32_t lds_dyn[];     // FE_SCATTER_STAGE records (12 KB), then the keys [col_slots][64]: sized by the ctx's slot count at
                                              // launch (12 KB at the default 48 slots), so that LDS does not cap the resident workgroups
    FeU4 *lout = reinterpret_cast<FeU4 *>(lds_dyn);
    uint32_t *lkeys = lds_dyn + FE_SCATTER_STAGE * 8;
    const int f = blockIdx.y;
    const int W = P.k.W;
    const int lx = (int)(threadIdx.x & 63);
    const int x0 = (int)(blockIdx.x * 64), x = x0 + lx;
    const uint32_t g = threadIdx.x >> 6;
    const uint32_t *coff = P.col_off + (size_t)f * (size_t)(W + 1);
    uint32_t off = 0, n = 0;
    if (x < W) { off = coff[x]; n = coff[x + 1] - off; }      // n = 0 for every column of a frame that did not fit
    const uint32_t t_first = coff[x0], t_last = coff[min(x0 + 64, W)];            // wave-uniform: the workgroup's range of the output
    const bool staged = t_last - t_first <= FE_SCATTER_STAGE;
    const FeU4 *src = P.cspans + (size_t)f * P.col_slots * (size_t)W + (size_t)x;
    for (uint32_t i = g; i < n; i += FE_SCATTER_GROUPS) lkeys[i * 64 + (uint32_t)lx] = src[(size_t)i * (size_t)W].x;
    __syncthreads();
    const DevFrame fr = P.frames[f];
    const FeFrame ff = P.fframes[f];
    FeU4 *out = reinterpret_cast<FeU4 *>(P.rspans + fr.span_base);
    for (uint32_t i = g; i < n; i += FE_SCATTER_GROUPS) {
        const FeU4 cs = src[(size_t)i * (size_t)W];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) {
            const uint32_t kj = lkeys[j * 64 + (uint32_t)lx];
            rank += (kj < cs.x || (kj == cs.x && j < i)) ? 1u : 0u;        // the tie-break keeps the scatter a permutation
        }
        const DevRSpan r = fe_resolve(P, fr, ff, x, cs);
        FeU4 *to = staged ? lout + 2 * (size_t)(off - t_first + rank) : out + 2 * (size_t)(off + rank);
        to[0] = FeU4{r.w[0], r.w[1], r.w[2], r.w[3]};
        to[1] = FeU4{r.w[4], r.w[5], r.w[6], r.w[7]};
    }
    if (!staged) return;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * (t_last - t_first); i += 64 * FE_SCATTER_GROUPS) out[2 * (size_t)t_first + i] = lout[i];
}

}  // namespace

hipError_t launch_fe(const FeParams &P, hipStream_t stream, hipEvent_t start, hipEvent_t stop) {
    if (P.n_frames <= 0) {
        hipError_t e = hipSuccess;
        if (start) e = hipEventRecord(start, stream);
        if (e == hipSuccess && stop) e = hipEventRecord(stop, stream);
        return e;
    }
    dim3 grid((unsigned)((P.k.W + FE_COL_THREADS - 1) / FE_COL_THREADS) * (unsigned)P.n_frames);
    hipExtLaunchKernelGGL(dg_fe_columns, grid, dim3(FE_COL_THREADS), 0, stream, start, nullptr, 0, P);
    if (P.max_sky_slots) hipLaunchKernelGGL(dg_fe_gaps, dim3(P.gap_waves ? std::min(P.gap_waves, P.max_sky_slots) : P.max_sky_slots, (unsigned)P.n_frames), dim3(64), 0, stream, P);
    hipLaunchKernelGGL(dg_fe_scan, dim3((unsigned)P.n_frames), dim3(FE_SCAN_THREADS), 0, stream, P);
    hipExtLaunchKernelGGL(dg_fe_scatter, dim3((unsigned)((P.k.W + 63) / 64), (unsigned)P.n_frames), dim3(64 * FE_SCATTER_GROUPS),
                          (uint32_t)((size_t)FE_SCATTER_STAGE * 32 + (size_t)P.col_slots * 64 * 4), stream, nullptr, stop, 0, P);
    return hipGetLastError();
}

}  // namespace dg
