// fs_frame.h — the device seg walk (DG_FE_DEVICE_SEGS): the ORDER-DEPENDENT half of the per-seg front end, one wavefront per frame.
//
// The reference walks the BSP front to back (src/renderer/mod.rs:61-104) and every step of it is cheap, branchy and order-dependent
// only through three things: the order itself, which earlier full-height walls hide a later part (the host walker's solid-column
// intervals, frontend.cpp), and the sprite / masked-wall draw sequence (renderer/map_objects.rs:216-240).  The GPU does it in two
// kernels per batch (fs_kernels.hip), with the arithmetic of fs_core.h:
//   dg_fs_segs    one lane per (frame, seg): process_seg + the tests of every process_sidedef call (transform, clip, x projection): the
//                 columns and flags of each call that reaches its column loop, written at the seg's VISIT POSITION in the frame's
//                 candidate row, with the entry's bit set in the frame's occupancy row — the live entries of the row are then the
//                 frame's candidate parts in the reference's order.  The position of a seg is that of its BSP leaf's first seg
//                 (fs_leaf_base: the sum, over the leaf's ancestors on whose BACK side it lies for this viewer, of the seg count of the
//                 ancestor's front subtree) plus its index in the leaf;
//   dg_fs_frame   one workgroup per frame, the phases below: hidden-part culling of the candidates -> the frame's FePart list (built
//                 here, for the survivors only: a tenth of the candidates);
//                 the map objects (FeSprite), their behind-bit rows and draw sequence; the column bins; the FeFrame header —
//                 exactly the arrays the host ships in DG_FE_DEVICE mode, which the column walk (fe_kernels.hip) then consumes unchanged.
// A frame the reference would panic on, or one that exceeds a capacity below, is flagged (FE_OVF_SEGS) and redone by the host.
//
// The phases are plain functions of (lane, shared state): inside a phase the 64 lanes touch disjoint data (or use atomics), between
// phases stands a barrier.  tests/emul runs them on the CPU — lanes one after another — against the host walker's records.
#pragma once
#include "../../include/doomgpu.h"
#include "fe_core.h"
#include "fs_core.h"

#if !defined(__HIPCC__)
struct uint2 { uint32_t x, y; };
#endif

namespace dg {

constexpr uint32_t FE_OVF_SEGS = 8;            // the device seg walk gave the frame up (reference panic / capacity): redone on the host
constexpr int FS_LANES = 256;                  // threads of dg_fs_frame's workgroup (four wavefronts per frame)
constexpr int FS_CALLS = 5;                    // process_sidedef calls a seg can make (segs.rs:493-588)
constexpr int FS_BLOCK = 16;                   // lanes per block of the two-level prefix sums
constexpr uint32_t FS_PART_CAP = 256;          // parts of one frame after the hidden-part culling
constexpr uint32_t FS_CL_CAP = 1536;           // candidate parts of one frame the culling stages in shared memory (more: the frame's rows in global memory, FsParams::cl_rows)
constexpr int FS_MAX_W = 2560;                 // widest frame the culling's per-column table holds (wider: DG_FE_DEVICE)
constexpr int FS_GROUP = 16;                   // lanes that share one candidate in the column passes of the culling
constexpr uint32_t FS_SPRITE_CAP = 512;        // visible map objects of one frame
constexpr uint32_t FS_SKY_CAP = 64;            // parts of one frame that may produce sky visplanes (event rows of dg_fe_gaps)
constexpr uint32_t FS_BIN_CAP = 4096;          // (part, column bin) pairs of one frame
constexpr uint32_t FS_SBIN_CAP = 4096;         // (sprite, column bin) pairs
constexpr uint32_t FS_BEHIND_WORDS = FS_PART_CAP / 32;

// One ancestor of one leaf: the partition line, and the seg count of the subtree on the OTHER side of it (visited before the leaf when the
// viewer stands on that side) | (the leaf lies in the LEFT subtree) << 31
struct FsAnc { float x, y, dx, dy; uint32_t front; };
static_assert(sizeof(FsAnc) == 20, "FsAnc layout");


struct FsParams {
    DevConsts k;
    // scene (immutable per upload)
    const FsSeg *segs; const uint16_t *seg_leaf; const uint32_t *leaf_first;
    const FsSector *sectors; const FsAnim *anims; const FsBitmap *bitmaps; const uint8_t *flat_sky;
    const FsMobj *mobjs; const FsSpriteFrame *sframes;
    const uint32_t *anc_off; const FsAnc *anc;                                   // per leaf: its ancestors, root first (entries [anc_off[leaf], anc_off[leaf + 1]))
    uint32_t n_segs, n_leaves, n_mobjs;
    // per-frame strides of the sprite arrays below: the scene's map-object count (no frame can show more), rounded up, at most FS_SPRITE_CAP /
    // FS_SBIN_CAP — a map with 40 things keeps its records 64 apart, not 512
    uint32_t sprite_stride, sbin_stride;
    // game state of this batch (scene-wide values as of submission)
    const int16_t *sector_light;               // [n_sectors] for the whole batch (light_stride 0), or [n_frames][light_stride]: per-view game state
    const int32_t *mobj_state;                 // [n_mobjs] / [n_frames][mstate_stride]: sprite_frame * 2 + full_bright, negative: S_NULL
    uint32_t light_stride, mstate_stride;      // elements per frame of the two arrays above; 0: one array for every frame (lights.rs:47-259, map_objects.rs:63-121)
    // per frame
    const dg_view *views;                      // [n_frames], trig filled
    int32_t n_frames;
    // scratch (shared by all slots: every kernel runs on the ctx's one stream)
    // [frame][visit position of the seg x FS_CALLS + call]: x = sx | ex << 16, y = FEP_* (bits 0-7) | 1 << 8 | seg << 12
    // for a call that reaches its column loop (dg_fs_segs): the frame's candidate parts, already in the reference's visit order.  Which
    // entries of the row hold a candidate of THIS batch says the frame's occupancy row (one bit per entry): the row itself is never cleared
    uint2 *lite;
    uint32_t *occ;                             // [frame][fs_occ_words(n_segs)], zero between batches (dg_fs_frame clears what dg_fs_segs set): bit e = entry e of the frame's row
    // a frame with more candidate parts than dg_fs_frame stages in shared memory (FS_CL_CAP) keeps its candidate list and keep bits here:
    // [frame][cl_row_cap] words + [frame][cl_row_cap / 32] words (cl_row_cap: a multiple of 32, = n_segs x FS_CALLS rounded up; 0: none)
    uint32_t *cl_rows, *keep_rows;
    uint32_t cl_row_cap;
    uint32_t *flags;                           // [frame] FE_OVF_* (the column walk's flag words)
    // outputs: the DG_FE_DEVICE record arrays with fixed per-frame strides
    FeFrame *fframes; FePart *parts; FeSprite *sprites; uint32_t *behind; uint32_t *sky_parts;
    uint32_t *bin_off; uint16_t *bin_parts; uint32_t *sbin_off; uint16_t *sbin_sprites;
    // launch order of dg_fe_columns (fe_core.h FeParams::order), built here because only the GPU knows the bins: every (frame, 256-column
    // group) is appended to the list of its weight class, heaviest class first; order_cnt zeroed per batch; nullptr: no lists wanted
    uint32_t *order_cnt;                       // [FS_ORDER_CLASSES]
    uint32_t *order_list;                      // [FS_ORDER_CLASSES][n_items]
    uint32_t n_items;                          // n_frames x ceil(W / 256)
};
constexpr uint32_t FS_ORDER_CLASSES = 4;       // longest bin of the group (parts + 2 x sprites): > 32, > 16, > 8, the rest

// ---- the visit position of a leaf (mod.rs:61-104 walks front to back from the root: the viewer's side of every partition first): the
// number of segs visited before the leaf's first one = the seg counts of the FRONT subtrees of the ancestors on whose back side the leaf
// lies.  Ten to fourteen ancestors; asked only for the segs that turn out to be candidates (dg_fs_segs), a few hundred per frame — a
// kernel of its own over every (frame, leaf) was a launch, 10 us per 1 000 frames and a scratch row per frame for the same sums.
DG_HD uint32_t fs_leaf_base(const FsParams &P, int f, uint32_t leaf) {
    const dg_view &v = P.views[f];
    const V2 ppos{v.x, v.y};
    uint32_t base = 0;
    // four ancestors per round, fetched together: a loop of one ancestor per turn pays a memory latency per ancestor, and a large map has
    // thirty of them above a leaf.  (The list carries each ancestor's line and count itself — a list of node indices costs a second,
    // dependent fetch per ancestor.)
    const uint32_t i1 = P.anc_off[leaf + 1];
    for (uint32_t i = P.anc_off[leaf]; i < i1; i += 4) {
        FsAnc a[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t j = 0; j < 4; j++) a[j] = P.anc[i + j < i1 ? i + j : i1 - 1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t j = 0; j < 4; j++) {
            if (i + j >= i1) continue;
            const bool is_left = left_of(ppos, Seg2{V2{a[j].x, a[j].y}, V2{a[j].x + a[j].dx, a[j].y + a[j].dy}});   // mod.rs:70-77: the viewer's side is visited first
            const bool leaf_left = (a[j].front >> 31) != 0;
            if (leaf_left != is_left) base += a[j].front & 0x7fffffffu;                            // the whole front subtree comes before this leaf
        }
    }
    return base;
}

// ---- dg_fs_segs: one (frame, seg) ----------------------------------------------------------------------------------------------------
DG_HD void fs_flag(const FsParams &P, int f, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(&P.flags[f], bits);
#else
    P.flags[f] |= bits;
#endif
}
DG_HD void fs_or_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
// The occupancy row of a frame: lane l of dg_fs_frame owns words [l * wpl, (l + 1) * wpl) of it — entries [l * wpl * 32, ..) of the candidate row
DG_HD uint32_t fs_occ_wpl(uint32_t n_segs) { return ((n_segs * FS_CALLS + 31u) / 32u + FS_LANES - 1) / FS_LANES; }
DG_HD uint32_t fs_occ_words(uint32_t n_segs) { return fs_occ_wpl(n_segs) * FS_LANES; }
DG_HD uint32_t fs_ctz(uint32_t v) { return (uint32_t)__builtin_ctz(v); }           // (v != 0)
DG_HD uint32_t fs_popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
// body(i, entry) for every occupied entry i of the lane's slice, in row order; the entries of up to eight set bits of a word are fetched
// together (a seg's calls sit next to each other in the row)
template <typename Body> DG_HD void fs_for_occupied(const uint32_t *occ, const uint2 *row, uint32_t wpl, int lane, Body body) {
    for (uint32_t w = (uint32_t)lane * wpl; w < ((uint32_t)lane + 1u) * wpl; w++) {
        uint32_t bits = occ[w];
        while (bits) {
            uint32_t e[8];
            uint2 q[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t j = 0; j < 8; j++) {
                e[j] = bits ? w * 32u + fs_ctz(bits) : 0xffffffffu;
                bits &= bits - 1u;                                            // (0 stays 0)
            }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t j = 0; j < 8; j++) q[j] = row[e[j] != 0xffffffffu ? e[j] : e[0]];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (uint32_t j = 0; j < 8; j++)
                if (e[j] != 0xffffffffu) body(e[j], q[j]);
        }
    }
}
// The whole 48-byte record in one fetch: field by field the compiler fetches what each step needs when it needs it, a chain of memory
// latencies in front of the clip test that turns most segs away.
DG_HD FsSeg fs_load_seg(const FsSeg *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    union U { uint4 w[3]; FsSeg s; __device__ U() {} } u;
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    u.w[0] = q[0]; u.w[1] = q[1]; u.w[2] = q[2];
    return u.s;
#else
    return *p;
#endif
}
static_assert(sizeof(FsSeg) == 48, "fs_load_seg");
DG_HD void fs_seg_lane(const FsParams &P, int f, uint32_t si) {
    const dg_view &v = P.views[f];
    const uint32_t leaf = P.seg_leaf[si];
    const FsSeg sg = fs_load_seg(P.segs + si);
    FsSegOut so;
    // (the sector's light level is not read here: nothing below looks at it, fs_ph_emit fetches it for the parts that are kept)
    const int32_t st = fs_seg(P.k, sg, P.sectors, P.anims, V2{v.x, v.y}, v.cos_na, v.sin_na, v.floor_height + 41.0f, v.timestamp, (int16_t)0, so);
    if (st == FS_SKIP || leaf == 0xffffu) return;                             // (a seg of no reachable leaf is never visited; asked here, after the
                                                                              // clip, so that the seg's record is not fetched behind its leaf's)
    if (st != FS_OK) { fs_flag(P, f, FE_OVF_SEGS); return; }
    // Only what the hidden-part culling reads is computed here — the columns and the flags of every call; the finished FePart of the
    // few calls that survive it (a tenth of them) is built afterwards, by dg_fs_frame, from (seg, call).
    const uint32_t pos = fs_leaf_base(P, f, leaf) + (si - P.leaf_first[leaf]);
    uint2 *lite = P.lite + ((size_t)f * P.n_segs + pos) * FS_CALLS;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (!((so.call_mask >> i) & 1u)) continue;
        int32_t sx, ex;
        uint32_t flags;
        const int32_t ps = fs_part_head(P.k, so, so.call[i], P.bitmaps, P.flat_sky, sx, ex, flags);
        if (ps == FS_SKIP) continue;                                          // nothing (a zero-width part)
        if (ps != FS_OK) { fs_flag(P, f, FE_OVF_SEGS); continue; }
        lite[i] = uint2{(uint32_t)sx | ((uint32_t)ex << 16), flags | 0x100u | (si << 12)};
        const uint32_t e = pos * FS_CALLS + (uint32_t)i;
        fs_or_u32(&P.occ[(size_t)f * fs_occ_words(P.n_segs) + (e >> 5)], 1u << (e & 31u));
    }
}

// ---- dg_fs_frame: one wavefront per frame --------------------------------------------------------------------------------------------
struct FsShared {                              // LDS on the GPU
    uint32_t lane_cnt[FS_LANES], block_sum[FS_LANES / FS_BLOCK];
    uint32_t first[FS_MAX_W];                  // per screen column: visit index of the first full-height solid candidate that spans it
    uint32_t cl[FS_CL_CAP];                    // the frame's candidate parts in visit order: sx | ex << 12 | FEP_* << 24 (more than FS_CL_CAP: FsParams::cl_rows)
    uint32_t keepw[FS_CL_CAP / 32];            // bit k: candidate k survives the hidden-part culling
    uint32_t cl_big;                           // the frame's list lives in global memory
    uint32_t lane_k0[FS_LANES];                // index in cl[] of the first candidate of the lane's slice of the lite row
    uint32_t n_cl;
    // kept parts
    uint32_t kept_src[FS_PART_CAP];            // seg << 3 | call
    int16_t kept_sky[FS_PART_CAP];
    uint16_t kept_sx[FS_PART_CAP], kept_ex[FS_PART_CAP];
    uint16_t kept_t[FS_PART_CAP];              // two-sided middle parts: order index of the first sprite they are drawn behind (0xffff: none; 0xfffe: not two-sided)
    float kline[FS_PART_CAP][4];
    uint32_t n_parts, n_sky;
    // sprites
    float s_centre[FS_SPRITE_CAP][2], s_mid[FS_SPRITE_CAP][2];
    int16_t s_key[FS_SPRITE_CAP];
    uint16_t s_order[FS_SPRITE_CAP], s_x0b[FS_SPRITE_CAP], s_x1b[FS_SPRITE_CAP];   // place in the far-to-near order; first / last column bin (x0b > x1b: no columns)
    uint32_t n_sprites;
    // `fail`: the frame is given up.  Invariant: it only ever goes from 0 to 1; a phase in which several lanes may raise it does so with an
    // atomic OR, a phase in which one lane decides stores it; lanes that read it in the SAME phase may see either value, which is harmless by
    // construction (everything a phase writes after such a read is guarded by a capacity test of its own, and the frame is flagged for the
    // host either way); from the next barrier on every lane sees it.
    uint32_t fail;
    uint8_t group_cls[FS_MAX_W / 256 + 6];     // weight class of each 256-column group (fs_ph_bin_prefix)
};

// phase 0 (lane 0): reset
DG_HD void fs_ph_init(FsShared &S) { S.n_cl = 0; S.n_parts = 0; S.n_sky = 0; S.n_sprites = 0; S.fail = 0; S.cl_big = 0; }
// The frame's candidate list / keep bits: shared memory, or the frame's rows in global memory when there are more than FS_CL_CAP (a map of
// doom2's scale seen down its long axis: ten thousand).  The same code walks both (a generic pointer); the phases are separated by
// workgroup barriers, which order the global accesses of a workgroup as well.
DG_HD uint32_t *fs_cl(const FsParams &P, FsShared &S, int f) { return S.cl_big ? P.cl_rows + (size_t)f * P.cl_row_cap : S.cl; }
DG_HD uint32_t *fs_keepw(const FsParams &P, FsShared &S, int f) { return S.cl_big ? P.keep_rows + (size_t)f * (P.cl_row_cap / 32) : S.keepw; }

// Two-level exclusive prefix over lane_cnt[]: a phase in which the first FS_LANES / FS_BLOCK lanes sum their block, then any lane adds
// the blocks before its own and the lanes before it in its block (at most 2 FS_BLOCK reads instead of FS_LANES).
DG_HD void fs_ph_block_sums(FsShared &S, int lane) {
    if (lane >= FS_LANES / FS_BLOCK) return;
    uint32_t sum = 0;
    for (int i = 0; i < FS_BLOCK; i++) sum += S.lane_cnt[lane * FS_BLOCK + i];
    S.block_sum[lane] = sum;
}
DG_HD uint32_t fs_lane_offset(const FsShared &S, int lane) {
    uint32_t at = 0;
    for (int b = 0; b < lane / FS_BLOCK; b++) at += S.block_sum[b];
    for (int l = lane / FS_BLOCK * FS_BLOCK; l < lane; l++) at += S.lane_cnt[l];
    return at;
}

// phases 1a / 1b: the candidates of the frame, in visit order, into cl[].  Lane l owns the slice [l * per, (l + 1) * per) of the frame's
// candidate row, found through its words of the occupancy row (the order of the live entries is the reference's visit order).
DG_HD void fs_ph_cand_count(const FsParams &P, FsShared &S, int f, int lane) {
    const uint32_t wpl = fs_occ_wpl(P.n_segs);
    const uint32_t *occ = P.occ + (size_t)f * wpl * FS_LANES + (size_t)lane * wpl;
    uint32_t c = 0;
    for (uint32_t w = 0; w < wpl; w++) c += fs_popc(occ[w]);
    S.lane_cnt[lane] = c;
}
DG_HD void fs_ph_cand_stage(const FsParams &P, FsShared &S, int f, int lane) {
    const uint32_t n = P.n_segs * FS_CALLS;
    uint32_t at = fs_lane_offset(S, lane);
    S.lane_k0[lane] = at;
    uint32_t total = 0;                                               // (every lane: where the list goes is decided before anything is written)
    for (int b = 0; b < FS_LANES / FS_BLOCK; b++) total += S.block_sum[b];
    const bool big = total > FS_CL_CAP, fits = !big || total <= P.cl_row_cap;
    if (lane == FS_LANES - 1) {
        S.n_cl = fits ? total : 0u;
        S.cl_big = big && fits;
        if (!fits) S.fail = 1;
    }
    if (!fits) return;
    uint32_t *cl = big ? P.cl_rows + (size_t)f * P.cl_row_cap : S.cl;
    uint32_t *keepw = big ? P.keep_rows + (size_t)f * (P.cl_row_cap / 32) : S.keepw;
    for (uint32_t w = (uint32_t)lane; w < (big ? (total + 31u) / 32u : FS_CL_CAP / 32u); w += FS_LANES) keepw[w] = 0;
    if (S.lane_cnt[lane] == 0u) return;
    fs_for_occupied(P.occ + (size_t)f * fs_occ_words(P.n_segs), P.lite + (size_t)f * n, fs_occ_wpl(P.n_segs), lane, [&](uint32_t, const uint2 q) {
        cl[at] = (q.x & 0xfffu) | ((q.x >> 16) << 12) | (q.y << 24);
        at++;
    });
}
// phase 2c: the column table starts empty
DG_HD void fs_ph_first_clear(const FsParams &P, FsShared &S, int lane) {
    for (int c = lane; c < P.k.W; c += FS_LANES) S.first[c] = 0xffffffffu;
}
DG_HD uint32_t fs_add_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    const uint32_t old = *p;
    *p = old + v;
    return old;
#endif
}
DG_HD void fs_min_u32(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
// phases 2d / 2e: FS_GROUP lanes share a candidate and stride over its columns; the groups take the candidates round robin, four of a group's
// candidates fetched together (a frame's list may live in global memory: one candidate per turn is one memory latency per candidate).
template <typename Body> DG_HD void fs_for_group_candidates(const uint32_t *cl, uint32_t n_cl, uint32_t g, Body body) {
    constexpr uint32_t STEP = FS_LANES / FS_GROUP;
    for (uint32_t k = g; k < n_cl; k += 4 * STEP) {
        uint32_t q[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t j = 0; j < 4; j++) q[j] = cl[k + j * STEP < n_cl ? k + j * STEP : k];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t j = 0; j < 4; j++)
            if (k + j * STEP < n_cl) body(k + j * STEP, q[j]);
    }
}
DG_HD void fs_ph_solids(const FsParams &P, FsShared &S, int f, int lane) {
    if (S.fail) return;
    const uint32_t g = (uint32_t)lane / FS_GROUP, sub = (uint32_t)lane % FS_GROUP;
    fs_for_group_candidates(fs_cl(P, S, f), S.n_cl, g, [&](uint32_t k, uint32_t q) {
        if (!fs_part_is_solid(q >> 24)) return;
        const uint32_t sx = q & 0xfffu, ex = (q >> 12) & 0xfffu;
        for (uint32_t c = sx + sub; c <= ex; c += FS_GROUP) fs_min_u32(&S.first[c], k);
    });
}
DG_HD void fs_ph_keep(const FsParams &P, FsShared &S, int f, int lane) {
    if (S.fail) return;
    const uint32_t g = (uint32_t)lane / FS_GROUP, sub = (uint32_t)lane % FS_GROUP;
    uint32_t *keepw = fs_keepw(P, S, f);
    fs_for_group_candidates(fs_cl(P, S, f), S.n_cl, g, [&](uint32_t k, uint32_t q) {
        const uint32_t sx = q & 0xfffu, ex = (q >> 12) & 0xfffu;
        bool open = false;
        for (uint32_t c = sx + sub; c <= ex; c += FS_GROUP) open |= S.first[c] >= k;
        if (open) fs_or_u32(&keepw[k >> 5], 1u << (k & 31u));
    });
}
// phases 2f / 2g: the survivors get their place in the frame's part list (and their sky event slot).  Lane l walks its slice of the candidate
// row again (candidate indices from lane_k0); lane_cnt packs (survivors | survivors that want a sky slot << 16).
DG_HD void fs_ph_kept_count(const FsParams &P, FsShared &S, int f, int lane) {
    // (flags of candidate k: cl[k] >> 24)
    uint32_t c = 0;
    if (!S.fail) {
        const uint32_t *cl = fs_cl(P, S, f), *keepw = fs_keepw(P, S, f);
        const uint32_t k1 = lane + 1 < FS_LANES ? S.lane_k0[lane + 1] : S.n_cl;
        for (uint32_t k = S.lane_k0[lane]; k < k1; k++)
            if ((keepw[k >> 5] >> (k & 31u)) & 1u) c += 1u + (fs_part_wants_sky_slot(cl[k] >> 24) ? 0x10000u : 0u);
    }
    S.lane_cnt[lane] = c;
}
DG_HD void fs_ph_kept_place(const FsParams &P, FsShared &S, int f, int lane) {
    const uint32_t n = P.n_segs * FS_CALLS;
    const uint32_t off = fs_lane_offset(S, lane);
    uint32_t o = off & 0xffffu, sky = off >> 16;
    if (lane == FS_LANES - 1) {
        const uint32_t end = off + S.lane_cnt[lane];
        S.n_parts = end & 0xffffu; S.n_sky = end >> 16;
        if (S.n_parts > FS_PART_CAP || S.n_sky > FS_SKY_CAP) { S.fail = 1; S.n_parts = 0; S.n_sky = 0; }
    }
    if (S.fail || S.lane_cnt[lane] == 0u) return;                        // (no survivor in this lane's slice: nothing to look up)
    const uint32_t *keepw = fs_keepw(P, S, f);
    uint32_t k = S.lane_k0[lane];
    fs_for_occupied(P.occ + (size_t)f * fs_occ_words(P.n_segs), P.lite + (size_t)f * n, fs_occ_wpl(P.n_segs), lane, [&](uint32_t i, const uint2 q) {
        const bool kept = (keepw[k >> 5] >> (k & 31u)) & 1u;
        k++;
        if (!kept) return;
        const bool wants = fs_part_wants_sky_slot(q.y & 0xffu);
        if (o < FS_PART_CAP && sky <= FS_SKY_CAP) {
            S.kept_src[o] = ((q.y >> 12) << 3) | (i % FS_CALLS);
            S.kept_sky[o] = wants ? (int16_t)sky : (int16_t)-1;
            S.kept_sx[o] = (uint16_t)(q.x & 0xffffu); S.kept_ex[o] = (uint16_t)(q.x >> 16);
            S.kept_t[o] = (q.y & FEP_TWO_SIDED_MID) ? 0xffffu : 0xfffeu;
        }
        o++; sky += wants;
    });
}
// phase 3: kept part o (lane-strided): process_seg + process_sidedef's head again for its (seg, call) — the finished FePart to its
// place, its clipped line to shared memory (is_behind_vertex), its sky slot recorded
DG_HD void fs_ph_emit(const FsParams &P, FsShared &S, int f, int lane) {
    if (S.fail) return;
    const dg_view &v = P.views[f];
    for (uint32_t o = (uint32_t)lane; o < S.n_parts; o += FS_LANES) {
        const uint32_t src = S.kept_src[o];
        const FsSeg &sg = P.segs[src >> 3];
        FsSegOut so;
        FePart p;
        const int32_t st = fs_seg(P.k, sg, P.sectors, P.anims, V2{v.x, v.y}, v.cos_na, v.sin_na, v.floor_height + 41.0f, v.timestamp, P.sector_light[(size_t)f * P.light_stride + (size_t)sg.front_sector], so);
        if (st != FS_OK || fs_part(P.k, so, fs_call(so, src & 7u), P.bitmaps, P.flat_sky, v.floor_height, p) != FS_OK) { fs_or_u32(&S.fail, 1u); continue; }   // (cannot happen: dg_fs_segs passed it)
        p.sky_slot = S.kept_sky[o];
        P.parts[(size_t)f * FS_PART_CAP + o] = p;
        S.kline[o][0] = so.cl.line.a.x; S.kline[o][1] = so.cl.line.a.y; S.kline[o][2] = so.cl.line.b.x; S.kline[o][3] = so.cl.line.b.y;
        if (S.kept_sky[o] >= 0) P.sky_parts[(size_t)f * FS_SKY_CAP + (uint32_t)S.kept_sky[o]] = o;
    }
}
// phase 4a: lane l looks at map object base + l (draw_map_objects' per-object part); lane_cnt = 1 when it shows
struct FsSpriteTmp { FsSpriteOut so; int32_t status; };
DG_HD void fs_ph_mobj(const FsParams &P, FsShared &S, int f, uint32_t base, int lane, FsSpriteTmp &T) {
    S.lane_cnt[lane] = 0;
    T.status = FS_SKIP;
    const uint32_t mi = base + (uint32_t)lane;
    if (S.fail || mi >= P.n_mobjs) return;
    const int32_t st = P.mobj_state[(size_t)f * P.mstate_stride + mi];
    if (st < 0) return;                                                // S_NULL (renderer/map_objects.rs:37)
    const dg_view &v = P.views[f];
    const FsMobj &m = P.mobjs[mi];
    T.status = fs_mobj(P.k, m, P.sframes[st >> 1], P.bitmaps, P.sectors, V2{v.x, v.y}, v.angle, v.cos_na, v.sin_na, v.floor_height + 41.0f, st & 1,
                       m.sector >= 0 ? P.sector_light[(size_t)f * P.light_stride + (size_t)m.sector] : (int16_t)0, T.so);
    if (T.status == FS_OK) S.lane_cnt[lane] = 1;
    else if (T.status != FS_SKIP) fs_or_u32(&S.fail, 1u);              // a failure (any lane may see one)
}
// phase 4c: the FeSprite records, sprite indices in map-object order (n_before: sprites of the chunks before this one)
DG_HD void fs_ph_mobj_emit(const FsParams &P, FsShared &S, int f, int lane, const FsSpriteTmp &T, uint32_t n_before) {
    const uint32_t si = n_before + fs_lane_offset(S, lane);
    if (lane == FS_LANES - 1) {
        S.n_sprites = si + S.lane_cnt[lane];
        if (S.n_sprites > P.sprite_stride) { S.fail = 1; S.n_sprites = 0; }
    }
    if (S.fail || S.lane_cnt[lane] != 1 || si >= P.sprite_stride) return;
    FeSprite sp = T.so.sp;
    sp.behind_off = si * FS_BEHIND_WORDS;
    P.sprites[(size_t)f * P.sprite_stride + si] = sp;
    S.s_centre[si][0] = T.so.centre.x; S.s_centre[si][1] = T.so.centre.y;
    const Seg2 &l = T.so.line;
    S.s_mid[si][0] = (l.a.x + l.b.x) / 2.0f; S.s_mid[si][1] = (l.a.y + l.b.y) / 2.0f;       // map_objects.rs:222-226
    S.s_key[si] = (int16_t)T.so.sort_key;
    const int nb = (P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    if (sp.x0 < sp.x1) { S.s_x0b[si] = (uint16_t)(sp.x0 / FE_BIN_W); S.s_x1b[si] = (uint16_t)((sp.x1 - 1) / FE_BIN_W); }
    else { S.s_x0b[si] = (uint16_t)nb; S.s_x1b[si] = 0; }
}
// phase 5: the behind-bit rows (which wall records do NOT clip a sprite: map_objects.rs:138-140), one (sprite, word) per step
DG_HD void fs_ph_behind(const FsParams &P, FsShared &S, int f, int lane) {
    if (S.fail) return;
    const uint32_t words = (S.n_parts + 31) / 32;
    for (uint32_t it = (uint32_t)lane; it < S.n_sprites * words; it += FS_LANES) {
        const uint32_t si = it / words, w = it % words;
        const V2 c{S.s_centre[si][0], S.s_centre[si][1]};
        uint32_t bits = 0;
        for (uint32_t b = 0; b < 32 && w * 32 + b < S.n_parts; b++) {
            const float *k = S.kline[w * 32 + b];
            if (fs_behind(Seg2{V2{k[0], k[1]}, V2{k[2], k[3]}}, c)) bits |= 1u << b;
        }
        P.behind[((size_t)f * P.sprite_stride + si) * FS_BEHIND_WORDS + w] = bits;
    }
}
// phase 6: place of every sprite in the far-to-near order: stable ascending sort on the key, reversed (map_objects.rs:216-217)
DG_HD void fs_ph_sprite_order(FsShared &S, int lane) {
    if (S.fail) return;
    for (uint32_t si = (uint32_t)lane; si < S.n_sprites; si += FS_LANES) {
        uint32_t before = 0;
        for (uint32_t t = 0; t < S.n_sprites; t++) before += (S.s_key[t] < S.s_key[si]) || (S.s_key[t] == S.s_key[si] && t < si);
        S.s_order[si] = (uint16_t)(S.n_sprites - 1 - before);
    }
}
// phase 7: a two-sided middle part is drawn just before the first sprite (in that order) whose midpoint it is behind (map_objects.rs:220-240)
DG_HD void fs_ph_masked_when(FsShared &S, int lane) {
    if (S.fail) return;
    for (uint32_t r = (uint32_t)lane; r < S.n_parts; r += FS_LANES) {
        if (S.kept_t[r] == 0xfffeu) continue;
        const Seg2 line{V2{S.kline[r][0], S.kline[r][1]}, V2{S.kline[r][2], S.kline[r][3]}};
        uint32_t t = 0xffffu;
        for (uint32_t si = 0; si < S.n_sprites; si++)
            if (S.s_order[si] < t && fs_behind(line, V2{S.s_mid[si][0], S.s_mid[si][1]})) t = S.s_order[si];
        S.kept_t[r] = (uint16_t)t;
    }
}
// phase 8: the draw sequence numbers (what the host walker counts while it replays the interleave, frontend.cpp map_objects)
DG_HD void fs_ph_seq(const FsParams &P, FsShared &S, int f, int lane) {
    if (S.fail) return;
    for (uint32_t si = (uint32_t)lane; si < S.n_sprites; si += FS_LANES) {
        const uint32_t m = S.s_order[si];
        uint32_t walls = 0;
        for (uint32_t r = 0; r < S.n_parts; r++) walls += S.kept_t[r] <= m;                 // (0xfffe / 0xffff never are)
        P.sprites[(size_t)f * P.sprite_stride + si].seq = m + walls;
    }
    for (uint32_t r = (uint32_t)lane; r < S.n_parts; r += FS_LANES) {
        const uint32_t t = S.kept_t[r];
        if (t == 0xfffeu) continue;
        uint32_t earlier = 0;
        for (uint32_t q = 0; q < S.n_parts; q++) {
            const uint32_t tq = S.kept_t[q];
            if (tq == 0xfffeu) continue;
            earlier += tq < t || (tq == t && q > r);                                          // same sprite: later records first (reversed list)
        }
        P.parts[(size_t)f * FS_PART_CAP + r].seq = (t == 0xffffu ? S.n_sprites : t) + earlier;
    }
}
// phases 9a .. 9e: the column bins (frontend.cpp bin_by_columns: for every FE_BIN_W-column strip the parts, and the sprites, that touch
// it, in order).  A bin's members are a bit mask over the part (sprite) indices — set by one lane per part, counted per bin, and a
// member's place in its bin's list is the number of mask bits below its own.  The masks live in first[], which is dead by now.
constexpr uint32_t FS_PMASK_WORDS = FS_PART_CAP / 32, FS_SMASK_WORDS = FS_SPRITE_CAP / 32;   // mask words per bin: parts, sprites
static_assert(FS_PART_CAP % 32 == 0 && FS_SPRITE_CAP % 32 == 0, "bin masks");
static_assert((FS_MAX_W / FE_BIN_W) * (FS_PMASK_WORDS + FS_SMASK_WORDS + 2) <= FS_MAX_W, "bin masks fit first[]");
DG_HD uint32_t fs_mask_words(uint32_t kind) { return kind ? FS_SMASK_WORDS : FS_PMASK_WORDS; }
DG_HD uint32_t *fs_bin_mask(FsShared &S, uint32_t nb, uint32_t kind, uint32_t b) {     // kind 0 parts, 1 sprites
    return kind ? S.first + nb * FS_PMASK_WORDS + b * FS_SMASK_WORDS : S.first + b * FS_PMASK_WORDS;
}
DG_HD uint32_t *fs_bin_off(FsShared &S, uint32_t nb, uint32_t kind) { return S.first + nb * (FS_PMASK_WORDS + FS_SMASK_WORDS) + kind * nb; }
DG_HD void fs_ph_bin_clear(const FsParams &P, FsShared &S, int lane) {
    const uint32_t nb = (uint32_t)(P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    for (uint32_t i = (uint32_t)lane; i < nb * (FS_PMASK_WORDS + FS_SMASK_WORDS + 2); i += FS_LANES) S.first[i] = 0u;
}
DG_HD void fs_ph_bin_mark(const FsParams &P, FsShared &S, int lane) {
    if (S.fail) return;
    const uint32_t nb = (uint32_t)(P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    for (uint32_t r = (uint32_t)lane; r < S.n_parts; r += FS_LANES)
        for (uint32_t b = (uint32_t)S.kept_sx[r] / FE_BIN_W; b <= (uint32_t)S.kept_ex[r] / FE_BIN_W; b++) fs_or_u32(&fs_bin_mask(S, nb, 0, b)[r >> 5], 1u << (r & 31u));
    for (uint32_t q = (uint32_t)lane; q < S.n_sprites; q += FS_LANES)
        for (uint32_t b = S.s_x0b[q]; b <= S.s_x1b[q] && b < nb; b++) fs_or_u32(&fs_bin_mask(S, nb, 1, b)[q >> 5], 1u << (q & 31u));
}
DG_HD void fs_ph_bin_count(const FsParams &P, FsShared &S, int lane) {
    const uint32_t nb = (uint32_t)(P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    for (uint32_t i = (uint32_t)lane; i < 2 * nb; i += FS_LANES) {
        const uint32_t *m = fs_bin_mask(S, nb, i / nb, i % nb);
        uint32_t c = 0;
        for (uint32_t w = 0; w < fs_mask_words(i / nb); w++) c += fs_popc(m[w]);
        fs_bin_off(S, nb, i / nb)[i % nb] = c;
    }
}
DG_HD void fs_ph_bin_prefix(const FsParams &P, FsShared &S, int f) {      // lane 0
    const uint32_t nb = (uint32_t)(P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    uint32_t *bo = P.bin_off + (size_t)f * (nb + 1), *so = P.sbin_off + (size_t)f * (nb + 1);
    uint32_t *cp = fs_bin_off(S, nb, 0), *cs = fs_bin_off(S, nb, 1);
    // (the frame's workgroups of dg_fe_columns — 256 columns = four bins each — are appended to the launch-order list of their weight
    // class here rather than with the header: the atomics' round trip then overlaps the other lanes' filling of the bin lists.  No
    // indexed local arrays: they would live in scratch memory.)
    uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    uint32_t rp = 0, rs = 0, w = 0;
    for (uint32_t b = 0; b < nb; b++) {
        const uint32_t np = cp[b], ns = cs[b];
        bo[b] = rp; so[b] = rs;
        cp[b] = rp; cs[b] = rs;
        rp += np; rs += ns;
        w = w > np + 2 * ns ? w : np + 2 * ns;
        if ((b & 3u) == 3u || b + 1 == nb) {
            const uint32_t cls = w > 32 ? 0u : w > 16 ? 1u : w > 8 ? 2u : 3u;
            S.group_cls[b >> 2] = (uint8_t)cls;
            n0 += cls == 0u; n1 += cls == 1u; n2 += cls == 2u; n3 += cls == 3u;
            w = 0;
        }
    }
    bo[nb] = rp; so[nb] = rs;
    if (rp > FS_BIN_CAP || rs > P.sbin_stride) S.fail = 1;
    if (P.order_cnt) {
        const uint32_t groups = (nb + 3) / 4;
        uint32_t a0 = n0 ? fs_add_u32(&P.order_cnt[0], n0) : 0u, a1 = n1 ? fs_add_u32(&P.order_cnt[1], n1) : 0u;
        uint32_t a2 = n2 ? fs_add_u32(&P.order_cnt[2], n2) : 0u, a3 = n3 ? fs_add_u32(&P.order_cnt[3], n3) : 0u;
        for (uint32_t g = 0; g < groups; g++) {
            const uint32_t cls = S.group_cls[g];
            const uint32_t at = cls == 0u ? a0++ : cls == 1u ? a1++ : cls == 2u ? a2++ : a3++;
            P.order_list[(size_t)cls * P.n_items + at] = (uint32_t)f * groups + g;
        }
    }
}
DG_HD void fs_ph_bin_fill(const FsParams &P, FsShared &S, int f, int lane) {
    if (S.fail) return;
    const uint32_t nb = (uint32_t)(P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    auto place = [&](uint32_t kind, uint32_t i, uint32_t b) {
        const uint32_t *m = fs_bin_mask(S, nb, kind, b);
        uint32_t rank = fs_popc(m[i >> 5] & ((1u << (i & 31u)) - 1u));
        for (uint32_t w = 0; w < (i >> 5); w++) rank += fs_popc(m[w]);
        return fs_bin_off(S, nb, kind)[b] + rank;
    };
    for (uint32_t r = (uint32_t)lane; r < S.n_parts; r += FS_LANES)
        for (uint32_t b = (uint32_t)S.kept_sx[r] / FE_BIN_W; b <= (uint32_t)S.kept_ex[r] / FE_BIN_W; b++)
            P.bin_parts[(size_t)f * FS_BIN_CAP + place(0, r, b)] = (uint16_t)r;
    for (uint32_t q = (uint32_t)lane; q < S.n_sprites; q += FS_LANES)
        for (uint32_t b = S.s_x0b[q]; b <= S.s_x1b[q] && b < nb; b++)
            P.sbin_sprites[(size_t)f * P.sbin_stride + place(1, q, b)] = (uint16_t)q;
}
// phase 10a: the frame's occupancy row goes back to zero for the next batch (the rows are zeroed once, at upload: a memset of all rows
// per batch is a fill kernel plus a launch in front of every walk).  The candidate row itself keeps its stale entries: no bit, no entry.
DG_HD void fs_ph_clean(const FsParams &P, int f, int lane) {
    const uint32_t wpl = fs_occ_wpl(P.n_segs);
    uint32_t *occ = P.occ + (size_t)f * wpl * FS_LANES + (size_t)lane * wpl;
    for (uint32_t w = 0; w < wpl; w++) occ[w] = 0u;
}
// phase 10 (lane 0): the frame header the column walk reads; a frame that was given up carries nothing and is flagged for the host
DG_HD void fs_ph_header(const FsParams &P, FsShared &S, int f) {
    const uint32_t nb = (uint32_t)(P.k.W + FE_BIN_W - 1) / FE_BIN_W;
    const bool bad = S.fail || (P.flags[f] & FE_OVF_SEGS);
    FeFrame ff;
    ff.part_base = (uint32_t)f * FS_PART_CAP; ff.n_parts = bad ? 0u : S.n_parts;
    ff.sprite_base = (uint32_t)f * P.sprite_stride; ff.n_sprites = bad ? 0u : S.n_sprites;
    ff.behind_base = (uint32_t)f * P.sprite_stride * FS_BEHIND_WORDS;
    ff.behind_words = FS_BEHIND_WORDS;
    ff.n_sky_slots = bad ? 0u : S.n_sky;
    ff.sky_base = (uint32_t)f * FS_SKY_CAP;
    ff.bin_base = (uint32_t)f * FS_BIN_CAP;
    ff.sbin_base = (uint32_t)f * P.sbin_stride;
    ff.pad[0] = ff.pad[1] = 0;
    P.fframes[f] = ff;
    if (bad) {
        uint32_t *bo = P.bin_off + (size_t)f * (nb + 1), *so = P.sbin_off + (size_t)f * (nb + 1);
        for (uint32_t b = 0; b <= nb; b++) { bo[b] = 0; so[b] = 0; }
        if (S.fail) fs_flag(P, f, FE_OVF_SEGS);
    }
}

}  // namespace dg
