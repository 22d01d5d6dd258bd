// fe_core.h — the per-column half of the reference's front end as host/device inline functions (one call = one screen
// column of one record).  fe_kernels.hip runs them with one lane per column; tests/emul compiles the same bodies for
// the CPU.  They restate, column for column:
//
//   fe_part_column    the body of the `for x in start.x..=end.x` loop of Segs::process_sidedef      src/renderer/segs.rs:202-345
//                     and SidedefVisPlanes::add_*/flush (as span emission + event bits)             src/renderer/sidedef_visplanes.rs:60-118
//   fe_sprite_column  the clip arrays and the column push of draw_map_objects                       src/renderer/map_objects.rs:130-209
//   fe_gap            the zero-filled Visplane entries between two adds of one visplane             src/renderer/visplanes.rs:28-38
//
// Spans are appended to the column's scratch list in a compact form (extent + which record they texture from) with a
// sort key that encodes the reference's draw order (frontend.hpp: walls in BSP order, visplanes in push
// order, then the sprite / masked-wall sequence).
#pragma once
#include <stddef.h>

#include "fe_dev.h"
#include "raster_core.h"

namespace dg {

struct alignas(16) FeU4 { uint32_t x, y, z, w; };

struct FeParams {
    DevScene scene;
    DevConsts k;
    const DevFrame *frames;       // [n_frames] view constants + span_base (= f * span_stride)
    const FeFrame *fframes;       // [n_frames]
    const FePart *parts;
    const uint32_t *bin_off;      // [frame][n_bins + 1]  column bins of the parts (fe_dev.h)
    const uint16_t *bin_parts;
    const uint32_t *sbin_off;     // [frame][n_bins + 1]  column bins of the sprites
    const uint16_t *sbin_sprites;
    const FeSprite *sprites;
    const uint32_t *behind;
    const uint32_t *sky_parts;    // per sky slot: index of its part within the frame
    // per-column scratch, [frame][slot][W] so that neighbouring lanes touch neighbouring addresses
    FeU4 *cspans;                 // col_slots slots of one compact span: x = draw-order key, y = ctop | cbot << 16,
                                  // z = top_y | bot_y << 16 (walls), w = FES_* source (resolved into a DevRSpan by dg_fe_scatter)
    FeColRec *recs;               // col_slots slots
    uint32_t *cnt;                // [frame][W] spans emitted per column
    uint64_t *events;             // [3][n_frames][max_sky_slots][W64] add-floor / add-ceiling / not-flushed bits per column,
                                  // zeroed by the host (= what a horizontally occluded column yields), see fe_event_words
    uint32_t *flags;              // [frame] FE_OVF_*, DEVICE memory (the walk's kernels OR into it), zeroed by the host with the event bits
    uint32_t *host_flags;         // [frame] the same word, published once per frame by dg_fe_scan — pinned host memory (plain stores only:
                                  // an OR is not one of the atomics PCIe carries)
    uint32_t *totals;             // [frame] spans of the frame, pinned host memory (written by dg_fe_scan)
    // outputs consumed by dg_raster_tiles
    uint32_t *col_off;            // [frame][W + 1]
    DevRSpan *rspans;
    int32_t n_frames;
    uint32_t span_stride;         // rspans reserved per frame
    uint32_t w64;                 // (W + 63) / 64 = number of column bins
    uint32_t max_sky_slots;       // max n_sky_slots over the frames of the batch (grid of dg_fe_gaps; stride of the event rows)
    uint32_t gap_waves;           // 0: one wave of dg_fe_gaps per sky slot; else this many per frame, striding over its slots (the bound is loose)
    uint32_t col_slots;           // span slots and wall-record slots per screen column in the scratch arrays (<= FE_MAX_COL_SLOTS)
    // launch order of dg_fe_columns: workgroup i takes item order[i] = frame * ceil(W / 256) + 256-column group, heaviest first (the kernel
    // lasts as long as its longest bin, and a long bin that starts late ends late); nullptr = natural order.  order_cnt == nullptr: one
    // list, sorted by the host.  Else (lists built by dg_fs_frame, fs_frame.h): FS_ORDER_CLASSES lists of order_cnt[k] entries each,
    // list k at order + k * n_items, heaviest class first.
    const uint32_t *order;
    const uint32_t *order_cnt;
};

struct FeColumn {                 // what one lane carries through the walk
    int32_t x;
    int32_t hor, fo, co;          // horizontal_ocl / floor_ver_ocl / ceiling_ver_ocl of this column (segs.rs:70-74)
    uint32_t nsp, nrec, ovf;
    size_t sp_at;                 // element index of the column's next free slot in cspans (advances by W per slot)
};

DG_HD FeColumn fe_column_start(const FeParams &P, int f, int32_t x) {         // Segs::new, segs.rs:97-99
    FeColumn c;
    c.x = x; c.hor = 0; c.fo = P.k.H; c.co = -1; c.nsp = 0; c.nrec = 0; c.ovf = 0;
    c.sp_at = (size_t)f * P.col_slots * (size_t)P.k.W + (size_t)x;
    return c;
}

// Where a column's wall records live between the walk of the parts and the clipping of the sprites.  The first FE_NEAR_RECS of a
// column sit next to the walker — on the GPU in the wavefront's LDS, [slot][lane] — because a sprite column reads ALL of them, and
// a round trip to the global scratch per eight records and sprite was half the life of the longest waves of dg_fe_columns (the ones
// its duration is made of: profiles/r04_column_walk.md); only records beyond that go to the global rows (slot = record number).
constexpr uint32_t FE_NEAR_RECS = 9;
struct FeRecStore {
    uint32_t *near_cand;          // [FE_NEAR_RECS][64]  top_cand | bottom_cand << 16 of the bin's 64 columns
    uint16_t *near_part;          // [FE_NEAR_RECS][64]
};
DG_HD void fe_put_rec(const FeParams &P, int f, FeColumn &c, const FeRecStore &st, const FeColRec &r) {
    if (c.nrec >= P.col_slots) { c.ovf |= FE_OVF_RECS; return; }
    if (c.nrec < FE_NEAR_RECS) {
        const uint32_t at = c.nrec * (uint32_t)FE_BIN_W + ((uint32_t)c.x & (uint32_t)(FE_BIN_W - 1));
        st.near_cand[at] = (uint32_t)(uint16_t)r.top_cand | ((uint32_t)(uint16_t)r.bottom_cand << 16);
        st.near_part[at] = r.part;
    } else {
        P.recs[((size_t)f * P.col_slots + c.nrec) * (size_t)P.k.W + (size_t)c.x] = r;
    }
    c.nrec++;
}

DG_HD int32_t fe_min(int32_t a, int32_t b) { return a < b ? a : b; }
DG_HD int32_t fe_max(int32_t a, int32_t b) { return a > b ? a : b; }

// FeU4.w of a compact span: which record its texture-mapping constants come from.
enum : uint32_t { FES_KIND_SHIFT = 30, FES_SPRITE = 1u << 29, FES_CEIL = 1u << 28, FES_INDEX_MASK = (1u << 28) - 1 };

DG_HD void fe_emit(const FeParams &P, int f, FeColumn &c, uint32_t key, int32_t ctop, int32_t cbot, int32_t top_y, int32_t bot_y, uint32_t src) {
    if (c.nsp >= P.col_slots) { c.ovf |= FE_OVF_SPANS; return; }
    P.cspans[c.sp_at] = FeU4{key, (uint32_t)ctop | ((uint32_t)cbot << 16), (uint32_t)(uint16_t)top_y | ((uint32_t)(uint16_t)bot_y << 16), src};
    c.sp_at += (size_t)P.k.W;
    c.nsp++;
}

DG_HD DevSpan fe_span(int32_t ctop, int32_t cbot, int32_t top_y, int32_t bot_y, uint8_t kind, int32_t x) {
    DevSpan s;
    s.ctop = (int16_t)ctop; s.cbot = (int16_t)cbot; s.top_y = (int16_t)top_y; s.bot_y = (int16_t)bot_y;
    s.rec = 0; s.kind = kind; s.pad0 = 0; s.x = (int16_t)x; s.pad1 = 0;
    return s;
}

// One Visplane::add_point (visplanes.rs:28-38) as draw_visplane / draw_sky will see it (visplanes.rs:61-62,95-101).
DG_HD void fe_plane(const FeParams &P, int f, FeColumn &c, uint32_t key, bool sky, uint32_t src, int32_t top, int32_t bottom) {
    const int32_t t = fe_max(top, 0), b = fe_min(bottom, P.k.H - 1);
    if (sky) {
        if (t > b) return;
        fe_emit(P, f, c, key, t, b, 0, 0, ((uint32_t)SPAN_SKY << FES_KIND_SHIFT) | src);
    } else {
        if (wrap_i16(b - t) <= 1) return;
        fe_emit(P, f, c, key, t, b, 0, 0, ((uint32_t)SPAN_FLAT << FES_KIND_SHIFT) | src);
    }
}

// The compact span of column x as the raster kernel wants it (what dg_setup_spans does for host-built lists).
DG_HD DevRSpan fe_resolve(const FeParams &P, const DevFrame &fr, const FeFrame &ff, int32_t x, const FeU4 &cs) {
    const uint32_t kind = cs.w >> FES_KIND_SHIFT, idx = cs.w & FES_INDEX_MASK;
    const int32_t ctop = (int32_t)(cs.y & 0xffffu), cbot = (int32_t)(cs.y >> 16);
    if (kind == SPAN_WALL) {
        const DevWallRec &r = (cs.w & FES_SPRITE) ? P.sprites[ff.sprite_base + idx].wall : P.parts[ff.part_base + idx].wall;
        return resolve_wall_span(fe_span(ctop, cbot, lo_i16(cs.z), hi_i16(cs.z), SPAN_WALL, x), r);
    }
    if (kind == SPAN_FLAT) {
        const FePart &p = P.parts[ff.part_base + idx];
        return resolve_flat_span(fe_span(ctop, cbot, 0, 0, SPAN_FLAT, x), (cs.w & FES_CEIL) ? p.ceil_plane : p.floor_plane, P.k, (uint32_t)(P.scene.flats - P.scene.texel_idx));
    }
    return resolve_sky_span(fe_span(ctop, cbot, 0, 0, SPAN_SKY, x), P.scene, P.k, fr);
}

DG_HD void fe_occlude(const FeParams &P, FeColumn &c) {                        // segs.rs:113-117
    c.hor = 1;
    c.fo = c.co = (int32_t)(int16_t)((int16_t)P.k.H / 2);
}

// Column c.x of part `pi` (sx <= x <= ex).  Returns FE_EV_* bits.
DG_HD uint32_t fe_part_column(const FeParams &P, int f, const FePart &p, uint32_t pi, FeColumn &c, const FeRecStore &st) {
    const int32_t hm1 = P.k.H - 1, x = c.x;
    const uint32_t fl = p.flags;
    const bool two = (fl & FEP_TWO_SIDED_MID) != 0, only = (fl & FEP_ONLY_OCCL) != 0, lower = (fl & FEP_LOWER) != 0, upper = (fl & FEP_UPPER) != 0;
    const bool drawc = (fl & FEP_DRAW_CEILING) != 0;
    const bool full = !lower && !upper && !only;
    const bool planes_here = !two && (full || only);
    uint32_t ev = 0;
    if (!c.hor) {
        const int32_t bottom_y = f32_as_i16(p.bsy + ((float)x - p.bsx) * p.bdelta);
        const int32_t top_y = f32_as_i16(p.tsy + ((float)x - p.tsx) * p.tdelta);
        const int32_t fo = c.fo, co = c.co;
        const int32_t cb = fe_min(hm1, fe_min(fo, bottom_y));
        const int32_t ct = fe_max(0, fe_max(co, top_y));
        const bool vis = cb >= ct;
        if (vis) {
            const bool ext_b = lower || (!two && full), ext_t = upper || (!two && full);
            if (two || ext_b || ext_t) {                                       // the records draw_map_objects clips against
                FeColRec r;
                r.part = (uint16_t)pi; r.pad = 0;
                if (two) {                                                     // ST_TWOSIDED branch, map_objects.rs:152-163
                    r.top_cand = (int16_t)(drawc ? top_y : -32768);
                    r.bottom_cand = (int16_t)bottom_y;
                } else {                                                       // solid: map_objects.rs:141-151
                    r.top_cand = (int16_t)(ext_t ? cb : -32768);
                    r.bottom_cand = (int16_t)(ext_b ? ct : 32767);
                }
                fe_put_rec(P, f, c, st, r);
            }
            if ((fl & FEP_HAS_BITMAP) && (two || !only)) {                     // inline draw (segs.rs:231-258) or masked replay (segs.rs:593-597)
                const uint32_t key = two ? (FE_KEY_LATE | (p.seq << 2)) : (FE_KEY_WALL | (pi << 2));
                fe_emit(P, f, c, key, ct, cb, top_y, bottom_y, ((uint32_t)SPAN_WALL << FES_KIND_SHIFT) | pi);
            }
        }
        if (planes_here && vis) {
            bool added = false;
            if (cb < fo && cb != hm1) {
                fe_plane(P, f, c, FE_KEY_PLANE | (pi << 2), (fl & FEP_FLOOR_SKY) != 0, pi, cb, fo);
                added = true; ev |= FE_EV_FADD;
            }
            if (drawc && ct > co && ct != -1) {
                fe_plane(P, f, c, FE_KEY_PLANE | (pi << 2) | 1u, (fl & FEP_CEIL_SKY) != 0, FES_CEIL | pi, co, ct);
                added = true; ev |= FE_EV_CADD;
            }
            if (!added) ev |= FE_EV_FLUSH;
        } else if (planes_here && !vis && fo > co) {                           // occluded wall, open vertical gap (segs.rs:293-318)
            if (bottom_y <= co) {
                fe_plane(P, f, c, FE_KEY_PLANE | (pi << 2), (fl & FEP_FLOOR_SKY) != 0, pi, co, fo);
                ev |= FE_EV_FADD;
                fe_occlude(P, c);
            }
            if (drawc && top_y >= fo) {
                fe_plane(P, f, c, FE_KEY_PLANE | (pi << 2) | 1u, (fl & FEP_CEIL_SKY) != 0, FES_CEIL | pi, co, fo);
                ev |= FE_EV_CADD;
                fe_occlude(P, c);
            }
        }
        if (!two && vis) {
            if (only) {
                c.fo = cb;
                if (drawc) c.co = ct;
            }
            if (lower) c.fo = ct;
            if (upper) c.co = cb;
        }
    } else {
        ev |= FE_EV_FLUSH;
    }
    if (!two && full) fe_occlude(P, c);
    return ev;
}

// Column c.x of one sprite (x0 <= x < x1): clip arrays from the wall records of this column that are not behind the
// sprite's centre, then the clipped column (map_objects.rs:130-209).  `row(w)` = word w of the sprite's behind-bit row (the GPU
// stages the rows of the sprites it is about to walk next to their records).
template <typename Row>
DG_HD void fe_sprite_column(const FeParams &P, int f, const FeFrame &ff, const FeSprite &s, uint32_t si, FeColumn &c, const FeRecStore &st, Row row) {
    const int32_t H = P.k.H, x = c.x;
    int32_t top_clip = -1, bottom_clip = H;
    const uint32_t n_near = c.nrec < FE_NEAR_RECS ? c.nrec : FE_NEAR_RECS, lane = (uint32_t)x & (uint32_t)(FE_BIN_W - 1);
    // eight records (then their eight behind-bit words) are read before any is used: independent reads, two round trips per eight records
    for (uint32_t i0 = 0; i0 < n_near; i0 += 8) {
        uint32_t cand[8], part[8], w[8];
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t at = (i0 + k < n_near ? i0 + k : n_near - 1) * (uint32_t)FE_BIN_W + lane;
            cand[k] = st.near_cand[at];
            part[k] = st.near_part[at];
        }
        for (uint32_t k = 0; k < 8; k++) w[k] = row(part[k] >> 5);
        for (uint32_t k = 0; k < 8; k++) {
            if (i0 + k >= n_near || ((w[k] >> (part[k] & 31)) & 1u)) continue;
            top_clip = fe_max(top_clip, (int32_t)(int16_t)(cand[k] & 0xffffu));
            bottom_clip = fe_min(bottom_clip, (int32_t)(int16_t)(cand[k] >> 16));
        }
    }
    const size_t at0 = (size_t)f * P.col_slots * (size_t)P.k.W + (size_t)x, W = (size_t)P.k.W;
    for (uint32_t i0 = FE_NEAR_RECS; i0 < c.nrec; i0 += 8) {                  // the far ones: global scratch rows
        FeColRec r[8];
        uint32_t w[8];
        for (uint32_t k = 0; k < 8; k++) r[k] = P.recs[at0 + (size_t)(i0 + k < c.nrec ? i0 + k : c.nrec - 1) * W];
        for (uint32_t k = 0; k < 8; k++) w[k] = row(r[k].part >> 5);
        for (uint32_t k = 0; k < 8; k++) {
            if (i0 + k >= c.nrec || ((w[k] >> (r[k].part & 31)) & 1u)) continue;
            top_clip = fe_max(top_clip, r[k].top_cand);
            bottom_clip = fe_min(bottom_clip, r[k].bottom_cand);
        }
    }
    const int32_t bottom_y = f32_as_i16(s.bsy + ((float)x - s.bsx) * s.bdelta);
    const int32_t top_y = f32_as_i16(s.tsy + ((float)x - s.tsx) * s.tdelta);
    const int32_t ct = fe_max(0, fe_max(top_y, top_clip));
    const int32_t cb = fe_min(H - 1, fe_min(bottom_y, bottom_clip));
    if (ct > cb) return;
    fe_emit(P, f, c, FE_KEY_LATE | (s.seq << 2), ct, cb, top_y, bottom_y, ((uint32_t)SPAN_WALL << FES_KIND_SHIFT) | FES_SPRITE | si);
}
// The sprite's behind-bit row straight from the batch's array (the CPU emulation; on the GPU rows of more than FE_NEAR_BEHIND words).
struct FeBehindGlobal {
    const uint32_t *row;
    DG_HD uint32_t operator()(uint32_t w) const { return row[w]; }
};
DG_HD FeBehindGlobal fe_behind_row(const FeParams &P, const FeFrame &ff, const FeSprite &s) { return FeBehindGlobal{P.behind + ff.behind_base + s.behind_off}; }

// Is column x a zero-filled entry of a visplane of this part: no add and no flush at x, and the nearest event on
// either side inside [sx, ex] is an add (the visplane was opened before x and extended after it).  `open` holds the
// complement of the flush bits (1 = the column was walked and did not flush).
DG_HD bool fe_gap(const uint64_t *add, const uint64_t *open, int32_t x, int32_t sx, int32_t ex) {
    const int32_t w0 = x >> 6, b0 = x & 63;
    if (((add[w0] | ~open[w0]) >> b0) & 1ull) return false;
    {   // nearest event to the left
        int32_t w = w0;
        uint64_t m = (add[w] | ~open[w]) & ((1ull << b0) - 1ull);
        while (!m) {
            if (w * 64 <= sx) return false;
            w--;
            m = add[w] | ~open[w];
        }
        const int32_t b = 63 - __builtin_clzll(m);
        if (w * 64 + b < sx || !((add[w] >> b) & 1ull)) return false;
    }
    {   // nearest event to the right
        int32_t w = w0;
        uint64_t m = (add[w] | ~open[w]) & (b0 == 63 ? 0ull : ~0ull << (b0 + 1));
        while (!m) {
            if (w * 64 + 63 >= ex) return false;
            w++;
            m = add[w] | ~open[w];
        }
        const int32_t b = __builtin_ctzll(m);
        if (w * 64 + b > ex || !((add[w] >> b) & 1ull)) return false;
    }
    return true;
}

// Event words of one (frame, sky slot): kind 0 = add-floor, 1 = add-ceiling, 2 = not flushed.  The arrays are zeroed
// before the walk: "no add, flushed" is exactly what every column of a part yields once the column is horizontally
// occluded (segs.rs:337-341), so the walk of a column (or of a wavefront whose 64 columns are all occluded) may stop early.
// Bits outside the part's [sx, ex] are never interpreted (fe_gap bounds every scan).
DG_HD uint64_t *fe_event_words(const FeParams &P, int f, int32_t sky_slot, int kind) {
    return P.events + (((size_t)kind * (size_t)P.n_frames + (size_t)f) * (size_t)P.max_sky_slots + (size_t)sky_slot) * (size_t)P.w64;
}

}  // namespace dg
