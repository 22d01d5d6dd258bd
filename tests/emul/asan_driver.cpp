// tests/emul/asan_driver.cpp — sanitizer stress for the product's HOST code (scene loader, list generation, binner),
// built with -fsanitize=address,undefined by tests/test_sanitizers.py (GPU sanitizers are not available on this pool).
//   usage: asan_driver <wad file> <camera path .f32> <map name>
// Renders lists for path frames at several sizes, for random off-map viewpoints, and loads truncated / bit-flipped
// copies of the WAD: every call must either succeed or fail with a clean error, never touch memory out of bounds.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../doom-rust-renderer_amd/csrc/binner.hpp"
#include "../../doom-rust-renderer_amd/csrc/fe_core.h"
#include "../../doom-rust-renderer_amd/csrc/frontend.hpp"
#include "../../doom-rust-renderer_amd/csrc/scene.hpp"

using namespace dg;

static uint32_t rng_state = 2463534242u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }
static float frand(float lo, float hi) { return lo + (hi - lo) * (float)(rnd() & 0xffffff) / 16777216.0f; }

static int walk_parts_on_host(const Scene &sc, int W, int H, const dg_view &v, FrameArena &A);

static int lists(const Scene &sc, int W, int H, dg_view v) {
    static FrameArena arena;
    static BinnedFrame bf;
    std::string err;
    dg_frame_lists fl;
    fill_view_trig(v);
    int rc = build_frame_lists(sc, W, H, v, arena, fl, err);
    if (rc) return rc;
    rc = bin_frame(sc, make_consts(W, H), fl, bf, err);
    if (rc) return rc;
    // parts mode (records for the device column walk) + a host replay of the column bodies over its output
    static FrameArena parts_arena;
    rc = build_frame_parts(sc, W, H, v, parts_arena, err);
    if (rc == kPartsUnsupported) return 0;
    if (rc) return rc;
    return walk_parts_on_host(sc, W, H, v, parts_arena);
}

// The bodies of dg_fe_columns over one frame's records, with bounds-checked scratch (sanitizer coverage of fe_core.h).
static int walk_parts_on_host(const Scene &sc, int W, int H, const dg_view &v, FrameArena &A) {
    const FrameConsts fk = make_consts(W, H);
    DevFrame fr = make_frame_header(v);
    FeFrame ff{0, (uint32_t)A.parts.size(), 0, (uint32_t)A.sprites.size(), 0, A.behind_words, A.n_sky_slots, 0, 0, 0, {0, 0}};
    const uint32_t w64 = (uint32_t)((W + 63) / 64), slots = 16;
    std::vector<FeU4> cspans((size_t)slots * W);
    std::vector<FeColRec> recs((size_t)slots * W);
    std::vector<uint32_t> cnt((size_t)W), flags(1, 0);
    std::vector<uint64_t> events((size_t)3 * (A.n_sky_slots + 1) * w64, 0);
    FeParams P{};
    P.k = DevConsts{fk.ARC, fk.GCFX, fk.CFX, fk.CFY, W, H};
    P.frames = &fr; P.fframes = &ff; P.parts = A.parts.data(); P.sprites = A.sprites.data(); P.behind = A.behind.data();
    P.cspans = cspans.data(); P.recs = recs.data(); P.cnt = cnt.data(); P.events = events.data(); P.flags = flags.data();
    P.n_frames = 1; P.max_sky_slots = A.n_sky_slots; P.w64 = w64; P.col_slots = slots;
    if (A.bin_off.size() != (size_t)w64 + 1 || A.sbin_off.size() != (size_t)w64 + 1) return -100;
    std::vector<uint32_t> near_cand((size_t)FE_NEAR_RECS * FE_BIN_W);
    std::vector<uint16_t> near_part((size_t)FE_NEAR_RECS * FE_BIN_W);
    const FeRecStore st{near_cand.data(), near_part.data()};
    for (int x = 0; x < W; x++) {
        FeColumn c = fe_column_start(P, 0, x);
        const uint32_t bin = (uint32_t)x / FE_BIN_W;
        for (uint32_t bi = A.bin_off[bin]; bi < A.bin_off[bin + 1]; bi++) {
            const uint32_t pi = A.bin_parts.at(bi);
            const FePart &p = A.parts.at(pi);
            if (x < p.sx || x > p.ex) continue;
            const uint32_t ev = fe_part_column(P, 0, p, pi, c, st);
            if (p.sky_slot >= 0 && (ev & FE_EV_FADD)) fe_event_words(P, 0, p.sky_slot, 0)[x >> 6] |= 1ull << (x & 63);
            if (c.hor) break;
        }
        for (uint32_t bi = A.sbin_off[bin]; bi < A.sbin_off[bin + 1]; bi++) {
            const uint32_t si = A.sbin_sprites.at(bi);
            const FeSprite &sp = A.sprites.at(si);
            if (x >= sp.x0 && x < sp.x1) fe_sprite_column(P, 0, ff, sp, si, c, st, fe_behind_row(P, ff, sp));
        }
        for (uint32_t i = 0; i < c.nsp; i++) (void)fe_resolve(P, fr, ff, x, cspans[(size_t)i * W + (size_t)x]);
    }
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> wad((std::istreambuf_iterator<char>(f)), {});
    std::ifstream pf(argv[2], std::ios::binary);
    std::vector<float> path(8000);
    pf.read((char *)path.data(), 32000);
    std::string err;
    Scene *sc = load_scene_from_wad(wad.data(), wad.size(), argv[3], err);
    if (!sc) { std::printf("load failed: %s\n", err.c_str()); return 1; }
    int ok = 0, refused = 0;
    const int sizes[][2] = {{320, 200}, {1280, 800}, {64, 48}, {132, 67}, {4, 4}, {16384, 8}};
    for (auto &s : sizes)
        for (int i = 0; i < 1000; i += 53) {
            const float *r = &path[(size_t)i * 8];
            dg_view v{r[0], r[1], r[2], r[7], r[3], r[4], r[5], r[6], (float)(i % 7) * 0.37f, 1};
            (lists(*sc, s[0], s[1], v) == 0 ? ok : refused)++;
        }
    for (int i = 0; i < 3000; i++) {
        dg_view v{frand(-40000, 40000), frand(-40000, 40000), frand(-50, 50), frand(-40000, 40000), 0, 0, 0, 0, frand(-5, 50), 0};
        if (i % 7 == 0) { v.x = frand(0, 4096); v.y = frand(0, 3072); v.floor_height = frand(-100, 100); }
        (lists(*sc, 320, 200, v) == 0 ? ok : refused)++;
    }
    // hostile trig values (a caller may pass anything when trig_valid = 1)
    for (int i = 0; i < 200; i++) {
        dg_view v{frand(0, 4096), frand(0, 3072), frand(-7, 7), 0, frand(-2, 2), frand(-2, 2), frand(-2, 2), frand(-2, 2), 0, 1};
        if (i % 10 == 0) v.cos_na = 0.0f / 0.0f;
        if (i % 10 == 1) v.sin_na = 1.0f / 0.0f;
        (lists(*sc, 320, 200, v) == 0 ? ok : refused)++;
    }
    int loaded = 0, rejected = 0;
    for (size_t cut = 0; cut < wad.size(); cut += wad.size() / 211 + 1) {
        Scene *s2 = load_scene_from_wad(wad.data(), cut, argv[3], err);
        if (s2) { loaded++; delete s2; } else rejected++;
    }
    for (int k = 0; k < 400; k++) {
        std::vector<uint8_t> b = wad;
        for (int j = 0; j < 1 + (k % 4); j++) b[12 + rnd() % (b.size() - 12)] ^= (uint8_t)(1 + rnd() % 255);
        Scene *s2 = load_scene_from_wad(b.data(), b.size(), argv[3], err);
        if (!s2) { rejected++; continue; }
        loaded++;
        const float *r = &path[0];
        dg_view v{r[0], r[1], r[2], r[7], r[3], r[4], r[5], r[6], 0.0f, 1};
        (lists(*s2, 160, 100, v) == 0 ? ok : refused)++;
        delete s2;
    }
    delete sc;
    std::printf("SANITIZER DRIVER OK lists ok=%d refused=%d wads loaded=%d rejected=%d\n", ok, refused, loaded, rejected);
    return 0;
}
