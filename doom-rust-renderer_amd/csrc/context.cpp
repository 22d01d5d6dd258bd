// context.cpp — dg_ctx: one GPU's resident scene, per-slot list slabs / framebuffer slabs / streams, and the
// host thread pool that builds the per-frame lists.  Implements the C-ABI declared in include/doomgpu.h.
//
// HBM layout (sized once at dg_create for 288 GB parts: nothing is reallocated on the submit path):
//   scene   : palette 1 KB | texel index plane | texel opacity plane | flats (4 KB each)      immutable per map
//   per slot: list slab  [DevFrame x F | col_off x F*(W+1) | DevWallRec.. | DevPlaneRec.. | DevSpan..]  one H2D copy
//             rspan slab DevRSpan 32 B per span (device-only, written by dg_setup_spans, walked by dg_raster_tiles)
//             framebuffer slab  F x 3*W*H bytes RGB24 (the reference's Pixels.pixels, one per frame)
//             DG_FE_DEVICE: record slab [DevFrame x F | FeFrame x F | FePart.. | FeSprite.. | behind bits.. | sky slot -> part.. | column bins..] (one H2D copy),
//             col_off F*(W+1) written by dg_fe_finalize, 2F status words (overflow flags, span totals)
//   per ctx : DG_FE_DEVICE column scratch [F][slot][W]: compact spans 16 B (48 slots), wall-record columns 8 B (48 slots),
//             counts, sky event bits — shared by the slots because their kernels run back to back
#include <hip/hip_runtime_api.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/doomgpu.h"
#include "binner.hpp"
#include "fe_kernels.hpp"
#include "fs_kernels.hpp"
#include "frontend.hpp"
#include "kernels.hpp"
#include "scene.hpp"

using namespace dg;

static thread_local std::string t_err;
static int set_err(int code, const std::string &m) { t_err = m; return code; }

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return set_err(DG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct dg_scene { Scene *sc; };

namespace {

// Minimal persistent pool: parallel_for over [0, n) with dynamic chunking.
class Pool {
public:
    explicit Pool(int n) {
        for (int i = 0; i < n; i++) workers_.emplace_back([this, i] { loop(i); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // fn(index, worker_id); worker ids are 0..size() (the caller participates as id size()).
    void parallel_for(int n, const std::function<void(int, int)> &fn) {
        if (n <= 0) return;
        { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; next_.store(0); pending_ = (int)workers_.size(); gen_++; }
        cv_.notify_all();
        run(fn, (int)workers_.size());
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void run(const std::function<void(int, int)> &fn, int wid) {
        for (;;) {
            int i = next_.fetch_add(1);
            if (i >= n_) break;
            fn(i, wid);
        }
    }
    void loop(int wid) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int, int)> *fn;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            if (fn) run(*fn, wid);
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct Slot {
    hipStream_t stream = nullptr;
    // timing events, attached to the dispatches themselves (kernels.hpp): first / last front-end kernel, raster launch; ev_raster is also
    // what "the slot's kernels are done" is waited on
    hipEvent_t ev_start = nullptr, ev_setup = nullptr, ev_rstart = nullptr, ev_raster = nullptr, ev_h2d = nullptr;
    bool raster_recorded = false;
    hipStream_t copy_stream = nullptr;   // dg_readback_async: D2H of this slot's frames while another slot's kernels run
    uint8_t *copy_out = nullptr;         // pending asynchronous readback (re-issued if the batch has to be redone)
    int copy_first = 0, copy_count = 0;
    bool copy_pending = false;
    uint8_t *h_lists = nullptr;   // pinned staging
    uint8_t *d_lists = nullptr;
    DevRSpan *d_rspans = nullptr;
    uint8_t *d_fb = nullptr;
    size_t lists_cap = 0;
    // last submission
    RasterParams P{};
    uint32_t max_spans = 0;
    uint64_t n_spans = 0, covered = 0, list_bytes = 0, n_walls = 0, n_planes = 0;
    int n_frames = 0;
    float host_ms = 0.0f;         // list generation + binning + packing of the last submission
    bool busy = false, timed = false;
    // device column walk (DG_FE_DEVICE)
    uint8_t *h_fe = nullptr, *d_fe = nullptr;   // record slab: pinned staging / HBM
    uint32_t *d_fe_coloff = nullptr;
    uint32_t *h_status = nullptr;               // pinned host memory the walk's kernels write: [F] overflow flags, [F] spans per frame
    uint64_t *d_events = nullptr;               // sky event bits (fe_event_words), zeroed before every walk
    size_t flags_bytes = 0, walk_state_bytes = 0;   // the flag words + order counters; the whole allocation with the event bits
    bool walk_state_clean = false;              // d_flags .. is all zero (dg_fe_scan cleans up after the walk; enqueue_kernels clears a slot that is not)
    uint32_t *d_order = nullptr;                // dg_fe_columns' launch-order lists as dg_fs_frame builds them (FsParams::order_list)
    uint32_t *d_flags = nullptr;                // [F] overflow flags the walk's kernels OR into; sits in front of d_events (one memset clears both)
    FeParams FP{};
    FsParams FSP{};               // DG_FE_DEVICE_SEGS: the device seg walk in front of the column walk
    bool fs_mode = false;         // the last submission's per-seg half ran on the GPU too
    bool harvested = true;        // DG_FE_AUTO has read this submission's GPU time
    bool fe_mode = false;         // the last submission went through the device column walk
    bool fe_check = false;        // ... and its overflow flags have not been looked at yet
    std::vector<dg_view> views;   // the views of that submission (to redo it on the host if a capacity overflowed)
    // ... and a private copy of their game-state snapshots (the caller's arrays need not outlive the call)
    std::vector<dg_view_state> states;
    std::vector<dg_sector_light> state_lights;
    std::vector<dg_mobj_state> state_mobjs;
    // The scene's own light levels and map-object states as they were when the batch was submitted: a frame that overflows a capacity is
    // redone at dg_wait time from the host walker, and dg_scene_set_sector_light / _mobj_state may have moved the scene on by then
    // (lights.rs:47-259 and map_objects.rs:63-121 run between two submissions of a pipelined caller).
    const Scene *snap_scene = nullptr;
    uint64_t snap_rev = 0;
    std::vector<dg_sector_light> snap_lights;
    std::vector<dg_mobj_state> snap_mobjs;
    void snapshot_scene(const Scene &sc) {
        if (snap_scene == &sc && snap_rev == sc.revision && snap_lights.size() == sc.sectors.size() && snap_mobjs.size() == sc.mobjs.size()) return;
        snap_lights.resize(sc.sectors.size()); snap_mobjs.resize(sc.mobjs.size());
        for (size_t i = 0; i < sc.sectors.size(); i++) snap_lights[i] = dg_sector_light{(int32_t)i, (int32_t)sc.sectors[i].light};
        for (size_t i = 0; i < sc.mobjs.size(); i++) snap_mobjs[i] = dg_mobj_state{(int32_t)i, sc.mobjs[i].sprite_frame, sc.mobjs[i].full_bright ? 1 : 0, 0};
        snap_scene = &sc; snap_rev = sc.revision;
    }
    // The game state frame i of the last submission was rendered with, for the host walker: nullptr = the scene as it is (unchanged since the
    // submission, no per-view snapshot); else the submit-time scene state with the view's own entries on top (later entries win).
    struct RedoState { std::vector<dg_sector_light> lights; std::vector<dg_mobj_state> mobjs; dg_view_state st{}; };
    const dg_view_state *state_for_redo(const Scene &sc, int i, RedoState &tmp) const {
        const dg_view_state *own = states.empty() ? nullptr : &states[(size_t)i];
        if (snap_scene != &sc || snap_rev == sc.revision) return own;
        tmp.lights = snap_lights; tmp.mobjs = snap_mobjs;
        if (own) { tmp.lights.insert(tmp.lights.end(), own->lights, own->lights + own->n_lights); tmp.mobjs.insert(tmp.mobjs.end(), own->mobjs, own->mobjs + own->n_mobjs); }
        tmp.st = dg_view_state{tmp.lights.data(), (uint32_t)tmp.lights.size(), tmp.mobjs.data(), (uint32_t)tmp.mobjs.size()};
        return &tmp.st;
    }
    void keep_states(const dg_view_state *st, int n) {
        states.clear(); state_lights.clear(); state_mobjs.clear();
        if (!st) return;
        for (int i = 0; i < n; i++) { state_lights.insert(state_lights.end(), st[i].lights, st[i].lights + st[i].n_lights); state_mobjs.insert(state_mobjs.end(), st[i].mobjs, st[i].mobjs + st[i].n_mobjs); }
        size_t lo = 0, mo = 0;
        for (int i = 0; i < n; i++) {
            states.push_back(dg_view_state{state_lights.data() + lo, st[i].n_lights, state_mobjs.data() + mo, st[i].n_mobjs});
            lo += st[i].n_lights; mo += st[i].n_mobjs;
        }
    }
};

struct FeFrameOut {               // parts-mode output of one frame, owned per batch index
    std::vector<FePart> parts;
    std::vector<FeSprite> sprites;
    std::vector<uint32_t> behind, sky_parts, bin_off, sbin_off;
    std::vector<uint16_t> bin_parts, sbin_sprites;
    uint32_t behind_words = 0, n_sky_slots = 0;
    DevFrame hdr{};
};

}  // namespace

struct dg_ctx {
    dg_config cfg{};
    FrameConsts fk{};
    DevConsts dk{};
    const Scene *scene = nullptr;
    size_t uploaded_texels = 0;         // texel pool size at dg_upload_scene time (grows when new sprite bitmaps are decoded)
    // device scene
    uint32_t *d_palette = nullptr;      // 256 x u32 RGBX, followed by 256 x (r, g, b, 0) f32
    uint8_t *d_texel_idx = nullptr, *d_texel_opq = nullptr, *d_flats = nullptr;
    unsigned long long *d_checksums = nullptr;   // dg_frame_checksums scratch, max_batch entries
    uint4 *d_row_tab = nullptr;         // per-row constants of the flat / sky mappers (dg_row_table), rebuilt per scene upload
    DevScene dscene{};
    std::vector<Slot> slots;
    hipStream_t kstream = nullptr;      // every kernel of every slot, in submission order (enqueue_kernels)
    hipStream_t rstream = nullptr;      // raster_overlap: the raster launches, so that the next batch's front-end kernels (kstream) run next to them
    bool raster_overlap = false;
    std::unique_ptr<Pool> pool;
    std::vector<std::unique_ptr<FrameArena>> arenas;   // one per worker (+ caller)
    std::vector<BinnedFrame> binned;                   // one per frame of a batch
    size_t span_cap_per_batch = 0, wall_cap_per_batch = 0, plane_cap_per_batch = 0;
    int n_threads = 1;
    // device column walk
    bool fe_enabled = false;            // cfg.front_end asks for it
    bool fe_scene_ok = false;           // ... and the uploaded scene allows it (sky bitmap >= 256x128, see bin_frame)
    // device seg walk (DG_FE_DEVICE_SEGS): the scene's per-seg tables + BSP tables in one allocation, per-batch scratch sized by the scene
    bool fs_enabled = false, fs_scene_ok = false;
    bool fs_rows_dirty = true;          // the seg walk's candidate rows may hold entries (fresh allocation, or a launch that failed half way)
    bool preparing = false;             // inside dg_prepare_views: the records are built once and replayed — host time is not in the loop
    bool fs_forced = false;             // DG_FE_DEVICE_SEGS: always; DG_FE_AUTO: when it is the faster way for the batch at hand (choose_fs)
    // what DG_FE_AUTO decides by (running means over batches of >= 64 frames, ms per frame): the host's per-seg half, and the whole of the
    // GPU work of a batch with / without the seg walk in it
    double ema_host = -1.0, ema_gpu_dev = -1.0, ema_gpu_fs = -1.0;
    int host_samples = 0;               // batches the host walker was timed on (the first one pays for cold caches and arena growth: not counted)
    int since_probe = 0;                // seg-walk batches since the host walker was last timed (it is timed again every 32 batches)
    int since_fs_probe = 0;             // host-walker batches since the seg walk was last timed (likewise)
    int gpu_samples[2] = {0, 0};        // finished batches seen per mode (host per-seg half / seg walk): the first of each runs on cold caches and clocks, not counted
    uint8_t *d_fs_scene = nullptr;
    uint8_t *d_fs_scratch = nullptr;    // occupancy rows (zero between batches) | candidate rows F x n_segs x 5 x 8 B | candidate lists + keep bits of frames beyond FS_CL_CAP
    size_t fs_zero_bytes = 0;
    FsParams fs_proto{};                // scene pointers and counts, filled at upload
    uint64_t fallbacks_fe = 0;          // batches in which frames were redone because a device-side capacity was exceeded (dg_ctx_fallbacks)
    uint64_t redone_frames = 0;         // frames redone through the host list path, one at a time (dg_ctx_redone_frames)
    DevRSpan *d_redo_rspans = nullptr;  // resolved spans of ONE frame being redone (allocated on first use)
    size_t redo_span_cap = 0;
    std::vector<FeFrameOut> fe_out;     // one per frame of a batch
    uint32_t fe_col_slots = FE_DEFAULT_COL_SLOTS;
    size_t fe_part_cap = 0, fe_sprite_cap = 0, fe_behind_cap = 0, fe_bin_cap = 0, fe_sbin_cap = 0, fe_slab_cap = 0;
    uint32_t *d_fe_cnt = nullptr;
    FeU4 *d_fe_cspans = nullptr;
    FeColRec *d_fe_recs = nullptr;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// CPUs' worth of run time the container allows this process (cgroup v2 cpu.max, v1 cfs quota), rounded up; 0 = no limit / unknown.
int cgroup_cpu_quota() {
    long long quota = -1, period = 0;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
        std::fclose(f);
    } else {
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)((quota + period - 1) / period);
}

// Everything queued for the slot so far has finished (its kernels run on the ctx's kernel stream, the rest on its own).
hipError_t slot_sync(Slot &s) {
    if (s.raster_recorded) { const hipError_t e = hipEventSynchronize(s.ev_raster); if (e != hipSuccess) return e; }
    return hipStreamSynchronize(s.stream);
}

void free_ctx(dg_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->kstream) (void)hipStreamSynchronize(c->kstream);        // every slot's kernels, before anything they use is freed
    if (c->rstream) (void)hipStreamSynchronize(c->rstream);
    for (Slot &s : c->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.h_lists) (void)hipHostFree(s.h_lists);
        if (s.d_lists) (void)hipFree(s.d_lists);
        if (s.d_rspans) (void)hipFree(s.d_rspans);
        if (s.d_fb) (void)hipFree(s.d_fb);
        if (s.h_fe) (void)hipHostFree(s.h_fe);
        if (s.d_fe) (void)hipFree(s.d_fe);
        if (s.d_fe_coloff) (void)hipFree(s.d_fe_coloff);
        if (s.d_order) (void)hipFree(s.d_order);
        if (s.d_flags) (void)hipFree(s.d_flags);    // (d_events lies inside this allocation)
        if (s.h_status) (void)hipHostFree(s.h_status);
        if (s.ev_start) (void)hipEventDestroy(s.ev_start);
        if (s.ev_setup) (void)hipEventDestroy(s.ev_setup);
        if (s.ev_raster) (void)hipEventDestroy(s.ev_raster);
        if (s.ev_rstart) (void)hipEventDestroy(s.ev_rstart);
        if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
        if (s.copy_stream) { (void)hipStreamSynchronize(s.copy_stream); (void)hipStreamDestroy(s.copy_stream); }
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (c->kstream) (void)hipStreamDestroy(c->kstream);
    if (c->rstream) (void)hipStreamDestroy(c->rstream);
    if (c->d_palette) (void)hipFree(c->d_palette);
    if (c->d_texel_idx) (void)hipFree(c->d_texel_idx);
    if (c->d_texel_opq) (void)hipFree(c->d_texel_opq);
    if (c->d_row_tab) (void)hipFree(c->d_row_tab);
    if (c->d_redo_rspans) (void)hipFree(c->d_redo_rspans);
    if (c->d_checksums) (void)hipFree(c->d_checksums);
    if (c->d_fe_cnt) (void)hipFree(c->d_fe_cnt);
    if (c->d_fe_cspans) (void)hipFree(c->d_fe_cspans);
    if (c->d_fe_recs) (void)hipFree(c->d_fe_recs);
    if (c->d_fs_scene) (void)hipFree(c->d_fs_scene);
    if (c->d_fs_scratch) (void)hipFree(c->d_fs_scratch);
    delete c;
}

// Build + bin the lists of n views in parallel, pack them into the slot's pinned slab, fill slot.P.
int build_batch_host(dg_ctx *c, Slot &s, const dg_view *views, const dg_frame_lists *given, int n, const dg_view_state *states = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!c->scene) return set_err(DG_ERR_INVALID, "no scene uploaded (dg_upload_scene)");
    if (n <= 0 || n > c->cfg.max_batch) return set_err(DG_ERR_CAPACITY, "batch size outside [1, max_batch]");
    const Scene &sc = *c->scene;
    if (sc.texel_idx.size() != c->uploaded_texels) return set_err(DG_ERR_INVALID, "the scene decoded new bitmaps since dg_upload_scene: upload it again");
    const int W = c->cfg.width, H = c->cfg.height;
    std::vector<int> rc((size_t)n, 0);
    std::vector<std::string> errs((size_t)n);
    c->pool->parallel_for(n, [&](int i, int wid) {
        BinnedFrame &bf = c->binned[(size_t)i];
        if (given) {
            dg_frame_lists fl = given[i];
            fill_view_trig(fl.view);
            rc[(size_t)i] = bin_frame(sc, c->fk, fl, bf, errs[(size_t)i]);
        } else {
            dg_view v = views[i];
            fill_view_trig(v);
            dg_frame_lists fl;
            rc[(size_t)i] = build_frame_lists(sc, W, H, v, *c->arenas[(size_t)wid], fl, errs[(size_t)i], states ? &states[i] : nullptr);
            if (!rc[(size_t)i]) rc[(size_t)i] = bin_frame(sc, c->fk, fl, bf, errs[(size_t)i]);
        }
    });
    for (int i = 0; i < n; i++)
        if (rc[(size_t)i]) return set_err(rc[(size_t)i], "frame " + std::to_string(i) + ": " + errs[(size_t)i]);

    // prefix sums -> bases
    uint64_t spans = 0, walls = 0, planes = 0, covered = 0;
    uint32_t max_spans = 0;
    for (int i = 0; i < n; i++) {
        BinnedFrame &bf = c->binned[(size_t)i];
        bf.hdr.span_base = (uint32_t)spans; bf.hdr.wall_base = (uint32_t)walls; bf.hdr.plane_base = (uint32_t)planes;
        spans += bf.spans.size(); walls += bf.walls.size(); planes += bf.planes.size(); covered += bf.covered_pixels;
        max_spans = std::max<uint32_t>(max_spans, (uint32_t)bf.spans.size());
    }
    if (spans > c->span_cap_per_batch || walls > c->wall_cap_per_batch || planes > c->plane_cap_per_batch)
        return set_err(DG_ERR_CAPACITY, "frame lists exceed the slot's list slab");
    const size_t off_frames = 0;
    const size_t off_col = align_up(off_frames + (size_t)n * sizeof(DevFrame), 256);
    const size_t off_walls = align_up(off_col + (size_t)n * (size_t)(W + 1) * 4, 256);
    const size_t off_planes = align_up(off_walls + walls * sizeof(DevWallRec), 256);
    const size_t off_spans = align_up(off_planes + planes * sizeof(DevPlaneRec), 256);
    const size_t total = off_spans + spans * sizeof(DevSpan);
    if (total > s.lists_cap) return set_err(DG_ERR_CAPACITY, "list slab too small");
    c->pool->parallel_for(n, [&](int i, int) {
        const BinnedFrame &bf = c->binned[(size_t)i];
        std::memcpy(s.h_lists + off_frames + (size_t)i * sizeof(DevFrame), &bf.hdr, sizeof(DevFrame));
        std::memcpy(s.h_lists + off_col + (size_t)i * (size_t)(W + 1) * 4, bf.col_off.data(), (size_t)(W + 1) * 4);
        if (!bf.walls.empty()) std::memcpy(s.h_lists + off_walls + (size_t)bf.hdr.wall_base * sizeof(DevWallRec), bf.walls.data(), bf.walls.size() * sizeof(DevWallRec));
        if (!bf.planes.empty()) std::memcpy(s.h_lists + off_planes + (size_t)bf.hdr.plane_base * sizeof(DevPlaneRec), bf.planes.data(), bf.planes.size() * sizeof(DevPlaneRec));
        if (!bf.spans.empty()) std::memcpy(s.h_lists + off_spans + (size_t)bf.hdr.span_base * sizeof(DevSpan), bf.spans.data(), bf.spans.size() * sizeof(DevSpan));
    });
    RasterParams &P = s.P;
    P.scene = c->dscene;
    P.k = c->dk;
    P.frames = reinterpret_cast<const DevFrame *>(s.d_lists + off_frames);
    P.col_off = reinterpret_cast<const uint32_t *>(s.d_lists + off_col);
    P.walls = reinterpret_cast<const DevWallRec *>(s.d_lists + off_walls);
    P.planes = reinterpret_cast<const DevPlaneRec *>(s.d_lists + off_planes);
    P.spans = reinterpret_cast<const DevSpan *>(s.d_lists + off_spans);
    P.rspans = s.d_rspans;
    P.fb = s.d_fb;
    P.row_tab = c->d_row_tab;
    P.n_frames = n;
    s.max_spans = max_spans; s.n_spans = spans; s.covered = covered; s.n_frames = n; s.n_walls = walls; s.n_planes = planes;
    s.list_bytes = total;
    s.fe_mode = false; s.fs_mode = false; s.fe_check = false;
    s.host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipMemcpyAsync(s.d_lists, s.h_lists, total, hipMemcpyHostToDevice, s.stream));
    return DG_OK;
}

// DG_FE_DEVICE: build the per-seg / per-sprite records of n views in parallel, pack them into the slot's record slab,
// fill slot.FP / slot.P.  Returns kPartsUnsupported when the batch has to go through build_batch_host instead.
int build_batch_fe(dg_ctx *c, Slot &s, const dg_view *views, int n, const dg_view_state *states) {
    const auto t0 = std::chrono::steady_clock::now();
    if (!c->scene) return set_err(DG_ERR_INVALID, "no scene uploaded (dg_upload_scene)");
    if (n <= 0 || n > c->cfg.max_batch) return set_err(DG_ERR_CAPACITY, "batch size outside [1, max_batch]");
    const Scene &sc = *c->scene;
    if (sc.texel_idx.size() != c->uploaded_texels) return set_err(DG_ERR_INVALID, "the scene decoded new bitmaps since dg_upload_scene: upload it again");
    const int W = c->cfg.width, H = c->cfg.height;
    std::vector<int> rc((size_t)n, 0);
    std::vector<std::string> errs((size_t)n);
    c->pool->parallel_for(n, [&](int i, int wid) {
        FrameArena &A = *c->arenas[(size_t)wid];
        FeFrameOut &o = c->fe_out[(size_t)i];
        dg_view v = views[i];
        fill_view_trig(v);
        rc[(size_t)i] = build_frame_parts(sc, W, H, v, A, errs[(size_t)i], states ? &states[i] : nullptr);
        if (rc[(size_t)i]) return;
        o.parts.swap(A.parts); o.sprites.swap(A.sprites); o.behind.swap(A.behind); o.sky_parts.swap(A.sky_parts);
        o.bin_off.swap(A.bin_off); o.bin_parts.swap(A.bin_parts); o.sbin_off.swap(A.sbin_off); o.sbin_sprites.swap(A.sbin_sprites);
        o.behind_words = A.behind_words; o.n_sky_slots = A.n_sky_slots;
        o.hdr = make_frame_header(v);
    });
    for (int i = 0; i < n; i++) {
        if (rc[(size_t)i] == kPartsUnsupported) return kPartsUnsupported;
        if (rc[(size_t)i]) return set_err(rc[(size_t)i], "frame " + std::to_string(i) + ": " + errs[(size_t)i]);
    }
    uint64_t parts = 0, sprites = 0, behind = 0, skies = 0, bins = 0, sbins = 0;
    uint32_t max_sky = 0;
    std::vector<FeFrame> ffs((size_t)n);
    for (int i = 0; i < n; i++) {
        const FeFrameOut &o = c->fe_out[(size_t)i];
        if (o.n_sky_slots > FE_MAX_SKY_SLOTS) return kPartsUnsupported;
        ffs[(size_t)i] = FeFrame{(uint32_t)parts, (uint32_t)o.parts.size(), (uint32_t)sprites, (uint32_t)o.sprites.size(), (uint32_t)behind,
                                 o.behind_words, o.n_sky_slots, (uint32_t)skies, (uint32_t)bins, (uint32_t)sbins, {0u, 0u}};
        parts += o.parts.size(); sprites += o.sprites.size(); behind += o.behind.size(); skies += o.n_sky_slots;
        bins += o.bin_parts.size(); sbins += o.sbin_sprites.size();
        max_sky = std::max(max_sky, o.n_sky_slots);
    }
    if (parts > c->fe_part_cap || sprites > c->fe_sprite_cap || behind > c->fe_behind_cap || bins > c->fe_bin_cap || sbins > c->fe_sbin_cap)
        return kPartsUnsupported;
    const size_t nb1 = (size_t)(W + FE_BIN_W - 1) / FE_BIN_W + 1;
    const uint32_t span_stride = (uint32_t)(c->span_cap_per_batch / (size_t)c->cfg.max_batch);
    const size_t off_frames = 0;
    const size_t off_ff = align_up(off_frames + (size_t)n * sizeof(DevFrame), 256);
    const size_t off_parts = align_up(off_ff + (size_t)n * sizeof(FeFrame), 256);
    const size_t off_sprites = align_up(off_parts + parts * sizeof(FePart), 256);
    const size_t off_behind = align_up(off_sprites + sprites * sizeof(FeSprite), 256);
    const size_t off_sky = align_up(off_behind + behind * 4, 256);
    const size_t off_boff = align_up(off_sky + skies * 4, 256);
    const size_t off_sboff = align_up(off_boff + (size_t)n * nb1 * 4, 256);
    const size_t off_bins = align_up(off_sboff + (size_t)n * nb1 * 4, 256);
    const size_t off_sbins = align_up(off_bins + bins * 2, 256);
    const size_t groups = (size_t)(W + 255) / 256;                  // dg_fe_columns: one workgroup per (frame, 256 columns)
    const size_t off_order = align_up(off_sbins + sbins * 2, 256);
    const size_t total = off_order + (size_t)n * groups * 4;
    if (total > c->fe_slab_cap) return kPartsUnsupported;
    {   // launch order of dg_fe_columns: heaviest workgroup first (weight = its longest bin: parts + 2 x sprites), counting sort
        std::vector<uint32_t> weight((size_t)n * groups), start(258, 0);
        for (int i = 0; i < n; i++) {
            const FeFrameOut &o = c->fe_out[(size_t)i];
            for (size_t g = 0; g < groups; g++) {
                uint32_t w = 0;
                for (size_t b = 4 * g; b < std::min(4 * g + 4, nb1 - 1); b++)
                    w = std::max(w, (o.bin_off[b + 1] - o.bin_off[b]) + 2u * (o.sbin_off[b + 1] - o.sbin_off[b]));
                w = 255u - std::min(w, 255u);                        // heaviest = smallest key
                weight[(size_t)i * groups + g] = w;
                start[w + 1]++;
            }
        }
        for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
        uint32_t *order = reinterpret_cast<uint32_t *>(s.h_fe + off_order);
        for (size_t it = 0; it < weight.size(); it++) order[start[weight[it]]++] = (uint32_t)it;
    }
    c->pool->parallel_for(n, [&](int i, int) {
        FeFrameOut &o = c->fe_out[(size_t)i];
        const FeFrame &ff = ffs[(size_t)i];
        o.hdr.span_base = (uint32_t)i * span_stride;
        std::memcpy(s.h_fe + off_frames + (size_t)i * sizeof(DevFrame), &o.hdr, sizeof(DevFrame));
        std::memcpy(s.h_fe + off_ff + (size_t)i * sizeof(FeFrame), &ff, sizeof(FeFrame));
        if (!o.parts.empty()) std::memcpy(s.h_fe + off_parts + (size_t)ff.part_base * sizeof(FePart), o.parts.data(), o.parts.size() * sizeof(FePart));
        if (!o.sprites.empty()) std::memcpy(s.h_fe + off_sprites + (size_t)ff.sprite_base * sizeof(FeSprite), o.sprites.data(), o.sprites.size() * sizeof(FeSprite));
        if (!o.behind.empty()) std::memcpy(s.h_fe + off_behind + (size_t)ff.behind_base * 4, o.behind.data(), o.behind.size() * 4);
        if (!o.sky_parts.empty()) std::memcpy(s.h_fe + off_sky + (size_t)ff.sky_base * 4, o.sky_parts.data(), o.sky_parts.size() * 4);
        std::memcpy(s.h_fe + off_boff + (size_t)i * nb1 * 4, o.bin_off.data(), nb1 * 4);
        std::memcpy(s.h_fe + off_sboff + (size_t)i * nb1 * 4, o.sbin_off.data(), nb1 * 4);
        if (!o.bin_parts.empty()) std::memcpy(s.h_fe + off_bins + (size_t)ff.bin_base * 2, o.bin_parts.data(), o.bin_parts.size() * 2);
        if (!o.sbin_sprites.empty()) std::memcpy(s.h_fe + off_sbins + (size_t)ff.sbin_base * 2, o.sbin_sprites.data(), o.sbin_sprites.size() * 2);
    });
    FeParams &F = s.FP;
    F.scene = c->dscene;
    F.k = c->dk;
    F.frames = reinterpret_cast<const DevFrame *>(s.d_fe + off_frames);
    F.fframes = reinterpret_cast<const FeFrame *>(s.d_fe + off_ff);
    F.parts = reinterpret_cast<const FePart *>(s.d_fe + off_parts);
    F.sprites = reinterpret_cast<const FeSprite *>(s.d_fe + off_sprites);
    F.behind = reinterpret_cast<const uint32_t *>(s.d_fe + off_behind);
    F.sky_parts = reinterpret_cast<const uint32_t *>(s.d_fe + off_sky);
    F.max_sky_slots = max_sky; F.gap_waves = 0;
    F.bin_off = reinterpret_cast<const uint32_t *>(s.d_fe + off_boff);
    F.sbin_off = reinterpret_cast<const uint32_t *>(s.d_fe + off_sboff);
    F.bin_parts = reinterpret_cast<const uint16_t *>(s.d_fe + off_bins);
    F.sbin_sprites = reinterpret_cast<const uint16_t *>(s.d_fe + off_sbins);
    F.order = reinterpret_cast<const uint32_t *>(s.d_fe + off_order); F.order_cnt = nullptr;
    F.cspans = c->d_fe_cspans; F.recs = c->d_fe_recs; F.cnt = c->d_fe_cnt;
    F.events = s.d_events;
    F.flags = s.d_flags;
    F.host_flags = s.h_status; F.totals = s.h_status + c->cfg.max_batch;  // pinned host memory, written by dg_fe_scan with plain stores
    F.col_off = s.d_fe_coloff; F.rspans = s.d_rspans;
    F.n_frames = n; F.span_stride = span_stride; F.w64 = (uint32_t)((W + 63) / 64); F.col_slots = c->fe_col_slots;
    RasterParams &P = s.P;
    P.scene = c->dscene;
    P.k = c->dk;
    P.frames = F.frames;
    P.col_off = s.d_fe_coloff;
    P.walls = nullptr; P.planes = nullptr; P.spans = nullptr;
    P.rspans = s.d_rspans;
    P.fb = s.d_fb;
    P.row_tab = c->d_row_tab;
    P.n_frames = n;
    s.max_spans = 0; s.n_spans = 0; s.covered = 0; s.n_frames = n; s.n_walls = parts; s.n_planes = sprites;
    s.list_bytes = total;
    s.fe_mode = true; s.fs_mode = false; s.fe_check = false;
    s.views.assign(views, views + n);
    s.keep_states(states, n);
    s.snapshot_scene(sc);
    s.host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (n >= 64 && !states && c->host_samples++ > 0) {        // DG_FE_AUTO's measurement of the host side (the first batch pays for cold caches: not counted)
        const double v = (double)s.host_ms / n;
        c->ema_host = c->host_samples == 2 ? v : 0.75 * c->ema_host + 0.25 * v;
    }
    HIP_TRY(hipMemcpyAsync(s.d_fe, s.h_fe, total, hipMemcpyHostToDevice, s.stream));
    return DG_OK;
}

// DG_FE_DEVICE_SEGS: the scene's per-seg / per-sprite tables and BSP tables (Scene::rebuild_fs_tables) in one device allocation, and the
// per-batch scratch of the seg walk, whose size follows the scene (segs, leaves).
int upload_fs_scene(dg_ctx *c, const Scene &sc) {
    if (c->d_fs_scene) { (void)hipFree(c->d_fs_scene); c->d_fs_scene = nullptr; }
    if (c->d_fs_scratch) { (void)hipFree(c->d_fs_scratch); c->d_fs_scratch = nullptr; }
    struct Piece { const void *src; size_t bytes; size_t at; };
    std::vector<Piece> pieces;
    size_t total = 0;
    auto add = [&](const void *src, size_t bytes) { pieces.push_back(Piece{src, bytes, total}); total = align_up(total + std::max<size_t>(bytes, 16), 256); return pieces.size() - 1; };
    const size_t i_segs = add(sc.fs_segs.data(), sc.fs_segs.size() * sizeof(FsSeg));
    const size_t i_leaf = add(sc.fs_seg_leaf.data(), sc.fs_seg_leaf.size() * 2);
    const size_t i_first = add(sc.fs_leaf_first.data(), sc.fs_leaf_first.size() * 4);
    const size_t i_sectors = add(sc.fs_sectors.data(), sc.fs_sectors.size() * sizeof(FsSector));
    const size_t i_anims = add(sc.fs_anims.data(), sc.fs_anims.size() * sizeof(FsAnim));
    const size_t i_bitmaps = add(sc.fs_bitmaps.data(), sc.fs_bitmaps.size() * sizeof(FsBitmap));
    const size_t i_sky = add(sc.flat_sky.data(), sc.flat_sky.size());
    const size_t i_mobjs = add(sc.fs_mobjs.data(), sc.fs_mobjs.size() * sizeof(FsMobj));
    const size_t i_sframes = add(sc.sprite_frames.data(), sc.sprite_frames.size() * sizeof(SpriteFrameRec));
    const size_t i_aoff = add(sc.fs_anc_off.data(), sc.fs_anc_off.size() * 4);
    const size_t i_anc = add(sc.fs_anc.data(), sc.fs_anc.size() * sizeof(FsAnc));
    HIP_TRY(hipMalloc((void **)&c->d_fs_scene, total));
    for (const Piece &p : pieces)
        if (p.bytes) HIP_TRY(hipMemcpy(c->d_fs_scene + p.at, p.src, p.bytes, hipMemcpyHostToDevice));
    FsParams &P = c->fs_proto;
    P = FsParams{};
    P.k = c->dk;
    auto at = [&](size_t i) { return c->d_fs_scene + pieces[i].at; };
    P.segs = reinterpret_cast<const FsSeg *>(at(i_segs)); P.seg_leaf = reinterpret_cast<const uint16_t *>(at(i_leaf)); P.leaf_first = reinterpret_cast<const uint32_t *>(at(i_first));
    P.sectors = reinterpret_cast<const FsSector *>(at(i_sectors)); P.anims = reinterpret_cast<const FsAnim *>(at(i_anims));
    P.bitmaps = reinterpret_cast<const FsBitmap *>(at(i_bitmaps)); P.flat_sky = at(i_sky);
    P.mobjs = reinterpret_cast<const FsMobj *>(at(i_mobjs)); P.sframes = reinterpret_cast<const FsSpriteFrame *>(at(i_sframes));
    P.anc_off = reinterpret_cast<const uint32_t *>(at(i_aoff)); P.anc = reinterpret_cast<const FsAnc *>(at(i_anc));
    P.n_segs = (uint32_t)sc.segs.size(); P.n_leaves = (uint32_t)sc.subsectors.size(); P.n_mobjs = (uint32_t)sc.mobjs.size();
    P.sprite_stride = std::min<uint32_t>(FS_SPRITE_CAP, std::max<uint32_t>(32u, (P.n_mobjs + 31u) / 32u * 32u));
    P.sbin_stride = std::min<uint32_t>(FS_SBIN_CAP, P.sprite_stride * (uint32_t)((c->cfg.width + FE_BIN_W - 1) / FE_BIN_W));
    // scratch: the occupancy rows (zero before every walk: dg_fs_frame leaves them so), then the candidate rows they index (never cleared)
    const size_t F = (size_t)c->cfg.max_batch;
    c->fs_zero_bytes = F * (size_t)fs_occ_words(P.n_segs) * 4;
    const size_t off_lite = align_up(c->fs_zero_bytes, 256);
    const size_t off_leaf = align_up(off_lite + F * (size_t)P.n_segs * FS_CALLS * sizeof(uint2), 256);
    // ... and, per frame, room for a candidate list longer than dg_fs_frame's shared memory holds (FS_CL_CAP) with its keep bits: sized by the
    // scene (every call of every seg), so that no frame of this map is handed back to the host for its number of candidates
    const uint32_t cl_row_cap = (P.n_segs * FS_CALLS + 31u) / 32u * 32u;
    const size_t off_cl = off_leaf;
    const size_t off_keep = align_up(off_cl + (cl_row_cap > FS_CL_CAP ? F * (size_t)cl_row_cap * 4 : 0), 256);
    HIP_TRY(hipMalloc((void **)&c->d_fs_scratch, off_keep + (cl_row_cap > FS_CL_CAP ? F * (size_t)(cl_row_cap / 32) * 4 : 0)));
    c->fs_rows_dirty = true;
    P.occ = reinterpret_cast<uint32_t *>(c->d_fs_scratch);
    P.lite = reinterpret_cast<uint2 *>(c->d_fs_scratch + off_lite);
    P.cl_rows = reinterpret_cast<uint32_t *>(c->d_fs_scratch + off_cl);
    P.keep_rows = reinterpret_cast<uint32_t *>(c->d_fs_scratch + off_keep);
    P.cl_row_cap = cl_row_cap > FS_CL_CAP ? cl_row_cap : 0u;
    c->fs_scene_ok = true;
    return DG_OK;
}

// DG_FE_AUTO: should this batch's per-seg half run on the GPU?  The host does it for free as long as it is done before the GPU has
// finished the batches queued ahead (its time hides under theirs); the GPU pays for it (dg_fs_*: ~0.1 ms per 1 000 frames) but needs
// no host time.  So: when nothing is in flight the host's time would be exposed in full — the GPU does it, unless the batch is so small
// that the kernels' fixed latency exceeds the host's few microseconds per frame; in a filled pipeline the GPU does it when the host has
// been measured to be the slower of the two (few host threads, small frames).  Small batches (< 64 frames) always go to the host walker.
void harvest_gpu_time(dg_ctx *c, Slot &s);
// First guess of the host walker's speed without spending a whole batch on it: a dozen of the batch's views on the calling thread
// (four untimed ones first), scaled by the pool size.  Whole batches that do go through the host walker refine it (build_batch_fe).
void calibrate_host(dg_ctx *c, const dg_view *views, int n) {
    const Scene &sc = *c->scene;
    FrameArena &A = *c->arenas[0];
    std::string err;
    const int warm = std::min(4, n), timed = std::min(8, n);
    for (int i = 0; i < warm; i++) { dg_view v = views[i]; fill_view_trig(v); (void)build_frame_parts(sc, c->cfg.width, c->cfg.height, v, A, err); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < timed; i++) { dg_view v = views[n - 1 - i]; fill_view_trig(v); (void)build_frame_parts(sc, c->cfg.width, c->cfg.height, v, A, err); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->ema_host = ms / timed / std::max(1, c->n_threads) * 1.25;     // (the pool does not scale perfectly)
    c->host_samples = std::max(c->host_samples, 2);
}
bool choose_fs(dg_ctx *c, const dg_view *views, int n) {
    if (c->fs_forced) return true;
    if (n < 64 || c->preparing) return false;
    if (c->ema_host < 0.0) calibrate_host(c, views, n);
    bool in_flight = false;
    for (Slot &s : c->slots) { harvest_gpu_time(c, s); in_flight |= s.busy; }
    if (!in_flight) return true;
    // No seg-walk batch timed yet: the GPU keeps the per-seg half until one has been (it costs the GPU ~0.1 ms per 1 000 frames; a batch on a
    // host that turns out to be the slower side costs the pipeline a millisecond).  From then on: whoever is the slower side of the pipeline.
    bool fs = c->ema_gpu_fs < 0.0 ? true : c->ema_host > c->ema_gpu_fs;
    // time the other side again now and then (the host's speed depends on who else uses the CPUs, the first seg-walk samples may have been
    // taken on a cold GPU); the host rarely when it was far behind
    if (fs && ++c->since_probe >= (c->ema_gpu_fs > 0.0 && c->ema_host > 2.0 * c->ema_gpu_fs ? 256 : 32)) fs = false;
    else if (!fs && ++c->since_fs_probe >= 32) fs = true;
    if (!fs) c->since_probe = 0; else c->since_fs_probe = 0;
    return fs;
}

// DG_FE_DEVICE_SEGS: nothing of the front end runs on the host.  Per frame it ships the view (trig filled) and the DevFrame header,
// per batch the scene's current light levels and map-object states; dg_fs_* then write the same record arrays build_batch_fe packs,
// with fixed per-frame strides (fs_frame.h), into the slot's record slab.
int build_batch_fs(dg_ctx *c, Slot &s, const dg_view *views, int n, const dg_view_state *states) {
    const auto t0 = std::chrono::steady_clock::now();
    if (n <= 0 || n > c->cfg.max_batch) return set_err(DG_ERR_CAPACITY, "batch size outside [1, max_batch]");
    const Scene &sc = *c->scene;
    if (sc.texel_idx.size() != c->uploaded_texels) return set_err(DG_ERR_INVALID, "the scene decoded new bitmaps since dg_upload_scene: upload it again");
    const int W = c->cfg.width;
    const size_t nb1 = (size_t)(W + FE_BIN_W - 1) / FE_BIN_W + 1;
    const uint32_t span_stride = (uint32_t)(c->span_cap_per_batch / (size_t)c->cfg.max_batch);
    // uploaded part
    const size_t off_frames = 0;
    const size_t off_views = align_up(off_frames + (size_t)n * sizeof(DevFrame), 256);
    const size_t off_lights = align_up(off_views + (size_t)n * sizeof(dg_view), 256);
    // per-view game state (dg_view_state): every frame gets its own copy of the two state arrays — the scene's values with the view's
    // entries on top — instead of one copy for the batch; the kernels index them with a per-frame stride
    const size_t state_frames = states ? (size_t)n : 1;
    const size_t off_mstate = align_up(off_lights + state_frames * sc.sectors.size() * 2, 256);
    const size_t upload = align_up(off_mstate + state_frames * sc.mobjs.size() * 4, 256);
    // device-written part
    const size_t off_ff = upload;
    const size_t off_parts = align_up(off_ff + (size_t)n * sizeof(FeFrame), 256);
    const size_t off_sprites = align_up(off_parts + (size_t)n * FS_PART_CAP * sizeof(FePart), 256);
    const size_t off_behind = align_up(off_sprites + (size_t)n * c->fs_proto.sprite_stride * sizeof(FeSprite), 256);
    const size_t off_sky = align_up(off_behind + (size_t)n * c->fs_proto.sprite_stride * FS_BEHIND_WORDS * 4, 256);
    const size_t off_boff = align_up(off_sky + (size_t)n * FS_SKY_CAP * 4, 256);
    const size_t off_sboff = align_up(off_boff + (size_t)n * nb1 * 4, 256);
    const size_t off_bins = align_up(off_sboff + (size_t)n * nb1 * 4, 256);
    const size_t off_sbins = align_up(off_bins + (size_t)n * FS_BIN_CAP * 2, 256);
    const size_t total = off_sbins + (size_t)n * c->fs_proto.sbin_stride * 2;
    if (total > c->fe_slab_cap) return kPartsUnsupported;
    s.views.assign(views, views + n);
    c->pool->parallel_for(n, [&](int i, int) {
        dg_view &v = s.views[(size_t)i];
        fill_view_trig(v);
        DevFrame hdr = make_frame_header(v);
        hdr.span_base = (uint32_t)i * span_stride;
        std::memcpy(s.h_fe + off_frames + (size_t)i * sizeof(DevFrame), &hdr, sizeof hdr);
        std::memcpy(s.h_fe + off_views + (size_t)i * sizeof(dg_view), &v, sizeof v);
    });
    int16_t *lights = reinterpret_cast<int16_t *>(s.h_fe + off_lights);
    int32_t *mstate = reinterpret_cast<int32_t *>(s.h_fe + off_mstate);
    std::atomic<int> bad_state{-1};
    c->pool->parallel_for((int)state_frames, [&](int i, int) {
        int16_t *l = lights + (size_t)i * sc.sectors.size();
        int32_t *m = mstate + (size_t)i * sc.mobjs.size();
        for (size_t k = 0; k < sc.sectors.size(); k++) l[k] = sc.sectors[k].light;
        for (size_t k = 0; k < sc.mobjs.size(); k++) m[k] = sc.mobjs[k].sprite_frame < 0 ? -1 : sc.mobjs[k].sprite_frame * 2 + (sc.mobjs[k].full_bright ? 1 : 0);
        if (!states) return;
        const dg_view_state &st = states[i];                        // the same rules as the host walker's (frontend.cpp: Walker::apply_state): later entries win
        for (uint32_t k = 0; k < st.n_lights; k++) {
            if (st.lights[k].sector < 0 || (size_t)st.lights[k].sector >= sc.sectors.size()) { bad_state = i; continue; }
            l[(size_t)st.lights[k].sector] = (int16_t)st.lights[k].light_level;
        }
        for (uint32_t k = 0; k < st.n_mobjs; k++) {
            const dg_mobj_state &ms = st.mobjs[k];
            if (ms.mobj < 0 || (size_t)ms.mobj >= sc.mobjs.size() || ms.sprite_frame >= (int32_t)sc.sprite_frames.size()) { bad_state = i; continue; }
            m[(size_t)ms.mobj] = (ms.sprite_frame < 0 ? -1 : ms.sprite_frame) * 2 + (ms.full_bright ? 1 : 0);
        }
    });
    if (bad_state >= 0) return set_err(DG_ERR_INVALID, "frame " + std::to_string(bad_state.load()) + ": view state: sector, map object or sprite frame index out of range");

    FsParams &Q = s.FSP;
    Q = c->fs_proto;
    Q.k = c->dk;
    Q.sector_light = reinterpret_cast<const int16_t *>(s.d_fe + off_lights);
    Q.mobj_state = reinterpret_cast<const int32_t *>(s.d_fe + off_mstate);
    Q.light_stride = states ? (uint32_t)sc.sectors.size() : 0u;
    Q.mstate_stride = states ? (uint32_t)sc.mobjs.size() : 0u;
    Q.views = reinterpret_cast<const dg_view *>(s.d_fe + off_views);
    Q.n_frames = n;
    Q.flags = s.d_flags;
    Q.fframes = reinterpret_cast<FeFrame *>(s.d_fe + off_ff);
    Q.parts = reinterpret_cast<FePart *>(s.d_fe + off_parts);
    Q.sprites = reinterpret_cast<FeSprite *>(s.d_fe + off_sprites);
    Q.behind = reinterpret_cast<uint32_t *>(s.d_fe + off_behind);
    Q.sky_parts = reinterpret_cast<uint32_t *>(s.d_fe + off_sky);
    Q.bin_off = reinterpret_cast<uint32_t *>(s.d_fe + off_boff);
    Q.sbin_off = reinterpret_cast<uint32_t *>(s.d_fe + off_sboff);
    Q.bin_parts = reinterpret_cast<uint16_t *>(s.d_fe + off_bins);
    Q.sbin_sprites = reinterpret_cast<uint16_t *>(s.d_fe + off_sbins);
    FeParams &F = s.FP;
    F.scene = c->dscene;
    F.k = c->dk;
    F.frames = reinterpret_cast<const DevFrame *>(s.d_fe + off_frames);
    F.fframes = Q.fframes; F.parts = Q.parts; F.sprites = Q.sprites; F.behind = Q.behind; F.sky_parts = Q.sky_parts;
    F.max_sky_slots = FS_SKY_CAP; F.gap_waves = 12;
    F.bin_off = Q.bin_off; F.sbin_off = Q.sbin_off; F.bin_parts = Q.bin_parts; F.sbin_sprites = Q.sbin_sprites;
    Q.order_cnt = s.d_flags + c->cfg.max_batch;            // zeroed with the flags (enqueue_kernels)
    Q.order_list = s.d_order;
    Q.n_items = (uint32_t)n * (uint32_t)((W + 255) / 256);
    F.order = s.d_order; F.order_cnt = Q.order_cnt;
    F.cspans = c->d_fe_cspans; F.recs = c->d_fe_recs; F.cnt = c->d_fe_cnt;
    F.events = s.d_events;
    F.flags = s.d_flags;
    F.host_flags = s.h_status; F.totals = s.h_status + c->cfg.max_batch;
    F.col_off = s.d_fe_coloff; F.rspans = s.d_rspans;
    F.n_frames = n; F.span_stride = span_stride; F.w64 = (uint32_t)((W + 63) / 64); F.col_slots = c->fe_col_slots;
    RasterParams &P = s.P;
    P.scene = c->dscene;
    P.k = c->dk;
    P.frames = F.frames;
    P.col_off = s.d_fe_coloff;
    P.walls = nullptr; P.planes = nullptr; P.spans = nullptr;
    P.rspans = s.d_rspans;
    P.fb = s.d_fb;
    P.row_tab = c->d_row_tab;
    P.n_frames = n;
    s.max_spans = 0; s.n_spans = 0; s.covered = 0; s.n_frames = n; s.n_walls = 0; s.n_planes = 0;
    s.list_bytes = upload;
    s.fe_mode = true; s.fs_mode = true; s.fe_check = false;
    s.keep_states(states, n);
    s.snapshot_scene(sc);
    s.host_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipMemcpyAsync(s.d_fe, s.h_fe, upload, hipMemcpyHostToDevice, s.stream));
    return DG_OK;
}

int build_batch(dg_ctx *c, Slot &s, const dg_view *views, const dg_frame_lists *given, int n, const dg_view_state *states = nullptr) {
    if (!given && c->fs_enabled && c->fs_scene_ok && choose_fs(c, views, n)) {
        const int rc = build_batch_fs(c, s, views, n, states);
        if (rc != kPartsUnsupported) return rc;
    }
    if (!given && c->fe_enabled && c->fe_scene_ok) {
        const int rc = build_batch_fe(c, s, views, n, states);
        if (rc != kPartsUnsupported) return rc;
    }
    return build_batch_host(c, s, views, given, n, states);
}

constexpr size_t kOverlapMaxPixels = 500000;          // frames up to this size overlap their raster launch with the next batch's front end (dg_create)

int enqueue_kernels(dg_ctx *c, Slot &s) {
    // All kernels of all slots run on ONE in-order stream (highest priority, so that it gets a hardware queue of its own): column walk
    // i, rasteriser i, column walk i + 1, ...  The slot's own stream carries its H2D copy (queued already; it overlaps the previous
    // slots' kernels), tied in with an event.  Letting the walk of batch i + 1 overlap the raster launch of batch i was measured to buy
    // nothing at 1280x800: its waves take slots a raster workgroup needs as a whole, the launch stretches by what the walk costs alone
    // (profiles/r03_column_walk.md) — for small frames it does pay, and the raster launches then go to a second stream (raster_overlap).  The walk's per-frame status words (overflow flags, span totals) live in pinned host memory and are
    // written by the kernels directly: nothing is queued behind the raster launch, so no stream ever holds a barrier that another
    // slot's upload could get stuck behind (streams share hardware queues).
    hipStream_t ks = c->kstream;
    // A HIP call that fails half way leaves the slot describing no submission at all: later calls on it return DG_ERR_INVALID
    // instead of waiting on a stale event or reading status words nobody wrote.
    struct Invalidate {
        Slot &s; bool armed = true;
        ~Invalidate() { if (armed) { s.n_frames = 0; s.timed = false; s.busy = false; s.fe_check = false; s.raster_recorded = false; } }
    } guard{s};
    const bool fe_mode = s.fe_mode;
    s.fe_check = false;
    HIP_TRY(hipEventRecord(s.ev_h2d, s.stream));
    HIP_TRY(hipStreamWaitEvent(ks, s.ev_h2d, 0));
    if (fe_mode) {
        std::memset(s.h_status, 0, (size_t)2 * (size_t)c->cfg.max_batch * 4);
        // the overflow flags, the launch-order counters and the event bits behind them start from zero: dg_fe_scan leaves them so (its
        // last lines), and only a slot that is new or whose last enqueue failed half way is cleared here, whole
        if (!s.walk_state_clean) HIP_TRY(hipMemsetAsync(s.d_flags, 0, s.walk_state_bytes, ks));
        s.walk_state_clean = false;
        if (s.fs_mode) {                                                                                // the seg walk writes what the column walk reads
            if (c->fs_rows_dirty) HIP_TRY(hipMemsetAsync(c->d_fs_scratch, 0, c->fs_zero_bytes, ks));    // (dg_fs_frame leaves its rows clean)
            c->fs_rows_dirty = true;
            HIP_TRY(launch_fs(s.FSP, ks, s.ev_start));
            c->fs_rows_dirty = false;
        }
        HIP_TRY(launch_fe(s.FP, ks, s.fs_mode ? nullptr : s.ev_start, s.ev_setup));
    } else {
        HIP_TRY(launch_setup(s.P, s.max_spans, ks, s.ev_start, s.ev_setup));
    }
    if (c->raster_overlap && fe_mode) {                   // the front end of the next batch may start while this launch runs (the column scratch is the front end's alone)
        HIP_TRY(hipStreamWaitEvent(c->rstream, s.ev_setup, 0));
        HIP_TRY(launch_raster(s.P, c->rstream, s.ev_rstart, s.ev_raster));
    } else {
        HIP_TRY(launch_raster(s.P, ks, s.ev_rstart, s.ev_raster));
    }
    if (fe_mode) s.walk_state_clean = true;               // everything was enqueued: dg_fe_scan will have cleaned up by the slot's next batch
    guard.armed = false;
    s.harvested = false;
    s.fe_check = fe_mode;
    s.raster_recorded = true;
    s.busy = true; s.timed = true;
    return DG_OK;
}

// One frame of a device-column-walk batch again, through the host list path, into its place in the slot's framebuffer.  The
// slot's own prepared state (records, column index, resolved spans of the other frames) is not touched: the host list slab of
// the slot, unused in device mode, carries the one frame, its resolved spans go to a ctx-wide scratch.  Returns DG_ERR_CAPACITY
// when the frame does not fit that scratch (the caller then redoes the whole batch).
int redo_frame_host(dg_ctx *c, Slot &s, int i) {
    const Scene &sc = *c->scene;
    const int W = c->cfg.width, H = c->cfg.height;
    BinnedFrame &bf = c->binned[0];
    std::string err;
    dg_view v = s.views[(size_t)i];
    fill_view_trig(v);
    dg_frame_lists fl;
    Slot::RedoState redo_state;
    int rc = build_frame_lists(sc, W, H, v, *c->arenas[0], fl, err, s.state_for_redo(sc, i, redo_state));
    if (!rc) rc = bin_frame(sc, c->fk, fl, bf, err);
    if (rc) return set_err(rc, "frame " + std::to_string(i) + ": " + err);
    bf.hdr.span_base = 0; bf.hdr.wall_base = 0; bf.hdr.plane_base = 0;
    const size_t off_col = align_up(sizeof(DevFrame), 256);
    const size_t off_walls = align_up(off_col + (size_t)(W + 1) * 4, 256);
    const size_t off_planes = align_up(off_walls + bf.walls.size() * sizeof(DevWallRec), 256);
    const size_t off_spans = align_up(off_planes + bf.planes.size() * sizeof(DevPlaneRec), 256);
    const size_t total = off_spans + bf.spans.size() * sizeof(DevSpan);
    if (total > s.lists_cap) return DG_ERR_CAPACITY;
    if (!c->d_redo_rspans) {
        c->redo_span_cap = (size_t)W * 64;                                   // 64 spans per column on average: far beyond any real frame
        if (hipMalloc((void **)&c->d_redo_rspans, c->redo_span_cap * sizeof(DevRSpan)) != hipSuccess) { c->d_redo_rspans = nullptr; return DG_ERR_CAPACITY; }
    }
    if (bf.spans.size() > c->redo_span_cap) return DG_ERR_CAPACITY;
    std::memcpy(s.h_lists, &bf.hdr, sizeof(DevFrame));
    std::memcpy(s.h_lists + off_col, bf.col_off.data(), (size_t)(W + 1) * 4);
    if (!bf.walls.empty()) std::memcpy(s.h_lists + off_walls, bf.walls.data(), bf.walls.size() * sizeof(DevWallRec));
    if (!bf.planes.empty()) std::memcpy(s.h_lists + off_planes, bf.planes.data(), bf.planes.size() * sizeof(DevPlaneRec));
    if (!bf.spans.empty()) std::memcpy(s.h_lists + off_spans, bf.spans.data(), bf.spans.size() * sizeof(DevSpan));
    HIP_TRY(hipMemcpyAsync(s.d_lists, s.h_lists, total, hipMemcpyHostToDevice, s.stream));
    RasterParams Q = s.P;
    Q.frames = reinterpret_cast<const DevFrame *>(s.d_lists);
    Q.col_off = reinterpret_cast<const uint32_t *>(s.d_lists + off_col);
    Q.walls = reinterpret_cast<const DevWallRec *>(s.d_lists + off_walls);
    Q.planes = reinterpret_cast<const DevPlaneRec *>(s.d_lists + off_planes);
    Q.spans = reinterpret_cast<const DevSpan *>(s.d_lists + off_spans);
    Q.rspans = c->d_redo_rspans;
    Q.fb = s.d_fb + (size_t)i * (size_t)3 * (size_t)W * (size_t)H;
    Q.n_frames = 1;
    HIP_TRY(launch_setup(Q, (uint32_t)bf.spans.size(), s.stream));
    HIP_TRY(launch_raster(Q, s.stream));
    HIP_TRY(slot_sync(s));                                 // the host slab is reused by the next frame
    return DG_OK;
}

// After the slot's stream has been synchronised: look at the overflow flags of a device-column-walk submission; a batch
// that overflowed a per-column / per-frame capacity is redone through the host list path (which has the larger limits).
int settle_slot(dg_ctx *c, Slot &s) {
    if (s.fe_check) {
        s.fe_check = false;
        bool overflow = false;
        uint64_t spans = 0;
        for (int i = 0; i < s.n_frames; i++) {
            overflow |= s.h_status[i] != 0;
            spans += s.h_status[c->cfg.max_batch + i];
        }
        s.n_spans = spans;
        bool whole_batch = overflow;
        if (overflow) {
            // Only the frames that overflowed are redone (through the host list path, one at a time); if one of them does not fit the
            // single-frame scratch either, the whole batch is.
            c->fallbacks_fe++;
            whole_batch = false;
            for (int i = 0; i < s.n_frames && !whole_batch; i++) {
                if (s.h_status[i] == 0) continue;
                const int rc = redo_frame_host(c, s, i);
                if (rc == DG_ERR_CAPACITY) whole_batch = true;
                else if (rc) return rc;
                else c->redone_frames++;
            }
        }
        if (whole_batch) {
            // (the whole batch again, with every frame's submit-time state: build_batch_host reads the states while it runs and keeps nothing of them)
            const std::vector<dg_view> views = s.views;
            std::vector<Slot::RedoState> redo_states(views.size());
            std::vector<dg_view_state> sts;
            bool any_state = false;
            for (size_t i = 0; i < views.size(); i++) {
                const dg_view_state *st = s.state_for_redo(*c->scene, (int)i, redo_states[i]);
                any_state |= st != nullptr;
                sts.push_back(st ? *st : dg_view_state{nullptr, 0, nullptr, 0});
            }
            int rc = build_batch_host(c, s, views.data(), nullptr, (int)views.size(), any_state ? sts.data() : nullptr);
            if (rc) return rc;
            rc = enqueue_kernels(c, s);
            if (rc) return rc;
            HIP_TRY(slot_sync(s));
        }
    }
    return DG_OK;
}

int enqueue_copy(dg_ctx *c, Slot &s) {
    const size_t fsz = (size_t)3 * (size_t)c->cfg.width * (size_t)c->cfg.height;
    HIP_TRY(hipStreamWaitEvent(s.copy_stream, s.ev_raster, 0));
    HIP_TRY(hipMemcpyAsync(s.copy_out, s.d_fb + (size_t)s.copy_first * fsz, (size_t)s.copy_count * fsz, hipMemcpyDeviceToHost, s.copy_stream));
    return DG_OK;
}

// DG_FE_AUTO's measurement of the GPU side: the span of a finished submission's kernels (never waits)
void harvest_gpu_time(dg_ctx *c, Slot &s) {
    if (s.harvested || !s.timed || !s.fe_mode || s.n_frames < 64 || c->fs_forced || !c->fs_enabled) return;
    if (hipEventQuery(s.ev_raster) != hipSuccess) return;
    s.harvested = true;
    float ms = 0.0f;
    if (c->raster_overlap) {                               // the raster launch may have waited behind the previous batch's: the two halves' own durations
        float fe_ms = 0.0f, r_ms = 0.0f;
        if (hipEventElapsedTime(&fe_ms, s.ev_start, s.ev_setup) != hipSuccess || hipEventElapsedTime(&r_ms, s.ev_rstart, s.ev_raster) != hipSuccess) return;
        ms = std::max(fe_ms, r_ms);                        // (they overlap with the neighbouring batches': the longer one sets the pace)
        if (!(ms > 0.0f)) return;
    } else if (hipEventElapsedTime(&ms, s.ev_start, s.ev_raster) != hipSuccess || !(ms > 0.0f)) return;
    if (c->gpu_samples[s.fs_mode ? 1 : 0]++ == 0) return;  // (the first batch of a mode: cold caches, code not yet resident, clocks down — a seg walk judged by it alone was never tried again)
    double &ema = s.fs_mode ? c->ema_gpu_fs : c->ema_gpu_dev;
    const double v = (double)ms / s.n_frames;
    ema = ema < 0.0 ? v : 0.75 * ema + 0.25 * v;
}

// Everything queued for the slot has finished: kernels, capacity checks (a batch that overflowed is redone here) and a
// pending asynchronous readback (re-issued after a redo: its first copy took frames of the overflowed run).
int finish_slot(dg_ctx *c, Slot &s) {
    HIP_TRY(slot_sync(s));
    harvest_gpu_time(c, s);
    s.busy = false;
    const uint64_t redone = c->fallbacks_fe;
    int rc = settle_slot(c, s);
    if (rc) return rc;
    if (s.copy_pending) {
        HIP_TRY(hipStreamSynchronize(s.copy_stream));
        if (c->fallbacks_fe != redone) {
            rc = enqueue_copy(c, s);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(s.copy_stream));
        }
        s.copy_pending = false;
    }
    return DG_OK;
}

int check_slot(dg_ctx *c, int slot) {
    if (!c) return set_err(DG_ERR_INVALID, "null ctx");
    if (slot < 0 || slot >= (int)c->slots.size()) return set_err(DG_ERR_INVALID, "slot out of range");
    return DG_OK;
}

}  // namespace

extern "C" {

const char *dg_last_error(void) { return t_err.c_str(); }
const char *dg_version(void) { return "doomgpu 0.4 (gfx950; ABI 4)"; }

int dg_scene_load_wad(const uint8_t *wad, size_t len, const char *map_name, dg_scene **out) {
    if (!wad || !map_name || !out) return set_err(DG_ERR_INVALID, "null argument");
    std::string err;
    Scene *sc = load_scene_from_wad(wad, len, map_name, err);
    if (!sc) return set_err(DG_ERR_WAD, err);
    *out = new dg_scene{sc};
    return DG_OK;
}
void dg_scene_free(dg_scene *s) { if (s) { delete s->sc; delete s; } }
int dg_scene_player_start(const dg_scene *s, float *x, float *y, float *angle) {
    if (!s || !x || !y || !angle) return set_err(DG_ERR_INVALID, "null argument");
    if (!s->sc->has_start) return set_err(DG_ERR_WAD, "Could not find thing of type 1 (src/map/things.rs:46-55)");
    *x = s->sc->start_x; *y = s->sc->start_y; *angle = s->sc->start_angle;
    return DG_OK;
}
int dg_scene_floor_height_at(const dg_scene *s, float x, float y, float *h) {
    if (!s || !h) return set_err(DG_ERR_INVALID, "null argument");
    int sec = s->sc->sector_from_vertex(x, y);
    if (sec < 0) return 1;
    *h = (float)s->sc->sectors[(size_t)sec].floor_h;
    return DG_OK;
}
int dg_scene_sector_count(const dg_scene *s) { return s ? (int)s->sc->sectors.size() : DG_ERR_INVALID; }
int dg_scene_set_sector_light(dg_scene *s, int sector, int16_t light) {
    if (!s || sector < 0 || (size_t)sector >= s->sc->sectors.size()) return set_err(DG_ERR_INVALID, "bad sector");
    s->sc->sectors[(size_t)sector].light = light;
    s->sc->revision++;
    return DG_OK;
}
int dg_scene_mobj_count(const dg_scene *s) { return s ? (int)s->sc->mobjs.size() : DG_ERR_INVALID; }
int dg_scene_set_mobj_state(dg_scene *s, int mobj, const char *sprite, uint8_t frame, int full_bright) {
    if (!s || mobj < 0 || (size_t)mobj >= s->sc->mobjs.size()) return set_err(DG_ERR_INVALID, "bad map object");
    MapObjectRec &m = s->sc->mobjs[(size_t)mobj];
    s->sc->revision++;
    if (!sprite) { m.sprite_frame = -1; return DG_OK; }
    std::string err;
    int sf = s->sc->find_or_add_sprite_frame(sprite, frame, err);
    if (sf < 0) return set_err(DG_ERR_WAD, err);
    m.sprite_frame = sf; m.full_bright = full_bright;
    return DG_OK;
}
int dg_scene_texture_id(const dg_scene *s, const char *name) { return (s && name) ? s->sc->texture_id(name) : DG_ERR_INVALID; }
int dg_scene_flat_id(const dg_scene *s, const char *name, float ts) { return (s && name) ? s->sc->flat_id(name, ts) : DG_ERR_INVALID; }
int dg_scene_sprite_bitmap_id(const dg_scene *s, const char *sprite, uint8_t frame, uint8_t rot) {
    return (s && sprite) ? s->sc->sprite_bitmap_id(sprite, frame, rot) : DG_ERR_INVALID;
}
int dg_scene_bitmap_size(const dg_scene *s, int bitmap, int *w, int *h) {
    if (!s || bitmap < 0 || (size_t)bitmap >= s->sc->bitmaps.size()) return set_err(DG_ERR_INVALID, "bad bitmap id");
    if (w) *w = s->sc->bitmaps[(size_t)bitmap].w;
    if (h) *h = s->sc->bitmaps[(size_t)bitmap].h;
    return DG_OK;
}

int dg_build_lists(const dg_scene *s, int width, int height, const dg_view *view, dg_frame_lists *out) {
    if (!s || !view || !out) return set_err(DG_ERR_INVALID, "null argument");
    static thread_local FrameArena arena;
    dg_view v = *view;
    fill_view_trig(v);
    std::string err;
    int rc = build_frame_lists(*s->sc, width, height, v, arena, *out, err);
    return rc ? set_err(rc, err) : DG_OK;
}

int dg_create(const dg_config *cfg, dg_ctx **out) {
    if (!cfg || !out) return set_err(DG_ERR_INVALID, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 16384 || cfg->height > 16384)
        return set_err(DG_ERR_INVALID, "width/height must be positive, both <= 16384");
    if (cfg->max_batch <= 0 || cfg->max_batch > 65535 || cfg->slots <= 0 || cfg->slots > 16)
        return set_err(DG_ERR_INVALID, "max_batch must be in [1, 65535], slots in [1, 16]");
    if (cfg->front_end < DG_FE_AUTO || cfg->front_end > DG_FE_DEVICE_SEGS) return set_err(DG_ERR_INVALID, "front_end must be DG_FE_AUTO, DG_FE_HOST, DG_FE_DEVICE or DG_FE_DEVICE_SEGS");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(DG_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return set_err(DG_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(DG_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 only");
    HIP_TRY(hipSetDevice(cfg->device));

    dg_ctx *c = new dg_ctx();
    c->cfg = *cfg;
    c->fk = make_consts(cfg->width, cfg->height);
    c->dk = DevConsts{c->fk.ARC, c->fk.GCFX, c->fk.CFX, c->fk.CFY, cfg->width, cfg->height};
    // Default: the process's CPU share — its affinity mask and, in a container, its cgroup CPU quota (threads beyond the quota only get
    // the process throttled) — capped at 16 per ctx: an 8-GPU node gives each rank ~1/8 of the cores.
    int nthreads = cfg->host_threads;
    if (nthreads <= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        int avail = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        const int quota = cgroup_cpu_quota();
        if (quota > 0) avail = std::min(avail, quota);
        nthreads = std::max(1, std::min(avail, 16));
    }
    nthreads = std::min(nthreads, 256);
    c->n_threads = nthreads;
    c->pool.reset(new Pool(nthreads - 1));
    for (int i = 0; i < nthreads; i++) c->arenas.emplace_back(new FrameArena());
    c->binned.resize((size_t)cfg->max_batch);

    const size_t W = (size_t)cfg->width, H = (size_t)cfg->height, F = (size_t)cfg->max_batch;
    c->span_cap_per_batch = F * W * 24;       // 24 spans per column on average; real scenes use 2-8
    c->wall_cap_per_batch = F * 4096;
    c->plane_cap_per_batch = F * 4096;
    const size_t lists_cap = align_up(F * sizeof(DevFrame), 256) + align_up(F * (W + 1) * 4, 256) +
                             align_up(c->wall_cap_per_batch * sizeof(DevWallRec), 256) +
                             align_up(c->plane_cap_per_batch * sizeof(DevPlaneRec), 256) + c->span_cap_per_batch * sizeof(DevSpan) + 1024;
    c->fe_enabled = cfg->front_end != DG_FE_HOST;
    c->fs_enabled = cfg->front_end == DG_FE_DEVICE_SEGS || cfg->front_end == DG_FE_AUTO;
    c->fs_forced = cfg->front_end == DG_FE_DEVICE_SEGS;
    if (c->fe_enabled) {
        // Scratch slots per screen column (spans and wall-record columns).  A column that needs more sends its batch through
        // the host list path; DOOMGPU_FE_COLUMN_SLOTS trades scratch HBM (24 B x slots x width x max_batch) against that.
        if (const char *e = std::getenv("DOOMGPU_FE_COLUMN_SLOTS")) {
            const long v = std::strtol(e, nullptr, 10);
            if (v >= 1 && v <= (long)FE_MAX_COL_SLOTS) c->fe_col_slots = (uint32_t)v;
        }
        c->fe_out.resize(F);
        size_t parts_per_frame = 2048;     // wall records per frame on average (e1m1-like maps ship 20-200 after culling)
        if (const char *e = std::getenv("DOOMGPU_FE_RECORDS_PER_FRAME")) {   // sizes the record slab; a batch that needs more goes through the host list path
            const long v = std::strtol(e, nullptr, 10);
            if (v >= 1 && v <= 65535) parts_per_frame = (size_t)v;
        }
        c->fe_part_cap = F * parts_per_frame;
        c->fe_sprite_cap = F * 256;
        c->fe_behind_cap = F * 256 * 32;   // one bit per (sprite, wall record)
        c->fe_bin_cap = F * 16384;         // column-bin entries (a record is listed in every 64-column strip it touches)
        c->fe_sbin_cap = F * 2048;
        c->fe_slab_cap = align_up(F * sizeof(DevFrame), 256) + align_up(F * sizeof(FeFrame), 256) + align_up(c->fe_part_cap * sizeof(FePart), 256) +
                         align_up(c->fe_sprite_cap * sizeof(FeSprite), 256) + align_up(c->fe_behind_cap * 4, 256) + align_up(F * FE_MAX_SKY_SLOTS * 4, 256) +
                         2 * align_up(F * ((W + FE_BIN_W - 1) / FE_BIN_W + 1) * 4, 256) + align_up(c->fe_bin_cap * 2, 256) + align_up(c->fe_sbin_cap * 2, 256) +
                         F * ((W + 255) / 256) * 4 + 1024;
    }
    c->slots.resize((size_t)cfg->slots);
    hipError_t e;
#define CTX_TRY(expr) if ((e = (expr)) != hipSuccess) { std::string m = std::string(#expr) + ": " + hipGetErrorString(e); free_ctx(c); return set_err(DG_ERR_HIP, m); }
    CTX_TRY(hipMalloc((void **)&c->d_checksums, F * 8));
    if (c->fe_enabled) {
        CTX_TRY(hipMalloc((void **)&c->d_fe_cspans, F * c->fe_col_slots * W * sizeof(FeU4)));
        CTX_TRY(hipMalloc((void **)&c->d_fe_recs, F * c->fe_col_slots * W * sizeof(FeColRec)));
        CTX_TRY(hipMalloc((void **)&c->d_fe_cnt, F * W * 4));
    }
    {
        int lo = 0, hi = 0;
        CTX_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CTX_TRY(hipStreamCreateWithPriority(&c->kstream, hipStreamNonBlocking, hi));
        // Small frames: the front-end kernels of a batch are chains of dependent steps that leave most of the chip idle, and at 320x200 they
        // last as long as the raster launch itself — letting the next batch's front end run next to this batch's raster launch (two streams,
        // tied by the front end's last dispatch event) is worth 17 % there (2.57 -> 3.02 M frames/s), 7 % at 640x400, 2 % at 800x600.  From 1024x768 up the raster launch
        // fills the chip and the overlap costs 1-2 % (profiles/r03_column_walk.md, r05_seg_walk.md).  DOOMGPU_RASTER_OVERLAP=0 / 1 overrides.
        c->raster_overlap = (size_t)cfg->width * (size_t)cfg->height <= kOverlapMaxPixels;
        if (const char *e = std::getenv("DOOMGPU_RASTER_OVERLAP")) c->raster_overlap = std::atoi(e) != 0;
        if (c->raster_overlap) CTX_TRY(hipStreamCreateWithPriority(&c->rstream, hipStreamNonBlocking, hi));
    }
    for (Slot &s : c->slots) {
        CTX_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        CTX_TRY(hipEventCreate(&s.ev_start));
        CTX_TRY(hipEventCreate(&s.ev_setup));
        CTX_TRY(hipEventCreate(&s.ev_raster));
        CTX_TRY(hipEventCreate(&s.ev_rstart));
        CTX_TRY(hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming));
        CTX_TRY(hipStreamCreateWithFlags(&s.copy_stream, hipStreamNonBlocking));
        CTX_TRY(hipHostMalloc((void **)&s.h_lists, lists_cap, hipHostMallocDefault));
        CTX_TRY(hipMalloc((void **)&s.d_lists, lists_cap));
        CTX_TRY(hipMalloc((void **)&s.d_rspans, c->span_cap_per_batch * sizeof(DevRSpan)));
        CTX_TRY(hipMalloc((void **)&s.d_fb, F * 3 * W * H));
        if (c->fe_enabled) {
            CTX_TRY(hipHostMalloc((void **)&s.h_fe, c->fe_slab_cap, hipHostMallocDefault));
            CTX_TRY(hipMalloc((void **)&s.d_fe, c->fe_slab_cap));
            CTX_TRY(hipMalloc((void **)&s.d_fe_coloff, F * (W + 1) * 4));
            s.flags_bytes = align_up(F * 4 + FS_ORDER_CLASSES * 4, 256);       // the flag words, then the seg walk's launch-order counters
            CTX_TRY(hipMalloc((void **)&s.d_order, FS_ORDER_CLASSES * F * ((W + 255) / 256) * 4));
            s.walk_state_bytes = s.flags_bytes + F * FE_MAX_SKY_SLOTS * 3 * ((W + 63) / 64) * 8;
            CTX_TRY(hipMalloc((void **)&s.d_flags, s.walk_state_bytes));
            s.d_events = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(s.d_flags) + s.flags_bytes);
            CTX_TRY(hipHostMalloc((void **)&s.h_status, 2 * F * 4, hipHostMallocDefault));
        }
        s.lists_cap = lists_cap;
    }
#undef CTX_TRY
    *out = c;
    return DG_OK;
}

void dg_destroy(dg_ctx *ctx) { free_ctx(ctx); }
int dg_ctx_host_threads(const dg_ctx *ctx) { return ctx ? ctx->n_threads : DG_ERR_INVALID; }

int dg_upload_scene(dg_ctx *c, const dg_scene *scene) {
    if (!c || !scene) return set_err(DG_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->cfg.device));
    for (Slot &s : c->slots) {             // nothing may still read the old scene or write into a buffer a pending readback is copying from
        // A slot with work in flight is FINISHED against the scene it was rendered from (still resident): a device-walk batch that
        // overflowed a capacity is redone and its pending readback re-issued, exactly as dg_wait would have done.  (If that redo is
        // impossible — the old scene object itself was changed since its upload — the slot is just drained.)
        if ((s.busy || s.copy_pending || s.fe_check) && c->scene && s.n_frames > 0 && finish_slot(c, s) == DG_OK) continue;
        HIP_TRY(slot_sync(s));
        HIP_TRY(hipStreamSynchronize(s.copy_stream));
        s.copy_pending = false;
    }
    const Scene &sc = *scene->sc;
    if (c->d_palette) { (void)hipFree(c->d_palette); c->d_palette = nullptr; }
    if (c->d_texel_idx) { (void)hipFree(c->d_texel_idx); c->d_texel_idx = nullptr; }
    if (c->d_texel_opq) { (void)hipFree(c->d_texel_opq); c->d_texel_opq = nullptr; }
    c->d_flats = nullptr;               // inside d_texel_idx's allocation
    // the slots' prepared records point into the device scene that was just freed: nothing may be replayed from them
    for (Slot &s : c->slots) { s.n_frames = 0; s.timed = false; s.fe_check = false; s.busy = false; s.snap_scene = nullptr; }   // (a new scene may reuse the old one's address and revision)
    uint32_t pal[256];
    for (int i = 0; i < 256; i++) pal[i] = (uint32_t)sc.palette[3 * i] | ((uint32_t)sc.palette[3 * i + 1] << 8) | ((uint32_t)sc.palette[3 * i + 2] << 16);
    const size_t nt = std::max<size_t>(sc.texel_idx.size(), 16), nf = std::max<size_t>(sc.flat_pool.size(), 16);
    float palf[256 * 4];
    for (int i = 0; i < 256; i++) { palf[4 * i] = (float)sc.palette[3 * i]; palf[4 * i + 1] = (float)sc.palette[3 * i + 1]; palf[4 * i + 2] = (float)sc.palette[3 * i + 2]; palf[4 * i + 3] = 0.0f; }
    HIP_TRY(hipMalloc((void **)&c->d_palette, sizeof pal + sizeof palf));
    // [column-major texel index plane | flats] share one allocation: the tile rasteriser gathers every kind's texel with one
    // 32-bit offset from texel_idx (flats at flats - texel_idx)
    const size_t flats_at = (nt + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void **)&c->d_texel_idx, flats_at + nf));
    HIP_TRY(hipMalloc((void **)&c->d_texel_opq, nt));
    c->d_flats = c->d_texel_idx + flats_at;
    HIP_TRY(hipMemcpy(c->d_palette, pal, sizeof pal, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_palette + 256, palf, sizeof palf, hipMemcpyHostToDevice));
    if (!sc.texel_idx.empty()) {
        HIP_TRY(hipMemcpy(c->d_texel_idx, sc.texel_idx.data(), sc.texel_idx.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_texel_opq, sc.texel_opq.data(), sc.texel_opq.size(), hipMemcpyHostToDevice));
    }
    if (!sc.flat_pool.empty()) HIP_TRY(hipMemcpy(c->d_flats, sc.flat_pool.data(), sc.flat_pool.size(), hipMemcpyHostToDevice));
    const BitmapInfo &sky = sc.bitmaps[(size_t)sc.sky_bitmap];
    c->dscene = DevScene{c->d_palette, reinterpret_cast<const float *>(c->d_palette + 256), c->d_texel_idx, c->d_texel_opq, c->d_flats, sky.texel_off, sky.w, sky.h, sky.has_holes};
    if (!c->d_row_tab) HIP_TRY(hipMalloc((void **)&c->d_row_tab, (size_t)c->cfg.height * sizeof(uint4)));
    HIP_TRY(launch_row_table(c->dscene, c->dk, c->d_row_tab, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    c->scene = &sc;
    c->fe_scene_ok = sky.w >= 256 && sky.h >= 128;    // a smaller sky bitmap is an index panic only when a sky visplane is drawn: host path
    c->uploaded_texels = sc.texel_idx.size();
    c->fs_scene_ok = false;
    if (c->fs_enabled && c->fe_scene_ok && sc.fs_ok && c->cfg.width <= FS_MAX_W) {
        const int rc = upload_fs_scene(c, sc);
        if (rc && c->fs_forced) return rc;                 // (DG_FE_AUTO simply keeps the host walker when the seg walk's memory cannot be had)
        if (rc) {
            static bool said = false;                      // once per process: the ctx works, but not the way it was asked to
            if (!said) { said = true; std::fprintf(stderr, "doomgpu: DG_FE_AUTO keeps the per-seg half on the host: no device memory for the seg walk's per-batch rows (%d views x %zu segs)\n", c->cfg.max_batch, sc.segs.size()); }
            if (c->d_fs_scene) { (void)hipFree(c->d_fs_scene); c->d_fs_scene = nullptr; }
            if (c->d_fs_scratch) { (void)hipFree(c->d_fs_scratch); c->d_fs_scratch = nullptr; }
            c->fs_scene_ok = false;
            (void)hipGetLastError();
        }
    }
    return DG_OK;
}

int dg_submit_views(dg_ctx *c, int slot, const dg_view *views, int n) { return dg_submit_views_state(c, slot, views, nullptr, n); }

int dg_scene_sprite_frame(dg_scene *s, const char *sprite, uint8_t frame) {
    if (!s || !sprite) return set_err(DG_ERR_INVALID, "null argument");
    std::string err;
    const int sf = s->sc->find_or_add_sprite_frame(sprite, frame, err);
    return sf < 0 ? set_err(DG_ERR_WAD, err) : sf;
}

int dg_render_views_state(dg_ctx *c, const dg_view *views, const dg_view_state *states, int n, uint8_t *out) {
    int rc = dg_submit_views_state(c, 0, views, states, n);
    if (rc) return rc;
    if (out) return dg_readback(c, 0, 0, n, out);
    return dg_wait(c, 0);
}

int dg_submit_views_state(dg_ctx *c, int slot, const dg_view *views, const dg_view_state *states, int n) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    if (!views) return set_err(DG_ERR_INVALID, "null views");
    if (states)
        for (int i = 0; i < n; i++)
            if ((states[i].n_lights && !states[i].lights) || (states[i].n_mobjs && !states[i].mobjs)) return set_err(DG_ERR_INVALID, "view state with a null array");
    HIP_TRY(hipSetDevice(c->cfg.device));
    Slot &s = c->slots[(size_t)slot];
    if (s.busy || s.copy_pending) { rc = finish_slot(c, s); if (rc) return rc; }
    s.fe_check = false;
    rc = build_batch(c, s, views, nullptr, n, states);
    if (rc) return rc;
    return enqueue_kernels(c, s);
}

int dg_wait(dg_ctx *c, int slot) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    return finish_slot(c, c->slots[(size_t)slot]);
}

int dg_readback_async(dg_ctx *c, int slot, int first, int count, uint8_t *out) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    Slot &s = c->slots[(size_t)slot];
    if (!out || first < 0 || count < 0 || first + count > s.n_frames) return set_err(DG_ERR_INVALID, "bad readback range");
    if (s.copy_pending) return set_err(DG_ERR_INVALID, "the slot already has a readback in flight (dg_wait it first)");
    HIP_TRY(hipSetDevice(c->cfg.device));
    s.copy_out = out; s.copy_first = first; s.copy_count = count;
    rc = enqueue_copy(c, s);
    if (rc) return rc;
    s.copy_pending = true;
    return DG_OK;
}

int dg_ctx_redone_frames(const dg_ctx *c, uint64_t *frames) {
    if (!c || !frames) return set_err(DG_ERR_INVALID, "null argument");
    *frames = c->redone_frames;
    return DG_OK;
}

int dg_ctx_fallbacks(const dg_ctx *c, uint64_t *front_end) {
    if (!c || !front_end) return set_err(DG_ERR_INVALID, "null argument");
    *front_end = c->fallbacks_fe;
    return DG_OK;
}

int dg_slot_framebuffer(dg_ctx *c, int slot, void **p) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    if (!p) return set_err(DG_ERR_INVALID, "null argument");
    *p = c->slots[(size_t)slot].d_fb;
    return DG_OK;
}

int dg_readback(dg_ctx *c, int slot, int first, int count, uint8_t *out) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    Slot &s = c->slots[(size_t)slot];
    if (!out || first < 0 || count < 0 || first + count > s.n_frames) return set_err(DG_ERR_INVALID, "bad readback range");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t fsz = (size_t)3 * (size_t)c->cfg.width * (size_t)c->cfg.height;
    HIP_TRY(slot_sync(s));
    if (s.fe_check) {
        rc = settle_slot(c, s);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(out, s.d_fb + (size_t)first * fsz, (size_t)count * fsz, hipMemcpyDeviceToHost, s.stream));
    HIP_TRY(slot_sync(s));
    s.busy = false;
    return DG_OK;
}

int dg_frame_checksums(dg_ctx *c, int slot, int first, int count, uint64_t *out) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    Slot &s = c->slots[(size_t)slot];
    if (!out || first < 0 || count < 0 || first + count > s.n_frames) return set_err(DG_ERR_INVALID, "bad frame range");
    if (count == 0) return DG_OK;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(slot_sync(s));                       // (the kernels run on the ctx's kernel stream: the copy below is not ordered behind them by its stream)
    if (s.fe_check) {
        rc = settle_slot(c, s);
        if (rc) return rc;
    }
    const size_t fsz = (size_t)3 * (size_t)c->cfg.width * (size_t)c->cfg.height;
    unsigned long long *d_sum = c->d_checksums;      // max_batch entries, allocated at dg_create
    hipError_t e = hipMemsetAsync(d_sum, 0, (size_t)count * 8, s.stream);
    if (e == hipSuccess) e = launch_checksums(s.d_fb + (size_t)first * fsz, fsz, count, d_sum, s.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_sum, (size_t)count * 8, hipMemcpyDeviceToHost, s.stream);
    if (e == hipSuccess) e = slot_sync(s);
    if (e != hipSuccess) return set_err(DG_ERR_HIP, std::string("dg_frame_checksums: ") + hipGetErrorString(e));
    s.busy = false;
    return DG_OK;
}

void *dg_alloc_host(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void dg_free_host(void *p) { if (p) (void)hipHostFree(p); }

int dg_render_views(dg_ctx *c, const dg_view *views, int n, uint8_t *out) {
    int rc = dg_submit_views(c, 0, views, n);
    if (rc) return rc;
    if (out) return dg_readback(c, 0, 0, n, out);
    return dg_wait(c, 0);
}

int dg_prepare_views(dg_ctx *c, int slot, const dg_view *views, int n) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    if (!views) return set_err(DG_ERR_INVALID, "null views");
    HIP_TRY(hipSetDevice(c->cfg.device));
    Slot &s = c->slots[(size_t)slot];
    if (s.busy || s.copy_pending) { rc = finish_slot(c, s); if (rc) return rc; }   // incl. a readback still copying out of the slot's framebuffer
    HIP_TRY(slot_sync(s));
    s.busy = false; s.fe_check = false;
    c->preparing = true;
    rc = build_batch(c, s, views, nullptr, n);
    c->preparing = false;
    if (rc) return rc;
    if (s.fe_mode) {              // run the column walk once so that a batch that has to go through the host list path as a whole
        rc = enqueue_kernels(c, s);   // is re-prepared that way now, not on a replay
        if (rc) return rc;
    }
    HIP_TRY(slot_sync(s));
    s.busy = false;
    return settle_slot(c, s);
}

int dg_replay_slot(dg_ctx *c, int slot) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    Slot &s = c->slots[(size_t)slot];
    if (s.n_frames <= 0) return set_err(DG_ERR_INVALID, "slot has no prepared lists");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (s.copy_pending) {   // a dg_readback_async is still reading the framebuffer these kernels are about to overwrite
        rc = finish_slot(c, s);
        if (rc) return rc;
    }
    if (s.fe_check) {   // a submission that was never waited for
        HIP_TRY(slot_sync(s));
        rc = settle_slot(c, s);
        if (rc) return rc;
    }
    return enqueue_kernels(c, s);  // (the overflow flags are looked at again: frames that overflowed are redone on every replay)
}

int dg_draw_lists(dg_ctx *c, int slot, const dg_frame_lists *frames, int n, uint8_t *out) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    if (!frames) return set_err(DG_ERR_INVALID, "null frames");
    HIP_TRY(hipSetDevice(c->cfg.device));
    Slot &s = c->slots[(size_t)slot];
    if (s.busy || s.copy_pending) { rc = finish_slot(c, s); if (rc) return rc; }
    s.fe_check = false;
    rc = build_batch(c, s, nullptr, frames, n);
    if (rc) return rc;
    rc = enqueue_kernels(c, s);
    if (rc) return rc;
    if (out) return dg_readback(c, slot, 0, n, out);
    return dg_wait(c, slot);
}

int dg_slot_timing(dg_ctx *c, int slot, dg_timing *out) {
    int rc = check_slot(c, slot);
    if (rc) return rc;
    if (!out) return set_err(DG_ERR_INVALID, "null argument");
    Slot &s = c->slots[(size_t)slot];
    if (!s.timed) return set_err(DG_ERR_INVALID, "slot has not run yet");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipEventSynchronize(s.ev_raster));
    if (s.fe_check) {
        HIP_TRY(slot_sync(s));
        rc = settle_slot(c, s);
        if (rc) return rc;
    }
    std::memset(out, 0, sizeof *out);
    HIP_TRY(hipEventElapsedTime(&out->setup_ms, s.ev_start, s.ev_setup));
    HIP_TRY(hipEventElapsedTime(&out->raster_ms, s.ev_rstart, s.ev_raster));
    HIP_TRY(hipEventElapsedTime(&out->total_ms, s.ev_start, s.ev_raster));
    out->n_spans = s.n_spans; out->n_frames = (uint64_t)s.n_frames; out->covered_pixels = s.covered;
    out->host_ms = s.host_ms; out->list_bytes = s.list_bytes;
    out->n_walls = s.n_walls; out->n_planes = s.n_planes;
    out->front_end = s.fs_mode ? DG_FE_DEVICE_SEGS : s.fe_mode ? DG_FE_DEVICE : DG_FE_HOST;
    return DG_OK;
}

}  // extern "C"
