"""The device seg walk (DG_FE_DEVICE_SEGS: dg_fs_segs / dg_fs_frame, csrc/fs_frame.h) at its limits: every per-frame capacity driven into
overflow by a map built for it, the widest frame it takes and the first one it does not.  A frame that exceeds a capacity is flagged by the
kernels and redone by the host walker at dg_wait (context.cpp: redo_frame_host): same pixels, `redone_frames` counts it.
Reference behaviour at stake: src/renderer/mod.rs:61-104 (every seg of the map is visited, whatever their number), segs.rs:353-590,
map_objects.rs:19-241 (every map object in view is drawn)."""
import importlib
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
from test_hand_wad import build_polygon_iwad  # noqa: E402

FS_PART_CAP, FS_CL_CAP, FS_SPRITE_CAP, FS_SKY_CAP, FS_BIN_CAP, FS_MAX_W = 256, 1536, 512, 64, 4096, 2560   # csrc/fs_frame.h


def _capacities_from_header():
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "doom-rust-renderer_amd", "csrc", "fs_frame.h")).read()
    return {k: int(v) for k, v in re.findall(r"constexpr (?:uint32_t|int) (FS_[A-Z_]+) = (\d+);", src)}


def test_the_capacities_this_file_assumes_are_the_header_s():
    c = _capacities_from_header()
    assert (c["FS_PART_CAP"], c["FS_CL_CAP"], c["FS_SPRITE_CAP"], c["FS_SKY_CAP"], c["FS_BIN_CAP"], c["FS_MAX_W"]) == \
        (FS_PART_CAP, FS_CL_CAP, FS_SPRITE_CAP, FS_SKY_CAP, FS_BIN_CAP, FS_MAX_W)


LIMIT_VIEWS = {                     # map -> (builder, views (x, y, angle), frame size, the capacity bit tests/emul reports: emul.cpp emul_fs_frame)
    "sprites": (lambda: build_polygon_iwad(n_walls=64, radius=3000, n_things=900), [(-2900.0, 10.0, 0.02), (2880.0, -25.0, math.pi - 0.1)], (1280, 200), 4),
    "sky parts": (lambda: build_polygon_iwad(n_walls=240, radius=3000, ceil_flat="F_SKY1"), [(-2950.0, 10.0, 0.02), (100.0, -2940.0, math.pi / 2 + 0.3)], (1280, 200), 8),
    "candidates": (lambda: _limit_wad("long flight"), [(-32.0, 0.0, 0.0), (-20.0, 300.0, 0.05)], (640, 160), 2),
    "part bins": (lambda: _limit_wad("staircase"), [(-32.0, 0.0, 0.0), (-20.0, 300.0, 0.05)], (2560, 120), 16),
}


@pytest.mark.parametrize("which", sorted(LIMIT_VIEWS))
def test_each_limit_map_exceeds_the_capacity_it_is_named_after(campath_mod, which):
    """On the CPU (tests/emul runs dg_fs_frame's phases lane by lane): the device walk gives these frames up, the host walker completes them
    identically to the oracle, and the capacity exceeded is the one the map was built for."""
    import doomref
    import emul_bind
    build, pts, (W, H), bit = LIMIT_VIEWS[which]
    wad = build()
    osc = doomref.Scene(wad, "e1m1")
    es = emul_bind.EmulScene(wad)
    for (x, y, a) in pts:
        rec = _view(campath_mod, osc, x, y, a)
        if which == "candidates":             # FS_CL_CAP is the limit only where the ctx has no global candidate rows (a scene with few segs, or no memory for them)
            emul_bind.lib().emul_fs_no_cl_rows(1)
        try:
            rc, st = es.fs_frame(W, H, rec)
        finally:
            emul_bind.lib().emul_fs_no_cl_rows(0)
        assert rc == 2 and (st[4] & bit), (which, rc, st)
        if which == "candidates":             # with them (context.cpp: upload_fs_scene sizes them by the scene) the list is no reason any more: these frames then exceed FS_PART_CAP
            rc, st = es.fs_frame(W, H, rec)
            assert rc == 2 and not (st[4] & bit) and (st[4] & 1) and st[5] > FS_CL_CAP, (rc, st)
        assert es.render(W, H, rec)[0] == osc.render(W, H, rec)
    if which == "part bins":                                  # the same frames at half the width fit
        rc, st = es.fs_frame(1280, H, _view(campath_mod, osc, *pts[0]))
        assert rc == 0 and st[0] == 181, (rc, st)


def _view(campath_mod, osc, x, y, a):
    return np.concatenate([campath_mod.view_record(np.float32(x), np.float32(y), np.float32(a), np.float32(osc.floor_height_at(x, y, 0.0))), np.zeros(1, dtype=np.float32)])


def _render_all_front_ends(dg, osc, sc, views, W, H, expect_redone):
    """oracle == host lists == device column walk == device seg walk; the seg walk must have handed `expect_redone` or more frames back."""
    refs = [np.frombuffer(osc.render(W, H, r), dtype=np.uint8).reshape(H, W, 3) for r in views]
    arr = dg.make_views(np.stack([r[:8] for r in views]))
    for fe in (dg.DG_FE_HOST, dg.DG_FE_DEVICE, dg.DG_FE_DEVICE_SEGS):
        ctx = dg.Context(W, H, max_batch=64, slots=1, front_end=fe)       # (the list slabs scale with max_batch: these frames hold thousands of records)
        ctx.upload_scene(sc)
        out = ctx.render(arr)
        for k, ref in enumerate(refs):
            assert np.array_equal(out[k], ref), f"front end {fe}, {W}x{H}, view {k}"
        if fe == dg.DG_FE_DEVICE_SEGS:
            fb, used = ctx.fallbacks(), ctx.timing(0)["front_end"]
            # frame by frame through the host walker — or, when a flagged frame does not fit the single-frame scratch either (the flight of
            # 900 steps: half a million spans), the whole batch through the host list path, which timing then reports
            assert (used == dg.DG_FE_DEVICE_SEGS and fb["redone_frames"] >= expect_redone) or (expect_redone and used == dg.DG_FE_HOST and fb["front_end"] >= 1), (fb, used)
        ctx.close()


@pytest.mark.gpu
def test_more_visible_map_objects_than_the_seg_walk_holds(dg, campath_mod):
    """FS_SPRITE_CAP: an open hall (64 walls: few parts) with 900 imps and barrels on five rings; from the wall, looking across, most are in view."""
    import doomref
    wad = build_polygon_iwad(n_walls=64, radius=3000, n_things=900)
    osc = doomref.Scene(wad, "e1m1")
    assert osc.mobj_count() == 900
    sc = dg.Scene(wad, "e1m1")
    views = [_view(campath_mod, osc, -2900.0, 10.0, 0.02), _view(campath_mod, osc, 2880.0, -25.0, math.pi - 0.1), _view(campath_mod, osc, 0.0, 0.0, 0.7)]
    _render_all_front_ends(dg, osc, sc, views, 1280, 200, expect_redone=2)      # (from the centre a quarter of them: that frame stays on the GPU)
    sc.close()


@pytest.mark.gpu
def test_more_sky_parts_than_the_seg_walk_holds(dg, campath_mod):
    """FS_SKY_CAP: a round room of 240 walls under a sky ceiling; every wall in view adds to the sky visplane (segs.rs:293-345), 120 of them from the wall."""
    import doomref
    wad = build_polygon_iwad(n_walls=240, radius=3000, ceil_flat="F_SKY1")
    osc = doomref.Scene(wad, "e1m1")
    sc = dg.Scene(wad, "e1m1")
    views = [_view(campath_mod, osc, -2950.0, 10.0, 0.02), _view(campath_mod, osc, 100.0, -2940.0, math.pi / 2 + 0.3), _view(campath_mod, osc, 0.0, 0.0, 0.7)]
    _render_all_front_ends(dg, osc, sc, views, 1280, 200, expect_redone=2)
    sc.close()


def _staircase(m, rng, names, n=60, spacing=16, half_width=4096, rise=2, ceil0=2000, ceil_drop=0):
    """A very wide flight of n low steps: from its foot every riser runs across the whole frame (the side walls are outside the 90-degree
    frustum), so every part touches every column bin; no BSP partition (the risers' lines) splits anything.  A riser makes three
    process_sidedef calls that reach their column loop (segs.rs:493-588: lower part, the zero-height upper part, the opening), four when
    the ceiling comes down a step per sector as well (`ceil_drop`)."""
    tex = names["wall_textures"][0]
    xs = [-64] + [spacing * k for k in range(n + 1)]
    for k in range(n + 1):
        m.sectors.append(dict(floor=rise * k, ceil=ceil0 - ceil_drop * k, ffl=names["floor_flats"][k % len(names["floor_flats"])], cfl=names["ceil_flats"][0], light=130 + (k * 5) % 100, special=0, tag=0))
    w = half_width
    for k in range(n + 1):                                            # sector k: [xs[k], xs[k + 1]] x [-w, w]; clockwise = inside on the right
        x0, x1 = xs[k], xs[k + 1]
        m.line((x0, w), (x1, w), m.sidedef(k, "-", "-", tex), -1, 1)
        m.line((x1, -w), (x0, -w), m.sidedef(k, "-", "-", tex), -1, 1)
        if k == 0:
            m.line((x0, -w), (x0, w), m.sidedef(k, "-", "-", tex), -1, 1)
        if k == n:
            m.line((x1, w), (x1, -w), m.sidedef(k, "-", "-", tex), -1, 1)
        else:
            m.line((x1, w), (x1, -w), m.sidedef(k, tex, tex, "-"), m.sidedef(k + 1, tex, tex, "-"), 4)
    m.things.append((-32, 0, 0, 1, 7))
    m.things.append((spacing * n // 2, 40, 0, names["sprite_defs"][0][0], 7))


def _limit_wad(which):
    sw = importlib.import_module("doom-rust-renderer_amd.synth_wad")
    long_flight = lambda m, rng, names: _staircase(m, rng, names, n=900, spacing=4, half_width=8000, rise=1, ceil0=4000, ceil_drop=1)   # noqa: E731
    return sw.build_synth_iwad(7, custom_map={"long flight": long_flight, "staircase": _staircase}[which], vanilla=True)


@pytest.mark.gpu
def test_more_candidate_parts_than_the_seg_walk_stages(dg, campath_mod):
    """FS_CL_CAP: a flight of 900 steps under a ceiling that comes down as the floor goes up: from its foot four calls of every riser reach
    their column loop (3 600 candidates) before the hidden-part culling has a say."""
    import doomref
    wad = _limit_wad("long flight")
    osc = doomref.Scene(wad, "e1m1")
    sc = dg.Scene(wad, "e1m1")
    views = [_view(campath_mod, osc, -32.0, 0.0, 0.0), _view(campath_mod, osc, -20.0, 300.0, 0.05)]
    _render_all_front_ends(dg, osc, sc, views, 640, 160, expect_redone=2)
    sc.close()


@pytest.mark.gpu
def test_more_part_bin_pairs_than_the_seg_walk_holds(dg, campath_mod):
    """FS_BIN_CAP at 2 560 columns (40 column bins): 60 risers in view, three parts each, every part across the whole frame = 7 200
    (part, bin) pairs from 181 parts (fewer than FS_PART_CAP).  At 1 280 columns the same frames fit the seg walk (3 600 pairs: the CPU test
    above) — the column walk behind it then meets more spans per column than its scratch has slots and hands them back for its own reason."""
    import doomref
    wad = _limit_wad("staircase")
    osc = doomref.Scene(wad, "e1m1")
    assert osc.sector_count() == 61
    sc = dg.Scene(wad, "e1m1")
    views = [_view(campath_mod, osc, -32.0, 0.0, 0.0), _view(campath_mod, osc, -20.0, 300.0, 0.05)]
    _render_all_front_ends(dg, osc, sc, views, 2560, 120, expect_redone=2)
    _render_all_front_ends(dg, osc, sc, views, 1280, 120, expect_redone=0)
    sc.close()


@pytest.mark.gpu
def test_the_widest_frame_the_seg_walk_takes_and_the_first_it_does_not(dg, wad1994, oracle_scene1994, path1994):
    """FS_MAX_W: at 2 560 columns the seg walk runs; at 2 564 the context quietly uses the device column walk with the host's per-seg half
    (constants.rs:3-17 makes any width legal): timing says which, the pixels are the oracle's either way."""
    scene_heavy, oracle_scene_heavy, path_heavy = dg.Scene(wad1994, "e1m1"), oracle_scene1994, path1994
    idx = [0, 333, 711, 905]
    for W, want in ((FS_MAX_W, dg.DG_FE_DEVICE_SEGS), (FS_MAX_W + 4, dg.DG_FE_DEVICE)):
        H = 160
        ctx = dg.Context(W, H, max_batch=len(idx), slots=1, front_end=dg.DG_FE_DEVICE_SEGS)
        ctx.upload_scene(scene_heavy)
        out = ctx.render(dg.make_views(path_heavy[idx]))
        assert ctx.timing(0)["front_end"] == want, (W, ctx.timing(0))
        for k, i in enumerate(idx):
            assert np.array_equal(out[k], np.frombuffer(oracle_scene_heavy.render(W, H, path_heavy[i]), dtype=np.uint8).reshape(H, W, 3)), f"{W} columns, frame {i}"
        ctx.close()
    scene_heavy.close()


def test_more_candidates_than_shared_memory_holds_stay_on_the_gpu_walk(synth, campath_mod):
    """A map of doom2's scale seen down its long axes: up to ten thousand candidate parts per frame, a hundred of which survive the hidden-part
    culling.  dg_fs_frame keeps such a list in the frame's global rows (FsParams::cl_rows) instead of shared memory; on the CPU (tests/emul)
    its records equal the host walker's byte for byte, and only frames with more than FS_SPRITE_CAP map objects in view are still given up."""
    import doomref
    import emul_bind
    wad = synth.build_synth_iwad(2002, heavy=True, vanilla=True, grid=(32, 24), n_things=500)
    osc = doomref.Scene(wad, "e1m1")
    es = emul_bind.EmulScene(wad)
    route = synth.synth_route(2002, heavy=True, vanilla=True, grid=(32, 24), n_things=500)
    path = campath_mod.make_camera_path(route, osc.floor_height_at, 4000)[::40]
    big = given_up = 0
    for rec in path:
        rc, st = es.fs_frame(1280, 800, rec)
        assert rc == 0 or (rc == 2 and st[4] == 4), (rc, st)          # identical records, or too many sprites — never the candidate list
        big += rc == 0 and st[5] > FS_CL_CAP
        given_up += rc != 0
    assert big >= 20 and given_up <= 30, (big, given_up)


@pytest.mark.parametrize("which", ["sky parts", "part bins", "doom2 scale"])
def test_the_second_opinion_on_this_rounds_maps(synth, campath_mod, which):
    """The geometry the limit maps and the doom2-scale map add — a round room under a sky, a wide flight of steps, a 768-room lattice with long
    sight lines — through the numpy renderer (tests/np_front_end.py + np_mappers.py: shares no code with the oracle or the product) against
    the oracle, whole frames with their map objects: the oracle's treatment of these shapes is not only its own opinion."""
    import doomref
    import np_front_end as nf
    import np_mappers as nm
    if which == "doom2 scale":
        wad = synth.build_synth_iwad(2002, heavy=True, vanilla=True, grid=(32, 24), n_things=500)
        osc = doomref.Scene(wad, "e1m1")
        route = synth.synth_route(2002, heavy=True, vanilla=True, grid=(32, 24), n_things=500)
        recs = list(campath_mod.make_camera_path(route, osc.floor_height_at, 4000)[[0, 1333, 2777]])
    else:
        build, pts, _, _ = LIMIT_VIEWS[which]
        wad = build()
        osc = doomref.Scene(wad, "e1m1")
        recs = [_view(campath_mod, osc, x, y, a) for (x, y, a) in pts]
    W, H = 160, 100
    np_map, np_wad, things, sprites = nf.Map(wad, "e1m1"), nm.Wad(wad), nf.load_things(wad, "e1m1"), nf.SpriteTable(wad)
    assert len(things) == osc.mobj_count()
    for r in recs:
        view = {"x": r[0], "y": r[1], "angle": r[2], "cos": r[3], "sin": r[4], "cos_neg": r[5], "sin_neg": r[6], "floor_height": r[7]}
        got = nf.render_frame(np_map, things, sprites, np_wad, nm, W, H, view)
        want = np.frombuffer(osc.render(W, H, r), dtype=np.uint8).reshape(H, W, 3)
        bad = np.argwhere(np.any(got != want, axis=2))
        assert len(bad) == 0, f"{which}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]})"
