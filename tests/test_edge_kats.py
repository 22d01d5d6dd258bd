"""Edge-case known-answer tests: hand-built draw lists aimed at the arithmetic corners of the three
texture mappers, with the expected frame computed by tests/np_mappers.py — a second restatement that shares no code with the
oracle or the product.  CPU tier: oracle (dr_draw_lists) == numpy.  GPU tier: dg_draw_lists == numpy.

Cases (reference lines):
  vy == 0 on row H/2          visplanes.rs:109-123: wx, wy = +-inf / NaN -> `as i16` saturates / 0, distance saturated
  bottom_y == top_y           bitmap_render.rs:256-257: ay = 0/0 or +-inf -> ty from NaN / saturated (zero-height sectors, segs.rs:222-225)
  uz0 == 0                    bitmap_render.rs:242-243: 0.0 / uz0 = NaN -> tx = 0 before the offsets
  saturated extents           misc.rs:147-155: top_y / bottom_y at the i16 limits, rows clipped to the frame
  x >= W, x < 0               pixels.rs:22-25: dropped (negative x wraps through `as usize`)
  transparent texels          bitmap_render.rs:265 (HOLEY1 / GRATE1 patches with holes; COMBO2 with negative patch origins)
  bottom - top <= 1           visplanes.rs:98-101: column skipped for flats, drawn for sky (visplanes.rs:61-66)
  light outside [0, 255], negative distance       bitmap_render.rs:190-208: no upper clamp, `as u8` saturates
  negative / large texture offsets, non-power-of-two heights (TALL72)      bitmap_render.rs:244-248,259-263
"""
import numpy as np
import pytest

import np_mappers as nm

W, H = 64, 40            # CFY = 20: row 20 has vy == 0


def view_dict(campath_mod, x, y, angle, floor_height):
    rec = campath_mod.view_record(np.float32(x), np.float32(y), np.float32(angle), np.float32(floor_height))
    return rec, {"x": rec[0], "y": rec[1], "angle": rec[2], "cos": rec[3], "sin": rec[4], "floor_height": rec[7]}


def wall(texture, light, line, start_x, end_x, bottom_h, top_h, cols, columns, offset_x=0, offset_y=0, start_offset=0.0):
    first = len(columns)
    columns += cols
    return {"texture": texture, "light_level": light, "offset_x": offset_x, "offset_y": offset_y, "line": line, "start_offset": start_offset,
            "start_x": start_x, "end_x": end_x, "bottom_height": bottom_h, "top_height": top_h, "first_column": first, "n_columns": len(cols)}


def build_cases():
    """-> list of (name, view (x, y, angle, floor), lists)."""
    cases = []

    # 1. floor + ceiling planes across the horizon row (vy == 0), lights outside [0, 255], the 1-row skip rule, a sky plane
    columns = []
    planes = [
        {"flat": "FLOOR1", "height": 0, "light_level": 300, "left": 0, "right": 31, "tb": [(10 + (x % 3), 39) for x in range(32)]},          # crosses row 20
        {"flat": "CEIL2", "height": 128, "light_level": -20, "left": 20, "right": 63, "tb": [(0, 20 + (x % 2)) for x in range(20, 64)]},     # ends on / after row 20
        {"flat": "NUKAGE1", "height": -24, "light_level": 144, "left": 5, "right": 40, "tb": [(30, 30 + (x % 4)) for x in range(5, 41)]},    # bottom - top = 0, 1 (skipped), 2, 3
        {"flat": "F_SKY1", "height": 128, "light_level": 255, "left": 0, "right": 63, "tb": [(0, x % 3) for x in range(64)]},               # sky: 1- and 2-row columns are drawn
        {"flat": "FLOOR0", "height": 8, "light_level": 200, "left": 40, "right": 63, "tb": [(-5, 200)] * 24},                               # top / bottom outside the frame
    ]
    cases.append(("planes_across_the_horizon", (1000.3, -740.8, 0.7, 0.0), {"renders": [], "columns": columns, "visplanes": planes, "order": [(1, i) for i in range(len(planes))]}))

    # 2. walls: bottom_y == top_y, uz0 == 0, saturated extents, columns outside the frame, offsets, TALL72 modulus
    columns = []
    renders = [
        wall("BRICK1", 160, (100.0, -30.0, 180.0, 50.0), 0, 63, -41.0, 87.0, [(x, 5, 35, 35 + x // 8, 5 - x // 16) for x in range(0, 64)], columns),
        wall("TALL72", 255, (60.0, 10.0, 90.0, -20.0), 10, 50, -10.0, 62.0, [(x, 18, 22, 20, 20) for x in range(10, 30)], columns, offset_y=-7),   # bottom_y == top_y
        wall("WIDE2", 96, (0.0, -12.0, 40.0, 8.0), 0, 40, -41.0, 15.0, [(x, 25, 39, 39, 25) for x in range(0, 41)], columns, offset_x=-300, start_offset=13.7),  # uz0 == 0
        wall("PANEL2", 224, (5.0, 1.0, 5.25, -1.0), 30, 63, -2000.0, 2000.0, [(x, 0, 39, 32767, -32768) for x in range(30, 64)], columns, offset_y=30000),   # saturated extents
        wall("BRICK3", 128, (50.0, 5.0, 70.0, -5.0), 60, 70, -41.0, 40.0, [(x, 2, 12, 12, 2) for x in (-3, 60, 63, 64, 70, 32767, -32768)], columns),    # x >= W / x < 0 dropped
        wall("METAL2", 40, (300.0, 0.0, 3000.0, 900.0), 0, 63, -41.0, 300.0, [(x, 0, 10, 30, -20) for x in range(0, 64, 2)], columns, offset_x=32767, offset_y=-32768),
        wall("BRICK2", 192, (64.0, 0.0, 64.0, 0.0), 7, 7, -41.0, 87.0, [(7, 12, 30, 30, 12)], columns),                                    # zero-length line, start_x == end_x: ax = 0/0
    ]
    cases.append(("wall_corners", (0.0, 0.0, -2.1, 16.0), {"renders": renders, "columns": columns, "visplanes": [], "order": [(0, i) for i in range(len(renders))]}))

    # 3. overwrite order + transparency: plane, then walls with holes over it, then an opaque wall over those, then a plane again
    columns = []
    planes = [{"flat": "FLOOR3", "height": -8, "light_level": 176, "left": 0, "right": 63, "tb": [(0, 39)] * 64},
              {"flat": "CEIL0", "height": 96, "light_level": 112, "left": 24, "right": 40, "tb": [(8, 16)] * 17}]
    renders = [
        wall("HOLEY1", 208, (80.0, -40.0, 120.0, 40.0), 0, 63, -41.0, 87.0, [(x, 4, 36, 36, 4) for x in range(0, 64)], columns, offset_x=5),
        wall("GRATE1", 255, (40.0, -20.0, 44.0, 20.0), 8, 56, -41.0, 87.0, [(x, 0, 39, 45, -6) for x in range(8, 57)], columns, offset_y=-200),
        wall("COMBO2", 144, (200.0, -90.0, 230.0, 90.0), 0, 63, -41.0, 87.0, [(x, 10, 30, 33, 7) for x in range(16, 48)], columns, offset_x=-17),   # patches at negative origins
        wall("STONE2", 96, (90.0, 0.0, 91.0, 30.0), 30, 34, -41.0, 87.0, [(x, 0, 39, 39, 0) for x in range(30, 35)], columns),
    ]
    cases.append(("overwrite_order_and_holes", (-512.0, 2048.5, 3.9, -24.0),
                  {"renders": renders, "columns": columns, "visplanes": planes, "order": [(1, 0), (0, 0), (0, 1), (0, 2), (0, 3), (1, 1)]}))
    return cases


CASES = build_cases()


@pytest.fixture(scope="module")
def np_wad(wad1993):
    return nm.Wad(wad1993)


@pytest.fixture(scope="module")
def expected(np_wad, campath_mod):
    out = {}
    for name, v, lists in CASES:
        rec, vd = view_dict(campath_mod, *v)
        out[name] = (rec, nm.draw_lists(np_wad, "SKY1", W, H, vd, lists))
    return out


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_oracle_equals_independent_restatement(oracle_scene1993, expected, name):
    lists = next(c[2] for c in CASES if c[0] == name)
    rec, want = expected[name]
    got = np.frombuffer(oracle_scene1993.draw_lists(W, H, rec, lists), dtype=np.uint8).reshape(H, W, 3)
    assert want.any(), "the case draws nothing"
    bad = np.argwhere(np.any(got != want, axis=2))
    assert len(bad) == 0, f"{len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): oracle {got[bad[0][0], bad[0][1]]} numpy {want[bad[0][0], bad[0][1]]}"


def test_the_cases_hit_their_corners(np_wad, campath_mod):
    """The numpy restatement really goes through the non-finite paths the cases are built for."""
    f32 = np.float32
    with np.errstate(all="ignore"):
        fr = nm.Frame(W, H)
        vy = fr.CFY - f32(20)
        assert vy == 0 and np.isinf(fr.GCFX * f32(-41.0) / vy)                          # case 1, row 20
        assert np.isnan(f32(0) / f32(0)) and nm.f_as_i16(f32(0) / f32(0)) == 0           # case 2: bottom_y == top_y at y == top_y
        assert np.isnan(f32(0.0) / f32(0.0) * f32(1.0))                                  # uz0 == 0: 0.0 / uz0
    assert nm.f_as_i16(f32(np.inf)) == 32767 and nm.f_as_i16(f32(-np.inf)) == -32768
    w, h, rows = np_wad.texture("HOLEY1")
    assert any(t is None for r in rows for t in r)
    w, h, rows = np_wad.texture("COMBO2")
    assert (w, h) == (128, 128) and any(t is None for r in rows for t in r) and any(t is not None for r in rows for t in r)


def to_dg_lists(dg, scene, rec, lists):
    """the list dict -> one dg_frame_lists (ctypes), plus the arrays that must stay alive."""
    cols = (dg.DgBitmapColumn * max(1, len(lists["columns"])))(*[dg.DgBitmapColumn(*[int(np.int16(np.clip(t, -32768, 32767))) for t in c]) for c in lists["columns"]])
    rs = (dg.DgBitmapRender * max(1, len(lists["renders"])))()
    for i, r in enumerate(lists["renders"]):
        tid = dg.lib().dg_scene_texture_id(scene._h, r["texture"].encode())
        assert tid >= 0
        rs[i] = dg.DgBitmapRender(tid, r["light_level"], r["offset_x"], r["offset_y"], 0, *[float(t) for t in r["line"]], float(r["start_offset"]),
                                  r["start_x"], r["end_x"], float(r["bottom_height"]), float(r["top_height"]), r["first_column"], r["n_columns"])
    vs = (dg.DgVisplane * max(1, len(lists["visplanes"])))()
    tb = []
    for i, p in enumerate(lists["visplanes"]):
        fid = dg.lib().dg_scene_flat_id(scene._h, p["flat"].encode(), 0.0)
        vs[i] = dg.DgVisplane(fid, p["height"], p["light_level"], p["left"], p["right"], len(tb) // 2)
        for (t, b) in p["tb"]:
            tb += [t, b]
    import ctypes
    tba = (ctypes.c_int16 * max(1, len(tb)))(*tb)
    order = (dg.DgDrawCmd * max(1, len(lists["order"])))(*[dg.DgDrawCmd(k, i) for k, i in lists["order"]])
    fl = dg.DgFrameLists(dg.make_views(rec[None, :])[0], rs, len(lists["renders"]), cols, len(lists["columns"]), vs, len(lists["visplanes"]), tba, len(tb),
                         order, len(lists["order"]))
    return fl, (cols, rs, vs, tba, order)


@pytest.mark.gpu
def test_gpu_equals_independent_restatement(dg, wad1993, expected):
    """dg_draw_lists (the literal "host feeds lists" boundary) on the hand-built lists, all cases in one batch."""
    scene = dg.Scene(wad1993, "e1m1")
    ctx = dg.Context(W, H, max_batch=len(CASES), slots=1)
    ctx.upload_scene(scene)
    keep = []
    frames = (dg.DgFrameLists * len(CASES))()
    for i, (name, v, lists) in enumerate(CASES):
        fl, k = to_dg_lists(dg, scene, expected[name][0], lists)
        frames[i] = fl
        keep.append(k)
    out = ctx.draw_lists(0, frames)
    for i, (name, v, lists) in enumerate(CASES):
        want = expected[name][1]
        bad = np.argwhere(np.any(out[i] != want, axis=2))
        assert len(bad) == 0, f"{name}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): gpu {out[i][bad[0][0], bad[0][1]]} numpy {want[bad[0][0], bad[0][1]]}"
    ctx.close()
