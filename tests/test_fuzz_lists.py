"""Seeded random draw lists through the three texture mappers: records, columns and visplanes with arbitrary f32 / i16 contents (degenerate
lines, saturated extents, equal bottom / top, empty and out-of-frame columns, lights outside [0, 255], non-power-of-two and holey bitmaps,
sky, every draw order) — what the hand-built KATs of test_edge_kats.py aim at one by one, here in bulk.  The expected frame comes from
tests/np_mappers.py (shares no code with the oracle or the product; reference: bitmap_render.rs:190-276, visplanes.rs:42-130).
CPU tier: oracle (dr_draw_lists) == numpy.  GPU tier: dg_draw_lists == numpy.
"""
import numpy as np
import pytest

import np_mappers as nm
from test_edge_kats import to_dg_lists, view_dict, wall

SIZES = [(64, 40), (72, 70)]     # one tile; two strips (the second 8 columns wide) x two tile rows (the second with 6 live rows)
TEXTURES = ["BRICK1", "BRICK2", "BRICK3", "STONE2", "METAL2", "PANEL2", "WIDE2", "TALL72", "HOLEY1", "GRATE1", "COMBO2"]
FLATS = ["FLOOR0", "FLOOR1", "FLOOR3", "CEIL0", "CEIL2", "NUKAGE1", "F_SKY1"]
N_CASES = 24                     # per size


def random_case(seed, W, H):
    rng = np.random.default_rng(seed)
    f = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    view = (f(-4000, 4000), f(-4000, 4000), f(-7, 7), float(rng.integers(-200, 200)))
    columns, renders, planes = [], [], []
    for _ in range(int(rng.integers(3, 11))):
        kind = rng.integers(0, 8)
        if kind == 0:
            sx = f(-50, 300); sy = f(-200, 200); line = (sx, sy, sx, sy)                         # zero-length line
        elif kind == 1:
            line = (0.0, f(-30, 30), f(0.01, 50), f(-30, 30))                                    # starts on the view plane: uz0 == 0
        elif kind == 2:
            line = (f(1e4, 1e6), f(-1e6, 1e6), f(1e4, 1e6), f(-1e6, 1e6))                        # far away
        else:
            line = (f(0.01, 600), f(-400, 400), f(0.01, 600), f(-400, 400))
        start_x, end_x = int(rng.integers(-10, 70)), int(rng.integers(-10, 90))
        if rng.integers(0, 6) == 0:
            end_x = start_x
        cols = []
        for x in sorted(rng.choice(np.arange(-2, W + 3), size=int(rng.integers(1, W)), replace=False).tolist()):
            ct = int(rng.integers(0, H)); cb = int(rng.integers(0, H))
            if rng.integers(0, 8) != 0 and ct > cb:
                ct, cb = cb, ct                                                                  # (one in eight stays empty: ct > cb)
            style = rng.integers(0, 6)
            if style == 0:
                ty = by = int(rng.integers(-50, 90))                                             # bottom_y == top_y
            elif style == 1:
                ty, by = -32768, 32767
            else:
                ty = ct - int(rng.integers(0, 120)); by = cb + int(rng.integers(0, 120))
            cols.append((x, ct, cb, by, ty))
        renders.append(wall(str(rng.choice(TEXTURES)), int(rng.integers(-60, 360)), line, start_x, end_x, f(-600, 200), f(-200, 600), cols, columns,
                            offset_x=int(rng.integers(-400, 400)) if rng.integers(0, 4) else int(rng.choice([-32768, 32767])),
                            offset_y=int(rng.integers(-400, 400)) if rng.integers(0, 4) else int(rng.choice([-32768, 32767])),
                            start_offset=f(-100, 1000)))
    for _ in range(int(rng.integers(1, 6))):
        left = int(rng.integers(0, W)); right = int(rng.integers(left, W))
        tb = []
        for x in range(left, right + 1):
            t = int(rng.integers(-8, H)); b = t + int(rng.integers(0, 30)) if rng.integers(0, 5) else t + int(rng.integers(-2, 3))
            tb.append((t, b))
        planes.append({"flat": str(rng.choice(FLATS)), "height": int(rng.integers(-300, 300)), "light_level": int(rng.integers(-60, 360)), "left": left, "right": right, "tb": tb})
    order = [(0, i) for i in range(len(renders))] + [(1, i) for i in range(len(planes))]
    order = [order[i] for i in rng.permutation(len(order))]
    return view, {"renders": renders, "columns": columns, "visplanes": planes, "order": order}


CASES = {(W, H): [random_case(7000 + 100 * k + i, W, H) for i in range(N_CASES)] for k, (W, H) in enumerate(SIZES)}


@pytest.fixture(scope="module")
def expected(wad1993, campath_mod):
    np_wad = nm.Wad(wad1993)
    out = {}
    for (W, H), cases in CASES.items():
        out[(W, H)] = []
        for view, lists in cases:
            rec, vd = view_dict(campath_mod, *view)
            out[(W, H)].append((rec, nm.draw_lists(np_wad, "SKY1", W, H, vd, lists)))
    return out


@pytest.mark.parametrize("W,H", SIZES)
def test_oracle_equals_independent_restatement_on_random_lists(oracle_scene1993, expected, W, H):
    drawn = 0
    for i, ((view, lists), (rec, want)) in enumerate(zip(CASES[(W, H)], expected[(W, H)])):
        got = np.frombuffer(oracle_scene1993.draw_lists(W, H, rec, lists), dtype=np.uint8).reshape(H, W, 3)
        bad = np.argwhere(np.any(got != want, axis=2))
        assert len(bad) == 0, f"case {i}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): oracle {got[bad[0][0], bad[0][1]]} numpy {want[bad[0][0], bad[0][1]]}"
        drawn += int(want.any(axis=2).sum())
    assert drawn > N_CASES * W * H // 4                                       # the cases do draw


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", SIZES)
def test_gpu_equals_independent_restatement_on_random_lists(dg, wad1993, expected, W, H):
    """All cases in ONE dg_draw_lists batch (frame i must not depend on what frame i - 1 left in LDS)."""
    scene = dg.Scene(wad1993, "e1m1")
    ctx = dg.Context(W, H, max_batch=N_CASES, slots=1)
    ctx.upload_scene(scene)
    keep, frames = [], (dg.DgFrameLists * N_CASES)()
    for i, ((view, lists), (rec, want)) in enumerate(zip(CASES[(W, H)], expected[(W, H)])):
        fl, k = to_dg_lists(dg, scene, rec, lists)
        frames[i] = fl
        keep.append(k)
    out = ctx.draw_lists(0, frames)
    for i, (rec, want) in enumerate(expected[(W, H)]):
        bad = np.argwhere(np.any(out[i] != want, axis=2))
        assert len(bad) == 0, f"case {i}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): gpu {out[i][bad[0][0], bad[0][1]]} numpy {want[bad[0][0], bad[0][1]]}"
    ctx.close()
    scene.close()
