"""The paths of dg_raster_tiles that only dense scenes reach (the reference replays any number of records,
src/renderer/bitmap_render.rs:101-135, so the kernel's staging limits must not show):

  many_records      64x40, 24 overlapping wall records on every column (1 536 spans in the one 64-column strip, SPAN_CAP = 512):
                    the strip's spans do not fit the LDS staging area -> column groups are staged one after another, a wave owns fewer
                    than eight columns of a group (per-wave `nk`), every column has more than 8 spans (big_column_owner / _overlays)
  seventy_spans     one column carrying 70 spans, opaque and holey ones interleaved (second 64-span trip of both big-column loops),
                    its neighbours carrying 1, 8 and 9 spans (both sides of the pre-filter's 8-span limit)
  packed_rows_*     the same two at 64x72 = one full tile row + one with 8 live rows: the eight-rows-packed pass meets a strip that
                    does not fit, big columns and overlays, and has to leave them to the general path

Expected frames come from tests/np_mappers.py (shares no code with the oracle or the product).  CPU tier: oracle == numpy; GPU
tier: dg_draw_lists == numpy.
"""
import numpy as np
import pytest

import np_mappers as nm
from test_edge_kats import to_dg_lists, view_dict, wall

OPAQUE = ["BRICK1", "BRICK2", "BRICK3", "STONE2", "METAL2", "PANEL2", "WIDE2", "TALL72"]
HOLEY = ["HOLEY1", "GRATE1", "COMBO2"]


def many_records(W, H, n_rec=24):
    """n_rec wall records, each with a column on EVERY screen column; row ranges staggered so that owners change down the column,
    every third record a texture with holes (drawn immediately after the opaque ones below it in the order list)."""
    columns, renders = [], []
    for r in range(n_rec):
        holey = r % 3 == 2
        tex = HOLEY[(r // 3) % len(HOLEY)] if holey else OPAQUE[r % len(OPAQUE)]
        top0, span = (r * 5) % (H - 6), 6 + (r * 7) % (H // 2)
        cols = []
        for x in range(W):
            ct = max(0, min(H - 1, top0 + (x + r) % 4 - 1))
            cb = max(ct, min(H - 1, ct + span + (x * (r + 1)) % 3))
            cols.append((x, ct, cb, cb + (r % 5), ct - (r % 3)))
        renders.append(wall(tex, 96 + (r * 13) % 160, (40.0 + 6 * r, -30.0 + r, 140.0 - 3 * r, 35.0 - 2 * r), 0, W - 1, -41.0 + r, 87.0 - r, cols, columns,
                            offset_x=(r * 11) % 64 - 20, offset_y=(r * 17) % 90 - 30, start_offset=0.5 * r))
    return {"renders": renders, "columns": columns, "visplanes": [], "order": [(0, i) for i in range(n_rec)]}


def seventy_spans(W, H):
    """Column 21 gets 70 spans; columns 20, 22 and 23 get 1, 8 and 9; a floor plane under everything so that uncovered rows are not black."""
    columns, renders = [], []
    per_column = {20: 1, 21: 70, 22: 8, 23: 9}
    for r in range(70):
        holey = r % 4 == 1
        tex = HOLEY[r % len(HOLEY)] if holey else OPAQUE[r % len(OPAQUE)]
        cols = []
        for x, n in per_column.items():
            if r < n:
                ct = (r * 3 + x) % (H - 4)
                cb = min(H - 1, ct + 2 + (r % 9))
                cols.append((x, ct, cb, cb + 2, ct - 1))
        renders.append(wall(tex, 255 - 2 * r, (50.0 + r, -10.0, 60.0 + r, 12.0), 20, 23, -41.0, 87.0, cols, columns, offset_x=r, offset_y=-r))
    planes = [{"flat": "FLOOR1", "height": -16, "light_level": 160, "left": 0, "right": W - 1, "tb": [(0, H - 1)] * W}]
    return {"renders": renders, "columns": columns, "visplanes": planes, "order": [(1, 0)] + [(0, i) for i in range(70)]}


CASES = [
    ("many_records", 64, 40, (0.0, 0.0, -2.1, 16.0), many_records(64, 40)),
    ("seventy_spans", 64, 40, (-100.0, 300.0, 0.4, 0.0), seventy_spans(64, 40)),
    ("packed_rows_many_records", 64, 72, (0.0, 0.0, -2.1, 16.0), many_records(64, 72)),
    ("packed_rows_seventy_spans", 64, 72, (-100.0, 300.0, 0.4, 0.0), seventy_spans(64, 72)),
]


@pytest.fixture(scope="module")
def expected(wad1993, campath_mod):
    np_wad = nm.Wad(wad1993)
    out = {}
    for name, W, H, v, lists in CASES:
        rec, vd = view_dict(campath_mod, *v)
        out[name] = (rec, nm.draw_lists(np_wad, "SKY1", W, H, vd, lists))
    return out


def test_the_cases_are_as_dense_as_they_claim():
    name, W, H, v, lists = CASES[0]
    per_col = np.zeros(W, dtype=int)
    for c in lists["columns"]:
        per_col[c[0]] += 1
    assert per_col.min() >= 20 and per_col.sum() > 512                       # > SPAN_CAP spans in the strip, > 8 on every column
    name, W, H, v, lists = CASES[1]
    per_col = np.zeros(W, dtype=int)
    for c in lists["columns"]:
        per_col[c[0]] += 1
    assert per_col[21] == 70 and per_col[20] == 1 and per_col[22] == 8 and per_col[23] == 9
    assert CASES[2][2] - 64 == 8                                             # a tile row with exactly 8 live rows


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_oracle_equals_independent_restatement(oracle_scene1993, expected, name):
    _, W, H, _, lists = next(c for c in CASES if c[0] == name)
    rec, want = expected[name]
    got = np.frombuffer(oracle_scene1993.draw_lists(W, H, rec, lists), dtype=np.uint8).reshape(H, W, 3)
    assert want.any(), "the case draws nothing"
    bad = np.argwhere(np.any(got != want, axis=2))
    assert len(bad) == 0, f"{len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): oracle {got[bad[0][0], bad[0][1]]} numpy {want[bad[0][0], bad[0][1]]}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_gpu_equals_independent_restatement(dg, wad1993, expected, name):
    """dg_draw_lists on the dense lists: the staging-area overflow (column groups, per-wave nk), the 64-span trips of the big-column
    loops and the packed-rows pass next to them, against the numpy frame."""
    _, W, H, _, lists = next(c for c in CASES if c[0] == name)
    scene = dg.Scene(wad1993, "e1m1")
    ctx = dg.Context(W, H, max_batch=2, slots=1)
    ctx.upload_scene(scene)
    rec, want = expected[name]
    fl, keep = to_dg_lists(dg, scene, rec, lists)
    frames = (dg.DgFrameLists * 2)(fl, fl)                                   # twice: the second frame must not depend on what the first left in LDS
    out = ctx.draw_lists(0, frames)
    for i in range(2):
        bad = np.argwhere(np.any(out[i] != want, axis=2))
        assert len(bad) == 0, f"{name} frame {i}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): gpu {out[i][bad[0][0], bad[0][1]]} numpy {want[bad[0][0], bad[0][1]]}"
    ctx.close()
    scene.close()
