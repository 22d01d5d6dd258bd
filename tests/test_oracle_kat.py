"""Known-answer tests of the oracle's scalar formulas (§0 cast semantics).

The expected values are derived here with numpy float32 following the cited reference lines — the
reference ships no vectors of its own ("parity unpinned")."""
import ctypes
import struct

import numpy as np
import pytest

f32 = np.float32


def bits(x) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def test_constants_table(oracle):
    # src/renderer/constants.rs:3-17 const-folded in f32; table from SURVEY.md §8 A1
    out = (ctypes.c_float * 5)()
    expect = {320: (384.0, 192.0, 160.0, 100.0, 200), 1024: (None, None, 512.0, 384.0, 768), 1280: (1536.0, 768.0, 640.0, 400.0, 800),
              2560: (3072.0, 1536.0, 1280.0, 800.0, 1600)}
    for W, (gsw, gcfx, cfx, cfy, H) in expect.items():
        oracle.lib().dr_constants(W, H, out)
        assert bits(out[0]) == 0x3F555555          # 200/240
        if gsw is not None:
            assert out[1] == gsw and out[2] == gcfx
        else:
            assert bits(out[1]) == 0x4499999A and bits(out[2]) == 0x4419999A   # 1228.8, 614.4
        assert out[3] == cfx and out[4] == cfy
        # independent numpy f32 evaluation
        arc = f32(200.0) / f32(240.0)
        assert f32(out[1]) == f32(W) / arc and f32(out[2]) == (f32(W) / arc) / f32(2.0)


@pytest.mark.parametrize("f,i16,i32,u8", [
    (0.0, 0, 0, 0), (-0.0, 0, 0, 0), (1.9, 1, 1, 1), (-1.9, -1, -1, 0), (255.9, 255, 255, 255), (256.0, 256, 256, 255),
    (32767.5, 32767, 32767, 255), (40000.0, 32767, 40000, 255), (-40000.0, -32768, -40000, 0),
    (3e9, 32767, 2147483647, 255), (-3e9, -32768, -2147483648, 0), (float("inf"), 32767, 2147483647, 255),
    (float("-inf"), -32768, -2147483648, 0), (float("nan"), 0, 0, 0), (2147483520.0, 32767, 2147483520, 255),
])
def test_rust_as_casts(oracle, f, i16, i32, u8):
    L = oracle.lib()
    assert L.dr_f32_as_i16(f) == i16
    assert L.dr_f32_as_i32(f) == i32
    assert L.dr_f32_as_u8(f) == u8


def np_diminish(rgb, light, dist):
    # bitmap_render.rs:190-208 in numpy f32
    factor = f32(light) / f32(255.0)
    factor = factor - f32(dist) * (f32(1.0) / (f32(16.0) * f32(256.0)))
    if factor < 0:
        factor = f32(0.0)
    out = []
    for c in rgb:
        v = f32(c) * factor
        out.append(int(min(255, max(0, int(v)))) if v == v else 0)
    return tuple(out), factor


@pytest.mark.parametrize("rgb,light,dist,expect", [
    ((255, 128, 64), 160, 512, (128, 64, 32)),
    ((255, 255, 255), 144, 1, (143, 143, 143)),
    ((200, 100, 50), 128, 3000, (0, 0, 0)),
    ((255, 255, 255), 300, 0, (255, 255, 255)),     # no upper clamp on factor; `as u8` saturates
    ((10, 20, 30), 255, -32768, (90, 180, 255)),    # negative distance brightens (factor 9.0)
    ((1, 2, 3), 0, 0, (0, 0, 0)),
])
def test_diminish_color(oracle, rgb, light, dist, expect):
    out = ctypes.create_string_buffer(3)
    oracle.lib().dr_diminish_color(bytes(rgb), light, dist, out)
    assert tuple(out.raw) == expect
    assert np_diminish(rgb, light, dist)[0] == expect


def test_diminish_color_sweep_matches_numpy(oracle):
    rng = np.random.default_rng(7)
    out = ctypes.create_string_buffer(3)
    for _ in range(3000):
        rgb = tuple(int(v) for v in rng.integers(0, 256, 3))
        light = int(rng.integers(-50, 400))
        dist = int(rng.integers(-32768, 32768))
        oracle.lib().dr_diminish_color(bytes(rgb), light, dist, out)
        assert tuple(out.raw) == np_diminish(rgb, light, dist)[0]


def test_spawn_tables_agree():
    """The thing type -> spawn state table exists twice, derived by two scripts that share no code: the product's
    (data/mobj_spawn_table.inc <- data/mobj_spawn.tsv <- tools/extract_mobj_table.py: states matched by name) and the oracle's
    (oracle/mobj_spawn_oracle.inc <- tools/extract_mobj_table_oracle.py: `STATES[spawn_state as usize]` by enum ordinal, as
    src/map_objects.rs:25-59 executes it).  A slip in one of them cannot hide in both sides of the parity tests: the rows must agree."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    row = re.compile(r'\{(-?\d+), "(\w+)", (\d+), (\d+), (\d+)\},')
    prod = row.findall(open(os.path.join(root, "data", "mobj_spawn_table.inc")).read())
    orac = row.findall(open(os.path.join(root, "oracle", "mobj_spawn_oracle.inc")).read())
    tsv = [tuple(l.split("\t")) for l in open(os.path.join(root, "data", "mobj_spawn.tsv")).read().splitlines() if l and not l.startswith("#")]
    assert len(prod) == len(orac) == len(tsv) > 100
    assert prod == orac == [tuple(t) for t in tsv]
    assert "mobj_spawn_oracle.inc" in open(os.path.join(root, "oracle", "doomref.c")).read()
    assert "data/mobj_spawn_table.inc" not in open(os.path.join(root, "oracle", "Makefile")).read()
