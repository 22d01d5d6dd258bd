"""A hand-assembled IWAD, built in this file with struct.pack from the byte layouts of SURVEY.md Appendix B — no import of
doom-rust-renderer_amd/synth_wad.py, whose output is the only WAD the two loaders (oracle/doomref.c, csrc/scene.cpp) and the numpy
restatement had ever parsed.  One map, two convex sectors joined by a NON-AXIS-ALIGNED portal that carries a masked middle texture,
upper and lower steps, a two-patch texture whose second patch punches transparent texels into the first (textures.rs:74-103), an
animated floor (flats.rs:30-111: NUKAGE1-3), a sky ceiling, a rotating sprite (eight rotations, sprites.rs:35-57) and a plain one.

    oracle (C)  ==  numpy renderer (tests/np_front_end.py + np_mappers.py)      CPU tier
    GPU         ==  oracle                                                      GPU tier, both front ends

References: src/wad.rs:57-63,86-195 (header, directory, map lumps by position), src/map/*.rs (record layouts),
src/graphics/pictures.rs:66-147 (picture / post format), src/graphics/textures.rs:182-255 (PNAMES / TEXTURE1),
src/graphics/flats.rs:116-136, src/graphics/sprites.rs:26-97, src/map_objects.rs:25-59 (things -> map objects)."""
import math
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _name8(s: str) -> bytes:
    b = s.encode("ascii")
    assert len(b) <= 8
    return b + b"\0" * (8 - len(b))


def _picture(w: int, h: int, left: int, top: int, texel) -> bytes:
    """Doom picture lump (pictures.rs:66-126): header w, h, left_offset, top_offset (i16 x 4), w column offsets (u32), then per column
    posts {ytop u8, len u8, pad u8, len texels, pad u8} ended by 0xFF.  texel(x, y) -> palette index or None (transparent)."""
    cols = []
    for x in range(w):
        col = b""
        y = 0
        while y < h:
            if texel(x, y) is None:
                y += 1
                continue
            y0 = y
            run = []
            while y < h and texel(x, y) is not None and len(run) < 120:
                run.append(texel(x, y))
                y += 1
            col += struct.pack("<BBB", y0, len(run), 0) + bytes(run) + b"\0"
        cols.append(col + b"\xff")
    hdr = struct.pack("<hhhh", w, h, left, top)
    off = len(hdr) + 4 * w
    table = b""
    for c in cols:
        table += struct.pack("<I", off)
        off += len(c)
    return hdr + table + b"".join(cols)


def _texture1(defs, pnames) -> bytes:
    """TEXTURE1 (textures.rs:208-255): u32 count, u32 offsets[], then maptexture: name[8], 4 B unused, w i16, h i16, 4 B unused,
    patchcount i16, patches {ox i16, oy i16, pname index i16, 4 B unused}."""
    bodies = []
    for (name, w, h, patches) in defs:
        b = _name8(name) + struct.pack("<I", 0) + struct.pack("<hh", w, h) + struct.pack("<I", 0) + struct.pack("<h", len(patches))
        for (ox, oy, pn) in patches:
            b += struct.pack("<hhhhh", ox, oy, pnames.index(pn), 1, 0)
        bodies.append(b)
    out = struct.pack("<I", len(bodies))
    off = 4 + 4 * len(bodies)
    for b in bodies:
        out += struct.pack("<I", off)
        off += len(b)
    return out + b"".join(bodies)


def _cross(a, b, c):
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


# vertices: room A = a0, p1, p2, a1 (counter-clockwise), room B = p1, b1, b2, p2 (counter-clockwise); the portal is the edge p1 - p2
VERT = {"a0": (0, 0), "a1": (-64, 320), "p1": (384, 96), "p2": (320, 400), "b1": (768, 32), "b2": (832, 448)}
VORDER = ["a0", "a1", "p1", "p2", "b1", "b2"]


def _graphics():
    playpal = bytes(v for i in range(256) for v in ((i * 7 + 3) % 256, (i * 13 + 40) % 256, (255 - i * 5) % 256))
    patches = {
        "PWALL": _picture(64, 128, 0, 0, lambda x, y: (x * 3 + y * 5) % 251 + 1),
        "PHOLE": _picture(64, 128, 0, 0, lambda x, y: None if (x // 8 + y // 8) % 3 == 0 else (x + 2 * y) % 200 + 20),
        "PSMALL": _picture(32, 64, 0, 0, lambda x, y: None if (x // 4 + y // 4) % 2 == 0 else (7 * x + y) % 100 + 100),
        "PSKY": _picture(256, 128, 0, 0, lambda x, y: (x // 2 + y) % 64 + 150),
    }
    pnames_list = list(patches)
    pnames = struct.pack("<I", len(pnames_list)) + b"".join(_name8(n) for n in pnames_list)
    texture1 = _texture1([
        ("WALLA", 64, 128, [(0, 0, "PWALL")]),
        ("TWOP", 64, 128, [(0, 0, "PWALL"), (16, 32, "PSMALL")]),          # the second patch's transparent texels overwrite the first's
        ("MASKED", 64, 128, [(0, 0, "PHOLE")]),
        ("STEP", 64, 128, [(0, 0, "PWALL"), (32, 64, "PWALL"), (-8, 100, "PSMALL")]),
        ("SKY1", 256, 128, [(0, 0, "PSKY")]),
    ], pnames_list)
    flats = {
        "FLOORA": bytes((x * 2 + y * 3) % 97 + 1 for y in range(64) for x in range(64)),
        "CEILA": bytes((x ^ y) % 61 + 120 for y in range(64) for x in range(64)),
        "NUKAGE1": bytes((x + y) % 16 + 32 for y in range(64) for x in range(64)),
        "NUKAGE2": bytes((x + 2 * y) % 16 + 64 for y in range(64) for x in range(64)),
        "NUKAGE3": bytes((2 * x + y) % 16 + 96 for y in range(64) for x in range(64)),
        "F_SKY1": bytes(200 for _ in range(4096)),
    }
    sprites = {}
    for rot in range(1, 9):                                   # a rotating sprite: eight different pictures (sprites.rs:35-57)
        sprites["TROOA%d" % rot] = _picture(24, 40, 12, 38, lambda x, y, r=rot: None if (x - 12) ** 2 + (y - 20) ** 2 > 150 + 10 * r else (x * r + y) % 60 + 10 * r)
    sprites["BAR1A0"] = _picture(20, 30, 10, 28, lambda x, y: None if (x in (0, 19) and y < 6) else (x + y) % 40 + 180)
    return playpal, pnames, texture1, patches, flats, sprites


def _pack_iwad(playpal, pnames, texture1, patches, flats, sprites, map_lumps) -> bytes:
    lumps = [("PLAYPAL", playpal), ("PNAMES", pnames), ("TEXTURE1", texture1)]
    lumps += [("P_START", b"")] + list(patches.items()) + [("P_END", b"")]
    lumps += [("F_START", b"")] + list(flats.items()) + [("F_END", b"")]
    lumps += [("S_START", b"")] + list(sprites.items()) + [("S_END", b"")]
    lumps += map_lumps
    # header (12 B: "IWAD", lump count, directory offset), lumps, directory (16 B: offset, size, name[8])
    body, directory = b"", b""
    for (name, data) in lumps:
        directory += struct.pack("<II", 12 + len(body), len(data)) + _name8(name)
        body += data
    return b"IWAD" + struct.pack("<II", len(lumps), 12 + len(body)) + body + directory


def build_hand_iwad() -> bytes:
    vi = {n: i for i, n in enumerate(VORDER)}
    A, B = ["a0", "p1", "p2", "a1"], ["p1", "b1", "b2", "p2"]
    for poly in (A, B):                                       # convex and counter-clockwise, so each is one BSP leaf
        for i in range(4):
            assert _cross(VERT[poly[i]], VERT[poly[(i + 1) % 4]], VERT[poly[(i + 2) % 4]]) > 0
    playpal, pnames, texture1, patches, flats, sprites = _graphics()
    # ---- the map -------------------------------------------------------------------------------------------------------------------
    # sectors (26 B: floor, ceiling, floor flat[8], ceiling flat[8], light, special, tag)
    sectors = struct.pack("<hh", 0, 128) + _name8("FLOORA") + _name8("CEILA") + struct.pack("<hhh", 160, 0, 0) + \
        struct.pack("<hh", 24, 104) + _name8("NUKAGE1") + _name8("F_SKY1") + struct.pack("<hhh", 208, 0, 0)
    # sidedefs (30 B: xoff, yoff, upper[8], lower[8], middle[8], sector)
    sd = []

    def side(xoff, yoff, upper, lower, middle, sector):
        sd.append(struct.pack("<hh", xoff, yoff) + _name8(upper) + _name8(lower) + _name8(middle) + struct.pack("<h", sector))
        return len(sd) - 1
    # one-sided walls run clockwise, so that the sector lies on their right (front) side
    walls = [("p1", "a0", side(0, 0, "-", "-", "WALLA", 0)), ("a0", "a1", side(5, 9, "-", "-", "TWOP", 0)), ("a1", "p2", side(-20, 3, "-", "-", "WALLA", 0)),
             ("b1", "p1", side(0, 0, "-", "-", "TWOP", 1)), ("b2", "b1", side(11, -7, "-", "-", "WALLA", 1)), ("p2", "b2", side(0, 64, "-", "-", "WALLA", 1))]
    # the portal p1 -> p2: room B on its right (front sidedef), room A on its left (back sidedef); masked middle texture on both sides,
    # and seen from A the step up to B's floor / down to B's ceiling needs a lower and an upper texture
    portal_front = side(0, 0, "-", "-", "MASKED", 1)
    portal_back = side(3, 0, "STEP", "STEP", "MASKED", 0)
    linedefs = b""
    for (v1, v2, s) in walls:                                  # 14 B: v1, v2, flags, special, tag, front sidedef, back sidedef
        linedefs += struct.pack("<hhhhhhh", vi[v1], vi[v2], 1, 0, 0, s, -1)
    linedefs += struct.pack("<hhhhhhh", vi["p1"], vi["p2"], 4, 0, 0, portal_front, portal_back)
    portal_ld = len(walls)
    # segs (12 B: v1, v2, angle, linedef, direction, offset): one per linedef side, no splits; subsector 0 = room A, 1 = room B
    segs_a = [(vi["p1"], vi["a0"], 0, 0), (vi["a0"], vi["a1"], 1, 0), (vi["a1"], vi["p2"], 2, 0), (vi["p2"], vi["p1"], portal_ld, 1)]
    segs_b = [(vi["b1"], vi["p1"], 3, 0), (vi["b2"], vi["b1"], 4, 0), (vi["p2"], vi["b2"], 5, 0), (vi["p1"], vi["p2"], portal_ld, 0)]
    segs = b"".join(struct.pack("<hhhhhh", v1, v2, 0, ld, d, 0) for (v1, v2, ld, d) in segs_a + segs_b)
    ssectors = struct.pack("<hh", len(segs_a), 0) + struct.pack("<hh", len(segs_b), len(segs_a))
    # one node: the partition is the portal line p1 -> p2; its right side is room B (subsector 1), its left side room A (subsector 0)
    def bbox(poly):
        xs, ys = [VERT[n][0] for n in poly], [VERT[n][1] for n in poly]
        return struct.pack("<hhhh", max(ys), min(ys), min(xs), max(xs))                   # top, bottom, left, right
    p1, p2 = VERT["p1"], VERT["p2"]
    assert _cross(p1, p2, VERT["b1"]) < 0 < _cross(p1, p2, VERT["a0"])                      # B right of the partition, A left
    nodes = struct.pack("<hhhh", p1[0], p1[1], p2[0] - p1[0], p2[1] - p1[1]) + bbox(B) + bbox(A) + struct.pack("<HH", 0x8000 | 1, 0x8000 | 0)
    vertexes = b"".join(struct.pack("<hh", *VERT[n]) for n in VORDER)
    # things (10 B: x, y, angle in degrees, type, flags): player 1 start, an imp in each room (3001, rotating), a barrel (2035)
    things = struct.pack("<hhhhh", 96, 180, 0, 1, 7) + struct.pack("<hhhhh", 600, 250, 135, 3001, 7) + struct.pack("<hhhhh", 250, 300, 270, 3001, 7) + \
        struct.pack("<hhhhh", 200, 120, 0, 2035, 7) + struct.pack("<hhhhh", 700, 120, 90, 2035, 7)
    return _pack_iwad(playpal, pnames, texture1, patches, flats, sprites,
                      [("E1M1", b""), ("THINGS", things), ("LINEDEFS", linedefs), ("SIDEDEFS", b"".join(sd)), ("VERTEXES", vertexes), ("SEGS", segs),
                       ("SSECTORS", ssectors), ("NODES", nodes), ("SECTORS", sectors), ("REJECT", b"\0"), ("BLOCKMAP", b"\0\0\0\0\0\0\0\0")])


def build_polygon_iwad(n_walls: int = 1200, radius: int = 3000, ceil_flat: str = "CEILA", n_things: int = 24) -> bytes:
    """One round room of n_walls one-sided walls (two BSP leaves: the halves above and below the x axis) with a ring of sprites: from
    a point next to the wall, looking across, several hundred walls are in view at once — more than 256 wall records in one frame, which
    is where the device column walk stops staging a sprite's behind-bit row (eight words) next to its record and the device seg walk
    hands the frame back to the host (FS_PART_CAP)."""
    playpal, pnames, texture1, patches, flats, sprites = _graphics()
    V = []
    for i in range(n_walls):                                  # counter-clockwise, integer coordinates, all distinct
        a = 2.0 * math.pi * i / n_walls                      # vertices 0 and n / 2 lie on the x axis: no wall straddles the partition
        V.append((int(round(radius * math.cos(a))), int(round(radius * math.sin(a)))))
    assert len(set(V)) == n_walls and n_walls % 2 == 0
    sectors = struct.pack("<hh", 0, 128) + _name8("FLOORA") + _name8(ceil_flat) + struct.pack("<hhh", 176, 0, 0)
    sidedefs, linedefs, upper, lower = b"", b"", [], []
    for i in range(n_walls):                                  # wall i runs clockwise: V[i + 1] -> V[i], the room on its right
        sidedefs += struct.pack("<hh", (i * 7) % 64, (i * 3) % 32) + _name8("-") + _name8("-") + _name8("WALLA" if i % 3 else "TWOP") + struct.pack("<h", 0)
        v1, v2 = (i + 1) % n_walls, i
        linedefs += struct.pack("<hhhhhhh", v1, v2, 1, 0, 0, i, -1)
        mid_y = V[v1][1] + V[v2][1]
        (upper if mid_y > 0 else lower).append((v1, v2, i, 0))
    assert len(upper) == len(lower) == n_walls // 2
    segs = b"".join(struct.pack("<hhhhhh", v1, v2, 0, ld, d, 0) for (v1, v2, ld, d) in upper + lower)
    ssectors = struct.pack("<hh", len(upper), 0) + struct.pack("<hh", len(lower), len(upper))
    # one node: partition along the x axis, direction +x: its right side is y < 0 (subsector 1), its left side y > 0 (subsector 0)
    r = radius + 1
    nodes = struct.pack("<hhhh", -r, 0, 2 * r, 0) + struct.pack("<hhhh", 0, -r, -r, r) + struct.pack("<hhhh", r, 0, -r, r) + struct.pack("<HH", 0x8000 | 1, 0x8000 | 0)
    vertexes = b"".join(struct.pack("<hh", *v) for v in V)
    things = struct.pack("<hhhhh", 0, 0, 0, 1, 7)
    for k in range(n_things):                                 # a ring of imps and barrels half way out, and a second one further out
        a = 2.0 * math.pi * k / n_things                      # (more than 24: five rings, so that they do not hide one another)
        rr = (radius // 2 if k % 2 else (3 * radius) // 4) if n_things <= 24 else radius * (3 + k % 5) // 8
        things += struct.pack("<hhhhh", int(rr * math.cos(a)), int(rr * math.sin(a)), (k * 45) % 360, 3001 if k % 3 else 2035, 7)
    return _pack_iwad(playpal, pnames, texture1, patches, flats, sprites,
                      [("E1M1", b""), ("THINGS", things), ("LINEDEFS", linedefs), ("SIDEDEFS", sidedefs), ("VERTEXES", vertexes), ("SEGS", segs),
                       ("SSECTORS", ssectors), ("NODES", nodes), ("SECTORS", sectors), ("REJECT", b"\0"), ("BLOCKMAP", b"\0\0\0\0\0\0\0\0")])


def _polygon_views(campath_mod, osc, radius: int = 3000):
    """Next to the wall looking across the room (half the walls in view), from the centre, and two oblique ones."""
    pts = [(-radius + 40.0, 10.0, 0.02), (radius - 60.0, -25.0, math.pi - 0.1), (0.0, 0.0, 0.7), (100.0, -radius + 50.0, math.pi / 2 + 0.3), (-1500.0, 900.0, -0.4)]
    return [np.concatenate([campath_mod.view_record(np.float32(x), np.float32(y), np.float32(a), np.float32(osc.floor_height_at(x, y, 0.0))), np.zeros(1, dtype=np.float32)])
            for (x, y, a) in pts]


def _views(campath_mod, osc):
    """(x, y, angle, timestamp): the start view, across the masked portal from both rooms, along it, next to it, with both animated-flat phases."""
    pts = [(96.0, 180.0, 0.0, 0.0), (96.0, 180.0, 0.35, 0.4), (600.0, 250.0, math.pi - 0.2, 0.0), (640.0, 330.0, math.pi + 0.5, 0.7),
           (300.0, 140.0, 1.2, 0.0), (420.0, 250.0, 2.9, 0.4), (200.0, 330.0, -0.6, 0.0), (352.5, 250.0, 0.1, 0.0)]
    out = []
    for (x, y, a, ts) in pts:
        rec = campath_mod.view_record(np.float32(x), np.float32(y), np.float32(a), np.float32(osc.floor_height_at(x, y, 0.0)))
        out.append((np.concatenate([rec, np.array([ts], dtype=np.float32)]), ts))
    return out


def test_hand_assembled_wad_oracle_equals_numpy_renderer(campath_mod):
    import doomref
    import np_front_end as nf
    import np_mappers as nm
    wad = build_hand_iwad()
    osc = doomref.Scene(wad, "e1m1")
    assert osc.sector_count() == 2 and osc.mobj_count() == 4
    sx, sy, sa = osc.player_start()
    assert (sx, sy, sa) == (96.0, 180.0, 0.0)
    np_map, np_wad, things, sprites = nf.Map(wad, "e1m1"), nm.Wad(wad), nf.load_things(wad, "e1m1"), nf.SpriteTable(wad)
    assert len(things) == 4
    # the two-patch texture really has holes where the second patch's transparent texels landed on the first patch
    tw, th, tpx = np_wad.texture("TWOP")
    assert (tw, th) == (64, 128) and tpx[32][16] is None and tpx[0][0] is not None and tpx[32][20] is not None
    W, H = 160, 100
    frames = []
    for (rec, ts) in _views(campath_mod, osc):
        view = {"x": rec[0], "y": rec[1], "angle": rec[2], "cos": rec[3], "sin": rec[4], "cos_neg": rec[5], "sin_neg": rec[6], "floor_height": rec[7]}
        got = nf.render_frame(np_map, things, sprites, np_wad, nm, W, H, view, timestamp=ts)
        want = np.frombuffer(osc.render(W, H, rec), dtype=np.uint8).reshape(H, W, 3)
        bad = np.argwhere(np.any(got != want, axis=2))
        assert len(bad) == 0, f"view {rec[:3]} t={ts}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): numpy {got[bad[0][0], bad[0][1]]} oracle {want[bad[0][0], bad[0][1]]}"
        assert (want != 0).any(axis=2).mean() > 0.5
        frames.append(want)
    # the animated floor shows: the same viewpoint at another timestamp is another frame
    rec0 = _views(campath_mod, osc)[2][0].copy()
    a = np.frombuffer(osc.render(W, H, rec0), dtype=np.uint8)
    rec0[8] = 0.4
    assert not np.array_equal(a, np.frombuffer(osc.render(W, H, rec0), dtype=np.uint8))


def test_hand_assembled_wad_product_loader_and_host_front_end(campath_mod):
    """The product's own loader (csrc/scene.cpp) and front end + kernel bodies on the CPU (tests/emul), both column-walk variants."""
    import doomref
    import emul_bind
    wad = build_hand_iwad()
    osc = doomref.Scene(wad, "e1m1")
    es = emul_bind.EmulScene(wad)
    for (W, H) in ((320, 200), (132, 68)):
        for (rec, ts) in _views(campath_mod, osc):
            ref = osc.render(W, H, rec)
            assert es.render(W, H, rec, ts)[0] == ref, (W, H, rec[:3], ts)
            got, st = es.render_fe(W, H, rec, ts)
            assert got == ref and st[3] == 0 and st[4] == 1, (W, H, rec[:3], ts, st)
            rc, st = es.fs_frame(W, H, rec, ts)                      # the device seg walk's bodies against the host walker's records
            assert rc == 0 and st[0] > 0, (W, H, rec[:3], ts, rc, st)


@pytest.mark.gpu
@pytest.mark.parametrize("front_end", [1, 2, 3], ids=["host-lists", "device-column-walk", "device-seg-walk"])
def test_hand_assembled_wad_on_the_gpu(dg, campath_mod, front_end):
    import doomref
    wad = build_hand_iwad()
    osc = doomref.Scene(wad, "e1m1")
    sc = dg.Scene(wad, "e1m1")
    assert sc.player_start() == osc.player_start() and sc.sector_count() == 2 and sc.mobj_count() == 4
    views = _views(campath_mod, osc)
    for (W, H) in ((320, 200), (1280, 800)):
        ctx = dg.Context(W, H, max_batch=len(views), slots=1, front_end=front_end)
        ctx.upload_scene(sc)
        arr = dg.make_views(np.stack([r[:8] for (r, _) in views]))
        for k, (_, ts) in enumerate(views):
            arr[k].timestamp = ts
        out = ctx.render(arr)
        assert ctx.timing(0)["front_end"] == front_end
        for k, (rec, ts) in enumerate(views):
            ref = np.frombuffer(osc.render(W, H, rec), dtype=np.uint8).reshape(H, W, 3)
            assert np.array_equal(out[k], ref), f"{W}x{H} view {k} t={ts}"
        ctx.close()


def test_round_room_of_1200_walls_on_the_cpu(campath_mod):
    """More wall records in one frame than a sprite's staged behind-bit row holds (256): oracle == product loader + host front end +
    the column walk's bodies on the CPU; the frames really are that crowded."""
    import doomref
    import emul_bind
    wad = build_polygon_iwad()
    osc = doomref.Scene(wad, "e1m1")
    assert osc.sector_count() == 1 and osc.mobj_count() == 24
    es = emul_bind.EmulScene(wad)
    for (W, H) in ((1280, 96), (320, 64)):                    # 320 wide: a part per column, 64 and more in every 64-column bin
        most = 0
        for rec in _polygon_views(campath_mod, osc):
            ref = osc.render(W, H, rec)
            assert es.render(W, H, rec, 0.0)[0] == ref, (W, rec[:3])
            got, st = es.render_fe(W, H, rec, 0.0)
            assert got == ref and st[3] == 0 and st[4] == 1, (W, rec[:3], st)
            most = max(most, st[1])
        assert most > 256, (W, most)


@pytest.mark.gpu
@pytest.mark.parametrize("front_end", [1, 2, 3], ids=["host-lists", "device-column-walk", "device-seg-walk"])
def test_round_room_of_1200_walls_on_the_gpu(dg, campath_mod, front_end):
    """The same on the GPU.  With the device column walk the sprites' behind-bit rows are read from the batch's array (ten words: too long
    to stage); the device seg walk flags the crowded frames (more than FS_PART_CAP parts) and they are redone through the host path."""
    import doomref
    wad = build_polygon_iwad()
    osc = doomref.Scene(wad, "e1m1")
    sc = dg.Scene(wad, "e1m1")
    views = _polygon_views(campath_mod, osc)
    for (W, H) in ((1280, 96), (320, 64)):                    # 320 wide: a part per column, 64 and more in every 64-column bin
        ctx = dg.Context(W, H, max_batch=len(views), slots=1, front_end=front_end)
        ctx.upload_scene(sc)
        out = ctx.render(dg.make_views(np.stack([r[:8] for r in views])))
        for k, rec in enumerate(views):
            ref = np.frombuffer(osc.render(W, H, rec), dtype=np.uint8).reshape(H, W, 3)
            assert np.array_equal(out[k], ref), f"{W}x{H} view {k}"
        if front_end == 3:
            assert ctx.fallbacks()["redone_frames"] >= 2
        ctx.close()
