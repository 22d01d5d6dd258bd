"""Deterministic synthetic IWAD generator (SURVEY.md §8d "Inputs — synthetic stand-in").

No id Software WAD can be shipped or fetched, so tests and bench.py render this instead.  The file
is a valid IWAD for the reference's loader (src/wad.rs:86-109: magic ``IWAD``, ``S_START``/``S_END``,
``PLAYPAL``, ``PNAMES``, ``TEXTURE1``, flats, one map with the lump order of src/wad.rs:8-19).

Shape ("e1m1-like"): a GX x GY grid of rectangular rooms with walls of real thickness, joined by
short door sectors (upper + lower textures, some with a masked middle grate), 45-degree chamfered
corners and free-standing pillars (so the BSP really has to split sectors), sky ceilings on part
of the rooms (adjacent sky rooms exercise the sky hack of src/renderer/segs.rs:463-477), animated
NUKAGE floors, non-power-of-two texture heights, rotating + mirrored + full-bright sprites.
Everything derives from xorshift32(seed); seed 1993 = default, ``heavy=True`` (seed 1994) = config 5.

BSP: recursive partition on seg lines with exact integer arithmetic (all coordinates are multiples
of 16, every line is axis-aligned or 45 degrees, so every split vertex is an integer).
"""
from __future__ import annotations

import struct
from fractions import Fraction

CELL = 512


class XorShift32:
    def __init__(self, seed: int):
        self.s = (seed & 0xFFFFFFFF) or 0x9E3779B9

    def next(self) -> int:
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s

    def below(self, n: int) -> int:
        return self.next() % n

    def choice(self, seq):
        return seq[self.below(len(seq))]

    def chance(self, num: int, den: int) -> bool:
        return self.below(den) < num


def _name8(s: str) -> bytes:
    b = s.encode("ascii")
    assert len(b) <= 8, s
    return b + b"\0" * (8 - len(b))


# ------------------------------------------------------------------------------------------------
# graphics
def _palette(rng: XorShift32) -> bytes:
    """16 hue ramps x 16 brightness steps, jittered (256 seeded RGB triples)."""
    hues = [(255, 255, 255), (255, 64, 48), (255, 160, 64), (240, 220, 80), (96, 220, 96), (64, 200, 200),
            (80, 120, 255), (180, 96, 255), (200, 150, 110), (150, 110, 80), (120, 130, 140), (90, 160, 90),
            (220, 120, 160), (160, 160, 90), (110, 90, 150), (70, 110, 170)]
    out = bytearray()
    for h in hues:
        for b in range(16):
            k = 40 + b * 14
            for c in h:
                v = c * k // 255 + rng.below(9) - 4
                out.append(max(0, min(255, v)))
    return bytes(out)


def _picture_lump(w: int, h: int, px, left=0, top=0) -> bytes:
    """Doom picture format (src/graphics/pictures.rs:66-126).  px[y][x] = 0..255 or None."""
    cols = []
    for x in range(w):
        col = bytearray()
        y = 0
        while y < h:
            if px[y][x] is None:
                y += 1
                continue
            y0 = y
            run = bytearray()
            while y < h and px[y][x] is not None and len(run) < 128:
                run.append(px[y][x])
                y += 1
            col += bytes([y0, len(run), 0]) + bytes(run) + b"\0"
        col.append(0xFF)
        cols.append(bytes(col))
    hdr = struct.pack("<hhhh", w, h, left, top)
    off = 8 + 4 * w
    table = bytearray()
    for c in cols:
        table += struct.pack("<I", off)
        off += len(c)
    return hdr + bytes(table) + b"".join(cols)


def _brick_patch(rng: XorShift32, w: int, h: int, hue: int, bw: int, bh: int):
    px = [[0] * w for _ in range(h)]
    for y in range(h):
        row = y // bh
        for x in range(w):
            xx = (x + (bw // 2 if row & 1 else 0)) % w
            mortar = (y % bh == 0) or (xx % bw == 0)
            b = 3 if mortar else 8 + ((row * 7 + xx // bw * 3) % 5) + rng.below(3)
            px[y][x] = hue * 16 + min(15, b)
    return px


def _panel_patch(rng: XorShift32, w: int, h: int, hue: int):
    px = [[0] * w for _ in range(h)]
    for y in range(h):
        for x in range(w):
            edge = min(x, w - 1 - x, y % 32, 31 - y % 32)
            b = 4 + min(8, edge) + rng.below(2) + (2 if (x // 8 + y // 8) & 1 else 0)
            px[y][x] = hue * 16 + min(15, b)
    return px


def _grate_patch(rng: XorShift32, w: int, h: int, hue: int):
    """Masked middle texture: bars with ~20 % transparent texels... inverted: mostly holes between bars."""
    px = [[None] * w for _ in range(h)]
    for y in range(h):
        for x in range(w):
            if x % 16 < 3 or y % 32 < 4 or (x + y) % 64 < 2:
                px[y][x] = hue * 16 + 6 + rng.below(6)
    return px


def _sky_patch(rng: XorShift32, w: int, h: int, idx: int):
    px = [[0] * w for _ in range(h)]
    for y in range(h):
        for x in range(w):
            gx = x + idx * w
            m = (gx * 3 + y * 5) % 97 < 18 + (y // 4)
            b = 15 - y // 10 if not m else 6 + (gx // 16 + y // 16) % 4
            px[y][x] = (6 if not m else 10) * 16 + max(0, min(15, b))
    # a few stars
    for _ in range(12):
        px[rng.below(h // 2)][rng.below(w)] = 15
    return px


def _sprite_px(rng: XorShift32, w: int, h: int, hue: int, rot: int):
    """Blob with an eye-stripe that moves with the rotation so the 8 views differ."""
    px = [[None] * w for _ in range(h)]
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    for y in range(h):
        for x in range(w):
            dx, dy = (x - cx) / (w / 2.0), (y - cy) / (h / 2.0)
            d = dx * dx + dy * dy
            if d <= 1.0:
                b = 12 - int(d * 8) + rng.below(2)
                px[y][x] = hue * 16 + max(1, min(15, b))
    sx = int(cx + (rot - 4) * w / 10.0)
    for y in range(h // 4, h // 4 + max(2, h // 8)):
        for x in range(max(0, sx - 2), min(w, sx + 3)):
            if px[y][x] is not None:
                px[y][x] = 15
    return px


def _flat(rng: XorShift32, hue: int, style: int) -> bytes:
    out = bytearray(4096)
    for y in range(64):
        for x in range(64):
            if style == 0:
                b = 5 + ((x // 16 + y // 16) & 1) * 5 + rng.below(2)
            elif style == 1:
                b = 4 + min(x % 32, 31 - x % 32, y % 32, 31 - y % 32) // 2
            elif style == 2:
                b = 6 + ((x * x + y * y) // 97) % 7
            else:
                b = 3 + (x ^ y) % 11
            out[y * 64 + x] = hue * 16 + max(0, min(15, b))
    # orientation mark so that mirrored / rotated mapping is visible
    for i in range(10):
        out[(2 + i) * 64 + 2] = 15
        out[2 * 64 + 2 + i // 2] = 15
    return bytes(out)


# ------------------------------------------------------------------------------------------------
# map geometry
class _Map:
    def __init__(self):
        self.vertexes: list[tuple[int, int]] = []
        self.vindex: dict[tuple[int, int], int] = {}
        self.linedefs: list[dict] = []
        self.sidedefs: list[dict] = []
        self.sectors: list[dict] = []
        self.things: list[tuple[int, int, int, int, int]] = []
        self.jitter = None

    def vertex(self, x: int, y: int) -> int:
        k = (int(x), int(y))
        if k not in self.vindex:
            self.vindex[k] = len(self.vertexes)
            self.vertexes.append(k)
        return self.vindex[k]

    def sidedef(self, sector: int, upper="-", lower="-", middle="-", xoff=0, yoff=0) -> int:
        self.sidedefs.append(dict(sector=sector, upper=upper, lower=lower, middle=middle, xoff=xoff, yoff=yoff))
        return len(self.sidedefs) - 1

    def line(self, a, b, front: int, back: int = -1, flags: int = 0) -> int:
        if self.jitter is not None:  # vanilla-shaped maps: every lattice point moves to an arbitrary integer nearby
            a, b = self.jitter(*a), self.jitter(*b)
        self.linedefs.append(dict(v1=self.vertex(*a), v2=self.vertex(*b), flags=flags, front=front, back=back))
        return len(self.linedefs) - 1


def _cross(ax, ay, bx, by):
    return ax * by - ay * bx


class _BspBuilder:
    """Segs: dict(v1,v2 (coords), linedef, direction, offset, sector)."""

    def __init__(self, m: _Map, rng: XorShift32, round_splits: bool = False):
        self.m = m
        self.rng = rng
        self.round_splits = round_splits  # vanilla node builders round split vertices to the integer VERTEXES grid
        self.segs_out: list[dict] = []
        self.ssectors: list[tuple[int, int]] = []
        self.nodes: list[dict] = []

    @staticmethod
    def _side(px, py, dx, dy, x, y) -> int:
        """>0: right of the partition (front), <0: left, 0: on the line."""
        return _cross(dx, dy, x - px, y - py) * -1

    def _classify(self, part, seg):
        px, py, dx, dy = part
        a = self._side(px, py, dx, dy, *seg["a"])
        b = self._side(px, py, dx, dy, *seg["b"])
        if self.round_splits and ((a > 0 and b < 0) or (a < 0 and b > 0)):
            # a crossing whose rounded split point lands on an end point is no crossing: the seg goes to the other end's side
            ip = self._split_point(part, seg)
            if ip == seg["a"]:
                a = 0
            elif ip == seg["b"]:
                b = 0
        return a, b

    @staticmethod
    def _split_point(part, seg):
        px, py, dx, dy = part
        ax, ay = seg["a"]
        sx, sy = seg["b"][0] - ax, seg["b"][1] - ay
        t = Fraction(_cross(dx, dy, px - ax, py - ay), _cross(dx, dy, sx, sy))  # point = a + t * s
        fx, fy = ax + t * sx, ay + t * sy
        return (int((2 * fx + 1) // 2), int((2 * fy + 1) // 2))  # round half up, exactly

    def _convex(self, segs) -> bool:
        sec = segs[0]["sector"]
        for s in segs:
            if s["sector"] != sec:
                return False
        for s in segs:
            ax, ay = s["a"]
            dx, dy = s["b"][0] - ax, s["b"][1] - ay
            for t in segs:
                if t is s:
                    continue
                for (x, y) in (t["a"], t["b"]):
                    if self._side(ax, ay, dx, dy, x, y) < 0:
                        return False
        return True

    def _pick(self, segs):
        cands = segs if len(segs) <= 24 else [segs[self.rng.below(len(segs))] for _ in range(24)]
        best = self._pick_from(segs, cands)
        if best is None and cands is not segs:
            best = self._pick_from(segs, segs)
        return best

    def _pick_from(self, segs, cands):
        best, best_cost = None, None
        for c in cands:
            ax, ay = c["a"]
            part = (ax, ay, c["b"][0] - ax, c["b"][1] - ay)
            l = r = sp = 0
            for s in segs:
                a, b = self._classify(part, s)
                if a >= 0 and b >= 0 and (a > 0 or b > 0):
                    r += 1
                elif a <= 0 and b <= 0 and (a < 0 or b < 0):
                    l += 1
                elif a == 0 and b == 0:
                    same = _cross(part[2], part[3], s["b"][0] - s["a"][0], s["b"][1] - s["a"][1]) == 0 and \
                        (part[2] * (s["b"][0] - s["a"][0]) + part[3] * (s["b"][1] - s["a"][1])) > 0
                    if same:
                        r += 1
                    else:
                        l += 1
                else:
                    sp += 1
            if l == 0 or r == 0:
                if sp == 0:
                    continue
            axis = 0 if (part[2] == 0 or part[3] == 0) else 6
            cost = abs(l - r) + 9 * sp + axis
            if best_cost is None or cost < best_cost:
                best, best_cost = part, cost
        return best

    def _split(self, part, seg):
        px, py, dx, dy = part
        ax, ay = seg["a"]
        bx, by = seg["b"]
        sx, sy = bx - ax, by - ay
        den = _cross(dx, dy, sx, sy)
        t = Fraction(_cross(dx, dy, px - ax, py - ay), den)  # point = a + t * s
        ix, iy = ax + t * sx, ay + t * sy
        if self.round_splits:
            ix, iy = self._split_point(part, seg)
        else:
            assert ix.denominator == 1 and iy.denominator == 1, "non-integer BSP split"
        ix, iy = int(ix), int(iy)
        first = dict(seg, b=(ix, iy))
        ln = round(((ix - ax) ** 2 + (iy - ay) ** 2) ** 0.5)
        second = dict(seg, a=(ix, iy), offset=seg["offset"] + ln)
        return first, second

    def _bbox(self, segs):
        xs = [p[0] for s in segs for p in (s["a"], s["b"])]
        ys = [p[1] for s in segs for p in (s["a"], s["b"])]
        return (max(ys), min(ys), min(xs), max(xs))  # top, bottom, left, right

    def build(self, segs) -> int:
        """Returns child reference (bit 15 set = subsector)."""
        part = None if self._convex(segs) else self._pick(segs)
        if part is None:
            first = len(self.segs_out)
            self.segs_out.extend(segs)
            self.ssectors.append((len(segs), first))
            return 0x8000 | (len(self.ssectors) - 1)
        rights, lefts = [], []
        for s in segs:
            a, b = self._classify(part, s)
            if a == 0 and b == 0:
                same = (part[2] * (s["b"][0] - s["a"][0]) + part[3] * (s["b"][1] - s["a"][1])) > 0
                (rights if same else lefts).append(s)
            elif a >= 0 and b >= 0:
                rights.append(s)
            elif a <= 0 and b <= 0:
                lefts.append(s)
            else:
                f, g = self._split(part, s)
                if a > 0:
                    rights.append(f)
                    lefts.append(g)
                else:
                    lefts.append(f)
                    rights.append(g)
        assert rights and lefts
        rb, lb = self._bbox(rights), self._bbox(lefts)
        rc = self.build(rights)
        lc = self.build(lefts)
        self.nodes.append(dict(x=part[0], y=part[1], dx=part[2], dy=part[3], rb=rb, lb=lb, rc=rc, lc=lc))
        return len(self.nodes) - 1


def _bam16(dx, dy) -> int:
    import math
    a = int(round(math.atan2(dy, dx) / (2 * math.pi) * 65536.0)) & 0xFFFF
    return a - 65536 if a >= 32768 else a


# ------------------------------------------------------------------------------------------------
WALL_TEXTURES = ["BRICK1", "BRICK2", "BRICK3", "PANEL1", "PANEL2", "WIDE1", "WIDE2", "TALL72", "COMBO1", "COMBO2",
                 "STONE1", "STONE2", "METAL1", "METAL2", "HOLEY1"]
FLOOR_FLATS = ["FLOOR0", "FLOOR1", "FLOOR2", "FLOOR3", "FLOOR4", "FLOOR5", "NUKAGE1"]
CEIL_FLATS = ["CEIL0", "CEIL1", "CEIL2", "CEIL3"]
# thing types with synthetic sprite lumps: (doomednum, sprite, rotating, w, h)
SPRITE_DEFS = [(2035, "BAR1", False, 23, 32), (2028, "COLU", False, 19, 47), (48, "ELEC", False, 37, 127),
               (2014, "BON1", False, 14, 18), (34, "CAND", False, 7, 14), (46, "TRED", False, 25, 91),
               (3004, "POSS", True, 41, 56), (3001, "TROO", True, 43, 57)]


def build_synth_iwad(seed: int = 1993, heavy: bool = False, map_name: str = "E1M1", quirks: bool = False,
                     vanilla: bool = False, grid=None, n_things=None, custom_map=None) -> bytes:
    """`vanilla` (used with seed 1995) bends the lattice map towards what real IWAD maps hold: every vertex moved to an arbitrary
    integer position (so walls run at arbitrary angles and BSP split points are rounded onto the integer grid like a node
    builder's), chamfers with unequal legs, irregular pillars, closed doors (door sector ceiling == floor, segs.rs:222-225),
    thing angles in 1 degree steps and wall textures whose patches start above/left of the texture or run past its bottom.
    The default maps do not consume any of the extra random numbers and stay byte-identical.
    `grid` = (columns, rows) of rooms and `n_things` override the two sizes (8 x 6 / 16 x 12 rooms, 40 / 300 things): (32, 24) is a map of
    doom2's scale — 768 rooms, over a thousand sectors, over ten thousand segs.  `custom_map(m, rng, names)` fills the `_Map` itself
    (sectors, lines, things) instead of the room lattice; graphics, BSP builder and lump packing stay the generator's."""
    rng = XorShift32(seed)
    gx, gy = grid if grid is not None else ((16, 12) if heavy else (8, 6))
    sky_pct = 50 if heavy else 25
    n_things = n_things if n_things is not None else (300 if heavy else 40)

    lumps: list[tuple[str, bytes]] = []
    lumps.append(("PLAYPAL", _palette(rng)))

    # ---- patches / textures ----
    patches: list[tuple[str, bytes]] = []

    def add_patch(name, w, h, px):
        patches.append((name, _picture_lump(w, h, px)))
        return len(patches) - 1

    p = {}
    p["BRK1"] = add_patch("PBRK1", 64, 128, _brick_patch(rng, 64, 128, 1, 32, 16))
    p["BRK2"] = add_patch("PBRK2", 64, 128, _brick_patch(rng, 64, 128, 8, 16, 8))
    p["BRK3"] = add_patch("PBRK3", 64, 128, _brick_patch(rng, 64, 128, 9, 64, 32))
    p["PNL1"] = add_patch("PPNL1", 64, 128, _panel_patch(rng, 64, 128, 10))
    p["PNL2"] = add_patch("PPNL2", 64, 128, _panel_patch(rng, 64, 128, 6))
    p["STN1"] = add_patch("PSTN1", 128, 128, _brick_patch(rng, 128, 128, 11, 32, 32))
    p["STN2"] = add_patch("PSTN2", 128, 128, _panel_patch(rng, 128, 128, 13))
    p["MTL1"] = add_patch("PMTL1", 64, 72, _panel_patch(rng, 64, 72, 14))
    p["MTL2"] = add_patch("PMTL2", 32, 64, _brick_patch(rng, 32, 64, 2, 8, 8))
    p["GRT1"] = add_patch("PGRT1", 64, 128, _grate_patch(rng, 64, 128, 3))
    for i in range(4):
        p["SKY%d" % i] = add_patch("PSKY%d" % i, 64, 128, _sky_patch(rng, 64, 128, i))

    # texture = (name, w, h, [(ox, oy, patch)])
    texdefs = [
        ("BRICK1", 64, 128, [(0, 0, p["BRK1"])]),
        ("BRICK2", 64, 128, [(0, 0, p["BRK2"])]),
        ("BRICK3", 64, 128, [(0, 0, p["BRK3"])]),
        ("PANEL1", 64, 128, [(0, 0, p["PNL1"])]),
        ("PANEL2", 64, 128, [(0, 0, p["PNL2"])]),
        ("WIDE1", 128, 128, [(0, 0, p["STN1"])]),
        ("WIDE2", 256, 128, [(0, 0, p["STN1"]), (128, 0, p["STN2"])]),
        ("TALL72", 64, 72, [(0, 0, p["MTL1"])]),
        ("COMBO1", 128, 128, [(0, 0, p["BRK1"]), (64, 0, p["PNL1"]), (48, 32, p["MTL2"])]),
        ("COMBO2", 128, 128, [(-16, -8, p["STN2"]), (100, 60, p["BRK2"])]),  # clipped patches
        ("STONE1", 128, 128, [(0, 0, p["STN1"])]),
        ("STONE2", 128, 128, [(0, 0, p["STN2"])]),
        ("METAL1", 64, 64, [(0, 0, p["MTL2"]), (32, 0, p["MTL2"])]),
        ("METAL2", 64, 128, [(0, 0, p["MTL1"]), (0, 72, p["MTL1"])]),
        ("HOLEY1", 64, 128, [(0, 0, p["BRK3"]), (16, 40, p["GRT1"])]),  # later patch punches None holes
        ("GRATE1", 64, 128, [(0, 0, p["GRT1"])]),                        # masked middle texture
        ("SKY1", 256, 128, [(64 * i, 0, p["SKY%d" % i]) for i in range(4)]),
    ]
    wall_textures = list(WALL_TEXTURES)
    if vanilla:
        texdefs += [
            # a 128-tall patch hung 100 rows above a 64-tall texture, a second one starting left of it and running past the bottom
            ("VANT1", 64, 64, [(0, -100, p["BRK1"]), (-40, 24, p["STN2"])]),
            # patch taller than the texture from a negative origin + one that only touches the last column / last rows
            ("VANT2", 128, 96, [(-8, -16, p["STN1"]), (127, 90, p["MTL2"]), (40, -60, p["GRT1"])]),
            # non-power-of-two width, patch origins that leave uncovered (None) columns at the right
            ("VANT3", 72, 128, [(-32, 0, p["PNL1"]), (32, 64, p["MTL2"])]),
        ]
        wall_textures += ["VANT1", "VANT2", "VANT3"]
    pnames = struct.pack("<I", len(patches)) + b"".join(_name8(n) for n, _ in patches)
    tex_blobs = []
    for name, w, h, pl in texdefs:
        b = _name8(name) + struct.pack("<IhhIh", 0, w, h, 0, len(pl))
        for ox, oy, pi in pl:
            b += struct.pack("<hhhhh", ox, oy, pi, 1, 0)
        tex_blobs.append(b)
    off = 4 + 4 * len(tex_blobs)
    tex1 = struct.pack("<I", len(tex_blobs))
    for b in tex_blobs:
        tex1 += struct.pack("<I", off)
        off += len(b)
    tex1 += b"".join(tex_blobs)
    lumps.append(("PNAMES", pnames))
    lumps.append(("TEXTURE1", tex1))
    lumps.append(("P_START", b""))
    lumps.extend(patches)
    lumps.append(("P_END", b""))

    # ---- flats ----
    lumps.append(("F_START", b""))
    for i, nm in enumerate(["FLOOR0", "FLOOR1", "FLOOR2", "FLOOR3", "FLOOR4", "FLOOR5"]):
        lumps.append((nm, _flat(rng, 8 + i % 6, i % 4)))
    for i, nm in enumerate(CEIL_FLATS):
        lumps.append((nm, _flat(rng, 10 + i, (i + 1) % 4)))
    for i in range(3):
        lumps.append(("NUKAGE%d" % (i + 1), _flat(rng, 4, i)))
    lumps.append(("F_SKY1", _flat(rng, 6, 3)))
    lumps.append(("F_END", b""))

    # ---- sprites ----
    lumps.append(("S_START", b""))
    for num, spr, rotating, w, h in SPRITE_DEFS:
        hue = 1 + (num % 7)
        if not rotating:
            lumps.append((spr + "A0", _picture_lump(w, h, _sprite_px(rng, w, h, hue, 4), w // 2, h - 2)))
        elif spr == "POSS":  # mirrored pairs (src/graphics/sprites.rs:48-56)
            lumps.append((spr + "A1", _picture_lump(w, h, _sprite_px(rng, w, h, hue, 1), w // 2, h - 4)))
            for a, b in ((2, 8), (3, 7), (4, 6)):
                lumps.append(("%sA%dA%d" % (spr, a, b), _picture_lump(w, h, _sprite_px(rng, w, h, hue, a), w // 2, h - 4)))
            lumps.append((spr + "A5", _picture_lump(w, h, _sprite_px(rng, w, h, hue, 5), w // 2, h - 4)))
        else:
            for r in range(1, 9):
                lumps.append(("%sA%d" % (spr, r), _picture_lump(w, h, _sprite_px(rng, w, h, hue, r), w // 2, h - 4)))
    lumps.append(("S_END", b""))

    # ---- map ----
    m = _Map()

    def _finish():
        # ---- BSP ----
        segs = []
        for li, ld in enumerate(m.linedefs):
            a, b = m.vertexes[ld["v1"]], m.vertexes[ld["v2"]]
            segs.append(dict(a=a, b=b, linedef=li, direction=0, offset=0, sector=m.sidedefs[ld["front"]]["sector"]))
            if ld["back"] >= 0:
                segs.append(dict(a=b, b=a, linedef=li, direction=1, offset=0, sector=m.sidedefs[ld["back"]]["sector"]))
        bsp = _BspBuilder(m, rng, round_splits=vanilla)
        root = bsp.build(segs)
        assert not (root & 0x8000) and root == len(bsp.nodes) - 1

        things = b"".join(struct.pack("<hhhhh", *t) for t in m.things)
        seg_bytes = bytearray()
        for s in bsp.segs_out:
            v1, v2 = m.vertex(*s["a"]), m.vertex(*s["b"])
            seg_bytes += struct.pack("<hhhhhh", v1, v2, _bam16(s["b"][0] - s["a"][0], s["b"][1] - s["a"][1]), s["linedef"],
                                     s["direction"], min(32767, s["offset"]))
        vertexes = b"".join(struct.pack("<hh", x, y) for x, y in m.vertexes)
        linedefs = b"".join(struct.pack("<hhhhhhh", ld["v1"], ld["v2"], ld["flags"], 0, 0, ld["front"], ld["back"]) for ld in m.linedefs)
        sidedefs = b"".join(struct.pack("<hh", sd["xoff"], sd["yoff"]) + _name8(sd["upper"]) + _name8(sd["lower"]) + _name8(sd["middle"]) +
                            struct.pack("<h", sd["sector"]) for sd in m.sidedefs)
        ssectors = b"".join(struct.pack("<hh", c, f) for c, f in bsp.ssectors)

        def child(c):
            return struct.pack("<H", c)

        nodes = b"".join(struct.pack("<hhhh", n["x"], n["y"], n["dx"], n["dy"]) + struct.pack("<hhhh", *n["rb"]) +
                         struct.pack("<hhhh", *n["lb"]) + child(n["rc"]) + child(n["lc"]) for n in bsp.nodes)
        sectors = b"".join(struct.pack("<hh", s["floor"], s["ceil"]) + _name8(s["ffl"]) + _name8(s["cfl"]) +
                           struct.pack("<hhh", s["light"], s["special"], s["tag"]) for s in m.sectors)
        assert len(m.vertexes) < 32768 and len(bsp.segs_out) < 32768 and len(m.sidedefs) < 32768
        out_lumps = lumps + [(map_name.upper(), b""), ("THINGS", things), ("LINEDEFS", linedefs), ("SIDEDEFS", sidedefs), ("VERTEXES", vertexes),
                  ("SEGS", bytes(seg_bytes)), ("SSECTORS", ssectors), ("NODES", nodes), ("SECTORS", sectors), ("REJECT", b""),
                  ("BLOCKMAP", b"")]

        if quirks:
            out_lumps = _apply_quirks(out_lumps, p, patches, map_name.upper())

        # ---- container ----
        body = bytearray()
        directory = bytearray()
        for name, data in out_lumps:
            directory += struct.pack("<II", 12 + len(body), len(data)) + _name8(name)
            body += data
        hdr = b"IWAD" + struct.pack("<II", len(out_lumps), 12 + len(body))
        return hdr + bytes(body) + bytes(directory)

    if custom_map is not None:
        custom_map(m, rng, dict(wall_textures=list(wall_textures), floor_flats=list(FLOOR_FLATS), ceil_flats=list(CEIL_FLATS), sprite_defs=list(SPRITE_DEFS)))
        return _finish()
    if vanilla:
        def jitter(x, y, _seed=seed):
            h = (x * 73856093 ^ y * 19349663 ^ _seed * 83492791) & 0xFFFFFFFF
            h = (h ^ (h >> 15)) * 0x2C1B3C6D & 0xFFFFFFFF
            h = (h ^ (h >> 12)) * 0x297A2D39 & 0xFFFFFFFF
            h ^= h >> 15
            return (x + h % 11 - 5, y + (h >> 8) % 11 - 5)
        m.jitter = jitter
    rooms = {}
    for j in range(gy):
        for i in range(gx):
            mx0, mx1, my0, my1 = (32 + 16 * rng.below(5) for _ in range(4))
            x0, x1 = i * CELL + mx0, (i + 1) * CELL - mx1
            y0, y1 = j * CELL + my0, (j + 1) * CELL - my1
            floor = 8 * (rng.below(17) - 8)
            ceil = floor + rng.choice([96, 128, 160, 256])
            border = i in (0, gx - 1) or j in (0, gy - 1)
            sky = rng.below(100) < (sky_pct * 2 if border else sky_pct // 2)
            light = rng.choice([96, 112, 128, 144, 160, 192, 224, 255])
            sec = len(m.sectors)
            m.sectors.append(dict(floor=floor, ceil=ceil, ffl=rng.choice(FLOOR_FLATS), cfl="F_SKY1" if sky else rng.choice(CEIL_FLATS),
                                  light=light, special=0, tag=0))
            chamfer = [64 if rng.chance(1, 3) else 0 for _ in range(4)]  # corners: (x0,y0) (x0,y1) (x1,y1) (x1,y0)
            chamfer = [(c, c) for c in chamfer]                           # legs along x and along y
            if vanilla:
                chamfer = [(16 * (2 + rng.below(4)), 16 * (2 + rng.below(4))) if rng.chance(1, 2) else (0, 0) for _ in range(4)]
            rooms[(i, j)] = dict(x0=x0, x1=x1, y0=y0, y1=y1, sec=sec, chamfer=chamfer, doors={}, sky=sky,
                                 cx=i * CELL + CELL // 2, cy=j * CELL + CELL // 2)

    # connectivity: random spanning tree + ~35 % of the remaining neighbour pairs
    edges = []
    for j in range(gy):
        for i in range(gx):
            if i + 1 < gx:
                edges.append(((i, j), (i + 1, j)))
            if j + 1 < gy:
                edges.append(((i, j), (i, j + 1)))
    order = list(range(len(edges)))
    for k in range(len(order) - 1, 0, -1):
        q = rng.below(k + 1)
        order[k], order[q] = order[q], order[k]
    parent = {c: c for c in rooms}

    def find(c):
        while parent[c] != c:
            parent[c] = parent[parent[c]]
            c = parent[c]
        return c

    tree_edges, extra_edges = [], []
    for k in order:
        a, b = edges[k]
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
            tree_edges.append((a, b))
        elif rng.chance(35, 100):
            extra_edges.append((a, b))

    wall_tex = lambda: rng.choice(wall_textures)  # noqa: E731
    doors = []
    for a, b in tree_edges + extra_edges:
        ra, rb = rooms[a], rooms[b]
        hw = 16 * (2 + rng.below(3))  # half-width 32..64
        sa, sb = m.sectors[ra["sec"]], m.sectors[rb["sec"]]
        floor = max(sa["floor"], sb["floor"]) + rng.choice([0, 0, 8, 16])
        ceil = min(sa["ceil"], sb["ceil"]) - rng.choice([0, 8, 16, 32])
        if ceil < floor + 64:
            ceil = floor + 64
        if vanilla and (a, b) in extra_edges and rng.chance(1, 2):
            ceil = floor                    # a closed door: its sector has no height (the route never passes through one)
        both_sky = ra["sky"] and rb["sky"]
        sec = len(m.sectors)
        m.sectors.append(dict(floor=floor, ceil=ceil, ffl=rng.choice(FLOOR_FLATS), cfl="F_SKY1" if both_sky else rng.choice(CEIL_FLATS),
                              light=rng.choice([112, 144, 176, 208]), special=0, tag=0))
        horizontal = a[1] == b[1]
        grate = rng.chance(1, 4)
        d = dict(sec=sec, hw=hw, horizontal=horizontal, a=a, b=b, grate=grate)
        doors.append(d)
        if horizontal:
            ra["doors"]["E"] = d
            rb["doors"]["W"] = d
        else:
            ra["doors"]["N"] = d
            rb["doors"]["S"] = d

    def two_sided(pa, pb, room_sec, door_sec, grate):
        flags = 4 | (8 if rng.chance(1, 3) else 0) | (16 if rng.chance(1, 3) else 0)
        f = m.sidedef(room_sec, wall_tex(), wall_tex(), "GRATE1" if grate else "-", 8 * rng.below(4), 8 * rng.below(3))
        bk = m.sidedef(door_sec, wall_tex(), wall_tex(), "GRATE1" if grate else "-", 0, 0)
        m.line(pa, pb, f, bk, flags)

    def one_sided(pa, pb, sec):
        if pa == pb:
            return
        flags = 1 | (16 if rng.chance(1, 4) else 0)
        m.line(pa, pb, m.sidedef(sec, "-", "-", wall_tex(), rng.choice([0, 0, 8, 16, -24]), rng.choice([0, 0, 8, -16])), -1, flags)

    for (i, j), r in rooms.items():
        x0, x1, y0, y1, sec = r["x0"], r["x1"], r["y0"], r["y1"], r["sec"]
        (c00x, c00), (c01x, c01), (c11x, c11), (c10x, c10) = r["chamfer"]
        cx, cy = r["cx"], r["cy"]
        # clockwise boundary (interior on the right of each line), y up: W side up, N side right, E side down, S side left
        sides = [
            ("W", (x0, y0 + c00), (x0, y1 - c01), "y", cy),
            ("N", (x0 + c01x, y1), (x1 - c11x, y1), "x", cx),
            ("E", (x1, y1 - c11), (x1, y0 + c10), "y", cy),
            ("S", (x1 - c10x, y0), (x0 + c00x, y0), "x", cx),
        ]
        corners = [((x0 + c00x, y0), (x0, y0 + c00), c00), ((x0, y1 - c01), (x0 + c01x, y1), c01),
                   ((x1 - c11x, y1), (x1, y1 - c11), c11), ((x1, y0 + c10), (x1 - c10x, y0), c10)]
        for k, (side, pa, pb, axis, centre) in enumerate(sides):
            ca, cb, cs = corners[k]
            if cs:
                one_sided(ca, cb, sec)
            d = r["doors"].get(side)
            if d is None:
                one_sided(pa, pb, sec)
                continue
            hw = d["hw"]
            if axis == "y":
                lo, hi = (pa[0], centre - hw), (pa[0], centre + hw)
            else:
                lo, hi = (centre - hw, pa[1]), (centre + hw, pa[1])
            first, second = (lo, hi) if (pb[0] - pa[0] + pb[1] - pa[1]) > 0 else (hi, lo)
            one_sided(pa, first, sec)
            two_sided(first, second, sec, d["sec"], d["grate"])
            one_sided(second, pb, sec)
        # pillars (counter-clockwise so the room is on the right)
        for (ox, oy) in ((-96, -96), (96, 96), (-96, 96), (96, -96)):
            if rng.chance(1, 5):
                px, py, h = cx + ox, cy + oy, 16
                if vanilla:  # irregular convex quad: bottom, right, top, left (counter-clockwise)
                    rr = [14 + rng.below(18) for _ in range(4)]
                    ee = [rng.below(13) - 6 for _ in range(4)]
                    pts = [(px + ee[0], py - rr[0]), (px + rr[1], py + ee[1]), (px + ee[2], py + rr[2]), (px - rr[3], py + ee[3])]
                elif rng.chance(1, 2):
                    pts = [(px - h, py - h), (px + h, py - h), (px + h, py + h), (px - h, py + h)]
                else:  # diamond
                    pts = [(px, py - 2 * h), (px + 2 * h, py), (px, py + 2 * h), (px - 2 * h, py)]
                for q in range(4):
                    one_sided(pts[q], pts[(q + 1) % 4], sec)

    for d in doors:
        ra, rb = rooms[d["a"]], rooms[d["b"]]
        hw, sec = d["hw"], d["sec"]
        if d["horizontal"]:
            xa, xb, cy = ra["x1"], rb["x0"], ra["cy"]
            one_sided((xa, cy + hw), (xb, cy + hw), sec)
            one_sided((xb, cy - hw), (xa, cy - hw), sec)
        else:
            ya, yb, cx = ra["y1"], rb["y0"], ra["cx"]
            one_sided((cx - hw, ya), (cx - hw, yb), sec)
            one_sided((cx + hw, yb), (cx + hw, ya), sec)

    # things: player start in room (0,0), then seeded decorations / monsters
    r0 = rooms[(0, 0)]
    m.things.append((r0["cx"], r0["cy"], 0, 1, 7))
    cells = list(rooms.keys())
    for _ in range(n_things):
        r = rooms[rng.choice(cells)]
        num = rng.choice(SPRITE_DEFS)[0]
        ox = rng.choice([-64, -48, -32, 32, 48, 64])
        oy = rng.choice([-64, -48, -32, 32, 48, 64])
        m.things.append((r["cx"] + ox, r["cy"] + oy, rng.below(360) if vanilla else 45 * rng.below(8), num, 7))

    return _finish()


def _apply_quirks(lumps, p, patches, map_marker):
    """Loader edge cases on top of a finished lump list (default WADs stay byte-identical):
      * a decoy FLOOR0 lump earlier in the file (the directory HashMap keeps the LAST lump of a name, src/wad.rs:153-154)
      * a second map marker + empty lumps at the end (map lumps come from the FIRST marker, src/wad.rs:175-183)
      * TEXTURE2 that redefines BRICK2 (the later definition wins, textures.rs:252-253) and adds EXTRA1 and the 8-character
        name LONGNAME (not NUL-terminated, src/wad.rs:112-126)
      * sidedef texture names in lower case (Textures::get upper-cases, textures.rs:155), sector flats in lower case
        (found through the upper-cased directory but not "SKY" / not animated: flats.rs:103-111, visplanes.rs:89)."""
    out = []
    for name, data in lumps:
        if name == "F_START":
            out.append(("FLOOR0", bytes((i * 7 + 3) & 0xFF for i in range(4096))))   # decoy, must lose
        if name == "P_END":
            pass
        if name == "TEXTURE1":
            out.append((name, data))
            defs = [("BRICK2", 128, 128, [(0, 0, p["STN2"])]), ("EXTRA1", 64, 128, [(0, 0, p["PNL2"]), (8, 24, p["MTL2"])]),
                    ("LONGNAME", 64, 64, [(0, 0, p["MTL2"]), (32, 0, p["MTL2"])]),
                    ("SKY2", 256, 128, [(64 * i, 0, p["SKY%d" % (3 - i)]) for i in range(4)]),       # episode 2 / MAP12-20 (game.rs:199-227)
                    ("SKY3", 256, 128, [(64 * i, 0, p["SKY%d" % ((i + 2) % 4)]) for i in range(4)])]  # episode 3 / MAP21+
            blobs = []
            for tn, w, h, pl in defs:
                b = _name8(tn) + struct.pack("<IhhIh", 0, w, h, 0, len(pl))
                for ox, oy, pi in pl:
                    b += struct.pack("<hhhhh", ox, oy, pi, 1, 0)
                blobs.append(b)
            off = 4 + 4 * len(blobs)
            t2 = struct.pack("<I", len(blobs))
            for b in blobs:
                t2 += struct.pack("<I", off)
                off += len(b)
            out.append(("TEXTURE2", t2 + b"".join(blobs)))
            continue
        if name == "SIDEDEFS":
            data = data.replace(b"BRICK3\0\0", b"EXTRA1\0\0").replace(b"PANEL1\0\0", b"panel1\0\0").replace(b"STONE1\0\0", b"LONGNAME")
        if name == "SECTORS":
            data = data.replace(b"FLOOR5\0\0", b"floor5\0\0").replace(b"NUKAGE1\0", b"nukage1\0", 2).replace(b"F_SKY1\0\0", b"f_sky1\0\0", 1)
        out.append((name, data))
    out.append((map_marker, b""))
    out += [(n, b"") for n in ("THINGS", "LINEDEFS", "SIDEDEFS", "VERTEXES", "SEGS", "SSECTORS", "NODES", "SECTORS", "REJECT", "BLOCKMAP")]
    return out


def synth_route(seed: int = 1993, heavy: bool = False, vanilla: bool = False, grid=None, n_things=None):
    """Waypoints (x, y) of a closed walk through every room that only crosses door sectors:
    depth-first traversal of the generator's spanning tree (same RNG stream as build_synth_iwad)."""
    # Re-run the generator's RNG consumption up to the spanning tree by building the WAD's
    # connectivity again; cheaper: rebuild and introspect.
    return _route_from_build(seed, heavy, vanilla, grid, n_things)


def _route_from_build(seed, heavy, vanilla=False, grid=None, n_things=None):
    # The route is derived from the door list, recovered from the WAD itself: two-sided linedefs.
    wad = build_synth_iwad(seed, heavy, vanilla=vanilla, grid=grid, n_things=n_things)
    lumps = wad_directory(wad)
    idx = [i for i, (n, _, _) in enumerate(lumps) if n == "E1M1"][0]
    def lump(k):
        n, off, size = lumps[idx + k]
        return wad[off:off + size]
    ld, sd, vx, secs = lump(2), lump(3), lump(4), lump(8)
    gx, gy = grid if grid is not None else ((16, 12) if heavy else (8, 6))
    nroom = gx * gy
    adj = {c: set() for c in range(nroom)}
    door_rooms: dict[int, set] = {}
    for i in range(len(ld) // 14):
        v1, v2, flags, _, _, f, b = struct.unpack_from("<hhhhhhh", ld, i * 14)
        if b < 0:
            continue
        sf = struct.unpack_from("<h", sd, f * 30 + 28)[0]
        sb = struct.unpack_from("<h", sd, b * 30 + 28)[0]
        room, door = (sf, sb) if sf < nroom else (sb, sf)
        dfloor, dceil = struct.unpack_from("<hh", secs, door * 26)
        if dceil <= dfloor:
            continue  # closed door
        door_rooms.setdefault(door, set()).add(room)
    for door, rs in door_rooms.items():
        a, b = sorted(rs)
        adj[a].add(b)
        adj[b].add(a)
    centre = lambda c: ((c % gx) * CELL + CELL // 2, (c // gx) * CELL + CELL // 2)  # noqa: E731
    seen, route = set(), []

    def dfs(c):
        seen.add(c)
        route.append(centre(c))
        for n in sorted(adj[c]):
            if n not in seen:
                ca, cn = centre(c), centre(n)
                route.append(((ca[0] + cn[0]) // 2, (ca[1] + cn[1]) // 2))
                dfs(n)
                route.append(((ca[0] + cn[0]) // 2, (ca[1] + cn[1]) // 2))
                route.append(centre(c))

    import sys
    sys.setrecursionlimit(max(10000, 8 * nroom))
    dfs(0)
    return route[:-1]  # closed loop: last point == first


def wad_directory(wad: bytes):
    n, off = struct.unpack_from("<II", wad, 4)
    out = []
    for i in range(n):
        o, s = struct.unpack_from("<II", wad, off + 16 * i)
        name = wad[off + 16 * i + 8: off + 16 * i + 16].split(b"\0")[0].decode("ascii").upper()
        out.append((name, o, s))
    return out
