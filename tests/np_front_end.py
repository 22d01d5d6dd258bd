"""np_front_end.py — an independent restatement of the PER-COLUMN half of the reference's front end, for tests only.

    Segs::process_sidedef, the `for x in bottom.start.x..=bottom.end.x` loop     /root/reference src/renderer/segs.rs:202-345
    Segs::occlude_vertical_line                                                   src/renderer/segs.rs:113-117
    SidedefVisPlanes (add_bottom_point / add_top_point / flush)                   src/renderer/sidedef_visplanes.rs:41-83
    Visplane::new (left = right = -1, zero-filled top / bottom)                   src/renderer/visplanes.rs:28-39

Written from those lines with numpy float32 scalars; it shares no code with oracle/doomref.c, with the product's host front end
(csrc/frontend.cpp) or with its device column walk (csrc/fe_core.h).  Input: the per-seg half's result for one frame — what the
reference has in hand when it enters the column loop of each process_sidedef call (integer endpoints, the two edge lines and their
deltas, the call's Flags) — in visit order.  The product ships exactly that to the GPU (FePart, csrc/fe_dev.h), which is where the
tests take it from.  Output: what the loop leaves behind — every call's BitmapColumns (bitmap_render.add_column, segs.rs:260) and the
visplanes in push order (segs.rs:263-318,336-347) — to be compared with the product's list builder (dg_build_lists).
"""
import numpy as np

F32 = np.float32

# Flags of one process_sidedef call (segs.rs:29-36) as the product packs them (csrc/fe_dev.h FEP_*)
ONLY_OCCLUSIONS, IS_LOWER_WALL, IS_UPPER_WALL, DRAW_CEILING, IS_TWO_SIDED_MIDDLE_WALL, HAS_TEXTURE = 1, 2, 4, 8, 16, 32


def as_i16(v) -> int:
    """Rust `f32 as i16`: truncate toward zero, saturate, NaN -> 0."""
    v = float(v)
    if v != v:
        return 0
    if v >= 32767.0:
        return 32767
    if v <= -32768.0:
        return -32768
    return int(v)


class _Visplane:
    def __init__(self, W):
        self.left = self.right = -1                       # visplanes.rs:28-39
        self.top = [0] * W
        self.bottom = [0] * W

    def entries(self):
        return [(self.top[x], self.bottom[x]) for x in range(self.left, self.right + 1)]


class _SidedefVisPlanes:
    """sidedef_visplanes.rs:7-83 (the flat / height / light fields do not influence the geometry and are left out)."""

    def __init__(self, W, out, tag):
        self.W, self.out, self.tag = W, out, tag
        self.bottom, self.top = _Visplane(W), _Visplane(W)
        self.bottom_used = self.top_used = False

    def flush(self):                                      # :41-57: floor first, then ceiling
        if self.bottom_used:
            self.out.append((self.tag, "floor", self.bottom.left, self.bottom.right, self.bottom.entries()))
            self.bottom, self.bottom_used = _Visplane(self.W), False
        if self.top_used:
            self.out.append((self.tag, "ceiling", self.top.left, self.top.right, self.top.entries()))
            self.top, self.top_used = _Visplane(self.W), False

    def add_bottom_point(self, x, top_y, bottom_y):       # :60-70
        if not self.bottom_used:
            self.bottom.left = x
        self.bottom.right = x
        self.bottom_used = True
        self.bottom.top[x], self.bottom.bottom[x] = top_y, bottom_y

    def add_top_point(self, x, top_y, bottom_y):          # :73-83
        if not self.top_used:
            self.top.left = x
        self.top.right = x
        self.top_used = True
        self.top.top[x], self.top.bottom[x] = top_y, bottom_y


def column_loops(W: int, H: int, calls):
    """calls: per process_sidedef call, in visit order, a dict with sx, ex (ints), bsy, bsx, bdelta, tsy, tsx, tdelta (np.float32) and
    flags.  Returns (columns per call: list of (x, clipped_top_y, clipped_bottom_y, bottom_y, top_y), visplanes in push order)."""
    hor_ocl = [False] * W                                  # Segs::new, segs.rs:96-98: fresh per frame
    floor_ver_ocl = [H] * W                                # SCREEN_HEIGHT as i16
    ceiling_ver_ocl = [-1] * W
    visplanes, all_columns = [], []

    def occlude_vertical_line(x):                         # segs.rs:113-117
        hor_ocl[x] = True
        floor_ver_ocl[x] = H // 2                         # SCREEN_HEIGHT as i16 / 2
        ceiling_ver_ocl[x] = H // 2

    for ci, c in enumerate(calls):
        fl = c["flags"]
        two, only = bool(fl & IS_TWO_SIDED_MIDDLE_WALL), bool(fl & ONLY_OCCLUSIONS)
        lower, upper, draw_ceiling = bool(fl & IS_LOWER_WALL), bool(fl & IS_UPPER_WALL), bool(fl & DRAW_CEILING)
        is_full_height_wall = not lower and not upper and not only                         # segs.rs:171-172
        sv = _SidedefVisPlanes(W, visplanes, ci)
        cols = []
        for x in range(c["sx"], c["ex"] + 1):                                              # segs.rs:202
            if not hor_ocl[x]:
                bottom_y = as_i16(c["bsy"] + (F32(x) - c["bsx"]) * c["bdelta"])            # :205-207
                top_y = as_i16(c["tsy"] + (F32(x) - c["tsx"]) * c["tdelta"])               # :208-209
                fvo, cvo = floor_ver_ocl[x], ceiling_ver_ocl[x]
                clipped_bottom_y = min(H - 1, min(fvo, bottom_y))                          # :216-220
                clipped_top_y = max(0, max(cvo, top_y))
                in_area = clipped_bottom_y >= clipped_top_y                                # :225
                if in_area:
                    cols.append((x, clipped_top_y, clipped_bottom_y, bottom_y, top_y))     # :260 (drawing itself is the mapper's business)
                if not two and in_area and (is_full_height_wall or only):                  # :263-266
                    added = False
                    if clipped_bottom_y < fvo and clipped_bottom_y != H - 1:               # :270-275
                        sv.add_bottom_point(x, clipped_bottom_y, fvo)
                        added = True
                    if draw_ceiling and clipped_top_y > cvo and clipped_top_y != -1:       # :278-287
                        sv.add_top_point(x, cvo, clipped_top_y)
                        added = True
                    if not added:                                                          # :289-292
                        sv.flush()
                elif not two and not in_area and (is_full_height_wall or only) and fvo > cvo:   # :293-297
                    if bottom_y <= cvo:                                                    # :303-308
                        sv.add_bottom_point(x, cvo, fvo)
                        occlude_vertical_line(x)
                    if draw_ceiling and top_y >= fvo:                                      # :310-317 (fvo, cvo: the values read at :212-213)
                        sv.add_top_point(x, cvo, fvo)
                        occlude_vertical_line(x)
                if not two and in_area and only:                                           # :320-326
                    floor_ver_ocl[x] = clipped_bottom_y
                    if draw_ceiling:
                        ceiling_ver_ocl[x] = clipped_top_y
                if not two and in_area and lower:                                          # :329-331
                    floor_ver_ocl[x] = clipped_top_y
                if not two and in_area and upper:                                          # :333-335
                    ceiling_ver_ocl[x] = clipped_bottom_y
            else:
                sv.flush()                                                                 # :336-339
            if not two and is_full_height_wall:                                            # :341-344
                occlude_vertical_line(x)
        sv.flush()                                                                         # :347
        all_columns.append(cols)
    return all_columns, visplanes


# ====================================================================================================================================
# The PER-SEG half: BSP visit order, seg transform, frustum clip, projection, the three parts of a wall / portal.
#
#     Renderer::render_node / process_subsector            src/renderer/mod.rs:61-104
#     Segs::process_seg                                     src/renderer/segs.rs:353-590
#     Segs::process_sidedef up to its column loop           src/renderer/segs.rs:121-200
#     clip_to_viewport, perspective_transform,
#     make_sidedef_non_vertical_line                        src/renderer/misc.rs:13-161
#     Line::intersection, Vertex::{rotate, is_left_of_line, distance_to}     src/geometry.rs:56-82, src/map/vertexes.rs:20-38
#     the map lumps                                         src/map/*.rs (record layouts), src/wad.rs:175-183 (first marker of the name)
#
# Again written from those lines with numpy float32 scalars and Python ints, sharing nothing with the oracle or the product.  Together
# with column_loops above this is a complete second front end for walls and visplanes (sprites are not restated).
# ====================================================================================================================================
import struct

TWOSIDED, DONTPEGTOP, DONTPEGBOTTOM = 4, 8, 16            # src/map/linedefs.rs:10-18


def _as_i32(v) -> int:
    v = float(v)
    if v != v:
        return 0
    if v >= 2147483647.0:
        return 2147483647
    if v <= -2147483648.0:
        return -2147483648
    return int(v)


def _i32_as_i16(v: int) -> int:
    return ((v + 32768) & 0xffff) - 32768                 # `as i16` of an i32 wraps


class Map:
    """The lumps process_seg reads, as plain tuples."""

    def __init__(self, wad: bytes, map_name: str):
        n, off = struct.unpack_from("<II", wad, 4)
        lumps = []
        for i in range(n):
            o, s = struct.unpack_from("<II", wad, off + 16 * i)
            lumps.append((wad[off + 16 * i + 8: off + 16 * i + 16].split(b"\0")[0].decode("ascii").upper(), o, s))
        m = next(i for i, l in enumerate(lumps) if l[0] == map_name.upper())     # wad.rs:175-183: the first marker of that name

        def lump(k):
            _, o, s = lumps[m + k]
            return wad[o:o + s]

        def name(b):
            return b.split(b"\0")[0].decode("ascii")                          # wad.rs:112-126 (no case folding)
        linedefs, sidedefs, vertexes, segs, ssectors, nodes, sectors = (lump(k) for k in (2, 3, 4, 5, 6, 7, 8))
        self.vertexes = [tuple(F32(t) for t in struct.unpack_from("<hh", vertexes, 4 * i)) for i in range(len(vertexes) // 4)]
        self.sectors = []
        for i in range(len(sectors) // 26):
            fh, ch = struct.unpack_from("<hh", sectors, 26 * i)
            self.sectors.append({"floor_height": fh, "ceiling_height": ch, "floor_texture": name(sectors[26 * i + 4: 26 * i + 12]),
                                 "ceiling_texture": name(sectors[26 * i + 12: 26 * i + 20]), "light_level": struct.unpack_from("<h", sectors, 26 * i + 20)[0]})
        self.sidedefs = []
        for i in range(len(sidedefs) // 30):
            b = sidedefs[30 * i: 30 * i + 30]
            self.sidedefs.append({"x_offset": struct.unpack_from("<h", b, 0)[0], "y_offset": struct.unpack_from("<h", b, 2)[0],
                                  "upper": name(b[4:12]), "lower": name(b[12:20]), "middle": name(b[20:28]), "sector": self.sectors[struct.unpack_from("<h", b, 28)[0]]})
        self.linedefs = []
        for i in range(len(linedefs) // 14):
            v1, v2, flags, _sp, _tag, fs, bs = struct.unpack_from("<hhhhhhh", linedefs, 14 * i)
            self.linedefs.append({"flags": flags, "front": None if fs == -1 else self.sidedefs[fs], "back": None if bs == -1 else self.sidedefs[bs]})
        self.segs = []
        for i in range(len(segs) // 12):
            v1, v2, _ang, ld, direction, seg_off = struct.unpack_from("<hhhhhh", segs, 12 * i)
            self.segs.append({"start": self.vertexes[v1], "end": self.vertexes[v2], "linedef": self.linedefs[ld], "direction": direction != 0, "offset": seg_off})
        self.subsectors = [struct.unpack_from("<hh", ssectors, 4 * i) for i in range(len(ssectors) // 4)]        # (seg_count, first_seg)
        self.nodes = []
        for i in range(len(nodes) // 28):
            x, y, dx, dy = struct.unpack_from("<hhhh", nodes, 28 * i)
            rc, lc = struct.unpack_from("<HH", nodes, 28 * i + 24)
            self.nodes.append((F32(x), F32(y), F32(dx), F32(dy), rc, lc))


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def _is_left_of_line(p, l0, l1):                           # vertexes.rs:32-34
    a, b = _sub(p, l0), _sub(l1, l0)
    return a[0] * b[1] - a[1] * b[0] <= F32(0.0)


def _intersection(s, e, o0, o1):                          # geometry.rs:56-82; None = "Lines are parallel"
    x1, y1, x2, y2 = s[0], s[1], e[0], e[1]
    x3, y3, x4, y4 = o0[0], o0[1], o1[0], o1[1]
    quot = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    if abs(quot) < F32(0.001):
        return None
    invquot = F32(1.0) / quot
    px = invquot * ((x1 * y2 - y1 * x2) * (x3 - x4) - (x1 - x2) * (x3 * y4 - y3 * x4))
    py = invquot * ((x1 * y2 - y1 * x2) * (y3 - y4) - (y1 - y2) * (x3 * y4 - y3 * x4))
    return (px, py)


def clip_to_viewport(start, end):                         # misc.rs:13-115; -> (start, end, start_offset) or None
    Z, ONE = F32(0.0), F32(1.0)
    left, right = ((Z, Z), (ONE, ONE)), ((Z, Z), (ONE, -ONE))
    start_outside_left, end_outside_left = _is_left_of_line(start, *left), _is_left_of_line(end, *left)
    start_outside_right, end_outside_right = not _is_left_of_line(start, *right), not _is_left_of_line(end, *right)
    start_in = start[0] > Z and not start_outside_left and not start_outside_right
    end_in = end[0] > Z and not end_outside_left and not end_outside_right
    if start_in and end_in:
        return start, end, Z
    li, ri = _intersection(start, end, *left), _intersection(start, end, *right)
    left_x = li is not None and li[0] >= Z
    right_x = ri is not None and ri[0] >= Z
    if not start_in and not end_in and not left_x and not right_x:
        return None
    if not start_in and not end_in and left_x != right_x:
        return None
    if (right_x and start_outside_right and end_outside_right) or (left_x and start_outside_left and end_outside_left):
        return None
    s, e, start_offset = start, end, Z
    if left_x:
        if start_outside_left:
            dx, dy = li[0] - s[0], li[1] - s[1]
            start_offset = np.sqrt(dx * dx + dy * dy)       # new_start.distance_to(&start), vertexes.rs:36-38
            s = li
        if end_outside_left:
            e = li
    if right_x:
        if start_outside_right:
            s = ri
        if end_outside_right:
            e = ri
    return s, e, start_offset


def per_seg_calls(m: Map, W: int, H: int, view):
    """view: x, y, cos(-angle), sin(-angle), floor_height as np.float32 (the trig values the reference's libm returned are an input, as for
    the product).  -> the process_sidedef calls that reach their column loop, in visit order, as column_loops() takes them."""
    arc = F32(200.0) / F32(240.0)                          # constants.rs:3-17
    gcfx = (F32(W) / arc) / F32(2.0)
    cfx, cfy = F32(W) / F32(2.0), F32(H) / F32(2.0)
    pos = (view["x"], view["y"])
    cn, sn = view["cos_neg"], view["sin_neg"]
    calls = []

    def non_vertical_line(s, e, height):                  # misc.rs:130-161
        out = []
        for v in (s, e):
            tx = gcfx * v[1] / v[0]                        # perspective_transform: x = v.y, z = v.x
            ty = gcfx * height / v[0]
            tx = tx * arc
            out.append((min(_as_i32(cfx - tx), W - 1), _as_i32(cfy - ty)))
        return out

    def process_sidedef(s, e, bottom_height, top_height, texture_name, only, lower, upper, draw_ceiling, two, rec=None, offset_y=0):   # segs.rs:121-200
        bottom, top = non_vertical_line(s, e, bottom_height), non_vertical_line(s, e, top_height)
        assert bottom[0][0] == top[0][0] and bottom[1][0] == top[1][0], "Wall start not vertical"
        if _i32_as_i16(bottom[0][0]) == _i32_as_i16(bottom[1][0]) or _i32_as_i16(top[0][0]) == _i32_as_i16(top[1][0]):
            return
        for (x, _) in bottom + top:
            assert 0 <= x < W, "Invalid line x (the reference panics)"
        bdelta = (F32(bottom[0][1]) - F32(bottom[1][1])) / (F32(bottom[0][0]) - F32(bottom[1][0]))
        tdelta = (F32(top[0][1]) - F32(top[1][1])) / (F32(top[0][0]) - F32(top[1][0]))
        flags = (ONLY_OCCLUSIONS if only else 0) | (IS_LOWER_WALL if lower else 0) | (IS_UPPER_WALL if upper else 0) | (DRAW_CEILING if draw_ceiling else 0) | \
                (IS_TWO_SIDED_MIDDLE_WALL if two else 0) | (HAS_TEXTURE if texture_name != "-" else 0)
        c = {"sx": bottom[0][0], "ex": bottom[1][0], "bsy": F32(bottom[0][1]), "bsx": F32(bottom[0][0]), "bdelta": bdelta,
             "tsy": F32(top[0][1]), "tsx": F32(top[0][0]), "tdelta": tdelta, "flags": flags}
        if rec is not None:                                # what BitmapRender::new (segs.rs:186-199) and SidedefVisPlanes::new (:163-169) are given
            c.update(rec)
            c.update({"texture": texture_name, "line": (s[0], s[1], e[0], e[1]), "start_x": bottom[0][0], "end_x": bottom[1][0],
                      "bottom_height": bottom_height, "top_height": top_height,
                      "offset_y": _i32_as_i16(rec["sidedef_y_offset"] + _i32_as_i16(offset_y))})
        calls.append(c)

    def process_seg(seg):                                 # segs.rs:353-590
        ld = seg["linedef"]
        front, back = (ld["back"], ld["front"]) if seg["direction"] else (ld["front"], ld["back"])
        if front is None:
            return
        fs = front["sector"]
        floor_height, ceiling_height = F32(fs["floor_height"]), F32(fs["ceiling_height"])
        portal_bottom = portal_top = None
        if back is not None:
            bsec = back["sector"]
            if bsec["floor_height"] > fs["floor_height"]:
                portal_bottom = F32(bsec["floor_height"])
            if bsec["ceiling_height"] < fs["ceiling_height"]:
                portal_top = F32(bsec["ceiling_height"])
        two_sided = (ld["flags"] & TWOSIDED) != 0
        rot = []
        for v in (seg["start"], seg["end"]):
            mx, my = v[0] - pos[0], v[1] - pos[1]
            rot.append((mx * cn - my * sn, my * cn + mx * sn))            # rotate(-angle), vertexes.rs:20-25
        clipped = clip_to_viewport(rot[0], rot[1])
        if clipped is None:
            return
        s, e, start_offset = clipped
        assert not (s[0] < F32(-0.01)), "Clipped line x < -0.01 (the reference panics)"
        top_unpegged, bottom_unpegged = (ld["flags"] & DONTPEGTOP) != 0, (ld["flags"] & DONTPEGBOTTOM) != 0
        player_height = view["floor_height"] + F32(41.0)
        floor = non_vertical_line(s, e, floor_height - player_height)
        if floor[0][0] > floor[1][0]:
            return
        draw_ceiling = True
        if back is not None and "SKY" in fs["ceiling_texture"] and "SKY" in back["sector"]["ceiling_texture"]:   # the sky hack
            portal_top = None
            ceiling_height = min(F32(back["sector"]["ceiling_height"]), ceiling_height)
            draw_ceiling = False
        rec = {"light_level": fs["light_level"], "start_offset": start_offset, "sidedef_y_offset": front["y_offset"],
               "offset_x": _i32_as_i16(front["x_offset"] + seg["offset"]),                          # sidedef.x_offset as i16 + sds.offset_x (wrapping)
               "floor_flat": fs["floor_texture"], "ceiling_flat": fs["ceiling_texture"],            # (get_animated at timestamp 0 = the first frame: draw side)
               "floor_height_i16": fs["floor_height"], "ceiling_height_i16": fs["ceiling_height"]}
        if not two_sided:
            oy = _as_i32(floor_height - ceiling_height) if bottom_unpegged else 0                     # segs.rs:496-502
            process_sidedef(s, e, floor_height - player_height, ceiling_height - player_height, front["middle"], False, False, False, draw_ceiling, False, rec, oy)
        else:
            process_sidedef(s, e, floor_height - player_height, ceiling_height - player_height, front["middle"], True, False, False, draw_ceiling, False, rec, 0)
            mid_floor = portal_bottom if portal_bottom is not None else floor_height
            mid_ceiling = portal_top if portal_top is not None else ceiling_height
            process_sidedef(s, e, mid_floor - player_height, mid_ceiling - player_height, front["middle"], False, False, False, draw_ceiling, True, rec, 0)
            if portal_bottom is not None:
                oy = _as_i32(ceiling_height - portal_bottom) if bottom_unpegged else 0                # segs.rs:552-558
                process_sidedef(s, e, floor_height - player_height, portal_bottom - player_height, front["lower"], False, True, False, draw_ceiling, False, rec, oy)
            if portal_top is not None:
                oy = 0 if top_unpegged else _as_i32(portal_top - ceiling_height)                      # segs.rs:572-578
                process_sidedef(s, e, portal_top - player_height, ceiling_height - player_height, front["upper"], False, False, True, draw_ceiling, False, rec, oy)

    def subsector(i):                                     # mod.rs:61-66
        count, first = m.subsectors[i]
        for k in range(first, first + count):
            process_seg(m.segs[k])

    # mod.rs:69-104, iteratively: front child first, then the back child — always both
    def render_node(ni):
        x, y, dx, dy, rc, lc = m.nodes[ni]
        v1 = (x, y)
        v2 = (x + dx, y + dy)
        is_left = _is_left_of_line(pos, v1, v2)
        front_c, back_c = (lc, rc) if is_left else (rc, lc)
        for c in (front_c, back_c):
            if c & 0x8000:
                subsector(c & 0x7fff)
            else:
                render_node(c)

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    render_node(len(m.nodes) - 1)
    return calls


# The flats that cycle (flats.rs:30-77): get_animated(name, timestamp) picks list[(timestamp * 3.0) as usize % len]
_ANIMATED = [["NUKAGE1", "NUKAGE2", "NUKAGE3"], ["FWATER1", "FWATER2", "FWATER3", "FWATER4"], ["SWATER1", "SWATER2", "SWATER3", "SWATER4"],
             ["LAVA1", "LAVA2", "LAVA3", "LAVA4"], ["BLOOD1", "BLOOD2", "BLOOD3"], ["RROCK05", "RROCK06", "RROCK07", "RROCK08"],
             ["SLIME01", "SLIME02", "SLIME03", "SLIME04"], ["SLIME05", "SLIME06", "SLIME07", "SLIME08"], ["SLIME09", "SLIME10", "SLIME11", "SLIME12"]]


def get_animated(name: str, timestamp: float) -> str:
    for lst in _ANIMATED:
        if name in lst:
            return lst[int(F32(timestamp) * F32(3.0)) % len(lst)]
    return name


def frame_lists(m: Map, W: int, H: int, view, timestamp: float = 0.0):
    """The whole front end for a frame WITHOUT map objects, as the list dict tests/np_mappers.draw_lists replays: inline walls in visit
    order (segs.rs:231-258), the visplanes in push order (mod.rs:106-116), then — no sprite being there to interleave them —
    the masked middle textures in reversed visit order (mod.rs:124, segs.rs:593-597, bitmap_render.rs:101-135)."""
    calls = per_seg_calls(m, W, H, view)
    columns, visplanes = column_loops(W, H, calls)
    lists = {"renders": [], "columns": [], "visplanes": [], "order": []}
    masked = []
    for c, cols in zip(calls, columns):
        if not (c["flags"] & HAS_TEXTURE) or not cols:
            continue
        two, only = bool(c["flags"] & IS_TWO_SIDED_MIDDLE_WALL), bool(c["flags"] & ONLY_OCCLUSIONS)
        if only:
            continue
        r = {k: c[k] for k in ("texture", "light_level", "offset_x", "offset_y", "line", "start_offset", "start_x", "end_x", "bottom_height", "top_height")}
        r["first_column"], r["n_columns"] = len(lists["columns"]), len(cols)
        lists["columns"] += cols
        lists["renders"].append(r)
        (masked if two else lists["order"]).append((0, len(lists["renders"]) - 1))
    for (ci, which, left, right, tb) in visplanes:
        c = calls[ci]
        flat = get_animated(c["floor_flat"] if which == "floor" else c["ceiling_flat"], timestamp)
        lists["visplanes"].append({"flat": flat, "height": c["floor_height_i16"] if which == "floor" else c["ceiling_height_i16"],
                                   "light_level": c["light_level"], "left": left, "right": right, "tb": tb})
        lists["order"].append((1, len(lists["visplanes"]) - 1))
    lists["order"] += masked[::-1]
    return lists


# ====================================================================================================================================
# Map objects: draw_map_objects (src/renderer/map_objects.rs:19-241), BitmapRender::is_behind_vertex / Ord (bitmap_render.rs:137-188),
# get_sector_from_vertex (bsp.rs:9-44), the sprite lump table (graphics/sprites.rs:26-117, pictures.rs:66-147), MapObjects::new
# (src/map_objects.rs:25-50; the thing type -> spawn state table is DATA taken from data/mobj_spawn.tsv, extracted from src/info.rs).
# With it the numpy code renders complete frames.
# ====================================================================================================================================
import math
import os

_PI = F32(math.pi)


def _as_u8(v) -> int:
    v = float(v)
    if v != v or v <= 0.0:
        return 0
    return 255 if v >= 255.0 else int(v)


def load_things(wad: bytes, map_name: str):
    """MapObjects::new: (x, y, angle in radians as f32::to_radians gives it, sprite, frame, full_bright) per thing that is not a start spot
    and whose spawn state is not S_NULL."""
    n, off = struct.unpack_from("<II", wad, 4)
    lumps = [(wad[off + 16 * i + 8: off + 16 * i + 16].split(b"\0")[0].decode("ascii").upper(),) + struct.unpack_from("<II", wad, off + 16 * i) for i in range(n)]
    mi = next(i for i, l in enumerate(lumps) if l[0] == map_name.upper())
    _, o, s = lumps[mi + 1]
    table = {}
    for line in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "mobj_spawn.tsv")):
        if line.startswith("#") or not line.strip():
            continue
        t = line.split("\t")
        table[int(t[0])] = (t[1], int(t[2]), int(t[3]) != 0, int(t[4]) != 0)
    out = []
    for i in range(s // 10):
        x, y, ang, typ, _flags = struct.unpack_from("<hhhhh", wad, o + 10 * i)
        if 1 <= typ <= 4 or typ == 11:
            continue
        sprite, frame, full_bright, is_null = table[typ]                        # (an unknown type is a panic in the reference)
        if is_null:
            continue
        out.append({"x": F32(x), "y": F32(y), "angle": F32(ang) * (_PI / F32(180.0)), "sprite": sprite, "frame": frame, "full_bright": full_bright})
    return out


class SpriteTable:
    """sprites.rs:26-117 for the sprites that are asked for."""

    def __init__(self, wad: bytes):
        self.wad = wad
        n, off = struct.unpack_from("<II", wad, 4)
        self.lumps = [(wad[off + 16 * i + 8: off + 16 * i + 16].split(b"\0")[0].decode("ascii").upper(),) + struct.unpack_from("<II", wad, off + 16 * i) for i in range(n)]
        last = {nm: i for i, (nm, _, _) in enumerate(self.lumps)}            # get_dir_entry: the last lump of a name (wad.rs:128-157)
        self.first, self.last_idx, self.by_name = last["S_START"], last["S_END"], last
        self.cache = {}

    def _picture(self, name):                                                 # pictures.rs:66-126: (w, h, top_offset, rows[y][x])
        _, o, _ = self.lumps[self.by_name[name]]
        d = self.wad
        w, h, _left, top = struct.unpack_from("<hhhh", d, o)
        px = [[None] * w for _ in range(h)]
        for x in range(w):
            p = o + struct.unpack_from("<I", d, o + 8 + 4 * x)[0]
            while d[p] != 0xFF:
                ytop, cnt = d[p], d[p + 1]
                for k in range(cnt):
                    if ytop + k < h:
                        px[ytop + k][x] = d[p + 3 + k]
                p += cnt + 4
        return w, h, top, px

    def get_picture(self, sprite: str, frame: int, rotation: int):
        key = (sprite, frame)
        if key not in self.cache:
            rots = {}
            for i in range(self.first, self.last_idx):
                nm = self.lumps[i][0]
                if not nm.startswith(sprite):
                    continue
                pic = self._picture(nm)
                if ord(nm[4]) - 65 == frame:
                    rots[ord(nm[5]) - 48] = pic
                if len(nm) > 6 and ord(nm[6]) - 65 == frame:
                    w, h, top, px = pic
                    rots[ord(nm[7]) - 48] = (w, h, top, [row[::-1] for row in px])          # Picture::mirror, pictures.rs:129-147
            if len(rots) != 1:
                assert len(rots) == 8, "Got something other than 8 rotations"
                self.cache[key] = [rots[r] for r in range(1, 9)]
            else:
                self.cache[key] = [rots[0]]
        pics = self.cache[key]
        assert rotation <= 7
        return pics[rotation] if len(pics) == 8 else pics[0]


def get_sector_from_vertex(m: Map, v):                    # bsp.rs:9-44
    ni = len(m.nodes) - 1
    while True:
        x, y, dx, dy, rc, lc = m.nodes[ni]
        c = lc if _is_left_of_line(v, (x, y), (x + dx, y + dy)) else rc
        if c & 0x8000:
            count, first = m.subsectors[c & 0x7fff]
            for k in range(first, first + count):
                seg = m.segs[k]
                sd = seg["linedef"]["back"] if seg["direction"] else seg["linedef"]["front"]
                if sd is not None:
                    return sd["sector"]
            return None
        ni = c


def render_frame(m: Map, things, sprites: SpriteTable, np_wad, nm, W: int, H: int, view, sky_name: str = "SKY1", timestamp: float = 0.0):
    """Renderer::render (mod.rs:118-136) in full: walls inline, visplanes, map objects with the masked walls behind them, the remaining
    masked walls.  np_wad / nm: tests/np_mappers.py's Wad and module (the three texture mappers).  -> H x W x 3 uint8."""
    calls = per_seg_calls(m, W, H, view)
    columns, visplanes = column_loops(W, H, calls)
    fr = nm.Frame(W, H)
    pal = np_wad.palette()
    tex_cache, flat_cache = {}, {}
    mview = {"x": view["x"], "y": view["y"], "angle": view["angle"], "cos": view["cos"], "sin": view["sin"], "floor_height": view["floor_height"]}

    def texture(name):
        if name not in tex_cache:
            tex_cache[name] = np_wad.texture(name)
        return tex_cache[name]

    def replay(rec, cols, bitmap):
        for col in cols:
            nm.render_vertical_bitmap_line(fr, bitmap, pal, rec, col)

    # the records Segs.segs holds after the BSP walk (one per process_sidedef call, segs.rs:186-199, 349)
    segs = []
    for c, cols in zip(calls, columns):
        fl = c["flags"]
        two, only = bool(fl & IS_TWO_SIDED_MIDDLE_WALL), bool(fl & ONLY_OCCLUSIONS)
        lower, upper = bool(fl & IS_LOWER_WALL), bool(fl & IS_UPPER_WALL)
        full = not lower and not upper and not only
        rec = {k: c[k] for k in ("light_level", "offset_x", "offset_y", "line", "start_offset", "start_x", "end_x", "bottom_height", "top_height")}
        bitmap = texture(c["texture"]) if (fl & HAS_TEXTURE) else None
        if bitmap is not None and not two and not only:
            replay(rec, cols, bitmap)                                         # drawn inline, in visit order (segs.rs:231-258)
        segs.append({"state": "two" if two else "solid", "bitmap": bitmap, "rec": rec, "cols": cols, "line": c["line"],
                     "extends_to_bottom": lower or (not two and full), "extends_to_top": upper or (not two and full), "draw_ceiling": bool(fl & DRAW_CEILING)})
    for (ci, which, left, right, tb) in visplanes:                            # mod.rs:106-116
        c = calls[ci]
        p = {"flat": get_animated(c["floor_flat"] if which == "floor" else c["ceiling_flat"], timestamp), "height": c["floor_height_i16"] if which == "floor" else c["ceiling_height_i16"],
             "light_level": c["light_level"], "left": left, "right": right, "tb": tb}
        if "SKY" in p["flat"]:
            nm.draw_sky(fr, texture(sky_name), pal, mview, p)
        else:
            if p["flat"] not in flat_cache:
                flat_cache[p["flat"]] = np_wad.flat(p["flat"])
            nm.draw_visplane(fr, flat_cache[p["flat"]], pal, mview, p)
    segs.reverse()                                                            # mod.rs:124

    def is_behind_vertex(seg, v):                                             # bitmap_render.rs:137-165
        sx, sy, ex, ey = seg["line"]
        if min(sx, ex) > v[0]:
            return True
        return max(sx, ex) > v[0] and not _is_left_of_line(v, (sx, sy), (ex, ey))

    def render_seg(seg):                                                      # bitmap_render.rs:101-135
        if seg["state"] != "two":
            return
        if seg["bitmap"] is not None:
            replay(seg["rec"], seg["cols"], seg["bitmap"])
        seg["state"] = "drawn"

    # draw_map_objects, renderer/map_objects.rs:19-241
    cn, sn = view["cos_neg"], view["sin_neg"]
    arc = F32(200.0) / F32(240.0)
    gcfx = (F32(W) / arc) / F32(2.0)
    cfx, cfy = F32(W) / F32(2.0), F32(H) / F32(2.0)

    def non_vertical_line(s, e, height):                                      # misc.rs:138-161
        out = []
        for v in (s, e):
            tx = gcfx * v[1] / v[0] * arc
            ty = gcfx * height / v[0]
            out.append((min(_as_i32(cfx - tx), W - 1), _as_i32(cfy - ty)))
        return out
    objects = []
    two_pi = F32(2.0) * _PI
    for t in things:
        angle = view["angle"] - t["angle"] - _PI                              # :55-67
        angle = angle + _PI / F32(16.0)
        angle = F32(math.fmod(float(angle), float(two_pi)))
        if angle < F32(0.0):
            angle = angle + two_pi
        angle = F32(math.fmod(float(angle), float(two_pi)))
        rotation = _as_u8(angle * F32(8.0) / two_pi)
        w, h, top_offset, px = sprites.get_picture(t["sprite"], t["frame"], rotation)
        mx, my = t["x"] - view["x"], t["y"] - view["y"]
        vp = (mx * cn - my * sn, my * cn + mx * sn)
        start = (vp[0] - F32(0.0), vp[1] - (-F32(w) / F32(2.0)))              # :81-82
        end = (vp[0] - F32(0.0), vp[1] - (F32(w) / F32(2.0)))
        clipped = clip_to_viewport(start, end)
        if clipped is None:
            continue
        s, e, start_offset = clipped
        assert not (s[0] < F32(-0.01))
        sector = get_sector_from_vertex(m, (t["x"], t["y"]))
        if sector is None:
            continue
        light_level = 255 if t["full_bright"] else sector["light_level"]
        player_height = view["floor_height"] + F32(41.0)
        z = sector["floor_height"]
        bottom_height = F32(z) - player_height
        top_height = F32(z) + F32(h) - F32(1.0) - player_height
        bottom_height = bottom_height + (F32(top_offset) - F32(h))
        top_height = top_height + (F32(top_offset) - F32(h))
        bottom, top = non_vertical_line(s, e, bottom_height), non_vertical_line(s, e, top_height)
        top_clip, bottom_clip = [-1] * W, [H] * W
        for seg in segs:                                                      # :130-166
            if is_behind_vertex(seg, vp):
                continue
            for (x, ct, cb, by, ty) in seg["cols"]:
                if seg["state"] == "solid":
                    if seg["extends_to_bottom"]:
                        bottom_clip[x] = min(bottom_clip[x], ct)
                    if seg["extends_to_top"]:
                        top_clip[x] = max(top_clip[x], cb)
                elif seg["state"] == "two":
                    if seg["draw_ceiling"]:
                        top_clip[x] = max(top_clip[x], ty)
                    bottom_clip[x] = min(bottom_clip[x], by)
        with np.errstate(all="ignore"):
            bdelta = (F32(bottom[0][1]) - F32(bottom[1][1])) / (F32(bottom[0][0]) - F32(bottom[1][0]))
            tdelta = (F32(top[0][1]) - F32(top[1][1])) / (F32(top[0][0]) - F32(top[1][0]))
            cols = []
            for x in range(_i32_as_i16(bottom[0][0]), _i32_as_i16(bottom[1][0])):      # :194: the end is exclusive
                by = as_i16(F32(bottom[0][1]) + (F32(x) - F32(bottom[0][0])) * bdelta)
                ty = as_i16(F32(top[0][1]) + (F32(x) - F32(top[0][0])) * tdelta)
                ct = max(0, max(ty, top_clip[x]))
                cb = min(H - 1, min(by, bottom_clip[x]))
                cols.append((x, ct, cb, by, ty))
        rec = {"light_level": light_level, "offset_x": 0, "offset_y": 0, "line": (s[0], s[1], e[0], e[1]), "start_offset": start_offset,
               "start_x": bottom[0][0], "end_x": bottom[1][0], "bottom_height": bottom_height, "top_height": top_height}
        objects.append({"rec": rec, "cols": cols, "bitmap": (w, h, px), "key": as_i16(s[0])})
    objects.sort(key=lambda o: o["key"])                                      # :216-217: stable sort on `start.x as i16`, then reversed
    objects.reverse()
    for o in objects:                                                         # :220-240
        sx, sy, ex, ey = o["rec"]["line"]
        mid = ((sx + ex) / F32(2.0), (sy + ey) / F32(2.0))
        for seg in segs:
            if is_behind_vertex(seg, mid):
                render_seg(seg)
        replay(o["rec"], o["cols"], o["bitmap"])
    for seg in segs:                                                          # draw_remaining_segs, segs.rs:593-597
        render_seg(seg)
    return fr.px
