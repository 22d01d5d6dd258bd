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
