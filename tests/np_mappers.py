"""np_mappers.py — a SECOND, independent restatement of the reference's three texture mappers, in numpy float32 / Python integers.

Test infrastructure.  It shares no code with oracle/doomref.c, with the product's host side or with its kernels: WAD lumps are
decoded here from the raw bytes, casts and wrapping arithmetic are spelled out with Python integers, and every f32 operation is a
numpy.float32 scalar operation in the reference's operand order.  tests/test_edge_kats.py aims hand-built draw lists at the edge
cases of SURVEY.md Appendix A and requires oracle == this == GPU.

Reference lines followed (paths relative to freewilll/doom-rust-renderer):
  Pixels::set                    src/renderer/pixels.rs:22-31
  diminish_color                 src/renderer/bitmap_render.rs:190-208
  render_vertical_bitmap_line    src/renderer/bitmap_render.rs:213-276   (BitmapRender::render :101-135)
  draw_visplane / draw_sky       src/renderer/visplanes.rs:82-152 / :42-80
  constants                      src/renderer/constants.rs:3-17
  WadFile / Picture / Texture    src/wad.rs:86-195, src/graphics/pictures.rs:66-126, src/graphics/textures.rs:74-103,182-255
  Flat / Palette                 src/graphics/flats.rs:116-136, src/graphics/palette.rs:11-28
"""
import math
import struct

import numpy as np

f32 = np.float32


# ---- Rust integer / cast semantics (release build) --------------------------------------------------------------------------

def w16(v: int) -> int:
    """i16 wrapping (`as i16` of a wider integer, and i16 + / * in release builds)."""
    return ((int(v) + 32768) & 0xFFFF) - 32768


def tdiv(a: int, b: int) -> int:
    """Rust integer `/`: truncates toward zero."""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def trem(a: int, b: int) -> int:
    """Rust integer `%`: sign of the dividend."""
    return a - b * tdiv(a, b)


def f_as_int(v, lo: int, hi: int) -> int:
    """float `as` integer: NaN -> 0, saturating, truncating toward zero."""
    v = float(v)
    if math.isnan(v):
        return 0
    if v <= lo:
        return lo
    if v >= hi:
        return hi
    return int(v)          # int() truncates toward zero


def f_as_i16(v) -> int:
    return f_as_int(v, -32768, 32767)


def f_as_u8(v) -> int:
    return f_as_int(v, 0, 255)


# ---- WAD decoding (only what the KAT needs) -----------------------------------------------------------------------------------

class Wad:
    def __init__(self, data: bytes):
        self.data = data
        assert data[:4] == b"IWAD"
        n, off = struct.unpack_from("<II", data, 4)
        self.lumps = []
        for i in range(n):
            o, s = struct.unpack_from("<II", data, off + 16 * i)
            name = data[off + 16 * i + 8: off + 16 * i + 16].split(b"\0")[0].decode("ascii").upper()
            self.lumps.append((name, o, s))

    def lump(self, name: str) -> bytes:
        # get_dir_entry: the reference's directory HashMap keeps the LAST entry of a name (src/wad.rs:128-157)
        for (n, o, s) in reversed(self.lumps):
            if n == name.upper():
                return self.data[o:o + s]
        raise KeyError(name)

    def palette(self):
        p = self.lump("PLAYPAL")
        return [(p[3 * i], p[3 * i + 1], p[3 * i + 2]) for i in range(256)]

    def flat(self, name: str):
        d = self.lump(name)
        return [[d[y * 64 + x] for x in range(64)] for y in range(64)]          # flat.pixels[y][x], flats.rs:126-130

    def picture(self, name: str):
        """-> (w, h, rows[y][x] with None = transparent) — column / post format, pictures.rs:66-126."""
        d = self.lump(name)
        w, h, _left, _top = struct.unpack_from("<hhhh", d, 0)
        px = [[None] * w for _ in range(h)]
        for x in range(w):
            o = struct.unpack_from("<I", d, 8 + 4 * x)[0]
            while d[o] != 0xFF:
                ytop, n = d[o], d[o + 1]
                for k in range(n):
                    if ytop + k < h:
                        px[ytop + k][x] = d[o + 3 + k]
                o += n + 4
        return w, h, px

    def texture(self, name: str):
        """Composite of the texture's patches in order, later patches overwrite (also with None) — textures.rs:74-103."""
        pn = self.lump("PNAMES")
        pnames = [pn[4 + 8 * i: 12 + 8 * i].split(b"\0")[0].decode("ascii").upper() for i in range(struct.unpack_from("<I", pn, 0)[0])]
        found = None
        for lump_name in ("TEXTURE1", "TEXTURE2"):
            try:
                t = self.lump(lump_name)
            except KeyError:
                continue
            for i in range(struct.unpack_from("<I", t, 0)[0]):
                o = struct.unpack_from("<I", t, 4 + 4 * i)[0]
                if t[o:o + 8].split(b"\0")[0].decode("ascii").upper() == name.upper():
                    found = (t, o)                                             # a later definition replaces an earlier one
        if found is None:
            raise KeyError(name)
        t, o = found
        w, h = struct.unpack_from("<hh", t, o + 12)
        px = [[None] * w for _ in range(h)]
        for j in range(struct.unpack_from("<h", t, o + 20)[0]):
            ox, oy, pi = struct.unpack_from("<hhh", t, o + 22 + 10 * j)
            pw, ph, ppx = self.picture(pnames[pi])
            for x in range(pw):
                for y in range(ph):
                    X, Y = w16(x + ox), w16(y + oy)
                    if 0 <= X < w and 0 <= Y < h:
                        px[Y][X] = ppx[y][x]
        return w, h, px


# ---- the renderer's pixel functions -----------------------------------------------------------------------------------------------

class Frame:
    def __init__(self, W: int, H: int):
        self.W, self.H = W, H
        self.px = np.zeros((H, W, 3), dtype=np.uint8)                         # Pixels::new: zeroed (pixels.rs:10-14)
        arc = f32(200.0) / f32(240.0)                                         # constants.rs:7
        gsw = f32(W) / arc
        self.ARC, self.GCFX, self.CFX, self.CFY = arc, gsw / f32(2.0), f32(W) / f32(2.0), f32(H) / f32(2.0)

    def set(self, x: int, y: int, rgb):
        # x, y arrive as `as usize` of signed values: negatives are huge and fail the x >= W test (pixels.rs:23)
        if x < 0 or x >= self.W or y < 0 or y > self.H:
            return
        assert y != self.H, "the reference would index out of bounds"
        self.px[y, x] = rgb


def diminish_color(rgb, light_level: int, distance: int):
    with np.errstate(all="ignore"):
        factor = f32(light_level) / f32(255.0)
        factor = factor - f32(distance) * (f32(1.0) / (f32(16.0) * f32(256.0)))
        if factor < f32(0.0):
            factor = f32(0.0)
        return tuple(f_as_u8(f32(c) * factor) for c in rgb)


def render_vertical_bitmap_line(fr: Frame, bitmap, palette, r: dict, col):
    """bitmap = (w, h, rows); r = a render record of the list dict; col = (x, clipped_top_y, clipped_bottom_y, bottom_y, top_y)."""
    bw, bh, rows = bitmap
    x, ctop, cbot, bottom_y, top_y = [int(t) for t in col]
    sx, sy, ex, ey = [f32(t) for t in r["line"]]
    with np.errstate(all="ignore"):
        dx, dy = sx - ex, sy - ey
        length = np.sqrt(dx * dx + dy * dy)                                   # geometry.rs:84-86 (powi(2) = x * x)
        ux0, ux1 = f32(0.0), f32(length)
        uy0, uy1 = f32(0.0), f32(r["top_height"]) - f32(r["bottom_height"])
        uz0, uz1 = sx, ex
        one = f32(1.0)
        ax = f32(x - r["start_x"]) / f32(r["end_x"] - r["start_x"])
        tx = f_as_i16(((one - ax) * (ux0 / uz0) + ax * (ux1 / uz1)) / ((one - ax) * (one / uz0) + ax * (one / uz1)))
        tx = w16(tx + w16(f_as_i16(f32(r["start_offset"])) + r["offset_x"]))
        if tx < 0:
            tx = w16(tx + w16(bw * w16(1 - tdiv(tx, bw))))
        tx = trem(tx, bw)
        z = f_as_i16(((one - ax) + ax) / ((one - ax) * (one / uz0) + ax * (one / uz1)))
        for y in range(ctop, cbot + 1):
            ay = f32(y - top_y) / f32(bottom_y - top_y)
            ty = f_as_i16(f32(bh) + (one - ay) * uy0 + ay * uy1)
            ty = w16(ty + r["offset_y"])
            if ty < 0:
                ty = w16(ty + w16(bh * w16(1 - tdiv(ty, bh))))
            ty = trem(ty, bh)
            texel = rows[ty][tx]
            if texel is None:
                continue
            fr.set(x, y, diminish_color(palette[texel], r["light_level"], z))


def draw_sky(fr: Frame, sky, palette, view, p: dict):
    sw, sh, rows = sky
    with np.errstate(all="ignore"):
        tx_offset = w16(f_as_i16(f32(-256.0) * f32(view["angle"]) / (f32(math.pi) / f32(2.0))) + 256)
        if tx_offset < 0:
            tx_offset = w16(tx_offset + w16(256 * w16(1 - tdiv(tx_offset, 256))))
        for i, x in enumerate(range(p["left"], p["right"] + 1)):
            top = max(p["tb"][i][0], 0)
            bottom = min(p["tb"][i][1], fr.H - 1)
            for y in range(top, bottom + 1):
                tx = trem(w16(f_as_i16(f32(x) * f32(256.0) / f32(fr.W)) + tx_offset), 256)
                ty = f_as_i16(f32(y) * f32(128.0) * f32(2.0) / f32(fr.H))
                if ty < 0:
                    ty = w16(ty + 128)
                ty = trem(ty, 128)
                texel = rows[ty][tx]
                if texel is not None:
                    fr.set(x, y, palette[texel])


def draw_visplane(fr: Frame, flat, palette, view, p: dict):
    with np.errstate(all="ignore"):
        c, s = f32(view["cos"]), f32(view["sin"])
        for i, x in enumerate(range(p["left"], p["right"] + 1)):
            top = max(p["tb"][i][0], 0)
            bottom = min(p["tb"][i][1], fr.H - 1)
            if w16(bottom - top) <= 1:
                continue
            for y in range(top, bottom + 1):
                vx = (fr.CFX - f32(x)) / fr.ARC
                vy = fr.CFY - f32(y)
                wz = f32(p["height"]) - f32(view["floor_height"]) - f32(41.0)
                wx = fr.GCFX * wz / vy
                wy = wz * vx / vy
                rx = wx * c - wy * s
                ry = wy * c + wx * s
                tx = w16(f_as_i16(rx) + f_as_i16(f32(view["x"]))) & 63
                ty = w16(f_as_i16(ry) + f_as_i16(f32(view["y"]))) & 63
                fr.set(x, y, diminish_color(palette[flat[ty][tx]], p["light_level"], f_as_i16(wx)))


def draw_lists(wad: Wad, sky_name: str, W: int, H: int, view: dict, lists: dict) -> np.ndarray:
    """Replays the list dict (renders / columns / visplanes / order) like Renderer::render would issue the calls."""
    fr = Frame(W, H)
    pal = wad.palette()
    tex_cache, flat_cache = {}, {}
    for kind, idx in lists["order"]:
        if kind == 0:
            r = lists["renders"][idx]
            if r["texture"] not in tex_cache:
                tex_cache[r["texture"]] = wad.texture(r["texture"])
            for col in lists["columns"][r["first_column"]: r["first_column"] + r["n_columns"]]:
                render_vertical_bitmap_line(fr, tex_cache[r["texture"]], pal, r, col)
        else:
            p = lists["visplanes"][idx]
            if "SKY" in p["flat"]:
                if "sky" not in tex_cache:
                    tex_cache["sky"] = wad.texture(sky_name)
                draw_sky(fr, tex_cache["sky"], pal, view, p)
            else:
                if p["flat"] not in flat_cache:
                    flat_cache[p["flat"]] = wad.flat(p["flat"])
                draw_visplane(fr, flat_cache[p["flat"]], pal, view, p)
    return fr.px
