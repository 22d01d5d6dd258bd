"""The front end against a code-disjoint restatement (tests/np_front_end.py, written from the reference's src/renderer/{mod,segs,misc,
sidedef_visplanes,visplanes}.rs, geometry.rs and the map loaders):
  * the numpy column loop on the process_sidedef calls the product's host walk hands to the device column walk (FePart records), and
  * the same column loop on calls produced by the numpy per-seg half (BSP visit order, seg transform, clip_to_viewport, projection, the
    parts of a wall / portal) — a complete second front end for walls and visplanes —
must leave behind what the product's list builder (dg_build_lists: BitmapRender columns, Visplane entries, draw order) produced for the
same view; and the numpy front end feeding the numpy mappers (tests/np_mappers.py) must render WHOLE FRAMES byte-identical to
oracle/doomref.c when the map objects are switched off.  Nothing in the numpy code is shared with the oracle or the product; what
still has no second restatement is draw_map_objects (sprite projection, clipping against the recorded wall columns, the sprite /
masked-wall interleave)."""
import ctypes

import numpy as np
import pytest

import emul_bind
import np_front_end as nf

CAP = 1 << 16


def _view(rec):
    return emul_bind.DgView(float(rec[0]), float(rec[1]), float(rec[2]), float(rec[7]), float(rec[3]), float(rec[4]), float(rec[5]), float(rec[6]), 0.0, 1)


def frame_parts(sc, W, H, rec):
    L = emul_bind.lib()
    L.emul_frame_parts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(emul_bind.DgView), ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
    buf = (ctypes.c_uint32 * (12 * 4096))()
    v = _view(rec)
    n = L.emul_frame_parts(sc._h, W, H, ctypes.byref(v), buf, 4096)
    assert n >= 0, L.emul_last_error().decode()
    a = np.frombuffer(buf, dtype=np.uint32)[:12 * n].reshape(n, 12)
    f = a.view(np.float32)
    return [{"sx": int(np.int32(a[i, 0])), "ex": int(np.int32(a[i, 1])), "bsy": f[i, 2], "bsx": f[i, 3], "bdelta": f[i, 4],
             "tsy": f[i, 5], "tsx": f[i, 6], "tdelta": f[i, 7], "flags": int(a[i, 8]), "seq": int(a[i, 9])} for i in range(n)]


def frame_lists(sc, W, H, rec):
    L = emul_bind.lib()
    I32, I16, U32 = ctypes.c_int32, ctypes.c_int16, ctypes.c_uint32
    L.emul_frame_lists.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(emul_bind.DgView), ctypes.POINTER(I32), ctypes.POINTER(I16),
                                   ctypes.POINTER(I32), ctypes.POINTER(I16), ctypes.POINTER(U32), ctypes.POINTER(U32), U32]
    r, c, vp, tb, od, cn = (I32 * (4 * CAP))(), (I16 * (5 * CAP))(), (I32 * (4 * CAP))(), (I16 * (2 * CAP))(), (U32 * (2 * CAP))(), (U32 * 5)()
    v = _view(rec)
    assert L.emul_frame_lists(sc._h, W, H, ctypes.byref(v), r, c, vp, tb, od, cn, CAP) == 0, L.emul_last_error().decode()
    nr, nc, nv, nt, no = list(cn)
    cols = np.frombuffer(c, dtype=np.int16)[:5 * nc].reshape(nc, 5)
    rr = np.frombuffer(r, dtype=np.int32)[:4 * nr].reshape(nr, 4)
    renders = [[tuple(int(t) for t in cols[j]) for j in range(rr[i, 2], rr[i, 2] + rr[i, 3])] for i in range(nr)]
    vv = np.frombuffer(vp, dtype=np.int32)[:4 * nv].reshape(nv, 4)
    tbl = np.frombuffer(tb, dtype=np.int16)[:nt].reshape(-1, 2)
    planes = [(int(vv[i, 0]), int(vv[i, 1]), [tuple(int(t) for t in tbl[vv[i, 2] + k]) for k in range(vv[i, 1] - vv[i, 0] + 1)]) for i in range(nv)]
    order = np.frombuffer(od, dtype=np.uint32)[:2 * no].reshape(no, 2)
    return renders, planes, [(int(k), int(i)) for k, i in order]


@pytest.mark.parametrize("per_seg", ["product-records", "numpy-per-seg-half"])
@pytest.mark.parametrize("seed,vanilla,size", [(1993, False, (320, 200)), (1993, False, (132, 67)), (1995, True, (320, 200)), (1994, False, (256, 160))])
def test_column_loop_restatement_agrees_with_the_list_builder(synth, campath_mod, oracle, seed, vanilla, size, per_seg):
    """product-records: the numpy column loop on the process_sidedef calls the product ships to the GPU (FePart).
    numpy-per-seg-half: the same column loop on calls produced by the numpy BSP walk / process_seg / clip / projection — a complete second
    front end (walls and visplanes) that never touches product or oracle code; it walks every seg like the reference, the product culls."""
    W, H = size
    wad = synth.build_synth_iwad(seed, heavy=(seed == 1994), vanilla=vanilla)
    sc = emul_bind.EmulScene(wad)
    osc = oracle.Scene(wad, "e1m1")
    path = campath_mod.make_camera_path(synth.synth_route(seed, heavy=(seed == 1994), vanilla=vanilla), osc.floor_height_at, 1000)
    np_map = nf.Map(wad, "e1m1") if per_seg == "numpy-per-seg-half" else None
    checked_cols = checked_planes = masked = 0
    for i in range(0, 1000, 83):
        if np_map is None:
            calls = frame_parts(sc, W, H, path[i])
        else:
            r = path[i]
            calls = nf.per_seg_calls(np_map, W, H, {"x": r[0], "y": r[1], "cos_neg": r[5], "sin_neg": r[6], "floor_height": r[7]})
        columns, visplanes = nf.column_loops(W, H, calls)
        renders, planes, order = frame_lists(sc, W, H, path[i])
        # phase 1 of the draw order = the inline wall draws (segs.rs:231-258), in visit order: textured, not two-sided, not occlusion-only
        first_plane = next((k for k, (kind, _) in enumerate(order) if kind == 1), len(order))
        inline = [renders[idx] for kind, idx in order[:first_plane]]
        want_inline = [cols for c, cols in zip(calls, columns) if (c["flags"] & nf.HAS_TEXTURE) and not (c["flags"] & (nf.IS_TWO_SIDED_MIDDLE_WALL | nf.ONLY_OCCLUSIONS)) and cols]
        assert inline == want_inline, f"frame {i}: inline wall columns differ"
        checked_cols += sum(len(r) for r in inline)
        # phase 2 = every visplane in push order (mod.rs:106-116)
        assert [idx for kind, idx in order if kind == 1] == list(range(len(planes)))
        assert [(l, r, tb) for (_, _, l, r, tb) in visplanes] == planes, f"frame {i}: visplanes differ"
        checked_planes += len(planes)
        # phases 3 / 4: the masked middle textures replayed later (segs.rs:593-597, map_objects.rs:216-240) carry the columns recorded here
        late = [renders[idx] for kind, idx in order[first_plane:] if kind == 0]
        for c, cols in zip(calls, columns):
            if (c["flags"] & nf.IS_TWO_SIDED_MIDDLE_WALL) and (c["flags"] & nf.HAS_TEXTURE) and cols:
                assert cols in late, f"frame {i}: a masked wall's recorded columns are not among the late draws"
                masked += 1
    assert checked_cols > 1000 and checked_planes > 50 and masked > 0


@pytest.mark.parametrize("seed,vanilla", [(1993, False), (1995, True), (1994, False)])
def test_numpy_renderer_equals_the_oracle_without_map_objects(synth, campath_mod, oracle, seed, vanilla):
    """Whole frames from code that shares nothing with the oracle or the product: the numpy front end (per-seg half + column loop,
    np_front_end.py) feeding the numpy mappers (np_mappers.py), against oracle/doomref.c with every map object switched off (sprites are
    the one part of the frame that has no second restatement).  160x100, a handful of path frames: pure-Python pixel loops."""
    import np_mappers as nm
    W, H = 160, 100
    wad = synth.build_synth_iwad(seed, heavy=(seed == 1994), vanilla=vanilla)
    osc = oracle.Scene(wad, "e1m1")
    for mo in range(osc.mobj_count()):
        osc.set_mobj_state(mo, None)                                          # S_NULL: draw_map_objects skips it (renderer/map_objects.rs:37)
    path = campath_mod.make_camera_path(synth.synth_route(seed, heavy=(seed == 1994), vanilla=vanilla), osc.floor_height_at, 1000)
    np_map, np_wad = nf.Map(wad, "e1m1"), nm.Wad(wad)
    for i in (0, 217, 431, 640, 858):
        r = path[i]
        lists = nf.frame_lists(np_map, W, H, {"x": r[0], "y": r[1], "cos_neg": r[5], "sin_neg": r[6], "floor_height": r[7]})
        got = nm.draw_lists(np_wad, "SKY1", W, H, {"x": r[0], "y": r[1], "angle": r[2], "cos": r[3], "sin": r[4], "floor_height": r[7]}, lists)
        want = np.frombuffer(osc.render(W, H, r), dtype=np.uint8).reshape(H, W, 3)
        bad = np.argwhere(np.any(got != want, axis=2))
        assert len(bad) == 0, f"frame {i}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): numpy {got[bad[0][0], bad[0][1]]} oracle {want[bad[0][0], bad[0][1]]}"
        assert want.any()


@pytest.mark.parametrize("seed,vanilla", [(1993, False), (1995, True), (1994, False)])
def test_numpy_renderer_equals_the_oracle_with_map_objects(synth, campath_mod, oracle, seed, vanilla):
    """The complete frame — Renderer::render (mod.rs:118-136) with draw_map_objects — from the numpy code alone against the oracle:
    sprite rotation and projection, the clip arrays from the recorded wall columns, the far-to-near order with the masked walls drawn
    behind each object, the remaining masked walls."""
    import np_mappers as nm
    W, H = 160, 100
    wad = synth.build_synth_iwad(seed, heavy=(seed == 1994), vanilla=vanilla)
    osc = oracle.Scene(wad, "e1m1")
    path = campath_mod.make_camera_path(synth.synth_route(seed, heavy=(seed == 1994), vanilla=vanilla), osc.floor_height_at, 1000)
    np_map, np_wad, things, sprites = nf.Map(wad, "e1m1"), nm.Wad(wad), nf.load_things(wad, "e1m1"), nf.SpriteTable(wad)
    assert len(things) == osc.mobj_count() and len(things) > 10
    sprite_pixels = 0
    for i in (0, 217, 431, 640, 858, 100, 323, 728):
        r = path[i]
        view = {"x": r[0], "y": r[1], "angle": r[2], "cos": r[3], "sin": r[4], "cos_neg": r[5], "sin_neg": r[6], "floor_height": r[7]}
        got = nf.render_frame(np_map, things, sprites, np_wad, nm, W, H, view)
        want = np.frombuffer(osc.render(W, H, r), dtype=np.uint8).reshape(H, W, 3)
        bad = np.argwhere(np.any(got != want, axis=2))
        assert len(bad) == 0, f"frame {i}: {len(bad)} pixels differ, first at (x={bad[0][1]}, y={bad[0][0]}): numpy {got[bad[0][0], bad[0][1]]} oracle {want[bad[0][0], bad[0][1]]}"
        nothings = nf.render_frame(np_map, [], sprites, np_wad, nm, W, H, view)
        sprite_pixels += int(np.any(got != nothings, axis=2).sum())
    assert sprite_pixels > 300                                                # the objects really are in these frames
