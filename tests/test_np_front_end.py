"""The per-column half of the front end against a code-disjoint restatement (tests/np_front_end.py, written from
src/renderer/segs.rs:202-345 and sidedef_visplanes.rs): the process_sidedef calls the product's host walk hands to the device column
walk (FePart records) are run through the numpy column loop, and what it leaves behind — wall columns and visplanes — must be what
the product's list builder (dg_build_lists: BitmapRender columns, Visplane entries, draw order) produced for the same view.
Together with tests/np_mappers.py (the three texture mappers) the reference's per-column and per-pixel work now has a second
opinion that shares no code with oracle/doomref.c or the product; the per-seg half (BSP order, clip, projection) has not."""
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


@pytest.mark.parametrize("seed,vanilla,size", [(1993, False, (320, 200)), (1993, False, (132, 67)), (1995, True, (320, 200)), (1994, False, (256, 160))])
def test_column_loop_restatement_agrees_with_the_list_builder(synth, campath_mod, oracle, seed, vanilla, size):
    W, H = size
    wad = synth.build_synth_iwad(seed, heavy=(seed == 1994), vanilla=vanilla)
    sc = emul_bind.EmulScene(wad)
    osc = oracle.Scene(wad, "e1m1")
    path = campath_mod.make_camera_path(synth.synth_route(seed, heavy=(seed == 1994), vanilla=vanilla), osc.floor_height_at, 1000)
    checked_cols = checked_planes = masked = 0
    for i in range(0, 1000, 83):
        calls = frame_parts(sc, W, H, path[i])
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
