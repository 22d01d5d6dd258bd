"""F1/F2 loader semantics: the product's scene loader (csrc/scene.cpp) and the oracle's
(oracle/doomref.c) are independent restatements of src/wad.rs, src/map/*.rs, src/graphics/*.rs — they must agree on a
WAD full of edge cases: duplicate lump names, a second map marker, TEXTURE2 redefinitions, 8-character names,
lower-case texture / flat names, and the map-name -> sky texture rule of src/game.rs:199-227."""
import numpy as np
import pytest

import emul_bind

SAMPLES = list(range(0, 1000, 50))


@pytest.mark.parametrize("map_name", ["E1M1", "e1m1", "e2m3", "e3m9", "e4m1", "map07", "MAP12", "MAP21", "level7", "e12m3"])
def test_product_loader_matches_oracle_on_quirky_wad(oracle, synth, path1993, map_name):
    wad = synth.build_synth_iwad(1993, map_name=map_name, quirks=True)
    osc = oracle.Scene(wad, map_name)
    es = emul_bind.EmulScene(wad, map_name)
    for i in SAMPLES:
        assert es.render(320, 200, path1993[i])[0] == osc.render(320, 200, path1993[i]), f"{map_name} frame {i}"


def test_quirks_are_visible_and_sky_follows_the_map_name(oracle, synth, path1993):
    """The decoy / redefinition rules change pixels (otherwise the test above proves nothing), and the three sky
    textures are really selected by episode / map number."""
    plain = oracle.Scene(synth.build_synth_iwad(1993), "e1m1")
    skies = {}
    for name in ("e1m1", "e2m1", "e3m1", "MAP05", "MAP15", "MAP25"):
        sc = oracle.Scene(synth.build_synth_iwad(1993, map_name=name, quirks=True), name)
        skies[name] = [sc.render(320, 200, path1993[i]) for i in (600, 610, 620, 700)]      # sky-heavy frames
    assert any(a != plain.render(320, 200, path1993[i]) for a, i in zip(skies["e1m1"], (600, 610, 620, 700)))
    assert skies["e1m1"] == skies["MAP05"] and skies["e2m1"] == skies["MAP15"] and skies["e3m1"] == skies["MAP25"]
    assert skies["e1m1"] != skies["e2m1"] and skies["e2m1"] != skies["e3m1"] and skies["e1m1"] != skies["e3m1"]


def test_missing_sky_texture_is_an_error_on_both_sides(dg, oracle, synth):
    """`Textures::get("SKY3")` panics when the WAD has no SKY3 (textures.rs:158): the plain synthetic WAD only has SKY1."""
    wad = synth.build_synth_iwad(1993, map_name="e3m1")
    with pytest.raises(oracle.OracleError):
        oracle.Scene(wad, "e3m1")
    with pytest.raises(dg.DoomGpuError) as e:
        dg.Scene(wad, "e3m1")
    assert e.value.code == dg.DG_ERR_WAD


def test_lookup_rules_through_the_c_abi(dg, synth):
    sc = dg.Scene(synth.build_synth_iwad(1993, quirks=True), "E1M1")
    L = dg.lib()
    assert L.dg_scene_texture_id(sc._h, b"LONGNAME") >= 0
    assert L.dg_scene_texture_id(sc._h, b"longname") == L.dg_scene_texture_id(sc._h, b"LONGNAME")
    import ctypes
    w, h = ctypes.c_int(), ctypes.c_int()
    # BRICK2 is 64x128 in TEXTURE1 and 128x128 in TEXTURE2: the later definition wins
    assert L.dg_scene_bitmap_size(sc._h, L.dg_scene_texture_id(sc._h, b"BRICK2"), w, h) == 0 and (w.value, h.value) == (128, 128)
    sc.close()
