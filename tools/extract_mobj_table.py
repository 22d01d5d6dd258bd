#!/usr/bin/env python3
"""Derive the thing-type -> spawn-state table (sprite name, frame, full_bright, is_null)
from the reference's generated data table (src/info.rs: MAP_OBJECT_INFOS :2258-, STATES :1276-).

This is DATA extraction (id Software's doomednum/state table as the reference ships it), run once
in the build container; the output (a compact TSV) is committed so nothing reads /root/reference
at test/bench time.  Usage: python tools/extract_mobj_table.py > <out.tsv>
"""
import re, sys
src = open("/root/reference/src/info.rs").read()
states = {}
order = []
for m in re.finditer(r"State\{id: StateId::(\w+), sprite: SpriteId::(\w+), frame: (\d+), full_bright: (\w+),", src):
    sid, spr, frame, fb = m.groups()
    states[sid] = (spr, int(frame), 1 if fb == "true" else 0)
    order.append(sid)
rows = []
for m in re.finditer(r"MapObjectInfo\{\s*id: (-?\d+),\s*spawn_state: StateId::(\w+),", src):
    num, st = int(m.group(1)), m.group(2)
    if num == -1:
        continue
    spr, frame, fb = states[st]
    rows.append((num, spr, frame, fb, 1 if st == "S_NULL" else 0))
# HashMap insert semantics (map_objects.rs:52-59): a later entry with the same id overwrites.
last = {}
for r in rows:
    last[r[0]] = r
print("# doomednum\tsprite\tframe\tfull_bright\tspawn_is_null   (from reference src/info.rs; see tools/extract_mobj_table.py)")
for num in sorted(last):
    print("\t".join(str(v) for v in last[num]))
