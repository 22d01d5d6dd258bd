#!/usr/bin/env python3
"""Static VALU class count of a kernel in hipcc -S output (gfx950): which vector instructions are of the 'simple' class that a SIMD
issues two of per ~4.25 clocks (measured: tools/microbench/issue_rates.hip, profiles/r05_issue_model.md) and which are 'complex'
(one per ~4.25 clocks).  Simple = v_fma/fmac/mul/add/sub_f32, v_mov_b32, v_add/sub/subrev_u32, v_and/or/xor_b32, v_ashrrev/lshrrev
with VGPR / inline-constant / literal operands only; anything else — and any of those with an SGPR, VCC or EXEC operand, SDWA or DPP —
is complex.   usage: valu_classes.py file.s kernel_symbol [--blocks]"""
import collections, re, sys
SIMPLE = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
          "v_and_b32", "v_or_b32", "v_xor_b32", "v_ashrrev_i32", "v_lshrrev_b32"}
def classify(line):
    toks = line.split(None, 1)
    op = re.sub(r"_e32$|_e64$", "", toks[0])
    if not op.startswith("v_"): return None, op
    if op.endswith("_sdwa") or op.endswith("_dpp"): return "complex", op
    args = toks[1] if len(toks) > 1 else ""
    args = args.split(";")[0]
    if op in SIMPLE and not re.search(r"\bs\d+\b|\bs\[|\bvcc|\bexec|\bm0\b|src_", args): return "simple", op
    return "complex", op
def main():
    fn, sym = sys.argv[1], sys.argv[2]
    lines = open(fn).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    tot = collections.Counter(); byop = collections.Counter(); other = collections.Counter()
    blocks = []; cur = ["entry", collections.Counter()]
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t): blocks.append(cur); cur = [t.split(":")[0], collections.Counter()]
            continue
        cls, op = classify(t)
        if cls: tot[cls] += 1; byop[(cls, op)] += 1; cur[1][cls] += 1
        else:
            k = "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"
            other[k] += 1; cur[1][k] += 1
    blocks.append(cur)
    print(f"{sym}: VALU simple {tot['simple']}  complex {tot['complex']}  | " + "  ".join(f"{k} {v}" for k, v in sorted(other.items())))
    for cls in ("complex", "simple"):
        print(f"  {cls}: " + ", ".join(f"{op} {n}" for (c, op), n in byop.most_common() if c == cls))
    if "--blocks" in sys.argv:
        for name, c in blocks:
            if sum(c.values()) >= 12: print(f"    {name:14s} " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())))
main()
