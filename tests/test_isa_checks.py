"""Static checks on the gfx950 ISA hipcc produces for the hand-scheduled parts of kernels.hip (no GPU needed: hipcc
cross-compiles here).

A kernel that issues loads with inline assembly and waits for them with an explicit `s_waitcnt vmcnt(N)` (so that a texel can
be in flight under other work) hides from the compiler that those registers are written asynchronously: were it to copy, spill
or overwrite one between the load and the wait, the kernel would read or lose a value that has not landed.  `check` runs a forward
data-flow analysis over a kernel's control-flow graph and reports any instruction that reads or overwrites a VGPR whose load may
still be in flight; it is applied to every raster kernel in kernels.hip (today dg_raster_tiles waits inside the same asm block
as its loads, so the check is a guard for future hand-scheduled variants), and its own unit tests below pin the model.

Model (MI355X_MICROARCH.md, `s_waitcnt vmcnt`, and what hipcc itself assumes on gfx9): vmcnt counts vector loads and stores
together; loads return in issue order among themselves, a store may complete before an older load.  Hence after
`s_waitcnt vmcnt(N)` a load is known to have landed iff at least N vector LOADS were issued after it.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "doom-rust-renderer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
def _makefile_flags():
    """The product's own compile flags (csrc/Makefile: FLAGS), minus what only matters for linking a shared library."""
    txt = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "doom-rust-renderer_amd", "csrc", "Makefile")).read()
    line = re.search(r"^FLAGS\s*=\s*(.*)$", txt, re.M).group(1)
    return [f.replace("$(ARCH)", "gfx950") for f in line.split() if f not in ("-fPIC",) and not f.startswith("-W")]


FLAGS = _makefile_flags()

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(operand: str):
    out = set()
    for m in REG.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernel_body(asm: str, mangled: str):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled + ":"))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


def parse(body):
    """-> list of basic blocks: {label, insts: [(mnemonic, [operands])], succ: [block index]}"""
    blocks, cur = [], {"label": None, "insts": []}
    for raw in body:
        l = raw.split(";")[0].strip()
        if not l or l.startswith(".") and not l.endswith(":"):
            continue
        if l.endswith(":"):
            if cur["insts"] or cur["label"]:
                blocks.append(cur)
            cur = {"label": l[:-1], "insts": []}
            continue
        mnem, _, rest = l.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        cur["insts"].append((mnem, ops))
        if mnem.startswith("s_cbranch") or mnem in ("s_branch", "s_endpgm"):
            blocks.append(cur)
            cur = {"label": None, "insts": []}
    if cur["insts"] or cur["label"]:
        blocks.append(cur)
    by_label = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    for i, b in enumerate(blocks):
        succ = []
        last = b["insts"][-1] if b["insts"] else None
        if last and last[0] == "s_endpgm":
            pass
        elif last and last[0] == "s_branch":
            succ.append(by_label[last[1][0]])
        else:
            if last and last[0].startswith("s_cbranch"):
                succ.append(by_label[last[1][0]])
            if i + 1 < len(blocks):
                succ.append(i + 1)
        b["succ"] = succ
    return blocks


def is_vload(m):
    return (m.startswith("global_load") or m.startswith("buffer_load") or m.startswith("flat_load")) and "lds" not in m


def is_vstore(m):
    return m.startswith("global_store") or m.startswith("buffer_store") or m.startswith("flat_store") or m.startswith("global_atomic")


def check(blocks):
    """state: dict reg -> min number of vector loads issued after the load that targets reg (over all paths)."""
    n = len(blocks)
    state_in = [None] * n
    state_in[0] = {}
    work = [0]
    violations = []
    seen = set()
    while work:
        i = work.pop()
        st = dict(state_in[i])
        for (m, ops) in blocks[i]["insts"]:
            if m == "s_waitcnt":
                txt = " ".join(ops)
                mm = re.search(r"vmcnt\((\d+)\)", txt)
                if mm:
                    nmax = int(mm.group(1))
                    st = {r: k for r, k in st.items() if k < nmax}
                continue
            if is_vload(m):
                dst = regs(ops[0])
                src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                for r in (src | dst) & set(st):
                    violations.append((blocks[i]["label"], m, " ".join(ops), f"v{r} may still be in flight"))
                st = {r: k + 1 for r, k in st.items()}
                for r in dst:
                    st[r] = 0
                continue
            if is_vstore(m):
                used = set().union(*[regs(o) for o in ops]) if ops else set()
            elif m.startswith("s_") and not m.startswith("s_waitcnt"):
                used = set().union(*[regs(o) for o in ops]) if ops else set()      # e.g. v_readlane results feed scalars; scalar ops name no VGPR
            else:
                used = set().union(*[regs(o) for o in ops]) if ops else set()      # reads and the destination alike
            for r in used & set(st):
                violations.append((blocks[i]["label"], m, " ".join(ops), f"v{r} may still be in flight"))
        for s in blocks[i]["succ"]:
            old = state_in[s]
            if old is None:
                new = dict(st)
            else:
                new = dict(old)
                for r, k in st.items():
                    new[r] = min(k, new[r]) if r in new else k
            if new != old:
                state_in[s] = new
                work.append(s)
        seen.add(i)
    return violations


@pytest.fixture(scope="module")
def kernels_asm(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    subprocess.check_call([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "kernels.hip")], stderr=subprocess.DEVNULL)
    return out.read_text()


def test_checker_catches_a_premature_read():
    body = """
    global_load_ubyte v5, v1, s[0:1]
    global_load_ubyte v6, v2, s[0:1]
    s_waitcnt vmcnt(1)
    v_mov_b32_e32 v7, v5
    v_mov_b32_e32 v8, v6
    s_endpgm
    """.splitlines()
    v = check(parse(body))
    assert len(v) == 1 and "v6" in v[0][3]


def test_checker_does_not_count_stores_as_younger_loads():
    body = """
    global_load_ubyte v5, v1, s[0:1]
    global_store_dword v2, v3, s[0:1]
    s_waitcnt vmcnt(1)
    v_mov_b32_e32 v7, v5
    s_endpgm
    """.splitlines()
    assert len(check(parse(body))) == 1


@pytest.mark.parametrize("mangled", ["_ZN2dg15dg_raster_tilesENS_12RasterParamsE"])
def test_raster_kernel_never_touches_a_texel_in_flight(kernels_asm, mangled):
    blocks = parse(kernel_body(kernels_asm, mangled))
    assert sum(1 for b in blocks for (m, _) in b["insts"] if m == "global_load_ubyte") >= 2, "the texel loads are gone?"
    v = check(blocks)
    assert not v, "\n".join(f"{lab}: {m} {ops}: {why}" for (lab, m, ops, why) in v[:20])


def test_tile_rasteriser_keeps_four_workgroups_per_cu():
    """dg_raster_tiles is sized for four resident 8-wave workgroups per CU: at most 64 VGPRs (8 waves per SIMD),
    no scratch (a spill makes every wave set up scratch), at most 40 KB of LDS (4 x 40 KB = the CU's 160 KB).  Losing one resident
    workgroup costs 17 % (profiles/r02_raster_tiles.md, occupancy experiment; a 72-VGPR build measured 0.667 against 0.58 ms)."""
    asm = subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, "kernels.hip")],
                         capture_output=True, text=True, check=True).stdout
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        name, body = m.group(1), m.group(2)
        if "dg_raster_tiles" not in name:
            continue
        seen += 1
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        assert vgpr <= 64 and scratch == 0 and lds <= 40960, (name, vgpr, scratch, lds)
    assert seen == 2          # dg_raster_tiles and dg_raster_tiles_anyw (widths that are not multiples of 4)


def test_column_walk_and_seg_walk_keep_their_workgroups_per_cu():
    """dg_fe_columns is sized for eight 4-wave workgroups per CU: at most 20 KB of LDS per workgroup (the wall records of a column and the
    staged record heads live there: profiles/r04_column_walk.md — with 34 KB the launch of 320x200 batches spread over twice the time) and
    no scratch; dg_fs_frame for four 256-thread workgroups per CU (all 1 000 frames of a batch resident at once): at most 40 KB (its scratch —
    the FePart / FeSprite records it assembles for the few survivors — is bounded, not zero); dg_fs_segs, one lane per (frame, seg), keeps the
    calls of a seg in registers (fs_core.h: fixed call slots)."""
    want = {"dg_fe_columns": ("fe_kernels.hip", 20 * 1024, 0), "dg_fs_frame": ("fs_kernels.hip", 40 * 1024 + 512, 160), "dg_fs_segs": ("fs_kernels.hip", 0, 16)}
    for kernel, (src, lds_max, scratch_max) in want.items():
        asm = subprocess.run([HIPCC, *FLAGS, "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, src)], capture_output=True, text=True, check=True).stdout
        seen = 0
        for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
            name, body = m.group(1), m.group(2)
            if kernel not in name:
                continue
            seen += 1
            lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
            scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
            vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
            assert lds <= lds_max and scratch <= scratch_max and vgpr <= 128, (name, lds, scratch, vgpr)
        assert seen == 1, kernel
