#!/usr/bin/env python3
"""Build-time check of gemm_nt8.hip's asynchronous ticket draw (k_gemm8, `draw_async`): the returning atomic is issued from inline asm ahead of the last two
K-tiles of a tile and its destination register is read -- again from inline asm -- behind the last counted vmcnt wait.  hipcc does not know that the register
is written behind its back, so nothing may copy, spill or reuse it in between.  This script reads the device ISA (hipcc -S --cuda-device-only) and, for every
k_gemm8 instantiation, follows the straight-line and loop code between the two statements:

  * exactly one `v_mov_b32 vN, -1` + `global_atomic_add vN, ... sc0` pair inside an ASM block, and one `v_readfirstlane_b32 sX, vM ; DRAWN` inside an ASM block;
  * N == M;
  * no instruction between them (in layout order, which covers the K loop's body) names vN as a destination or source, and no range v[a:b] covers it.

k_attn_fwd3 (attention.hip) draws its slabs the same way, with the register live around its slab loop: see check_carried.

usage: check_async_regs.py gemm_nt8.s attention.s      exit status 1 on a violation"""
import re
import sys


def regs_of(tok):
    """registers named by an operand token: v12 -> {12}; v[8:11] -> {8..11}"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def check(name, lines):
    draw = read = None
    in_asm = False
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_atomic_add") and "sc0" in t:
            m = re.match(r"global_atomic_add v(\d+), v(\d+),", t)
            if m and i >= 9 and any(re.match(rf"v_mov_b32 v{m.group(1)}, -1$", x.strip()) for x in lines[i - 3:i]):
                if draw is not None:
                    return f"{name}: more than one asynchronous draw"
                draw = (i, int(m.group(1)))
        elif in_asm and t.startswith("v_readfirstlane_b32") and "DRAWN" in t and draw is not None and read is None:
            m = re.match(r"v_readfirstlane_b32 s\d+, v(\d+)", t)
            if m:
                read = (i, int(m.group(1)))
    if draw is None and read is None:
        return None                                         # a static-list instantiation: no draw in it
    if draw is None or read is None:
        return f"{name}: draw / read statements not found"
    if draw[1] != read[1]:
        return f"{name}: the draw lands in v{draw[1]} but v{read[1]} is read: a copy was made while the atomic was in flight"
    reg = draw[1]
    for i in range(draw[0] + 1, read[0]):
        t = lines[i].split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        if reg in regs_of(t):
            return f"{name}: v{reg} is touched between the draw and its use: `{t}` (line {i})"
    # the loop: the draw statement is executed once per K-tile pair (empty EXEC mask but for the last); its own v_mov is the only other writer
    return None


def check_carried(name, lines):
    """k_attn_fwd3: the draw is issued in one iteration of the slab loop and read at the top of the next (behind the slab's vmcnt(0)): the register is live
    around the loop, so NO other instruction of the function may name it, except moves of the constant 0 into it (its initial value)."""
    draw = read = None
    in_asm = False
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and t.startswith("global_atomic_add") and "sc0" in t:
            m = re.match(r"global_atomic_add v(\d+), v(\d+), v(\d+),", t)
            if m and any(re.match(rf"v_mov_b32 v{m.group(1)}, -1$", x.strip()) for x in lines[max(0, i - 3):i]):
                if draw is not None:
                    return f"{name}: more than one asynchronous draw"
                draw = (i, int(m.group(1)))
        elif in_asm and t.startswith("v_readfirstlane_b32") and "DRAWN" in t:
            m = re.match(r"v_readfirstlane_b32 s\d+, v(\d+)", t)
            if m:
                if read is not None:
                    return f"{name}: more than one read of the drawn ticket"
                read = (i, int(m.group(1)))
    if draw is None or read is None:
        return f"{name}: draw / read statements not found"
    if draw[1] != read[1]:
        return f"{name}: the draw lands in v{draw[1]} but v{read[1]} is read: the value is copied between iterations"
    reg = draw[1]
    in_asm = False
    asm_ok = 0
    # the slab loop: from its header (the last depth-1 loop header in front of the read) on; in front of it the register only receives its initial value
    heads = [i for i, l in enumerate(lines) if "This Loop Header: Depth=1" in l and i < min(draw[0], read[0])]
    if not heads:
        return f"{name}: slab loop header not found"
    for i, l in enumerate(lines):
        if i < heads[-1]:
            continue
        t = l.split(";")[0].strip()
        if l.strip().startswith(";;#ASMSTART"):
            in_asm = True
            asm_ok = 1 if any(("DRAWN" in x or "global_atomic_add" in x) for x in lines[i:i + 14]) else 0
            continue
        if l.strip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.endswith(":") or t.startswith("."):
            continue
        if reg in regs_of(t):
            if in_asm and asm_ok:
                continue                                    # the draw / read statements themselves
            if re.match(rf"v_mov_b32(_e32)? v{reg}, 0$", t):
                continue                                    # the initial value
            return f"{name}: v{reg} (the drawn ticket, live around the slab loop) is touched by `{t}` (line {i})"
    return None


def main(paths):
    bad = n = 0
    for path in paths:
        s = open(path).read()
        for m in re.finditer(r"^(_ZN2g87k_gemm8\w+|_Z11k_attn_fwd3ILi\d+E\w+):[^\n]*\n", s, re.M):
            name = m.group(1)
            end = s.index(".Lfunc_end", m.end())
            body = s[m.end():end].split("\n")
            err = check_carried(name, body) if name.startswith("_Z11k_attn_fwd3") else check(name, body)
            n += 1
            if err:
                print("ASYNC-REG:", err)
                bad += 1
    if n == 0:
        print("ASYNC-REG: no kernel with an asynchronous draw found in", paths)
        return 1
    print(f"check_async_regs: {n} kernels with an asynchronous ticket draw, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
