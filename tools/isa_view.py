#!/usr/bin/env python3
"""Compress a gfx950 kernel's ISA into one letter per instruction, per basic block.
usage: tools/isa_view.py file.hip|file.s <kernel-substring> [extra hipcc flags]
M mfma | x v_exp | v other VALU | a v_accvgpr_* | L v_readlane/v_writelane (SGPR spill traffic) |
r ds_read | w ds_write | g global/buffer load | D LDS-DMA | G store | c s_waitcnt | B s_barrier |
s other SALU | p permlane/dpp | j branch | n s_nop | ! scratch"""
import re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
if src.endswith(".s"):
    asm = open(src).read()
else:
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I/root/repo",
                          "-S", "--cuda-device-only", src, "-o", "-"] + flags, capture_output=True, text=True).stdout
lines = asm.split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\w*:", l) and all(x in l for x in pat.split("+")):
        start = i
        break
assert start is not None, "kernel not found"
out, cur, name = [], [], "entry"
def flush():
    global cur
    if cur:
        s = "".join(cur)
        out.append(f"{name:10s} n={len(cur):4d} " + " ".join(s[i:i+64] for i in range(0, len(s), 64)))
    cur = []
for l in lines[start+1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        cur.append("E"); break
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        flush(); name = m.group(1); continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"): c = "M"
    elif op.startswith("v_exp"): c = "x"
    elif op.startswith("v_accvgpr"): c = "a"
    elif op.startswith("v_readlane") or op.startswith("v_writelane"): c = "L"
    elif "permlane" in op or "dpp" in t: c = "p"
    elif op.startswith("ds_read") or op.startswith("ds_load"): c = "r"
    elif op.startswith("ds_write") or op.startswith("ds_store"): c = "w"
    elif (op.startswith("global_load") or op.startswith("buffer_load")) and t.endswith(" lds"): c = "D"
    elif op.startswith("global_load") or op.startswith("buffer_load"): c = "g"
    elif op.startswith("global_store") or op.startswith("buffer_store"): c = "G"
    elif op.startswith("scratch"): c = "!"
    elif op.startswith("s_waitcnt"): c = "c"
    elif op.startswith("s_barrier"): c = "B"
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): c = "j"
    elif op.startswith("s_nop"): c = "n"
    elif op.startswith("s_"): c = "s"
    elif op.startswith("v_"): c = "v"
    else: c = "?"
    cur.append(c)
flush()
print("\n".join(out))
