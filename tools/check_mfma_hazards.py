#!/usr/bin/env python3
"""Static check of the hazards hipcc cannot see inside the inline-asm MFMAs of prefill_w4_kernel.hip
(cdna_hip_programming.md section 5.7): for every v_mfma in the kernels of the given .s file

  (2) no VALU instruction that WRITES one of the MFMA's A / B / C operand registers may sit within the two
      instructions in front of it (VALU write -> MFMA operand read needs two wait states; an s_nop N in between
      counts N + 1 states);
  (1) no VALU / LDS / VMEM instruction may READ or WRITE the MFMA's destination registers in the instruction
      right behind it unless that instruction is the next MFMA of the same accumulation chain.

usage: tools/check_mfma_hazards.py file.s [kernel-substring]     exit status 1 when a hazard is found."""
import re
import sys


def regs(tok):
    """register set of one operand token: v12, v[4:7], a[0:15], s3 ... -> {('v', 12), ...}"""
    out = set()
    for kind, lo, hi in re.findall(r"\b([vas])\[(\d+):(\d+)\]", tok):
        out |= {(kind, i) for i in range(int(lo), int(hi) + 1)}
    for kind, i in re.findall(r"\b([vas])(\d+)\b", tok):
        out.add((kind, int(i)))
    return out


def parse(line):
    t = line.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def check(path, pat=""):
    bad = 0
    cur, in_k = None, False
    window = []                      # the last instructions: (op, ops, states) -- states = wait states it provides
    prev_mfma = None
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, in_k, window, prev_mfma = m.group(1), pat in m.group(1), [], None
            continue
        if not in_k:
            continue
        if re.match(r"^\.LBB\d+_\d+:", line.strip()):
            window, prev_mfma = [], None      # block boundary: predecessors unknown; hstep never ends a block on a producer
            continue
        ins = parse(line)
        if ins is None:
            continue
        op, ops = ins
        if prev_mfma is not None:
            dst, pl = prev_mfma
            touched = set().union(*[regs(o) for o in ops]) if ops else set()
            same_chain = op.startswith("v_mfma") and regs(ops[0]) == dst and regs(ops[-1]) == dst
            if (touched & dst) and not same_chain and not op.startswith("s_nop"):
                print(f"{path}:{ln}: {cur[:60]}: `{line.strip()}` touches the result of the MFMA at line {pl} right behind it")
                bad += 1
            prev_mfma = None
        if op.startswith("v_mfma"):
            srcs = set().union(*[regs(o) for o in ops[1:]])
            states = 0
            for pop, pops, pst in reversed(window):
                if states >= 2:
                    break
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pops and (regs(pops[0]) & srcs):
                    print(f"{path}:{ln}: {cur[:60]}: `{pop} {', '.join(pops)}` writes an operand of the MFMA {states} wait states ahead of it")
                    bad += 1
                states += pst
            prev_mfma = (regs(ops[0]), ln)
        st = 1
        if op == "s_nop":
            st = int(ops[0]) + 1
        window.append((op, ops, st))
        window = window[-4:]
    return bad


if __name__ == "__main__":
    n = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print(f"{n} hazard(s)")
    sys.exit(1 if n else 0)
