#!/usr/bin/env python3
"""Static check of the hazards hipcc cannot see inside the inline-asm MFMAs of the one-wave-per-SIMD prefill
kernels (cdna_hip_programming.md section 5.7).  For every v_mfma in the kernels of the given .s file:

  (1) the result of an MFMA may be read or written by a non-MFMA instruction (VALU, LDS, VMEM, v_accvgpr_*) only
      once the MFMA has drained.  hipcc pads `s_nop 11` behind a v_mfma_f32_32x32x16 (8 passes) and `s_nop 7`
      behind a v_mfma_f32_16x16x32 (4 passes) for its own code, i.e. the consumer may issue passes + 5 wait
      states after the MFMA.  The checker keeps a clock in wait states: every instruction takes one, `s_nop N`
      N + 1, and an MFMA cannot issue before the previous MFMA's passes are over (the matrix pipe is paced) --
      so two independent MFMAs between producer and consumer cover the distance, one does not.  The next MFMA of
      the same accumulation chain (destination = C operand = the result) needs nothing.
      The window follows fall-through labels and forward branches (at a label the state is the worst of the
      fall-through path and of every forward branch seen that targets it); a backward branch target (a loop
      header) starts from the fall-through state.
  (2) no VALU instruction that WRITES one of the MFMA's A / B / C operand registers may sit within the two
      wait states in front of it (an s_nop N in between counts N + 1 states).

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


def mfma_passes(op):
    m = re.match(r"v_mfma_\w+?_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return 8
    mm = int(m.group(1))
    return 8 if mm == 32 else 4


class State:
    """clock in wait states; pending = MFMAs whose result may not be touched yet: (dst regs, issue time, passes, line)"""

    def __init__(self):
        self.dead = False               # behind an unconditional branch: nothing falls through to the next label
        self.clock = 0
        self.pending = []
        self.pipe_free = 0              # earliest issue time of the next MFMA
        self.window = []                # the last few instructions, for hazard (2): (op, ops, states)

    def snapshot(self):
        s = State()
        s.dead = False
        s.clock, s.pipe_free = self.clock, self.pipe_free
        s.pending = list(self.pending)
        s.window = list(self.window)
        return s

    def merge(self, other):
        """worst case of two histories: re-base `other` on this clock; an MFMA pending in either stays pending with the
        smaller elapsed time"""
        shift = self.clock - other.clock
        mine = {ln: (dst, t, p, ln) for dst, t, p, ln in self.pending}
        for dst, t, p, ln in other.pending:
            t2 = t + shift
            if ln not in mine or t2 > mine[ln][1]:
                mine[ln] = (dst, t2, p, ln)
        self.pending = list(mine.values())
        self.pipe_free = max(self.pipe_free, other.pipe_free + shift)
        if len(other.window) < len(self.window):
            self.window = list(other.window)    # the shorter known history is the stricter one for hazard (2)


def check(path, pat=""):
    bad = 0
    cur, in_k = None, False
    st = State()
    fwd = {}                            # label -> [snapshots taken at forward branches to it]
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, in_k = m.group(1), pat in m.group(1)
            st, fwd = State(), {}
            continue
        if not in_k:
            continue
        lab = re.match(r"^(\.LBB\d+_\d+):", line.strip())
        if lab:
            snaps = fwd.pop(lab.group(1), [])
            if st.dead:                 # reached by branches only: start from the first of them (none: a backward target --
                st = snaps.pop(0) if snaps else State()      # a loop header, entered with everything drained)
                st.dead = False
            for snap in snaps:
                st.merge(snap)
            continue
        if st.dead:
            continue
        ins = parse(line)
        if ins is None:
            continue
        op, ops = ins
        is_mfma = op.startswith("v_mfma")
        touched = set().union(*[regs(o) for o in ops]) if ops else set()
        # ---- hazard (1) ----
        if not op.startswith("s_nop") and not op.startswith("s_") or is_mfma:
            issue = max(st.clock, st.pipe_free) if is_mfma else st.clock
            keep = []
            for dst, t, p, pl in st.pending:
                if issue - t >= p + 5:
                    continue                    # drained
                if touched & dst:
                    same_chain = is_mfma and regs(ops[0]) == dst and regs(ops[-1]) == dst
                    if not same_chain:
                        print(f"{path}:{ln}: {cur[:60]}: `{line.strip()}` touches the result of the MFMA at line {pl} "
                              f"{issue - t} wait states behind it (needs {p + 5})")
                        bad += 1
                        continue
                    continue                    # the chain's next link takes over the register
                keep.append((dst, t, p, pl))
            st.pending = keep
        # ---- hazard (2) ----
        if is_mfma:
            srcs = set().union(*[regs(o) for o in ops[1:]])
            states = 0
            for pop, pops, pst in reversed(st.window):
                if states >= 2:
                    break
                if pop.startswith("v_") and not pop.startswith("v_mfma") and pops and (regs(pops[0]) & srcs):
                    print(f"{path}:{ln}: {cur[:60]}: `{pop} {', '.join(pops)}` writes an operand of the MFMA {states} wait states ahead of it")
                    bad += 1
                states += pst
        # ---- advance the clock ----
        if is_mfma:
            issue = max(st.clock, st.pipe_free)
            p = mfma_passes(op)
            st.pending.append((regs(ops[0]), issue, p, ln))
            st.pipe_free = issue + p
            st.clock = issue + 1
            n = 1
        elif op == "s_nop":
            n = int(ops[0]) + 1
            st.clock += n
        else:
            n = 1
            st.clock += 1
        st.window.append((op, ops, n))
        st.window = st.window[-4:]
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = ops[0] if ops else ""
            fwd.setdefault(tgt, []).append(st.snapshot())     # (a backward target has been passed already: never popped)
            if op == "s_branch":
                st = st.snapshot()
                st.dead = True
    return bad


if __name__ == "__main__":
    n = check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print(f"{n} hazard(s)")
    sys.exit(1 if n else 0)
