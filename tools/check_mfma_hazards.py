#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc -S --cuda-device-only) for the hazard class behind the K1e
determinism bug: a non-MFMA instruction that reads or writes a destination register of a v_mfma
fewer than WAIT wait states after it.  On gfx950 these wait states are software's job; the compiler inserts them for its
own instructions but cannot see the operands of inline asm -- this checker reads the final listing, asm bodies included.

Model (conservative): every instruction is one wait state, `s_nop N` is N + 1, an intervening v_mfma also counts as one
(it really occupies the pipe for its passes, so the true distance is larger).  Paths: fall-through and taken branches are
both followed until WAIT wait states have passed.  WAIT = 12 for the 8-pass fp4 form of v_mfma_scale_f32_32x32x64_f8f6f4:
the compiler emits `s_nop 11` (= 12 wait states) between such an MFMA and an immediately following read of its result.

usage: check_mfma_hazards.py file.s [wait]      exit code 1 and a report if anything is found
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    ins, labels = [], {}
    for raw in open(path, errors="replace"):
        line = raw.rstrip("\n")
        m = re.match(r"^(\.LBB\d+_\d+|[A-Za-z_][\w$.]*):", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not line.startswith("\t") or line.startswith("\t.") or line.lstrip().startswith(";"):
            continue
        body = line.strip().split(";")[0].strip()
        if body:
            ins.append(body)
    return ins, labels


def check(path, wait=12):
    ins, labels = parse(path)
    findings = []
    for i, text in enumerate(ins):
        if not text.startswith("v_mfma"):
            continue
        dst = regs(text.split(",")[0])
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < wait:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                t = ins[j]
                op = t.split()[0]
                if op == "s_endpgm":
                    break
                if op == "s_nop":
                    ws += int(t.split()[1]) + 1
                    j += 1
                    continue
                if op.startswith("s_cbranch") or op == "s_branch":
                    tgt = t.split()[-1]
                    if tgt in labels:
                        stack.append((labels[tgt], ws + 1))
                    if op == "s_branch":
                        break
                elif not op.startswith("v_mfma") and not op.startswith("s_") and regs(t) & dst:
                    findings.append((i, text, j, t, ws))
                    break
                ws += 1
                j += 1
    return findings


if __name__ == "__main__":
    f = check(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
    for i, a, j, b, ws in f[:40]:
        print(f"[{i}] {a[:70]}\n    -> [{j}] {b[:90]}   after {ws} wait state(s)")
    print(f"{len(f)} finding(s)")
    sys.exit(1 if f else 0)
