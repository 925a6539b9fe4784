#!/usr/bin/env python3
"""EXPERIMENT TOOLING (tools/isa_edit_build.sh with EDIT=pad): one wait state (s_nop 0) between two adjacent vector instructions of
which the second reads a register the first writes -- the wave steps aside for a cycle instead of holding the pipe."""
import re
import sys

reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in reg.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


src = open(sys.argv[1]).read().split("\n")
out = []
prev_dst = None
added = 0
for ln in src:
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith("."):
        out.append(ln)
        continue
    if s.endswith(":") or not re.match(r"[a-z_0-9]+(\s|$)", s):
        prev_dst = None
        out.append(ln)
        continue
    op = s.split()[0]
    if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        body = s[len(op):].split(";")[0]
        ops = [o.strip() for o in body.split(",")]
        dst = regs(ops[0]) if ops else set()
        srcs = regs(",".join(ops[1:]))
        if "UNUSED_PRESERVE" in s:
            srcs |= dst
        if prev_dst and (srcs & prev_dst):
            out.append("\ts_nop 0")
            added += 1
        prev_dst = dst
    else:
        prev_dst = None
    out.append(ln)
open(sys.argv[2], "w").write("\n".join(out))
print("padded", added, file=sys.stderr)
