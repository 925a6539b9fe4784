import sys, re
# drop the single wait state the compiler puts behind an inline-asm block on the assumption that the block writes a partial
# register (dst_sel forwarding): only behind blocks that hold no SDWA / op_sel destination
src = open(sys.argv[1]).read().split("\n")
out = []; removed = kept = 0
i = 0
last_asm_partial = None
inasm = False; body = []
for ln in src:
    s = ln.strip()
    if s.startswith(";;#ASMSTART"):
        inasm = True; body = []
    elif s.startswith(";;#ASMEND"):
        inasm = False
        last_asm_partial = any(("dst_sel:WORD" in b or "dst_sel:BYTE" in b or "op_sel" in b) for b in body)
        out.append(ln); continue
    elif inasm:
        body.append(s)
    else:
        if s == "s_nop 0" and last_asm_partial is False:
            removed += 1; last_asm_partial = None; continue
        if s == "s_nop 0": kept += 1
        if s and not s.startswith(";"): last_asm_partial = None
    out.append(ln)
open(sys.argv[2], "w").write("\n".join(out))
print("removed", removed, "kept", kept, file=sys.stderr)
