#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of a translation unit, from the compiler's own assembly (no GPU needed).

    tools/kernel_resources.py ntsc-crt_amd/csrc/crt_decode.hip [substring ...]

Compiles the file device-only for gfx950 with -save-temps into a temp directory and prints, per kernel whose demangled name
contains every given substring, what the compiler's resource comment block behind the kernel says: `; NumVgprs`, `; NumAgprs`,
`; TotalNumSgprs`, `; ScratchSize`, `; LDSByteSize` and `; Occupancy` (waves per SIMD), plus the waves per CU the LDS alone allows.

(this tool used to print `.amdhsa_next_free_vgpr`, which is NOT the register count -- the compiler raises it to
the first value that enforces the occupancy the kernel's LDS / launch bounds leave anyway: 129 for k_decode_wide<..., 16>, which
uses 74 registers at 3 waves per SIMD.  Both are printed now, the directive as `alloc`.)"""
import os
import re
import subprocess
import sys
import tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1])
want = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="kres_")
base = os.path.splitext(os.path.basename(src))[0]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-fPIC", "-I" + os.path.join(root, "include"),
       "-I" + os.path.join(root, "ntsc-crt_amd", "csrc"), "--cuda-device-only", "-save-temps", "-c", src, "-o", base + ".dev.o"]
subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
asm = open(os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name


# the directive block: what the hardware is told to allocate
alloc = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
    v = re.search(r"\.amdhsa_next_free_vgpr (\S+)", m.group(2))
    alloc[m.group(1)] = int(v.group(1)) if v else 0
# the comment block behind every kernel: what the code uses
rows = []
for m in re.finditer(r"^\s*\.section\s+\.AMDGPU\.csdata.*?\n; Kernel info:\n(.*?)(?=^\s*\.(?:text|section|protected|globl|type|ident)|\Z)", asm, re.S | re.M):
    body = m.group(1)

    def g(key, body=body):
        v = re.search(r"^; %s: (\d+)" % key, body, re.M)
        return int(v.group(1)) if v else -1
    # the kernel this block belongs to: the last label defined before it
    head = asm[:m.start()]
    lab = re.findall(r"^(_Z\w+|k_\w+):\s*(?:;.*)?$", head, re.M)
    name = lab[-1] if lab else "?"
    dn = demangle(name)
    if all(w in dn for w in want):
        lds = g("LDSByteSize")
        rows.append((dn, g("NumVgprs"), g("NumAgprs"), alloc.get(name, 0), g("TotalNumSgprs"), lds, g("ScratchSize"), g("Occupancy"),
                     (160 * 1024 // lds) if lds > 0 else 0))
for r in sorted(rows):
    print("%-108s vgpr %3d agpr %3d (alloc %3d) sgpr %3d lds %6d scratch %4d | occupancy %d waves/SIMD; waves/CU by lds %d"
          % ((r[0][:108],) + r[1:]))
