#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel of a translation unit, from the compiler's own assembly (no GPU needed).

    tools/kernel_resources.py ntsc-crt_amd/csrc/crt_decode.hip [substring ...]

Compiles the file device-only for gfx950 with -save-temps into a temp directory and prints, per kernel whose demangled
name contains every given substring: VGPRs, AGPR offset, SGPRs, static LDS, scratch, and the waves per SIMD those allow
(512 VGPRs per SIMD lane, allocation granule 8; 160 KB of LDS per CU = 4 SIMDs)."""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1])
want = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="kres_")
base = os.path.splitext(os.path.basename(src))[0]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-fPIC", "-I" + os.path.join(root, "include"),
       "-I" + os.path.join(root, "ntsc-crt_amd", "csrc"), "--cuda-device-only", "-save-temps", "-c", src, "-o", base + ".dev.o"]
subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
asm = open(os.path.join(tmp, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
filt = "c++filt"
rows = []
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.amdhsa_%s (\S+)" % k, body) or [None, "0"])[1]
    dn = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
    if all(w in dn for w in want):
        vg, lds = int(g("next_free_vgpr")), int(g("group_segment_fixed_size"))
        scratch = int(g("private_segment_fixed_size"))
        w_v = min(8, 512 // max(8, (vg + 7) // 8 * 8))
        w_l = (160 * 1024 // lds) // 4 if lds else 8            # workgroups of one wave: waves per CU / 4 SIMDs (rounded down)
        rows.append((dn, vg, int(g("accum_offset")), int(g("next_free_sgpr")), lds, scratch, w_v, (160 * 1024 // lds) if lds else 0))
for r in sorted(rows):
    print("%-110s vgpr %3d (accum %3d) sgpr %3d lds %6d scratch %4d | waves/SIMD by vgpr %d, waves/CU by lds %d" % ((r[0][:110],) + r[1:]))
