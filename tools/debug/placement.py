#!/usr/bin/env python3
"""Does k_decode_wide's time depend on WHERE the pictures lie?  (round 5: 2.57-2.85 ms for the same launch from process to process on
one box.)  One process, one batch of 2048 fields of 1920x1080; the 17 GB picture buffer is re-allocated / re-based / re-strided
between measurements and the decoder's kernel time (HIP events of the library's profile mode, 5 steps) is printed with the buffer's
address:  tools/debug/placement.py [fields]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntsc-crt_amd"))
import torch
import crtlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
w, h = 1920, 1080
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
base = torch.randint(0, 256, (64, h + 1, w, 4), dtype=torch.uint8, device=dev, generator=gen)
images = base.repeat(n // 64, 1, 1, 1)[:, :h]
crt = crtlib.CRT(n, w, h, crtlib.FMT_BGRA, "ntsc", device=0)
crt.scanlines = 1
crt.reserve(n)
s = crtlib.Settings(images, format=crtlib.FMT_BGRA, as_color=1, hue=0, field=[k & 1 for k in range(n)], frame=0)
p = crt.params(s, 0)
crt._load_field_state(s)
pic = h * w * 4


def measure(tag):
    for _ in range(2):
        crt.fieldpass(s, 0, params=p)
    crt.synchronize()
    crt.profile(True)
    for _ in range(5):
        crt.fieldpass(s, 0, params=p)
    prof = crt.profile_read()
    crt.profile(False)
    print("%-46s out @ 0x%012x stride %9d  decode %.3f  active %.3f ms" % (tag, crt.out.data_ptr(), crt.out.stride(0), prof["decode"][0] / 5, prof["active"][0] / 5), flush=True)


measure("as allocated by the context")
keep = []
for i in range(6):
    keep.append(torch.empty(((i * 977 + 131) << 20,), dtype=torch.uint8, device=dev))      # shift what the allocator hands out next
    crt.out = torch.zeros((n, h, w, 4), dtype=torch.uint8, device=dev)
    measure("fresh allocation %d (after %d MB of filler)" % (i, (i * 977 + 131)))
big = torch.zeros((n * (pic + (1 << 20)) + (64 << 20),), dtype=torch.uint8, device=dev)
for off in (0, 4096, 65536, 1 << 20, (1 << 21) + 4096, 3 << 20):
    crt.out = big[off:off + n * pic].view(n, h, w, 4)
    measure("one big buffer, offset %d" % off)
for pad in (4096, 65536, 1 << 18, 1 << 20):
    v = big[:n * (pic + pad)].view(n, pic + pad)[:, :pic].view(n, h, w, 4)
    crt.out = v
    measure("picture stride padded by %d" % pad)
