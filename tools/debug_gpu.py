import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "ntsc-crt_amd")
import numpy as np, torch
import crtref as R, crtlib
n, w, h = 3, 640, 480
imgs = np.stack([R.synth_image(w, h, 4, 777 + 13 * k, "random" if k % 2 == 0 else "bars") for k in range(n)])
pad = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if pad:
    full = torch.zeros((n, h + 1, w, 4), dtype=torch.uint8, device="cuda:0")
    full[:, :h] = torch.from_numpy(imgs).cuda()
    d = full[:, :h]
else:
    d = torch.from_numpy(imgs).cuda()
g = crtlib.CRT(n, 640, 480, crtlib.FMT_BGRA, "ntsc")
fields = [0, 1, 0]; frames = [0, 0, 1]
s = crtlib.Settings(d, format=crtlib.FMT_BGRA, field=fields, frame=frames)
print("stride", g._image_stride(s), d.data_ptr())
g.modulate(s); g.synchronize()
an = g.analog.cpu().numpy()
orc = R.Oracle("ntsc")
for k in range(n):
    c = orc.new_crt(640, 480, R.FMT_BGRA)
    c.settings(imgs[k], format=R.FMT_BGRA, w=w, h=h, as_color=1, field=fields[k], frame=frames[k])
    c.modulate()
    a = an[k, :orc.input_size].reshape(262, 910); b = c.analog.reshape(262, 910)
    bad = a != b
    bl = np.nonzero(bad.any(1))[0]; bc = np.nonzero(bad.any(0))[0]
    print("field", k, "bad", bad.sum(), "bad lines", bl[:5], bl[-5:], "bad cols", bc[:5], bc[-5:])
    if len(bl):
        ln = bl[0]
        print(" line", ln, "gpu", a[ln, 150:180], "\n         orc", b[ln, 150:180])
