import sys, time, numpy as np
sys.path.insert(0, ".")
from rpg_open_remode_amd import api
w, h = 640, 480
rng = np.random.default_rng(1)
imgs = []
for lo, hi in ((1, 2), (1e-6, 1e-2), (1, 30), (1, 30)):
    im = api.DeviceImage(w, h, np.float32); im.setDevData(rng.uniform(lo, hi, (h, w)).astype(np.float32)); imgs.append(im)
d = api.DepthmapDenoiser(w, h); d.setLargeSigmaSq(1.0)
for timing in (0, 1):
    d.setOption(api.DENOISE_OPT_TIMING, timing)
    for dl in (False, True):
        for it in (200, 20):
            d.denoise(*imgs, 0.5, it, download=dl)
            ts = []
            for _ in range(5):
                t = time.perf_counter(); d.denoise(*imgs, 0.5, it, download=dl); ts.append((time.perf_counter() - t) * 1e3)
            print("timing", timing, "download", dl, "iters", it, "wall ms min %.2f med %.2f" % (min(ts), sorted(ts)[2]))
