import sys, numpy as np
sys.path.insert(0, "/root/repo")
from rpg_open_remode_amd import api, synth
W, H, F = 640, 480, 200
seq = synth.Sequence(W, H, F)
frames = []
for im in seq.images:
    d = api.DeviceImage(W, H, np.float32); d.setDevData(im); frames.append(d)
for rep in range(2):
    s = api.SeedMatrix(W, H, api.PinholeCamera(*seq.K), patch_side=9)
    s.setReferenceImageDevice(frames[0].data, frames[0].stride, seq.T_curr_world[0], seq.min_depth, seq.max_depth)
    s.sync(); s.setOption(api.OPT_COLLECT_STATS, 2)
    for k in range(1, F):
        s.updateDevice(frames[k].data, frames[k].stride, seq.T_curr_world[k])
    s.sync()
tot = np.zeros(4)
print("wave 0 of every search workgroup: rounds without / with an evaluation outside the LDS window, mean us per round")
for k in range(F - 1):
    t = s.frameTraceDownload(k).astype(np.uint64)
    t = t[(t[:, 0] != 0) & (t[:, 5] > 0)]
    w = t[:, 6]
    t_no, t_fb = (w & np.uint64(0xffffff)).astype(np.float64).sum() / 100, ((w >> np.uint64(24)) & np.uint64(0xffffff)).astype(np.float64).sum() / 100
    n_fb, n_no = ((w >> np.uint64(48)) & np.uint64(0xff)).sum(), ((w >> np.uint64(56)) & np.uint64(0xff)).sum()
    tot += (t_no, t_fb, n_no, n_fb)
    if k + 1 in (4, 25, 35, 40, 60, 90, 100, 125, 160, 190):
        print(f"frame {k+1:3d}: {int(n_no):6d} rounds {t_no / max(n_no, 1):6.2f} us | {int(n_fb):5d} rounds {t_fb / max(n_fb, 1):6.2f} us")
        t0 = t[:, 0].min()
        ex = (t[:, 3] - t0).astype(np.float64) / 100
        order = np.argsort(-ex)[:4]
        ph = t[:, 2]
        pol, stg, rnd = (ph & np.uint64(0xfffff)).astype(np.float64) / 100, ((ph >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64) / 100, ((ph >> np.uint64(40)) & np.uint64(0xffffff)).astype(np.float64) / 100
        rdy = (t[:, 1] - t[:, 0]).astype(np.float64) / 100
        life = (t[:, 3] - t[:, 0]).astype(np.float64) / 100
        print(f"   all busy workgroups, mean us: first unit ready {rdy.mean():.2f}, window policy {pol.mean():.2f}, staging {stg.mean():.2f}, rounds+barrier {rnd.mean():.2f}, lifetime {life.mean():.2f}")
        for i in order:
            print(f"   slow workgroup: end {ex[i]:6.1f} | ready {rdy[i]:5.1f} policy {pol[i]:5.1f} staging {stg[i]:5.1f} rounds+barrier {rnd[i]:5.1f} | items {int(t[i,4])} units {int(t[i,5])} windows {int(t[i,7] >> np.uint64(32))}")
print(f"all frames: {int(tot[2])} rounds {tot[0]/tot[2]:.2f} us | {int(tot[3])} rounds {tot[1]/max(tot[3],1):.2f} us; time in fallback rounds {tot[1]/(tot[0]+tot[1])*100:.1f} % of wave-0 round time")
