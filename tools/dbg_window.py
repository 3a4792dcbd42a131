import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V = importlib.import_module("old-vpic_amd")
n, ppc = 64, 32
dt = np.float32(0.95 / np.sqrt(3.0))
e = V.Engine(V.make_grid(n, n, n, float(n), float(n), float(n), dt))
e.set_vacuum()
q = -float((0.2 / float(dt)) ** 2 / (2 * ppc))
for k, s in enumerate((1.0, -1.0)):
    sp = e.new_species(-1.0, n ** 3 * ppc, 1024)
    e.load_maxwellian(sp, ppc, 1 + k, q, (s * 0.2, 0.0, 0.0), 0.02)
e.load_interpolator()
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    e.step(step, 0)
WX, WM = 80, 8
sy, sz = n + 2, (n + 2) ** 2
for sp in (0, 1):
    key = e.get_particles(sp)["i"].astype(np.int64)
    ch = key.reshape(-1, 2048)
    kmin, kmax = ch.min(1), ch.max(1)
    s64 = np.sort(ch[:, :64], axis=1)
    med = s64[:, 31]
    cand = np.where(s64 >= (med - WM)[:, None], s64, 1 << 40)
    wb = cand.min(1) - WM
    d = ch - wb[:, None]
    inw = np.zeros_like(d, bool)
    for off in (0, sy, -sy, sz, -sz):
        inw |= (d - off >= 0) & (d - off < WX)
    miss = ~inw
    print("species", sp, "miss fraction", miss.mean(), "chunks with >5% misses", (miss.mean(1) > 0.05).mean())
    dd = d[miss]
    u, cnt = np.unique(dd // 20 * 20, return_counts=True)
    o = np.argsort(-cnt)[:14]
    print("  commonest miss offsets (x20):", [(int(u[k]), int(cnt[k])) for k in o])
    bad = np.argsort(-miss.mean(1))[:3]
    for c in bad:
        dd = d[c][miss[c]]
        print("  chunk", c, "miss", miss[c].mean(), "wbase", wb[c], "first keys", np.sort(ch[c, :64])[[0, 1, 31, 62, 63]], "span", kmin[c], kmax[c], "miss offsets sample", np.unique(dd)[:12])
