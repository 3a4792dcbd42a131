"""The default deck over a long run, pushes/s per 20-step window: no warm-up ramp (window 1 is already the fastest), and a
physical decline -- the two-stream instability heats the beams, more particles change cell per step: 52 G pushes/s at steps
20-60, 45 G from step 160 on.  bench.py times steps 5-24."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
V = importlib.import_module("old-vpic_amd")
args = bench.argparse.Namespace(config=2, grid=None, ppc=0, deck="two-stream", vth=None, sort_interval=10, push="exact", accumulation="float", topology=None)
d = bench.deck(args, 1)
dt = np.float32(d["dt"]) if "dt" in d else np.float32(0.95 / np.sqrt(3.0))
n = d["gx"]
e = V.Engine(V.make_grid(d["gx"], d["gy"], d["gz"], float(d["gx"]), float(d["gy"]), float(d["gz"]), dt))
e.set_vacuum(); e.set_sort_order("engine")
sps = []
for k, u in enumerate(d["species"]):
    sp = e.new_species(-1.0, int(d["gx"] * d["gy"] * d["gz"] * d["ppc"] * 1.02), 4096)
    e.load_maxwellian(sp, d["ppc"], 1 + k, d["q"], u, d["vth"]); sps.append(sp)
e.load_interpolator()
total = sum(e.np(sp) for sp in sps)
step = 0
for w in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    e.sync(); t0 = time.perf_counter()
    for _ in range(20):
        e.step(step, 10); step += 1
    e.sync(); t = time.perf_counter() - t0
    print("window %2d (steps %3d-%3d): %.2f G pushes/s" % (w, step - 20, step - 1, total * 20 / t / 1e9), flush=True)
