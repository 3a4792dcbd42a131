#!/usr/bin/env python3
"""Two slab domains of the two-stream deck in ONE process on ONE GPU (a thread each), the RCCL transport's stream
choreography with device-to-device copies doing the moving (old-vpic_amd/domain.py, LoopbackTransport): what a trace shows
to run concurrently.  Under the profiler:

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/overlap -- python3 tools/overlap_trace.py
    python3 tools/overlap_trace.py --analyse gpurun_out/overlap

The analysis lists, per domain (= per engine stream), how much of the time its messages were in flight (copies on the
communication stream) fell inside one of its own advance_p launches -- the interior push behind which the exchange hides.
RCCL refuses two ranks on one device, so this is a rehearsal of the ordering, not of xGMI."""
import argparse
import csv
import glob
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(steps, grid):
    import torch
    domain = importlib.import_module("old-vpic_amd.domain")
    gx, gy, gz = grid
    ppc = 64
    dt = np.float32(0.95 / np.sqrt(3.0))
    deck = dict(gx=gx, gy=gy, gz=gz, ppc=ppc, dt=dt, q=-float((0.2 / float(dt)) ** 2 / (2 * ppc)), drift=0.2, vth=0.02,
                sort_interval=10, topology=(2, 1, 1), species=[(0.2, 0.0, 0.0), (-0.2, 0.0, 0.0)])
    tr = domain.LoopbackTransport()
    doms, errs = [None, None], []

    def work(rank):
        try:
            torch.cuda.set_device(0)
            d = domain.SlabDomain(deck, rank, 2, 0, loopback=tr)
            doms[rank] = d
            d.trace_reset(True)
            for step in range(steps):
                d.step(step)
            d.engine.sync()
            torch.cuda.synchronize()
        except Exception as exc:                             # noqa: BLE001
            errs.append((rank, repr(exc)))
            raise
    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise SystemExit(f"a domain failed: {errs}")
    for r, d in enumerate(doms):
        print(f"domain {r}: np {[d.engine.np(sp) for sp in d.species]}  exchange {d.trace_report()}")


def analyse(folder):
    kfile = glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True)
    cfile = glob.glob(os.path.join(folder, "**", "*memory_copy_trace.csv"), recursive=True)
    if not kfile:
        raise SystemExit("no kernel trace under " + folder)
    K = list(csv.DictReader(open(kfile[0])))
    C = list(csv.DictReader(open(cfile[0]))) if cfile else []
    name = lambda r: r.get("Kernel_Name", r.get("Name", ""))
    t0 = min(int(r["Start_Timestamp"]) for r in K)
    push = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in K if "advance_p_kernel" in name(r)]
    other = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, name(r)[:40], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in K
             if any(k in name(r) for k in ("boundary_classify", "boundary_inject", "exchange_", "plane_kernel", "pack", "unpack"))]
    # the messages: device-to-device copies on the communication streams -- blit kernels (__amd_rocclr_copyBuffer) on queues
    # that never run advance_p, or entries of the memory-copy trace when a copy engine does them
    push_queues = {q for _, _, q in push}
    copies = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in K
              if "copyBuffer" in name(r) and r.get("Queue_Id", r.get("Stream_Id", "?")) not in push_queues]
    copies += [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in C if "DEVICE_TO_DEVICE" in r.get("Direction", "").upper()]

    def union(spans):
        out = []
        for s, e in sorted((s, e) for s, e, *_ in spans):
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out

    def inside(a, b, spans):
        return sum(max(0, min(b, e) - max(a, s)) for s, e in union(spans))
    print(f"{len(push)} advance_p launches, {sum(e - s for s, e, _ in push) / 1e6:.1f} ms; {len(other)} exchange kernels, {len(copies)} device-to-device copies")
    if copies:
        tot = sum(e - s for s, e in copies)
        ov = sum(inside(s, e, push) for s, e in copies)
        print(f"copies: {tot / 1e6:.3f} ms in flight, {ov / 1e6:.3f} ms of it while an advance_p launch was running ({100.0 * ov / max(tot, 1):.0f} %)")
    tot = sum(e - s for s, e, *_ in other)
    ov = sum(inside(s, e, [p for p in push if p[2] != q]) for s, e, _, q in other)
    print(f"exchange kernels (classify / inject / header / plane): {tot / 1e6:.3f} ms, {ov / 1e6:.3f} ms of it while an advance_p launch of ANOTHER queue was running ({100.0 * ov / max(tot, 1):.0f} %)")
    by = {}
    for s, e, n, q in other:
        by.setdefault(n, [0, 0.0]); by[n][0] += 1; by[n][1] += (e - s) / 1e3
    for n, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"  {n:42s} {c:6d} launches {us / 1e3:9.3f} ms")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--grid", type=int, nargs=3, default=[64, 128, 128])
    ap.add_argument("--analyse", default=None)
    a = ap.parse_args()
    if a.analyse:
        analyse(a.analyse)
    else:
        run(a.steps, a.grid)
