#!/usr/bin/env python3
"""bench.py -- the hot path (advance_p + field solve + glue, i.e. one vpic_simulation::advance)
on synthetic two-stream decks, N GPUs of one node, one process per GPU.

    python bench.py                      # N=1: BASELINE.json configs[1]: 128^3, 2 species, 32 ppc
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `value` = particle pushes per second of the whole job over K
full steps (sort included when due); `roofline` prices the advance_p kernel alone against HBM;
`cpu_baseline` is the reference's own executable under mpiexec (the oracle port when that binary
is not there) on the box's host cores, one 24^3 block of the same deck per core, for about 10 s.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12           # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def b_push(ppc_species):
    """Algorithmic bytes per push: 32 B particle read + 32 B write, plus per occupied
    cell 72 B of interpolator read and 48 B of accumulator write amortised over the cell's particles."""
    return 64.0 + 120.0 / ppc_species


def deck(args, world):
    """Periodic two-stream deck: cvac = eps0 = 1, cubic cells of size 1, dt = 0.95 Courant,
    2 electron species drifting at +-0.2 c with 0.02 c thermal spread."""
    if args.grid:
        gx, gy, gz = args.grid
    elif world == 1:
        gx = gy = gz = 128
    else:
        gx = gy = gz = 256            # configs[2]: slab-decomposed in x over the GPUs
    ppc = args.ppc if args.ppc else (32 if world == 1 else 64)
    assert gx % world == 0, "x cells must divide over the ranks"
    dt = np.float32(0.95 / np.sqrt(3.0))
    wp_dt = 0.2                                       # plasma frequency * dt of both beams together
    q = -float((wp_dt / float(dt)) ** 2 / (2 * ppc))  # wp^2 = n |q| with n = 2*ppc macro-particles per unit volume, q/m = -1
    d = dict(gx=gx, gy=gy, gz=gz, ppc=ppc, dt=dt, q=q, drift=0.2, vth=0.02, sort_interval=args.sort_interval,
             kind=args.deck, species=[(0.2, 0.0, 0.0), (-0.2, 0.0, 0.0)])
    if args.deck == "drift":
        d.update(vth=0.0, species=[(0.1, 0.05, 0.02)], q=-float((wp_dt / float(dt)) ** 2 / ppc))
    if args.vth is not None:
        d.update(vth=args.vth)
    if args.deck == "sheet":
        if not args.grid and world == 1:
            d.update(gx=128, gy=128, gz=64)
        # (q/m, sign of the macro-charge, drift, thermal spread): drifting current-carrying pair + background pair
        d.update(species4=[(-1.0, -1, (0.0, 0.05, 0.0), 0.1), (0.04, 1, (0.0, -0.002, 0.0), 0.02),
                           (-1.0, -1, (0.0, 0.0, 0.0), 0.1), (0.04, 1, (0.0, 0.0, 0.0), 0.02)],
                 species=[None] * 4, q=-float((wp_dt / float(dt)) ** 2 / (2 * ppc)))
    return d


def cpu_baseline_reference(d, cores, seconds):
    """THE REFERENCE's own scalar code (oracle/_ref/twostream.exe, built from its sources where the reference
    tree is, see oracle/Makefile; the binary travels with the repo) under mpiexec, one 24^3 block of the
    bench deck per rank and core: the way the reference scales.  None when the executable, the launcher or
    the deck parameters it was compiled for are not there."""
    import re
    import subprocess
    import tempfile
    exe, mpiexec = os.path.join(ROOT, "oracle", "_ref", "twostream.exe"), "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)) or d["ppc"] != 32 or d["kind"] != "two-stream" or d["vth"] != 0.02:
        return None
    per_rank_step = 2 * 24 ** 3 * 32

    def run(ranks, steps):
        with tempfile.TemporaryDirectory() as tmp:
            out = subprocess.run([mpiexec, "-n", str(ranks), exe, "-tpp=1", str(steps)], cwd=tmp, capture_output=True, text=True, timeout=600)
        m = re.search(r"simulation time: ([0-9.eE+-]+)", out.stderr + out.stdout)
        return float(m.group(1)) if m and out.returncode == 0 else None
    try:
        t = run(1, 10)
        if not t:
            return None
        steps1 = max(10, int(0.4 * seconds / (t / 10)))
        one = steps1 * per_rank_step / run(1, steps1)
        t = run(cores, 10)
        steps = max(10, int(seconds / (t / 10)))
        allc = steps * per_rank_step * cores / run(cores, steps)
    except Exception:                                        # noqa: BLE001 -- any launcher trouble: fall back to the port
        return None
    return dict(value=allc, unit="particle-pushes/s", cores=cores, kind="reference", one_core=one,
                sample=f"the reference's scalar build (oracle/_ref/twostream.exe, gcc -O2) under mpiexec -n {cores}: one 24^3 block of the "
                       f"bench deck per rank ({24 * cores}x24x24 periodic two-stream, 2 species x 32 ppc, sort every 10 steps), {steps} full steps; "
                       f"1-rank figure from {steps1} steps of a 24^3 box")


def cpu_baseline(d, seconds=10.0):
    """The oracle (oracle/vpic_oracle.c, scalar) on the box's host cores, parallelised the way the
    reference scales: one independent 24^3 two-stream domain per core (same ppc and physics, full
    steps).  The domains run in Python threads -- the C calls release the GIL -- first one alone
    (the 1-core figure), then one per core of this process's CPU share, each for about `seconds`."""
    import threading
    from oracle import pyorc
    L = importlib.import_module("old-vpic_amd.layout")
    n = 24

    def make(seed):
        rng = np.random.default_rng(seed)
        g = pyorc.make_grid(n, n, n, float(n), float(n), float(n), d["dt"])
        st = dict(g=g, f=np.zeros(g.nv, L.field_t), fi=np.zeros(g.nv, L.interpolator_t), a=np.zeros(g.nv, L.accumulator_t),
                  m=pyorc.vacuum_coefficients(), species=[], steps=0)
        for s in (1, -1):
            npart = n * n * n * d["ppc"]
            p = np.zeros(npart, L.particle_t)
            cell = np.repeat(np.arange(n * n * n), d["ppc"])
            p["i"] = L.voxel(cell % n + 1, (cell // n) % n + 1, cell // (n * n) + 1, n, n, n)
            for c in ("dx", "dy", "dz"):
                p[c] = rng.uniform(-1, 1, npart).astype(np.float32)
            p["ux"] = (s * d["drift"] + d["vth"] * rng.standard_normal(npart)).astype(np.float32)
            p["uy"] = (d["vth"] * rng.standard_normal(npart)).astype(np.float32)
            p["uz"] = (d["vth"] * rng.standard_normal(npart)).astype(np.float32)
            p["q"] = d["q"]
            st["species"].append(dict(p=p, np=npart, q_m=-1.0, pm=np.zeros(npart // 8, L.particle_mover_t),
                                      partition=np.zeros(g.nv + 1, np.int32)))
        pyorc.load_interpolator(st["fi"], st["f"], g)
        pyorc.step(st["f"], st["fi"], st["a"], st["m"], st["species"], g)       # warm-up
        return st

    def work(st, t_end):
        while time.perf_counter() < t_end:
            pyorc.step(st["f"], st["fi"], st["a"], st["m"], st["species"], st["g"], sort=(st["steps"] % d["sort_interval"] == 0))
            st["steps"] += 1

    def timed(domains):
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(st, t0 + seconds)) for st in domains]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        return sum(st["steps"] for st in domains) * 2 * n * n * n * d["ppc"] / dt

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))                    # a one-GPU box's CPU share
    ref = cpu_baseline_reference(d, cores, seconds)
    one = timed([make(5)])
    doms = [make(5 + k) for k in range(cores)]
    allc = timed(doms) if cores > 1 else one
    if ref:                                           # the reference itself is the baseline; the port's figures ride along
        ref.update(port_value=allc, port_one_core=one)
        return ref
    return dict(value=allc, unit="particle-pushes/s", cores=cores, kind="port", one_core=one,
                sample=f"{cores} independent 24^3 periodic two-stream domains (one per core, {sum(st['steps'] for st in doms)} full steps in all), "
                       f"2 species x {d['ppc']} ppc ({2 * n * n * n * d['ppc']} particles each), oracle/vpic_oracle.c -O2 scalar; "
                       f"1-core figure from one such domain run alone")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, nargs=3, default=None, help="global cells (default: BASELINE config)")
    ap.add_argument("--ppc", type=int, default=0, help="particles per cell per species")
    ap.add_argument("--sort-interval", type=int, default=10, help="> 0: every N steps; < 0: adaptive (engine decides from window misses), at the latest every -N steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vth", type=float, default=None, help="two-stream: thermal spread per component in units of c (default 0.02; "
                    "reconnection decks run at 0.25-0.6, i.e. 0.13-0.34 cells per step)")
    ap.add_argument("--deck", default="two-stream", choices=["two-stream", "drift", "sheet"],
                    help="two-stream (configs[1..2]); the cold uniform drift of configs[4] (1 species, u=(0.1,0.05,0.02)); "
                         "sheet: the boundary conditions and species mix of configs[3] (trecon) on one GPU -- 4 species "
                         "(2 electron, 2 ion populations, mi/me = 25), periodic x,y, conducting walls that reflect "
                         "particles in z, 128x128x64 cells")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (one-GPU rehearsal, host-staged)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if os.environ.get("VPIC_HIP_SINGLE_DEVICE"):      # rehearsal: every rank on the one GPU of the box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    V = importlib.import_module("old-vpic_amd")
    d = deck(args, world)

    if world == 1:
        from importlib import import_module
        L = importlib.import_module("old-vpic_amd.layout")
        kw = {}
        if d["kind"] == "sheet":                     # turbulence.cxx:265-269: conducting walls in z that reflect particles
            kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
        g = V.make_grid(d["gx"], d["gy"], d["gz"], float(d["gx"]), float(d["gy"]), float(d["gz"]), d["dt"], **kw)
        e = V.Engine(g, local_rank)
        e.set_vacuum()
        n_sp = d["gx"] * d["gy"] * d["gz"] * d["ppc"]
        if d["kind"] == "sheet":
            for k, (q_m, sgn, u, vth) in enumerate(d["species4"]):
                sp = e.new_species(q_m, n_sp, max(n_sp // 16, 1024))
                e.load_maxwellian(sp, d["ppc"], 1 + k, sgn * abs(d["q"]), u, vth)
        else:
          for k, u in enumerate(d["species"]):
            sp = e.new_species(-1.0, n_sp, max(n_sp // 16, 1024))
            e.load_maxwellian(sp, d["ppc"], 1 + k, d["q"], u, d["vth"])
        e.load_interpolator()
        stepper = lambda n: e.step(n, d["sort_interval"])
        engine = e
    else:
        domain = importlib.import_module("old-vpic_amd.domain")
        dom = domain.SlabDomain(d, rank, world, local_rank)
        stepper = dom.step
        engine = dom.engine
        n_sp = dom.n_per_species

    step = 0
    for _ in range(args.warmup):
        stepper(step)
        step += 1
    engine.profile_enable(True)
    if world > 1:
        dist.barrier()
    engine.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper(step)
        step += 1
    engine.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    push_ms, launches, pushed = engine.profile_read()

    local_np = sum(engine.np(sp) for sp in range(len(d["species"])))
    if world > 1:
        rdev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([float(local_np), push_ms, float(pushed)], dtype=torch.float64, device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_np = c[0].item()
        t2 = torch.tensor([push_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        push_ms_max = float(t2.item())
        pushed_all = c[2].item()
    else:
        total_np, push_ms_max, pushed_all = float(local_np), push_ms, float(pushed)

    if rank == 0:
        bp = b_push(d["ppc"])
        # dominant kernel: advance_p.  achieved = algorithmic bytes per launch / mean launch time
        per_launch_particles = pushed / max(launches, 1)
        per_launch_s = push_ms * 1e-3 / max(launches, 1)
        achieved = bp * per_launch_particles / per_launch_s / 1e9
        kernel_rate = pushed_all / (push_ms_max * 1e-3)
        traffic = None
        try:                                   # HBM bytes per launch from the committed PMC run of this workload
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
            wl = f"{d['gx']}x{d['gy']}x{d['gz']} periodic two-stream, 2 species x {d['ppc']} ppc, dt=0.95 Courant, sort_interval={d['sort_interval']}"
            if t["workload"] == wl and world == 1 and d["kind"] == "two-stream":
                traffic = t["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "particle-pushes/sec (full step: advance_p + sort when due + field solve + glue)",
            "value": total_np * args.steps / elapsed,
            "unit": "particle-pushes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{d['gx']}x{d['gy']}x{d['gz']} " + {"two-stream": "periodic two-stream, 2 species", "drift": "periodic cold uniform drift, 1 species", "sheet": "periodic x,y / conducting reflecting z, 4 species (mi/me=25)"}[d["kind"]] + f" x {d['ppc']} ppc, "
                                   f"dt=0.95 Courant, sort_interval={d['sort_interval']}"
                                   + (f", vth={args.vth}" if args.vth is not None else "")
                                   + (f", x-slabs over {world} GPUs" if world > 1 else ""),
                       "particles": int(total_np), "decomposition": f"{world}x1x1"},
            "advance_p_pushes_per_s": kernel_rate,
            "full_step_ns_per_particle": elapsed / args.steps / total_np * 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bp * per_launch_particles,
                         "kernel": "advance_p_kernel", "bytes_per_push": bp,
                         "avg_launch_ms": per_launch_s * 1e3, "launches": int(launches)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
