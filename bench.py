#!/usr/bin/env python3
"""bench.py -- the hot path (advance_p + field solve + glue, i.e. one vpic_simulation::advance)
on synthetic two-stream decks, N GPUs of one node, one process per GPU.

    python bench.py                      # BASELINE.json configs[2]: 256^3 two-stream, 2 species x 64 ppc -- the deck the
                                         # north star's targets are quoted on (>= 0.50 of roofline at 64 ppc, strong scaling
                                         # 1 -> 8 GPUs), on ONE GPU; a short run of configs[1] (128^3, 32 ppc) rides along
    python bench.py --config 1           # configs[1] alone
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # the SAME 256^3 box in N x-slabs

Prints ONE JSON line on rank 0.  `value` = particle pushes per second of the whole job over K
full steps (sort included when due); `roofline` prices the advance_p kernel alone against HBM;
`cpu_baseline` is the reference's own executable under mpiexec (the oracle port when that binary
is not there) on the box's host cores, one 24^3 block of the same deck per core, for about 10 s.
"""
import argparse
import importlib
import json
import os
import subprocess
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
    # the workload does NOT depend on the number of GPUs: N = 1 ... 8 run the same global box (strong scaling)
    if args.grid:
        gx, gy, gz = args.grid
    elif args.config == 1:
        gx = gy = gz = 128            # configs[1]
    else:
        gx = gy = gz = 256            # configs[2]: one domain at N = 1, x-slabs over the GPUs otherwise
    ppc = args.ppc if args.ppc else (32 if args.config == 1 else 64)
    topo = tuple(args.topology) if getattr(args, "topology", None) else (world, 1, 1)
    assert topo[0] * topo[1] * topo[2] == world, "--topology must multiply to the number of ranks"
    assert gx % topo[0] == 0 and gy % topo[1] == 0 and gz % topo[2] == 0, "the cells must divide over the ranks"
    dt = np.float32(0.95 / np.sqrt(3.0))
    wp_dt = 0.2                                       # plasma frequency * dt of both beams together
    q = -float((wp_dt / float(dt)) ** 2 / (2 * ppc))  # wp^2 = n |q| with n = 2*ppc macro-particles per unit volume, q/m = -1
    d = dict(gx=gx, gy=gy, gz=gz, ppc=ppc, dt=dt, q=q, drift=0.2, vth=0.02, sort_interval=args.sort_interval, topology=topo,
             kind=args.deck, species=[(0.2, 0.0, 0.0), (-0.2, 0.0, 0.0)])
    if args.deck == "drift":
        d.update(vth=0.0, species=[(0.1, 0.05, 0.02)], q=-float((wp_dt / float(dt)) ** 2 / ppc))
    if args.vth is not None:
        d.update(vth=args.vth)
    if args.deck == "trecon":
        # BASELINE configs[3] as ONE of its 8 x-slabs: decks/trecon-part scaled to 256 x 256 x 128 -> 32 x 256 x 128 cells per
        # GPU; the pair plasma of turbulence.cxx:95-98 (mi/me = 1, vthe = vthi = 0.6 c) at 64 ppc and its two charge-0 tracer
        # copies (tracer.cxx:64-70: pushed, never deposited); conducting walls that reflect particles in z (:265-269)
        if not args.grid:
            d.update(gx=32 if world == 1 else 256, gy=256, gz=128)      # N > 1: configs[3] whole (256 x 256 x 128) in N x-slabs
        d.update(walls={2: (-1, -1)})                                   # pec_fields, reflect_particles (multi-GPU driver: domain.py)
        qq = abs(float((wp_dt / float(dt)) ** 2 / (2 * ppc)))
        d.update(vth=0.6 if args.vth is None else args.vth, species=[None] * 4, q=-qq,
                 species4=[(-1.0, 0, (0.0, 0.0, 0.0), d["vth"] if args.vth is not None else 0.6), (1.0, 0, (0.0, 0.0, 0.0), d["vth"] if args.vth is not None else 0.6),
                           (1.0, 1, (0.0, 0.0, 0.0), d["vth"] if args.vth is not None else 0.6), (-1.0, -1, (0.0, 0.0, 0.0), d["vth"] if args.vth is not None else 0.6)])
    if args.deck == "sheet":
        if not args.grid and world == 1:
            d.update(gx=128, gy=128, gz=64)
        # (q/m, sign of the macro-charge, drift, thermal spread): drifting current-carrying pair + background pair
        d.update(species4=[(-1.0, -1, (0.0, 0.05, 0.0), 0.1), (0.04, 1, (0.0, -0.002, 0.0), 0.02),
                           (-1.0, -1, (0.0, 0.0, 0.0), 0.1), (0.04, 1, (0.0, 0.0, 0.0), 0.02)],
                 species=[None] * 4, q=-float((wp_dt / float(dt)) ** 2 / (2 * ppc)))
    return d


def cpu_baseline_reference(d, cores, seconds):
    """THE REFERENCE's own code (oracle/_ref/twostream*.exe, built from its sources where the reference tree is,
    see oracle/Makefile; the binaries travel with the repo) under mpiexec, one 24^3 block of the bench deck per
    rank and core: the way the reference scales.  The scalar build is the baseline proper (it is what the oracle
    and every parity test are pinned on); the SSE build (V4=1: the pipelines the reference's shipped machine
    configs select, config/cray-haswell.conf:18) is timed beside it.  None when the executable, the launcher or
    the deck parameters it was compiled for are not there."""
    import re
    import subprocess
    import tempfile
    tag = {32: "", 64: "64"}.get(d["ppc"])
    mpiexec = "/opt/conda/bin/mpiexec"
    if tag is None or d["kind"] != "two-stream" or d["vth"] != 0.02 or not os.path.exists(mpiexec):
        return None
    exe = os.path.join(ROOT, "oracle", "_ref", f"twostream{tag}.exe")
    exe_v4 = os.path.join(ROOT, "oracle", "_ref", f"twostream{tag}_v4.exe")
    if not os.path.exists(exe):
        return None
    per_rank_step = 2 * 24 ** 3 * d["ppc"]

    def run(binary, ranks, steps):
        with tempfile.TemporaryDirectory() as tmp:
            out = subprocess.run([mpiexec, "-n", str(ranks), binary, "-tpp=1", str(steps)], cwd=tmp, capture_output=True, text=True, timeout=600)
        m = re.search(r"simulation time: ([0-9.eE+-]+)", out.stderr + out.stdout)
        return float(m.group(1)) if m and out.returncode == 0 else None

    def rate(binary, ranks, budget):
        t = run(binary, ranks, 6)
        if not t:
            return None, 0
        steps = max(6, int(budget / (t / 6)))
        return steps * per_rank_step * ranks / run(binary, ranks, steps), steps
    try:
        one, steps1 = rate(exe, 1, 0.3 * seconds)
        allc, steps = rate(exe, cores, 0.7 * seconds)
        if not one or not allc:
            return None
        v4 = {}
        if os.path.exists(exe_v4):
            v4_one, _ = rate(exe_v4, 1, 0.15 * seconds)
            v4_all, v4_steps = rate(exe_v4, cores, 0.35 * seconds)
            if v4_one and v4_all:
                v4 = dict(sse_value=v4_all, sse_one_core=v4_one,
                          sse_sample=f"the same deck on the reference's SSE build (-DUSE_V4_SSE -msse2), {v4_steps} steps on {cores} ranks")
    except Exception:                                        # noqa: BLE001 -- any launcher trouble: fall back to the port
        return None
    out = dict(value=allc, unit="particle-pushes/s", cores=cores, kind="reference", one_core=one,
               sample=f"the reference's scalar build (oracle/_ref/twostream{tag}.exe, gcc -O2) under mpiexec -n {cores}: one 24^3 block of the "
                      f"bench deck per rank ({24 * cores}x24x24 periodic two-stream, 2 species x {d['ppc']} ppc, sort every 10 steps), {steps} full steps; "
                      f"1-rank figure from {steps1} steps of a 24^3 box")
    out.update(v4)
    return out


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


def workload_name(d, args, world):
    kinds = {"two-stream": "periodic two-stream, 2 species", "drift": "periodic cold uniform drift, 1 species",
             "sheet": "periodic x,y / conducting reflecting z, 4 species (mi/me=25)",
             "trecon": "one x-slab of configs[3] (trecon-part at 256x256x128 over 8 GPUs): periodic x,y / conducting reflecting z, pair plasma vth=0.6c + its 2 tracer copies, 4 species"}
    return (f"{d['gx']}x{d['gy']}x{d['gz']} {kinds[d['kind']]} x {d['ppc']} ppc, dt=0.95 Courant, sort_interval={d['sort_interval']}"
            + (f", vth={args.vth}" if args.vth is not None else ""))


def run_workload(args, d, world, rank, local_rank, steps, warmup):
    """Load the deck, do `warmup` untimed steps, time exactly `steps` steps between barriers; returns the
    measurements of this rank reduced over the job (elapsed = max over ranks)."""
    import torch
    import torch.distributed as dist
    V = importlib.import_module("old-vpic_amd")
    if world == 1:
        L = importlib.import_module("old-vpic_amd.layout")
        kw = {}
        if d["kind"] in ("sheet", "trecon"):         # turbulence.cxx:265-269: conducting walls in z that reflect particles
            kw = dict(fbc=[0, 0, L.PEC_FIELDS, 0, 0, L.PEC_FIELDS], pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
        g = V.make_grid(d["gx"], d["gy"], d["gz"], float(d["gx"]), float(d["gy"]), float(d["gz"]), d["dt"], **kw)
        e = V.Engine(g, local_rank)
        e.set_vacuum()
        e.set_push_mode(args.push)
        if args.accumulation != "float":
            e.set_accumulation(args.accumulation, abs(d["q"]))
        e.set_sort_order("engine")               # the order of a sorted species is the engine's business (tile order, include/vpic_hip.h)
        n_sp = d["gx"] * d["gy"] * d["gz"] * d["ppc"]
        if d["kind"] in ("sheet", "trecon"):
            for k, (q_m, sgn, u, vth) in enumerate(d["species4"]):
                sp = e.new_species(q_m, n_sp, max(n_sp // 16, 1024))
                e.load_maxwellian(sp, d["ppc"], 1 + k, sgn * abs(d["q"]), u, vth)
        else:
            for k, u in enumerate(d["species"]):
                sp = e.new_species(-1.0, n_sp, max(n_sp // 16, 1024))
                e.load_maxwellian(sp, d["ppc"], 1 + k, d["q"], u, d["vth"])
        e.load_interpolator()
        stepper = lambda n: e.step(n, d["sort_interval"])
        engine, dom = e, None
    else:
        domain = importlib.import_module("old-vpic_amd.domain")
        if args.accumulation != "float":
            d = dict(d, accumulation=args.accumulation)
        dom = domain.SlabDomain(d, rank, world, local_rank, push_mode=args.push)
        stepper = dom.step
        engine = dom.engine

    # A box that has been idle runs its first seconds of kernels ~5 % slower (five bench runs back to back on a fresh box: 17.51 ms
    # per advance_p launch in the first, 16.60-16.67 in the others; with 3 s of warm-up 17.12 against 16.7-16.9, with 8 s level):
    # the device is kept busy for eight seconds with launches that change nothing (energy_p reads every particle, load_interpolator
    # rewrites what is there) BEFORE the W warm-up steps -- the timed region stays steps W .. W + K of the deck; reported as
    # `device_warmup_s`; the ride-along runs come after it and skip it
    t_w = time.perf_counter()
    while args.device_warmup_s > 0 and time.perf_counter() - t_w < args.device_warmup_s:
        for sp in range(len(d["species"])):
            engine.energy_p(sp)
        if dom is None:
            engine.load_interpolator()                     # (what it writes is what is there already)
    step = 0
    for _ in range(warmup):
        stepper(step)
        step += 1
    # the run checks itself, outside the timed region: particle count and total energy (kinetic of every species + field)
    # before and after the timed steps (energy_p.cxx:124-157, energy_f.c:139-179)
    n_species = len(d["species"])

    def census():
        if world > 1:                                      # (particles migrate between ranks: the job's total is what is conserved)
            t = torch.tensor([float(sum(engine.np(sp) for sp in range(n_species)))], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return dict(particles=int(t.item()))
        kin = [engine.energy_p(sp) for sp in range(n_species)]
        return dict(particles=int(sum(engine.np(sp) for sp in range(n_species))), kinetic=float(sum(kin)), field=float(engine.energy_f().sum()))
    check_before = census()
    engine.profile_enable(True)
    if dom is not None:
        dom.trace_reset(True)
    # an event behind every step on the engine's own stream: the median step beside the mean the contract asks for
    es = torch.cuda.ExternalStream(engine.stream(), device=torch.device("cuda", local_rank))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if world > 1:
        dist.barrier()
    engine.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record(es)
    for k in range(steps):
        stepper(step)
        step += 1
        marks[k + 1].record(es)
    engine.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    check_after = census()
    check = dict(before=check_before, after=check_after, particles_conserved=check_before["particles"] == check_after["particles"])
    if hasattr(engine, "species_stats"):
        check["species"] = [engine.species_stats(sp) for sp in range(n_species)]
    if "kinetic" in check_before:
        e0, e1 = check_before["kinetic"] + check_before["field"], check_after["kinetic"] + check_after["field"]
        check.update(total_energy_drift=(e1 - e0) / e0 if e0 else None, steps=steps,
                     note="energies in the deck's units, rank-local; the two-stream instability moves kinetic into field energy, the sum drifts at the 1e-4 level per 20 steps")
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))
    median_ms = per_step[len(per_step) // 2] if steps % 2 else 0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2])
    exchange = dom.trace_report() if dom is not None else None
    push_ms, launches, pushed = engine.profile_read()
    try:                                                   # launches that sorted the species as they pushed it: booked apart
        sort_ms, sort_launches, sort_pushed = engine.profile_read_sorting()
    except AttributeError:                                 # (an older build under VPIC_HIP_LIB)
        sort_ms, sort_launches, sort_pushed = 0.0, 0, 0
    local_np = sum(engine.np(sp) for sp in range(len(d["species"])))
    by_species = None
    if "species4" in d and hasattr(engine, "profile_read_species"):
        # a deck whose species differ (configs[3]: the charged pair and its charge-0 tracer copies, which deposit nothing): the
        # plain launches of every species on their own, priced with the same bytes per push
        by_species = []
        for sp, (q_m, sgn, u, vth) in enumerate(d["species4"]):
            ms_k, n_k, parts_k = engine.profile_read_species(sp)
            if n_k:
                bps = b_push(d["ppc"]) if sgn else 64.0 + 72.0 / d["ppc"]        # (a tracer copy writes no accumulator)
                by_species.append({"species": sp, "q_m": q_m, "charged": bool(sgn), "launches": int(n_k), "avg_launch_ms": ms_k / n_k,
                                   "bytes_per_push": bps, "frac": bps * (parts_k / n_k) / (ms_k * 1e-3 / n_k) / HBM_PEAK})
    if world > 1:
        rdev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, push_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, push_ms_max = float(t[0].item()), float(t[1].item())
        c = torch.tensor([float(local_np), float(pushed)], dtype=torch.float64, device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        total_np, pushed_all = c[0].item(), c[1].item()
    else:
        total_np, push_ms_max, pushed_all = float(local_np), push_ms, float(pushed)
    host_syncs = dom.host_syncs_per_step() if dom is not None else None
    transport = dom.transport if dom is not None else None
    if exchange is not None:                               # the slowest rank's figures beside rank 0's
        keys = ["host_issue_ms_per_step", "host_blocked_ms_per_step", "exchange_ms_per_step", "exchange_exposed_ms_per_step"]
        t = torch.tensor([exchange[k] for k in keys], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange["max_over_ranks"] = {k: float(v) for k, v in zip(keys, t.tolist())}
    engine.close()
    del engine, dom
    torch.cuda.empty_cache()

    bp = b_push(d["ppc"])
    # dominant kernel: advance_p.  achieved = algorithmic bytes per launch / mean launch time (HIP events on the
    # engine's own stream around every launch of the timed region, rank 0's launches)
    per_launch_particles = pushed / max(launches, 1)
    per_launch_s = push_ms * 1e-3 / max(launches, 1)
    achieved = bp * per_launch_particles / per_launch_s / 1e9
    traffic, traffic_source = None, None
    try:                                   # HBM bytes per launch from the committed PMC run of this workload
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        if world == 1 and args.push == "exact":
            traffic = t.get(workload_name(d, args, world), {}).get("hbm_bytes_per_launch")
            if traffic is not None:                # NOT measured in this run: counters need their own rocprofv3 --pmc passes
                traffic_source = "profiles/traffic_latest.json: TCC EA read + write bytes of a separate rocprofv3 --pmc run of this workload (tools/pmc_traffic.sh), not of this run"
    except Exception:
        pass
    sorting = None
    if sort_launches:
        # what a launch that sorts as it pushes moves per particle: the push's 32 B read, all eight arrays (32 B) written to the
        # second buffer, the cells read once more for the places (4 B), the same share of interpolators and accumulators
        bs = 68.0 + 120.0 / d["ppc"]
        s_particles, s_s = sort_pushed / sort_launches, sort_ms * 1e-3 / sort_launches
        sorting = {"kernel": "advance_p_kernel<.., SORT>: the species' sort happens inside this push (vpic_hip_step; the places come from "
                             "the counts the push before it took)", "launches": int(sort_launches), "avg_launch_ms": s_s * 1e3,
                   "bytes_per_push": bs, "achieved": bs * s_particles / s_s / 1e9, "unit": "GB/s", "frac": bs * s_particles / s_s / HBM_PEAK,
                   "note": "not part of roofline.avg_launch_ms / advance_p_pushes_per_s, which cover the plain launches; part of value and ms_per_step"}
    return dict(total_np=total_np, elapsed=elapsed, host_syncs=host_syncs, median_ms=median_ms, exchange=exchange, transport=transport, check=check,
                kernel_rate=pushed_all / (push_ms_max * 1e-3), sorting=sorting, by_species=by_species,
                roofline={"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                          "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                          "algorithmic_bytes_per_launch": bp * per_launch_particles,
                          "kernel": "advance_p_kernel", "bytes_per_push": bp, "ppc": d["ppc"], "push_arithmetic": args.push,
                          "avg_launch_ms": per_launch_s * 1e3, "launches": int(launches),
                          "frac_of_measured_copy_ceiling": achieved * 1e9 / 6.29e12})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2],
                    help="BASELINE.json configs[]: 2 = 256^3 two-stream x 64 ppc (default, at every N: the deck the targets are "
                         "quoted on; one domain at N = 1, x-slabs otherwise), 1 = 128^3 x 32 ppc")
    ap.add_argument("--grid", type=int, nargs=3, default=None, help="global cells (default: BASELINE config)")
    ap.add_argument("--ppc", type=int, default=0, help="particles per cell per species")
    ap.add_argument("--device-warmup-s", type=float, default=8.0, help="seconds of state-preserving launches before the warm-up steps (an idle box starts ~5 %% slow)")
    ap.add_argument("--sort-interval", type=int, default=10, help="> 0: every N steps; < 0: adaptive (engine decides from window misses), at the latest every -N steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-config", action="store_true", help="N = 1, default deck: skip the short configs[1] run that rides along")
    ap.add_argument("--push", default="exact", choices=["exact", "fast"],
                    help="arithmetic of advance_p: exact = the reference's scalar pipeline bit for bit (default); fast = contracted "
                         "multiply-adds and refined v_rsq/v_rcp (what the reference's own V4 pipelines do), momenta within 8 ulp")
    ap.add_argument("--accumulation", default="float", choices=["float", "deterministic"],
                    help="how deposits are summed: float atomics (default) or 64-bit fixed point (bit-identical from run to run; include/vpic_hip.h)")
    ap.add_argument("--vth", type=float, default=None, help="two-stream: thermal spread per component in units of c (default 0.02; "
                    "reconnection decks run at 0.25-0.6, i.e. 0.13-0.34 cells per step)")
    ap.add_argument("--deck", default="two-stream", choices=["two-stream", "drift", "sheet", "trecon"],
                    help="two-stream (configs[1..2]); the cold uniform drift of configs[4] (1 species, u=(0.1,0.05,0.02)); "
                         "sheet: the boundary conditions and species mix of configs[3] (trecon) on one GPU -- 4 species "
                         "(2 electron, 2 ion populations, mi/me = 25), periodic x,y, conducting walls that reflect "
                         "particles in z, 128x128x64 cells")
    ap.add_argument("--topology", type=int, nargs=3, default=None, help="domains along x y z (default: x-slabs, N 1 1)")
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
            kw = {}
            try:                                      # RCCL's own stream above the push kernels in the hardware queues
                from torch.distributed import ProcessGroupNCCL
                opts = ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw["pg_options"] = opts
            except Exception:                         # noqa: BLE001 -- an older torch: the default stream priority
                pass
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(args.backend)

    d = deck(args, world)
    default_deck = args.config == 2 and not args.grid and not args.ppc and args.deck == "two-stream" and args.vth is None
    second = None
    c3 = None
    r = run_workload(args, d, world, rank, local_rank, args.steps, args.warmup)
    other = None
    if world == 1 and default_deck and not args.no_second_config:
        # the same deck with the other arithmetic of advance_p (include/vpic_hip.h: exact is the engine's default and the
        # headline; fast = contracted multiply-adds and 1-ulp rsq / rcp, momenta within 8 ulp), a short run
        a2 = argparse.Namespace(**vars(args))
        a2.push = "fast" if args.push == "exact" else "exact"
        r2 = run_workload(a2, d, 1, rank, local_rank, 10, 5)
        other = {"push_arithmetic": a2.push, "value": r2["total_np"] * 10 / r2["elapsed"], "steps": 10, "warmup": 5,
                 "ms_per_step": r2["elapsed"] / 10 * 1e3, "advance_p_pushes_per_s": r2["kernel_rate"], "roofline": r2["roofline"]}

    si20 = None
    if world == 1 and default_deck and not args.no_second_config and args.sort_interval != 20:
        # SURVEY.md 8d quotes the metric with sort_interval = 20; the headline keeps 10 (what rounds 1-2 reported), this block is
        # the same deck sorted every 20 steps (round 3: the tile windows follow the beams between sorts, and the two intervals
        # come out within 2 % of each other)
        a3 = argparse.Namespace(**vars(args))
        a3.sort_interval = 20
        d3 = deck(a3, 1)
        r3 = run_workload(a3, d3, 1, rank, local_rank, 40, 5)      # (two whole sort cycles, like the headline's 20 steps at interval 10)
        si20 = {"workload": workload_name(d3, a3, 1), "value": r3["total_np"] * 40 / r3["elapsed"], "steps": 40, "warmup": 5,
                "ms_per_step": r3["elapsed"] / 40 * 1e3, "ms_per_step_median": r3["median_ms"],
                "advance_p_pushes_per_s": r3["kernel_rate"], "roofline": r3["roofline"]}
    if world == 1 and default_deck and not args.no_second_config:
        # Two smaller decks ride along, each in a FRESH PROCESS started once this one has released its engines: in one process,
        # whichever deck runs second pays for the first one's freed memory (behind the 137 GB deck the 128^3 deck's kernel took
        # 1.8 x as long; ahead of it, the small decks cost the large one 1.5-2 %: profiles/r03_* of round 3).
        def rider(extra):
            cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-second-config", "--push", args.push,
                   "--accumulation", args.accumulation, "--device-warmup-s", "0"] + extra      # (the device is warm by now)
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            for line in reversed(res.stdout.splitlines()):
                if line.startswith("{"):
                    return json.loads(line)
            raise RuntimeError("rider %s printed no result:\n%s" % (" ".join(extra), res.stderr[-2000:]))
        rider_errors = []
        def try_rider(extra):                              # (a rider that fails must not take the headline with it)
            try:
                return rider(extra)
            except Exception as exc:
                rider_errors.append("%s: %s" % (" ".join(extra), str(exc)[-500:]))
                return None
        # configs[1] (128^3, 32 ppc): a short run with its own roofline block
        j1 = try_rider(["--config", "1", "--steps", "10", "--warmup", "5"])
        if j1: second = {"workload": j1["config"]["workload"], "value": j1["value"], "steps": 10, "warmup": 5, "ms_per_step": j1["ms_per_step"],
                  "advance_p_pushes_per_s": j1["advance_p_pushes_per_s"], "roofline": j1["roofline"], "advance_p_sorting": j1.get("advance_p_sorting")}
        # configs[3] at its real per-GPU size (one of the 8 x-slabs: 32 x 256 x 128 cells, 4 species x 64 ppc, vth = 0.6 c,
        # reflecting conducting z walls), the engine's own sort policy: the hot regime of the reconnection deck
        j4 = try_rider(["--deck", "trecon", "--sort-interval", "-20", "--steps", "10", "--warmup", "8"])
        if j4: c3 = {"workload": j4["config"]["workload"], "value": j4["value"], "steps": 10, "warmup": 8, "particles": j4["config"]["particles"],
              "ms_per_step": j4["ms_per_step"], "ms_per_step_median": j4["ms_per_step_median"],
              "advance_p_pushes_per_s": j4["advance_p_pushes_per_s"], "roofline": j4["roofline"], "advance_p_by_species": j4.get("advance_p_by_species"),
              "note": "advance_p figures average over the 2 charged species and their 2 charge-0 tracer copies (which deposit nothing)"}
    deck_host = None
    if world == 1 and default_deck and not args.no_second_config:
        # configs[1] through the DECK API: oracle/decks/twostream.cxx (the very file cpu_baseline runs on the reference's
        # build) compiled against the C++ deck host with -DTS_N=128 (old-vpic_amd/host/twostream128.hip.exe, built by
        # __graft_entry__.build()): begin_initialization loads the particles on the host, vpic_simulation::advance runs on the GPU
        exe = os.path.join(ROOT, "old-vpic_amd", "host", "twostream128.hip.exe")
        if os.path.exists(exe):
            import re
            import tempfile
            try:
                n_steps = 40
                with tempfile.TemporaryDirectory() as tmp:
                    res = subprocess.run([exe, "-tpp=1", str(n_steps)], cwd=tmp, capture_output=True, text=True, timeout=600,
                                         env=dict(os.environ, VPIC_HIP_HOST_TIMING="1"))
                m = re.search(r"simulation time: ([0-9.eE+-]+)", res.stderr + res.stdout)
                if m and res.returncode == 0:
                    t = float(m.group(1))
                    deck_host = {"deck": "oracle/decks/twostream.cxx -DTS_N=128 -DTS_PPC=32 on old-vpic_amd/host (C++ deck API, one rank)",
                                 "value": 2 * 128 ** 3 * 32 * n_steps / t, "unit": "particle-pushes/s", "steps": n_steps, "ms_per_step": t / n_steps * 1e3,
                                 "note": "wall clock of the deck's whole advance loop from step 0 (its first sort included), no warm-up; compare config1_128cubed_32ppc"}
                else:
                    rider_errors.append("deck_host: rc %d: %s" % (res.returncode, (res.stderr or "")[-300:]))
            except Exception as exc:                       # noqa: BLE001
                rider_errors.append("deck_host: %s" % str(exc)[-300:])
    sustained = None
    if world == 1 and default_deck and not args.no_second_config:
        # the headline times steps 5..25 of the deck -- its quiet phase.  The two-stream instability heats the beams from about
        # step 100 on (more cell crossings, deposits further from their tiles): steps 160..200 of the same deck, a fresh process
        j5 = try_rider(["--steps", "40", "--warmup", "160"])
        if j5: sustained = {"workload": j5["config"]["workload"], "value": j5["value"], "steps": 40, "warmup": 160, "ms_per_step": j5["ms_per_step"],
                            "advance_p_pushes_per_s": j5["advance_p_pushes_per_s"], "roofline": j5["roofline"], "check": j5.get("check")}
    drift512 = None
    if world == 1 and default_deck and not args.no_second_config:
        # BASELINE configs[4] (the deposition-bound stress: cold uniform drift, 1 species x 512 ppc) at 128^3 -- the per-GPU
        # particle count of the 256^3 deck on 8 GPUs (2^30), one launch segment short of the 2^30 limit
        j6 = try_rider(["--deck", "drift", "--grid", "128", "128", "128", "--ppc", "512", "--steps", "10", "--warmup", "3"])
        if j6: drift512 = {"workload": j6["config"]["workload"], "value": j6["value"], "steps": 10, "warmup": 3, "particles": j6["config"]["particles"],
                           "ms_per_step": j6["ms_per_step"], "advance_p_pushes_per_s": j6["advance_p_pushes_per_s"], "roofline": j6["roofline"]}
    if rank == 0:
        out = {
            "metric": "particle-pushes/sec (full step: advance_p + sort when due + field solve + glue)",
            "value": r["total_np"] * args.steps / r["elapsed"],
            "unit": "particle-pushes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["elapsed"] / args.steps * 1e3,
            "ms_per_step_median": r["median_ms"],
            "device_warmup_s": args.device_warmup_s,      # state-preserving launches before the W warm-up steps (an idle box starts ~5 % slow)
            "higher_is_better": True,
            # the global box is the same at every N (x-slabs for N > 1): total work fixed
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(d, args, world) + ((", x-slabs" if d["topology"][1:] == (1, 1) else ", bricks") + f" over {world} GPUs" if world > 1 else ""),
                       "baseline_config": ("configs[2]" if args.config == 2 else "configs[1]") if not (args.grid or args.ppc or args.deck != "two-stream" or args.vth is not None) else "custom",
                       "particles": int(r["total_np"]), "decomposition": "%dx%dx%d" % d["topology"], "push_arithmetic": args.push,
                       "accumulation": args.accumulation,
                       "sort_order": "engine's choice: by 4x4x4-cell tile (vpic_hip_set_sort_order)"},
            "advance_p_pushes_per_s": r["kernel_rate"],
            "full_step_ns_per_particle": r["elapsed"] / args.steps / r["total_np"] * 1e9,
            "roofline": r["roofline"],
            "check": r["check"],
        }
        if r["sorting"] is not None:
            out["advance_p_sorting"] = r["sorting"]
        if r["by_species"]:
            out["advance_p_by_species"] = r["by_species"]
        if world == 1 and default_deck and not args.no_second_config and rider_errors:
            out["riders_failed"] = rider_errors
        if r["host_syncs"] is not None:
            out["host_syncs_per_step"] = r["host_syncs"]
        if r["exchange"] is not None:
            # N > 1: what the step loop costs the host, what the particle / field messages cost, how much of it was hidden
            out["transport"] = r["transport"]
            out["exchange"] = r["exchange"]
        if second:
            out["config1_128cubed_32ppc"] = second
            out["roofline_32ppc"] = second["roofline"]
        if c3:
            out["config3_slab"] = c3
            out["roofline_config3_slab"] = c3["roofline"]
        if drift512:
            out["config4_drift_128cubed_512ppc"] = drift512
        if sustained:
            out["sustained_steps_160_200"] = sustained
        if deck_host:
            out["deck_host"] = deck_host
        if si20:
            out["same_deck_sort_interval_20"] = si20
        if other:
            out["same_deck_" + other["push_arithmetic"] + "_arithmetic"] = other
            out["roofline_" + other["push_arithmetic"]] = other["roofline"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
