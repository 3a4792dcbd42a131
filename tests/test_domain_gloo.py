"""world_size-2 run of the slab-domain driver (old-vpic_amd/domain.py) over gloo, with the CPU
oracle plugged in behind the Engine interface, against a single-domain oracle run of the same box.
Checks the exchange choreography: tangential-B ghosts, the three-pass jf synchronisation, particle
migration with up to three rounds, counts-then-payload messaging."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GX, GY, GZ, PPC, STEPS = 8, 6, 4, 6, 12
DT = np.float32(0.95 / np.sqrt(3.0))


# the decks: "thermal" one hot species (0.4 c thermal spread: every third particle changes cell each step, many
# cross the slab boundary); "twostream" BASELINE.json configs[2] in small -- two beams drifting at +-0.2 c with
# 0.02 c thermal spread, 64 particles per cell and species
DECKS = {"thermal": dict(ppc=PPC, q=-0.02, species=[(0.0, 0.4, 0)]),
         "twostream": dict(ppc=64, q=-0.002, species=[(0.2, 0.02, 0), (-0.2, 0.02, 5000)])}


def make_particles(L, rng, org, dims, ppc=PPC, q=-0.02, drift=0.0, vth=0.4, seed=0):
    """Particles of the cells org+1..org+dims (global indices) of a brick, drawn per GLOBAL cell so that the union over
    bricks is the same set whatever the decomposition."""
    out = []
    (x0, y0, z0), (nxl, nyl, nzl) = org, dims
    for z in range(1, GZ + 1):
        for y in range(1, GY + 1):
            for x in range(1, GX + 1):
                r = np.random.default_rng(seed + 1000 * z + 100 * y + x)
                p = np.zeros(ppc, L.particle_t)
                for c in ("dx", "dy", "dz"):
                    p[c] = r.uniform(-1, 1, ppc).astype(np.float32)
                p["ux"] = (drift + vth * r.standard_normal(ppc)).astype(np.float32)
                p["uy"] = (vth * r.standard_normal(ppc)).astype(np.float32)
                p["uz"] = (vth * r.standard_normal(ppc)).astype(np.float32)
                p["q"] = q
                if x0 < x <= x0 + nxl and y0 < y <= y0 + nyl and z0 < z <= z0 + nzl:
                    p["i"] = L.voxel(x - x0, y - y0, z - z0, nxl, nyl, nzl)
                    out.append(p)
    return np.concatenate(out)


def deck_species(L, name, org, dims):
    D = DECKS[name]
    return [make_particles(L, None, org, dims, D["ppc"], D["q"], drift, vth, seed) for drift, vth, seed in D["species"]]


def brick(rank, topo):
    """Origin and size of the rank's brick (ranks ordered ix + gpx*(iy + gpy*iz), partition.c:41-45)."""
    gp = topo
    c = (rank % gp[0], (rank // gp[0]) % gp[1], rank // (gp[0] * gp[1]))
    dims = (GX // gp[0], GY // gp[1], GZ // gp[2])
    return tuple(c[a] * dims[a] for a in range(3)), dims


def deck(clean=False, name="thermal", legacy=False, rehearsal=False, topo=None):
    d = dict(comm_stream_rehearsal=rehearsal, **({"topology": topo} if topo else {}), gx=GX, gy=GY, gz=GZ, ppc=DECKS[name]["ppc"], dt=DT, q=DECKS[name]["q"], drift=0.0, vth=0.0, sort_interval=5,
             species=DECKS[name]["species"], legacy_exchange=legacy)
    if clean:
        d.update(clean_div_e_interval=4, clean_div_b_interval=4, sync_shared_interval=4)
    return d


def initial_fields(L):
    """A GLOBAL initial field that is not divergence free, so that the cleaning has work to do."""
    r = np.random.default_rng(77)
    f = np.zeros((GZ + 2, GY + 2, GX + 2), L.field_t)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        v = (0.05 * r.standard_normal((GZ, GY, GX))).astype(np.float32)
        f[c][1:-1, 1:-1, 1:-1] = v
        # the values on the far faces are those of the near faces of the periodic image
        f[c][-1, :, :] = f[c][1, :, :]; f[c][:, -1, :] = f[c][:, 1, :]; f[c][:, :, -1] = f[c][:, :, 1]
    return f


def brick_of(F, org, dims):
    (x0, y0, z0), (nxl, nyl, nzl) = org, dims
    return np.ascontiguousarray(F[z0:z0 + nzl + 2, y0:y0 + nyl + 2, x0:x0 + nxl + 2]).reshape(-1)


def worker(rank, world, port, q, use_hip=False, clean=False, name="thermal", legacy=False, rehearsal=False, topo=None, opts=None):
    """opts: resident_oracle (the CPU stand-in speaks the device-resident protocol), deck (extra deck keys), np_factor /
    nm_factor (capacity of the species' arrays relative to their initial size), expect (what the run must have done)."""
    opts = opts or {}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine, ResidentOracleEngine
    L = importlib.import_module("old-vpic_amd.layout")
    domain = importlib.import_module("old-vpic_amd.domain")
    d = deck(clean, name, legacy, rehearsal, topo)
    d.update(opts.get("deck", {}))
    factory = None if use_hip else (ResidentOracleEngine if opts.get("resident_oracle") else OracleEngine)
    dom = domain.SlabDomain(d, rank, world, engine_factory=factory, load=False)
    e = dom.engine
    assert not rehearsal or dom.comm is not None
    org, dims = brick(rank, topo or (world, 1, 1))
    dom.species = []
    for p in deck_species(L, name, org, dims):
        sp = e.new_species(-1.0, int(opts.get("np_factor", 4) * len(p)), int(opts.get("nm_factor", 2) * len(p)))
        e.set_particles(sp, p)
        dom.species.append(sp)
    if clean:
        e.set_fields(brick_of(initial_fields(L), org, dims))
        dom.initialize_fields()
    e.load_interpolator()
    en = []
    for step in range(STEPS):
        dom.step(step)
        en.append(np.concatenate([e.energy_f(), [e.energy_p(sp) for sp in dom.species]]))
    if rehearsal and not use_hip:
        # per step: one jf exchange per cut axis and one for tang-B -- and, with the device-resident protocol, one
        # particle message round per species and one per later round; each waits for the engine's stream, and the
        # engine's for each of them
        n_x = (len(dom.axes) + 1) * STEPS
        if dom.resident:
            n_x += (len(dom.species) + min(len(dom.axes) + 1, 3) - 1) * STEPS
        assert dom.comm.waited == n_x and dom.comm.recorded == n_x
        assert dom.estream.waited == n_x and dom.estream.recorded == n_x
    if opts.get("resident_oracle"):
        # the order of one step (SlabDomain.push_and_exchange): per species the boundary launch, its movers packed and
        # started, the interior launch; then every species' arrivals land; then the straggler round(s) carry all species
        ns, calls = len(dom.species), e.calls
        per_step = len(calls) // STEPS
        one = calls[:per_step]
        head = [c for k in range(ns) for c in (("push", k, 1), ("pack", (k,)), ("push", k, 2))]
        assert one[:3 * ns] == head, one
        rest = [c[0] for c in one[3 * ns:]]
        n_dir = len(dom.dirs)
        assert rest[:ns * n_dir] == ["inject"] * (ns * n_dir)
        assert rest[ns * n_dir:] == (["pack"] + ["inject"] * n_dir) * (min(len(dom.axes) + 1, 3) - 1)
    q.put((rank, e.get_fields(), [e.np(sp) for sp in dom.species], np.array(en), dom.host_syncs_per_step(),
           dict(recovery=getattr(dom, "n_recovery", 0), reserved=getattr(dom, "n_reserved", 0))))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_domains_match_one(orc, L):
    run_and_compare(orc, L, use_hip=False)


@pytest.mark.gpu
def test_two_hip_domains_match_one(orc, L):
    """The same comparison with the real HIP engines: two processes share the one GPU of the box,
    messages are staged through the host because gloo moves host memory only."""
    run_and_compare(orc, L, use_hip=True)


def test_two_slab_two_stream_matches_one_domain(orc, L):
    """BASELINE.json configs[2] in small: two beams (+-0.2 c, 0.02 c thermal), 64 particles per cell and species,
    the box cut into two x-slabs, against the one-domain oracle."""
    run_and_compare(orc, L, use_hip=False, name="twostream")


@pytest.mark.gpu
def test_two_slab_two_stream_hip_domains(orc, L):
    run_and_compare(orc, L, use_hip=True, name="twostream")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["thermal", "twostream"])
def test_two_hip_domains_regroup_their_arrivals(orc, L, name):
    """The same runs with the arrivals regrouped by tile before every push however few they are (the engine does that
    from 4096 appended particles on: push.hip, k_tail_sort), in the exchange's flow: counts on the device, removals
    back-filling from the appended part, two species."""
    os.environ["VPIC_HIP_TAIL_SORT_MIN"] = "1"
    try:
        run_and_compare(orc, L, use_hip=True, name=name)
    finally:
        del os.environ["VPIC_HIP_TAIL_SORT_MIN"]


@pytest.mark.gpu
def test_two_hip_domains_reference_protocol(orc, L):
    """The count-then-payload protocol of the reference (boundary_p.c:341-384) on the HIP engines (what the CPU
    tests run on the oracle): kept alive next to the device-resident one."""
    run_and_compare(orc, L, use_hip=True, legacy=True)


@pytest.mark.gpu
@pytest.mark.parametrize("clean", [False, True])
def test_two_hip_domains_communication_stream_plumbing(orc, L, clean):
    """The stream plumbing of the RCCL transport (the engine's stream as a torch external stream, events between
    it and the communication stream) driven by the staged gloo transport on the one GPU of the box: the calls the
    8-GPU run makes, without its overlap (SlabDomain.__init__, comm_stream_rehearsal)."""
    run_and_compare(orc, L, use_hip=True, clean=clean, name="twostream" if not clean else "thermal", rehearsal=True)


def test_two_domains_transport_call_sequence(orc, L):
    """The un-staged branch of the RCCL transport (tensors handed to the backend as they are, an event recorded on
    the engine's stream before and on the communication stream after every exchange) with gloo on CPU tensors and
    counting stand-ins for the streams: every exchange that was started is finished before its buffers are used."""
    run_and_compare(orc, L, use_hip=False, rehearsal=True)


@pytest.mark.parametrize("topo", [(2, 2, 1), (1, 2, 2), (2, 1, 2)])
def test_four_bricks_match_one_domain(orc, L, topo):
    """Brick decompositions (src/grid/partition.c:35-85): four domains, two cut axes -- the ordered jf passes with their
    edges, tang-B ghosts over all shared faces, particles that change domain twice in a step."""
    run_and_compare(orc, L, use_hip=False, topo=topo)


def test_four_bricks_with_divergence_cleaning_match_one_domain(orc, L):
    run_and_compare(orc, L, use_hip=False, clean=True, topo=(2, 2, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("topo,clean", [((2, 2, 1), False), ((1, 2, 2), False), ((2, 2, 1), True)])
def test_four_hip_bricks_match_one_domain(orc, L, topo, clean):
    """The same with HIP engines (four processes share the box's GPU, messages staged through the host): the
    device-resident particle exchange over four faces, three rounds."""
    run_and_compare(orc, L, use_hip=True, clean=clean, topo=topo)


def test_two_domains_overlapped_exchange_order(orc, L):
    """SlabDomain.push_and_exchange on two ranks over gloo with the CPU stand-in that speaks the device-resident protocol:
    two species, per-species messages started between a species' boundary and interior launches, arrivals landed after the
    last push, one straggler round -- the call order is asserted in the worker, the result against the one-domain oracle."""
    run_and_compare(orc, L, use_hip=False, name="twostream", rehearsal=True, opts=dict(resident_oracle=True))


def test_four_bricks_overlapped_exchange_order(orc, L):
    run_and_compare(orc, L, use_hip=False, topo=(2, 2, 1), opts=dict(resident_oracle=True))


@pytest.mark.gpu
@pytest.mark.parametrize("name,topo", [("thermal", None), ("twostream", None), ("thermal", (2, 2, 1))])
def test_hip_domains_exchange_that_grows(orc, L, name, topo):
    """boundary_p.c:416-448 grows its arrays when they run out; the device-resident exchange parks the movers of a full
    message (the particles stay where they are), both ends read `wanted > count` in its header and run an extra round
    over that face, and the species' arrays are enlarged (or compacted by an early sort) between steps from 85 % full.
    Forced here: messages of 8 injectors to begin with, arrays with 12 % head room (60 % for the four small bricks, which
    receive a quarter of their particles per step: the room must hold ONE step's arrivals, it is enlarged between steps).
    Against the one-domain oracle."""
    run_and_compare(orc, L, use_hip=True, name=name, topo=topo,
                    opts=dict(deck=dict(exchange_cap_granule=8, exchange_cap0=8), np_factor=1.6 if topo else 1.12, nm_factor=1.0,
                              expect=dict(recovery=True, reserved=True)))


@pytest.mark.gpu
@pytest.mark.parametrize("clean", [False, True])
def test_two_hip_domains_deterministic_accumulation(orc, L, clean):
    """Two slabs with 64-bit fixed-point sums (vpic_hip_set_accumulation): against the one-domain oracle like every other
    run here, and -- twice over -- against themselves: the fields of the two runs agree to the last bit, whatever order
    the neighbour's particles arrived in, the atomics landed in and the sort left the arrays in."""
    opts = dict(deck=dict(accumulation="deterministic"))
    a = run_and_compare(orc, L, use_hip=True, clean=clean, name="thermal" if clean else "twostream", opts=opts)
    b = run_and_compare(orc, L, use_hip=True, clean=clean, name="thermal" if clean else "twostream", opts=opts)
    for r in a:
        assert a[r][0].tobytes() == b[r][0].tobytes(), "fields of rank %d differ between two deterministic runs" % r


def test_two_domains_with_divergence_cleaning_match_one(orc, L):
    """Non-solenoidal initial fields, initialize()'s checks, then cleaning of E and B and the shared-face
    synchronisation every 4 steps: rho / normal-E / div-B / tang-E-norm-B messages between the slabs."""
    run_and_compare(orc, L, use_hip=False, clean=True)


@pytest.mark.gpu
def test_two_hip_domains_with_divergence_cleaning_match_one(orc, L):
    run_and_compare(orc, L, use_hip=True, clean=True)


def run_and_compare(orc, L, use_hip, clean=False, name="thermal", legacy=False, rehearsal=False, topo=None, opts=None):
    world = topo[0] * topo[1] * topo[2] if topo else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, use_hip, clean, name, legacy, rehearsal, topo, opts)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, f, n, en, syncs, did = q.get(timeout=120)
        res[r] = (f, n, en, syncs, did)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    for what in ((opts or {}).get("expect", {})):             # what the run was meant to go through, on some rank at least
        assert sum(res[r][4][what] for r in range(world)) > 0, what

    # single-domain reference
    gkw = {}
    walls = (opts or {}).get("deck", {}).get("walls")
    if walls:
        fbc, pbc = [0] * 6, [0] * 6
        for ax, (fcode, pcode) in walls.items():
            fbc[ax] = fbc[ax + 3] = fcode
            pbc[ax] = pbc[ax + 3] = pcode
        gkw = dict(fbc=fbc, pbc=pbc)
    g = orc.make_grid(GX, GY, GZ, float(GX), float(GY), float(GZ), DT, **gkw)
    f = np.zeros(g.nv, L.field_t)
    fi = np.zeros(g.nv, L.interpolator_t)
    a = np.zeros(g.nv, L.accumulator_t)
    m = orc.vacuum_coefficients()
    species = [dict(p=p.copy(), np=len(p), q_m=-1.0, pm=np.zeros(len(p), L.particle_mover_t),
                    partition=np.zeros(g.nv + 1, np.int32)) for p in deck_species(L, name, (0, 0, 0), (GX, GY, GZ))]
    ns = len(species)
    if clean:
        f[:] = initial_fields(L).reshape(-1)
        orc.initialize_fields(f, m, species, g)
    orc.load_interpolator(fi, f, g)
    en1 = []
    for step in range(STEPS):
        c = clean and step % 4 == 0
        orc.step(f, fi, a, m, species, g, sort=(step % 5 == 0), clean_e=c, clean_b=c, sync_shared=c)
        en1.append(np.concatenate([orc.energy_f(f, m, g), [orc.energy_p(s["p"], s["np"], -1.0, fi, g) for s in species]]))
    en1 = np.array(en1)

    for k in range(ns):                                       # no particle lost or duplicated, species by species
        assert sum(res[r][1][k] for r in range(world)) == species[k]["np"]
    en2 = sum(res[r][2] for r in range(world))                # energies add over domains
    np.testing.assert_allclose(en2[:, 6:], en1[:, 6:], rtol=2e-6)        # kinetic energy of every species
    if use_hip and not legacy and not clean:                  # the device-resident protocol: one read-back per step (cleaning adds its all-reduces)
        assert res[0][3] is not None and res[0][3] <= 2.0 + 1e-9, res[0][3]   # (+ the staged transport's own, counted apart)
    np.testing.assert_allclose(en2[:, :6], en1[:, :6], rtol=2e-4, atol=1e-9)
    # fields, interior voxels, brick by brick
    F1 = f.reshape(GZ + 2, GY + 2, GX + 2)
    for r in range(world):
        (x0, y0, z0), (nxl, nyl, nzl) = brick(r, topo or (world, 1, 1))
        Fr = res[r][0].reshape(nzl + 2, nyl + 2, nxl + 2)
        for c in ("ex", "ey", "ez", "cbx", "cby", "cbz", "jfx", "jfy", "jfz") + (("rhob", "rhof", "tcax", "tcay", "tcaz") if clean else ()):
            ref = F1[c][1 + z0:1 + z0 + nzl, 1 + y0:1 + y0 + nyl, 1 + x0:1 + x0 + nxl]
            got = Fr[c][1:nzl + 1, 1:nyl + 1, 1:nxl + 1]
            scale = max(np.abs(F1[c]).max(), 1e-12)
            assert np.abs(got - ref).max() <= 2e-4 * scale, (r, c)
    return res


def test_two_slabs_between_conducting_walls_match_one_domain(orc, L):
    """BASELINE configs[3]'s boundary conditions in small: the box cut into two x-slabs, periodic in y, conducting walls that
    reflect particles in z (deck key `walls`; turbulence.cxx:265-269) -- against the one-domain oracle with the same walls."""
    run_and_compare(orc, L, use_hip=False, opts=dict(deck=dict(walls={2: (-1, -1)})))


@pytest.mark.gpu
def test_two_hip_slabs_between_conducting_walls(orc, L):
    run_and_compare(orc, L, use_hip=True, opts=dict(deck=dict(walls={2: (-1, -1)})))
