"""The kernel instances of the HEADLINE, at the size the bench runs them: BASELINE.json configs[2] whole on one GPU
(256^3 cells, 2 species x 64 ppc = 2.15e9 particles, 262 144 tiles), driven by vpic_hip_step(e, step, 10) in the engine's
own order exactly as bench.py drives it, for twelve steps -- so that the plain launch, the launch that takes the next
sort's histogram (advance_p_kernel<.., HIST>, step 9), the launch that sorts as it pushes (<.., SORT>, step 10) and the
first launch on the freshly written array (step 11) all fire on arrays of 2^30 particles with byte offsets up to 2^32 - 4.

What is checked, around each of those steps (reference: advance_p.cxx:68-177, move_p.c:34-134, sort_p.c:48-101):
  * particles, bit for bit against the oracle: ~200 000 of them per probe.  Plain / HIST / following launch: the same
    array ranges before and after (the push works in place).  Sorting launch: WHOLE TILES -- every particle whose cell
    lies in a sampled tile before the push is gathered from the old order (the tile's range and its 26 neighbours': the
    beams drift 1.1 cells in ten steps), pushed by the oracle, and must be exactly what the new order holds in that
    tile's range, cell range by cell range (vpic_hip_species_get_tile_partition).  That the ranges have the right
    lengths is the histogram of step 9 checked at this size; that they hold the right particles is the tile order;
  * ALL particles at once, through the property the deposition exists for: it conserves charge.  With rho from
    accumulate_rho_p (rho_p.c:23-86) before and after the step and jf as unload_accumulator left it,
    (rho_1 - rho_0) / dt + div jf = 0 at every node (verified on the CPU oracle in test_oracle_golden.py); a single
    particle deposited in the wrong cell, twice or not at all breaks it by ~1e-2 of rho / dt;
  * the particle counts.
GPU box only; VPIC_HIP_HEADLINE_GRID="nx,ny,nz" runs a smaller box (development)."""
import importlib
import os

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
PPC = 64
SORT_INTERVAL = 10


def tile_key(i, nx, ny, nz):
    sy, sz = nx + 2, (nx + 2) * (ny + 2)
    z, r = np.divmod(i.astype(np.int64), sz)
    y, x = np.divmod(r, sy)
    x, y, z = x - 1, y - 1, z - 1
    ntx, nty = (nx + 3) // 4, (ny + 3) // 4
    return (((z >> 2) * nty + (y >> 2)) * ntx + (x >> 2)) * 64 + ((z & 3) << 4 | (y & 3) << 2 | (x & 3))


def records(p):
    a = np.stack([p[n].view(np.uint32) for n in ("i", "dx", "dy", "dz", "ux", "uy", "uz", "q")], axis=1)
    return a[np.lexsort(a.T[::-1])]


def continuity_residual(rho0, rho1, f, dims, dt):
    """max |(rho1 - rho0) / dt + div jf| over the nodes 2 .. n-1 of every axis (clear of the ghost bookkeeping), and the scale"""
    nx, ny, nz = dims
    shape = (nz + 2, ny + 2, nx + 2)
    s = (slice(2, nz), slice(2, ny), slice(2, nx))
    res = (rho1.astype(np.float64) - rho0.astype(np.float64)).reshape(shape)[s] / float(dt)
    for comp, ax in (("jfx", 2), ("jfy", 1), ("jfz", 0)):
        j = f[comp].astype(np.float64).reshape(shape)
        lo = [slice(2, nz), slice(2, ny), slice(2, nx)]
        lo[ax] = slice(1, (nz, ny, nx)[ax] - 1)
        res += j[s] - j[tuple(lo)]                      # cells of size 1 (the bench deck)
    return float(np.abs(res).max()), float(np.abs(rho1).max()) / float(dt)


@pytest.fixture(scope="module")
def headline(orc, L):
    old = os.environ.get("VPIC_HIP_RHO_PER_PARTICLE")
    os.environ["VPIC_HIP_RHO_PER_PARTICLE"] = "1"          # accumulate_rho_p must not re-sort the species it only reads
    V = importlib.import_module("old-vpic_amd")
    NX, NY, NZ = (int(x) for x in os.environ.get("VPIC_HIP_HEADLINE_GRID", "256,256,256").split(","))
    dt = np.float32(0.95 / np.sqrt(3.0))
    e = V.Engine(V.make_grid(NX, NY, NZ, float(NX), float(NY), float(NZ), dt))
    e.set_vacuum()
    e.set_sort_order("engine")
    q = -float((0.2 / float(dt)) ** 2 / (2 * PPC))
    n_sp = NX * NY * NZ * PPC
    sps = []
    for k, u in enumerate(((0.2, 0.0, 0.0), (-0.2, 0.0, 0.0))):        # bench.py's deck
        sp = e.new_species(-1.0, n_sp, max(n_sp // 16, 1024))
        e.load_maxwellian(sp, PPC, 1 + k, q, u, 0.02)
        sps.append(sp)
    nv = e.nv
    f = np.zeros(nv, L.field_t)                                         # a smooth field: the push is not a free flight
    idx = np.arange(nv)
    x, y, z = idx % (NX + 2), (idx // (NX + 2)) % (NY + 2), idx // ((NX + 2) * (NY + 2))
    for c, (a, b) in {"ex": (0.02, 3), "ey": (0.015, 5), "ez": (0.01, 7), "cbx": (0.03, 2), "cby": (0.02, 4), "cbz": (0.025, 6)}.items():
        f[c] = (a * np.sin(2 * np.pi * b * (x + 2 * y + 3 * z) / 128)).astype(np.float32)
    e.set_fields(f)
    del f, idx, x, y, z
    e.load_interpolator()
    og = orc.make_grid(NX, NY, NZ, float(NX), float(NY), float(NZ), dt)
    sp = sps[-1]
    rng = np.random.default_rng(17)

    def rho():
        e.clear_rhof()
        for s in sps:
            e.accumulate_rho_p(s)
        return e.get_fields()["rhof"].copy()

    def oracle_push(p, fi):
        p = p.copy()
        a = np.zeros(og.nv, L.accumulator_t)
        pm = np.zeros(64, L.particle_mover_t)
        assert orc.advance_p(p, len(p), -1.0, pm, a, fi, og) == 0
        return p

    ntx, nty, ntz = (NX + 3) // 4, (NY + 3) // 4, (NZ + 3) // 4
    out = {}
    e.profile_enable(True)
    for step in range(12):
        probe = {5: "plain", 9: "hist", 10: "sort", 11: "after_sort"}.get(step)
        if probe:
            assert e.species_order(sp) == "tile"
            rho0 = rho()
            fi = e.get_interpolator()
            sorting_before = e.profile_read_sorting()[1]
            if probe == "sort":
                tp_old = e.get_tile_partition(sp)
                tiles = np.unique(np.concatenate([rng.integers(0, ntx * nty * ntz, 44), [0, ntx - 1, ntx * nty * ntz - 1, (ntz // 2 * nty + nty // 2) * ntx]]))
                gathered = {}
                for t in tiles:
                    tx, ty, tz = t % ntx, (t // ntx) % nty, t // (ntx * nty)
                    parts = []
                    for nb in {(((tz + dz) % ntz) * nty + (ty + dy) % nty) * ntx + (tx + dx) % ntx for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)}:
                        lo, hi = int(tp_old[64 * nb]), int(tp_old[64 * nb + 64])
                        if hi > lo:
                            parts.append(e.get_particles_range(sp, lo, hi - lo))
                    p = np.concatenate(parts)
                    gathered[int(t)] = p[tile_key(p["i"], NX, NY, NZ) // 64 == t]
            else:
                n = e.np(sp)
                spots = [(int(a), 4096) for a in np.sort(rng.integers(0, n - 4096, 46))] + [(0, 4096), (n - 4096, 4096)]
                before = [e.get_particles_range(sp, a, c) for a, c in spots]
        e.step(step, SORT_INTERVAL)
        if not probe:
            continue
        r = dict(np=[e.np(s) for s in sps], sorting_launches=e.profile_read_sorting()[1] - sorting_before, order=e.species_order(sp))
        fields = e.get_fields()
        r["residual"], r["scale"] = continuity_residual(rho0, rho(), fields, (NX, NY, NZ), dt)
        del fields
        if probe == "sort":
            tp_new = e.get_tile_partition(sp)
            r["tiles"] = []
            for t, p0 in gathered.items():
                lo, hi = int(tp_new[64 * t]), int(tp_new[64 * t + 64])
                got = e.get_particles_range(sp, lo, hi - lo) if hi > lo else p0[:0]
                key0 = tile_key(p0["i"], NX, NY, NZ)
                ref = oracle_push(p0, fi)
                # the new order: cell range by cell range, the particles that were in that cell BEFORE this push
                got_key = np.repeat(np.arange(64 * t, 64 * t + 64), np.diff(tp_new[64 * t:64 * t + 65]))
                ok = len(got) == len(ref) and all(
                    np.array_equal(records(got[got_key == k]), records(ref[key0 == k])) for k in range(64 * t, 64 * t + 64))
                r["tiles"].append((t, len(ref), len(got), ok))
        else:
            r["ranges"] = [(a, bits_equal(oracle_push(p0, fi), e.get_particles_range(sp, a, c))) for (a, c), p0 in zip(spots, before)]
        out[probe] = r
    out["n_species"] = n_sp
    yield out
    e.close()
    if old is None:
        os.environ.pop("VPIC_HIP_RHO_PER_PARTICLE", None)
    else:
        os.environ["VPIC_HIP_RHO_PER_PARTICLE"] = old


@pytest.mark.parametrize("probe", ["plain", "hist", "after_sort"])
def test_in_place_launches_are_bit_exact_at_bench_size(headline, probe):
    r = headline[probe]
    assert r["np"] == [headline["n_species"]] * 2 and r["order"] == "tile"
    assert r["sorting_launches"] == 0
    assert len(r["ranges"]) == 48 and all(ok for _, ok in r["ranges"]), [a for a, ok in r["ranges"] if not ok]


def test_the_sorting_launch_is_bit_exact_and_in_tile_order_at_bench_size(headline):
    r = headline["sort"]
    assert r["np"] == [headline["n_species"]] * 2 and r["order"] == "tile"
    assert r["sorting_launches"] == 2                       # both species were sorted INSIDE their push (so step 9 counted for them)
    assert sum(n for _, n, _, _ in r["tiles"]) > 150000
    bad = [(t, n, m) for t, n, m, ok in r["tiles"] if not ok]
    assert not bad, bad


@pytest.mark.parametrize("probe", ["plain", "hist", "sort", "after_sort"])
def test_every_deposit_conserves_charge_at_bench_size(headline, probe):
    r = headline[probe]
    print(probe, "continuity residual", r["residual"], "of", r["scale"])
    assert r["residual"] <= 3e-5 * r["scale"]
