"""TILE order of crossing-heavy species (engine.h, push.hip Window<2>): the sort groups the array by 4x4x4-cell tile,
advance_p gives every tile one workgroup whose LDS window is the tile and its halo.  Only the array order differs from
the reference's sort, so everything per particle stays BIT-EXACT against the oracle run on the same array; sums carry
the usual summation-order tolerance.  GPU box only."""
import importlib
import os

import numpy as np
import pytest

from conftest import bits_equal

pytestmark = pytest.mark.gpu
ACC_TOL = 2e-6


@pytest.fixture(scope="module")
def V():
    v = importlib.import_module("old-vpic_amd")
    assert v.lib().vpic_hip_device_count() > 0, "no HIP device"
    return v


@pytest.fixture(params=["by cell within a tile", "by tile only"])
def tiles(request):
    """Forces the tile order for every sort; in its two flavours (a species whose particles mostly change cell every step
    is grouped by tile only, by kernels of their own: particles.hip)."""
    old = {k: os.environ.get(k) for k in ("VPIC_HIP_WINDOW", "VPIC_HIP_TILE_COARSE")}
    os.environ["VPIC_HIP_WINDOW"] = "tile"
    os.environ["VPIC_HIP_TILE_COARSE"] = "1" if request.param == "by tile only" else "0"
    yield request.param
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def hot_particles(L, rng, nx, ny, nz, ppc, vth=0.5, q=-0.01):
    n = nx * ny * nz * ppc
    p = np.zeros(n, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    x, y, z = rng.integers(1, nx + 1, n), rng.integers(1, ny + 1, n), rng.integers(1, nz + 1, n)
    p["i"] = L.voxel(x, y, z, nx, ny, nz)
    for c in ("ux", "uy", "uz"):
        p[c] = (rng.standard_normal(n) * vth).astype(np.float32)
    p["q"] = (q * rng.uniform(0.5, 1.5, n)).astype(np.float32)
    p["tag"] = np.arange(n) + 1
    return p


def tile_key(i, nx, ny, nz):
    sy, sz = nx + 2, (nx + 2) * (ny + 2)
    z, r = np.divmod(i, sz)
    y, x = np.divmod(r, sy)
    x, y, z = x - 1, y - 1, z - 1
    ntx, nty = (nx + 3) // 4, (ny + 3) // 4
    return (((z >> 2) * nty + (y >> 2)) * ntx + (x >> 2)) * 64 + ((z & 3) << 4 | (y & 3) << 2 | (x & 3))


def random_interpolator(orc, L, og, rng, amp=0.05):
    f = np.zeros(og.nv, L.field_t)
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        f[c] = (rng.standard_normal(og.nv) * amp).astype(np.float32)
    fi = np.zeros(og.nv, L.interpolator_t)
    orc.load_interpolator(fi, f, og)
    return fi


def acc_close(a, ref, tol=ACC_TOL):
    a = np.stack([a["jx"], a["jy"], a["jz"]]).astype(np.float64)
    r = np.stack([ref["jx"], ref["jy"], ref["jz"]]).astype(np.float64)
    assert np.abs(a - r).max() <= tol * np.abs(r).max()


@pytest.mark.parametrize("dims", [(10, 9, 7), (4, 4, 4), (3, 5, 2), (16, 8, 12), (24, 24, 24)])
def test_tile_sort_order_and_content(V, L, tiles, dims):
    """The sort under the tile policy: the same particles (tags follow), grouped tile by tile and cell by cell within a
    tile, on grids whose sides are not multiples of the tile edge too.  (The particles come in random order: on the
    largest grid a workgroup of the sort by tile only meets more destination tiles than its table holds and places the
    overflow one by one.)"""
    nx, ny, nz = dims
    rng = np.random.default_rng(3)
    p = hot_particles(L, rng, nx, ny, nz, 11)
    e = V.Engine(V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.3)))
    sp = e.new_species(-1.0, len(p) + 8, 64)
    e.set_particles(sp, p)
    e.sort_p(sp)
    got = e.get_particles(sp)
    k = tile_key(got["i"].astype(np.int64), nx, ny, nz)
    assert np.all(np.diff(k if tiles == "by cell within a tile" else k // 64) >= 0)
    assert bits_equal(got[np.argsort(got["tag"], kind="stable")], p)
    with pytest.raises(V.VpicHipError):
        e.get_partition(sp)                                   # partition[] is the reference's order's; not valid after a tile sort


@pytest.mark.parametrize("case", ["periodic", "reflecting_z", "small"])
def test_advance_p_on_tile_order_matches_oracle_per_particle(V, orc, L, tiles, case):
    """Several pushes of a hot species (most particles leave their cell, many their tile's halo after a few steps) between
    tile sorts: every particle bit for bit the oracle's on the same array, accumulators within the summation tolerance."""
    nx, ny, nz = (5, 3, 2) if case == "small" else (10, 9, 7)
    kw = {}
    if case == "reflecting_z":
        kw = dict(pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    rng = np.random.default_rng(11)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 24, vth=0.6)
    e = V.Engine(g)
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, len(p) + 64, 4096)
    e.set_particles(sp, p)
    pm = np.zeros(64, L.particle_mover_t)
    for cycle in range(2):
        e.sort_p(sp)
        ref = e.get_particles(sp)
        for step in range(4):
            ref_a = np.zeros(og.nv, L.accumulator_t)
            assert orc.advance_p(ref, len(ref), -1.0, pm, ref_a, fi, og) == 0
            e.clear_accumulators()
            assert e.advance_p(sp) == 0
            assert bits_equal(e.get_particles(sp), ref), (cycle, step)
            acc_close(e.get_accumulator(), ref_a)


@pytest.mark.parametrize("n_extra_ppc", [1, 8])
def test_particles_appended_after_a_tile_sort(V, orc, L, tiles, n_extra_ppc):
    """Particles that join the species after the sort (injection, arrivals from a neighbour) sit behind the tiles' ranges.
    A handful is pushed by workgroups of its own as it is; more than that is regrouped by tile among themselves before
    every push (their order changes, nothing else) and pushed by a second workgroup per tile."""
    nx, ny, nz = 10, 9, 7
    rng = np.random.default_rng(5)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 16, vth=0.6)
    extra = hot_particles(L, rng, nx, ny, nz, n_extra_ppc, vth=0.6)
    extra["tag"] += len(p)
    e = V.Engine(g)
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, len(p) + 5 * len(extra) + 64, 4096)
    e.set_particles(sp, p)
    e.sort_p(sp)
    e.append_particles(sp, extra[:3])
    e.append_particles(sp, extra[3:])
    pm = np.zeros(64, L.particle_mover_t)
    by_tag = lambda a: a[np.argsort(a["tag"], kind="stable")]
    for step in range(3):
        ref = e.get_particles(sp)
        n = len(ref)
        ref_a = np.zeros(og.nv, L.accumulator_t)
        assert orc.advance_p(ref, n, -1.0, pm, ref_a, fi, og) == 0
        e.clear_accumulators()
        assert e.advance_p(sp) == 0
        got = e.get_particles(sp)
        assert bits_equal(got[:len(p)], ref[:len(p)])                  # the sorted part stays where it is
        assert bits_equal(by_tag(got), by_tag(ref))
        acc_close(e.get_accumulator(), ref_a)
        more = hot_particles(L, rng, nx, ny, nz, n_extra_ppc, vth=0.6)    # and more arrive every step
        more["tag"] += 10 ** 6 * (step + 1)
        e.append_particles(sp, more)


def test_absorbing_walls_remove_particles_from_tiles(V, orc, L, tiles):
    """Absorbing x walls: movers are left on the faces, boundary_p removes them by back-filling from the end of the array
    (boundary_p.c:264), which moves particles into other tiles' ranges and shortens the last ones; then more pushes."""
    nx, ny, nz = 10, 9, 7
    pbc = [L.ABSORB_PARTICLES, 0, 0, L.ABSORB_PARTICLES, 0, 0]
    rng = np.random.default_rng(9)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), pbc=pbc)
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), pbc=pbc)
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 16, vth=0.6)
    e = V.Engine(g)
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, len(p) + 64, len(p))
    e.set_particles(sp, p)
    e.sort_p(sp)
    ref = e.get_particles(sp)
    n = len(ref)
    for step in range(3):
        pm = np.zeros(n, L.particle_mover_t)
        ref_a = np.zeros(og.nv, L.accumulator_t)
        nm = orc.advance_p(ref, n, -1.0, pm, ref_a, fi, og)
        e.clear_accumulators()
        assert e.advance_p(sp) == nm and nm > 0
        got = e.get_particles(sp)
        assert bits_equal(got, ref[:n]), step
        acc_close(e.get_accumulator(), ref_a)
        e.boundary_p_pack()
        assert e.np(sp) == n - nm
        # the survivors, whatever their order
        gone = np.zeros(n, bool)
        gone[pm["i"][:nm]] = True
        keep = ref[:n][~gone]
        got = e.get_particles(sp)
        assert bits_equal(got[np.argsort(got["tag"], kind="stable")], keep[np.argsort(keep["tag"], kind="stable")])
        ref, n = got.copy(), len(got)


def test_adaptive_policy_switches_a_hot_species_to_tiles(V, L):
    """Without the override: vpic_hip_step with adaptive sorting moves a species that keeps crossing cells to tile order on
    its own, and the run's energies stay those of the run that sorts by voxel every step."""
    nx = ny = nz = 16
    dt = np.float32(0.95 / np.sqrt(3.0))
    runs = {}
    for mode in ("voxel", "adaptive"):
        if mode == "voxel":
            os.environ["VPIC_HIP_WINDOW"] = "wide"
        else:
            os.environ.pop("VPIC_HIP_WINDOW", None)
        try:
            e = V.Engine(V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), dt))
            e.set_vacuum()
            sps = []
            for k, drift in enumerate((0.2, -0.2)):
                sp = e.new_species(-1.0, nx * ny * nz * 40, 4096)
                e.load_maxwellian(sp, 32, 1 + k, -1.0 / 64, (drift, 0.0, 0.0), 0.6)
                sps.append(sp)
            e.load_interpolator()
            en = []
            for step in range(24):
                e.step(step, 1 if mode == "voxel" else -20)
                en.append(list(e.energy_f()) + [e.energy_p(sp) for sp in sps])
            runs[mode] = (np.array(en), [e.species_order(sp) for sp in sps])
        finally:
            os.environ.pop("VPIC_HIP_WINDOW", None)
    a, b = runs["voxel"][0], runs["adaptive"][0]
    np.testing.assert_allclose(b[:, 6:], a[:, 6:], rtol=2e-6)
    np.testing.assert_allclose(b[:, :6], a[:, :6], rtol=5e-4, atol=1e-9)
    assert all(o == "tile" for o in runs["adaptive"][1]) and all(o != "tile" for o in runs["voxel"][1])


def test_a_clumped_species_leaves_the_tile_order(V, orc, L):
    """Every particle in one tile: that tile's workgroup would do the whole launch alone.  The push notices (the fullest
    tile's count rides back from the sort), runs the row-window kernel on the array as it is -- same particles bit for
    bit -- and the next sort is by voxel."""
    nx = ny = nz = 16
    rng = np.random.default_rng(21)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    fi = random_interpolator(orc, L, og, rng)
    n = 80000
    p = np.zeros(n, L.particle_t)
    for c in ("dx", "dy", "dz"):
        p[c] = rng.uniform(-1, 1, n).astype(np.float32)
    p["i"] = L.voxel(rng.integers(5, 9, n), rng.integers(5, 9, n), rng.integers(5, 9, n), nx, ny, nz)   # cells 5..8: tile (1,1,1)
    for c in ("ux", "uy", "uz"):
        p[c] = (rng.standard_normal(n) * 0.3).astype(np.float32)
    p["q"] = -0.01
    e = V.Engine(g)
    e.set_sort_order("engine")
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, n + 64, 4096)
    e.set_particles(sp, p)
    e.sort_p(sp)
    assert e.species_order(sp) == "tile"
    ref = e.get_particles(sp)
    ref_a = np.zeros(og.nv, L.accumulator_t)
    pm = np.zeros(64, L.particle_mover_t)
    assert orc.advance_p(ref, n, -1.0, pm, ref_a, fi, og) == 0
    e.clear_accumulators()
    assert e.advance_p(sp) == 0
    assert bits_equal(e.get_particles(sp), ref)
    acc_close(e.get_accumulator(), ref_a)
    e.sort_p(sp)
    assert e.species_order(sp) == "voxel"


def test_thin_grids_keep_the_reference_order(V, L):
    """A grid thinner than a tile on some axis (a 2-D deck) is not sorted by tile: the engine's choice is the reference's
    order there."""
    rng = np.random.default_rng(2)
    for dims in ((12, 1, 12), (3, 8, 8)):
        nx, ny, nz = dims
        e = V.Engine(V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.3)))
        e.set_sort_order("engine")
        p = hot_particles(L, rng, nx, ny, nz, 8)
        sp = e.new_species(-1.0, len(p) + 8, 64)
        e.set_particles(sp, p)
        e.sort_p(sp)
        assert e.species_order(sp) == "voxel"
        assert np.all(np.diff(e.get_particles(sp)["i"]) >= 0)


@pytest.mark.parametrize("case", ["periodic", "reflecting_z", "with_appended"])
def test_the_push_before_a_sort_counts_for_it(V, orc, L, case):
    """The histogram of the next sort taken inside advance_p (Species::hist; vpic_hip_species_sort_hint, vpic_hip_step does it
    by itself): the sort that follows starts at its scan.  After it the array is in tile order by cell, holds the same
    particles (by tag, bit for bit those of the oracle pushed on the same arrays), and a second sort -- which counts for
    itself -- changes nothing about the cells' ranges.  Hot species: half the particles end the step in another cell, some
    outside their tile's window."""
    nx, ny, nz = 12, 9, 8
    kw = {}
    if case == "reflecting_z":
        kw = dict(pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    rng = np.random.default_rng(23)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 24, vth=0.5)
    e = V.Engine(g)
    e.set_sort_order("engine")
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, 2 * len(p), 4096)
    e.set_particles(sp, p)
    e.sort_p(sp)
    if case == "with_appended":
        extra = hot_particles(L, rng, nx, ny, nz, 2, vth=0.5)
        extra["tag"] += len(p)
        e.append_particles(sp, extra)
    pm = np.zeros(64, L.particle_mover_t)
    for step in range(3):                                   # a few steps of disorder first
        e.clear_accumulators()
        assert e.advance_p(sp) == 0
    ref = e.get_particles(sp)
    ref_a = np.zeros(og.nv, L.accumulator_t)
    assert orc.advance_p(ref, len(ref), -1.0, pm, ref_a, fi, og) == 0
    V.lib().vpic_hip_species_sort_hint(e._h, sp)
    e.clear_accumulators()
    assert e.advance_p(sp) == 0                             # this push counts
    assert bits_equal(e.get_particles(sp), ref)
    acc_close(e.get_accumulator(), ref_a)
    e.sort_p(sp)                                            # ... for this sort
    got = e.get_particles(sp)
    k = tile_key(got["i"].astype(np.int64), nx, ny, nz)
    assert np.all(np.diff(k) >= 0)
    assert bits_equal(got[np.argsort(got["tag"], kind="stable")], ref[np.argsort(ref["tag"], kind="stable")])
    e.sort_p(sp)                                            # counts for itself: same cells in the same places
    again = e.get_particles(sp)
    assert np.array_equal(again["i"], got["i"])


def _records(p):
    """particle records without tags, in one canonical order (bit patterns, every field a sort key)"""
    a = np.stack([p[n].view(np.uint32) for n in ("i", "dx", "dy", "dz", "ux", "uy", "uz", "q")], axis=1)
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("case", ["cold_beams", "hot", "reflecting_z"])
def test_a_sort_that_finds_the_counts_happens_inside_the_push(V, orc, L, case, monkeypatch):
    """vpic_hip_sort_advance_p (vpic_hip_step does the same): after a push that counted for the sort,
    the sort moves nothing and the push writes every particle to its sorted place (advance_p_kernel<.., SORT>).  Against sort_p
    + advance_p on a second engine: the same particles bit for bit, the same accumulators to float-sum
    tolerance, the array in tile order by the cells BEFORE the push, and the next push and sort work on it.
    (VPIC_HIP_SORT_IN_PUSH=1: every time it can be; left alone the engine times both ways and keeps the cheaper one -- the test below.)"""
    monkeypatch.setenv("VPIC_HIP_SORT_IN_PUSH", "1")
    nx, ny, nz = 16, 12, 8
    kw = {}
    if case == "reflecting_z":
        kw = dict(pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    rng = np.random.default_rng(29)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 40, vth=0.5 if case in ("hot", "reflecting_z") else 0.05)
    p["tag"] = 0                                            # (tags ride outside the push: a tagged species sorts the ordinary way)
    if case == "cold_beams":
        p["ux"] += np.float32(0.3)
    engines = []
    for _ in range(2):                                          # a: sort_advance_p; b: sort_p, then advance_p
        e = V.Engine(g)
        e.set_sort_order("engine")
        e.set_interpolator(fi)
        sp = e.new_species(-1.0, 2 * len(p), 4096)
        e.set_particles(sp, p)
        e.sort_p(sp)
        engines.append((e, sp))
    for e, sp in engines:
        for step in range(3):
            if step == 2:
                V.lib().vpic_hip_species_sort_hint(e._h, sp)    # the third push counts for the sort
            e.clear_accumulators()
            assert e.advance_p(sp) == 0
    (a, spa), (b, spb) = engines
    before = a.get_particles(spa)
    assert np.array_equal(_records(before), _records(b.get_particles(spb)))   # (the order inside a cell is the sort's atomics' order)
    a.profile_enable(True)
    a.clear_accumulators()
    assert a.sort_advance_p(spa) == 0
    b.clear_accumulators()
    b.sort_p(spb)
    assert b.advance_p(spb) == 0
    ms, launches, parts = a.profile_read_sorting()
    assert launches == 1 and parts == len(p)                    # engine a did sort inside the push
    pa, pb = a.get_particles(spa), b.get_particles(spb)
    assert len(pa) == len(pb) == len(p)
    assert np.array_equal(_records(pa), _records(pb))
    acc_close(a.get_accumulator(), b.get_accumulator())
    # the order: by tile and cell of where each particle was BEFORE this push -- engine b's array (sorted, then pushed in
    # place) shows both states side by side; a's array holds the same particles cell range by cell range
    kb = tile_key(before["i"].astype(np.int64), nx, ny, nz)
    counts = np.bincount(kb, minlength=kb.max() + 1)
    starts = np.concatenate([[0], np.cumsum(counts)])
    b_sorted_before = before[np.argsort(kb, kind="stable")]
    ref = b_sorted_before.copy()
    ref_a = np.zeros(og.nv, L.accumulator_t)
    pm = np.zeros(4096, L.particle_mover_t)
    orc.advance_p(ref, len(ref), -1.0, pm, ref_a, fi, og)       # pushed in place: range k of ref = the particles that were in key k
    for k in np.flatnonzero(counts)[:: max(1, len(np.flatnonzero(counts)) // 200)]:
        lo, hi = starts[k], starts[k + 1]
        assert np.array_equal(_records(pa[lo:hi]), _records(ref[lo:hi])), k
    # ... and life goes on: a plain push, a counting push, another sort inside a push
    for e, sp in engines:
        e.clear_accumulators()
        assert e.advance_p(sp) == 0
        V.lib().vpic_hip_species_sort_hint(e._h, sp)
        assert e.advance_p(sp) == 0
        if e is a:
            assert e.sort_advance_p(sp) == 0
        else:
            e.sort_p(sp)
            assert e.advance_p(sp) == 0
    assert a.profile_read_sorting()[1] == 2
    assert np.array_equal(_records(a.get_particles(spa)), _records(b.get_particles(spb)))
    a.sort_p(spa)
    k = tile_key(a.get_particles(spa)["i"].astype(np.int64), nx, ny, nz)
    assert np.all(np.diff(k) >= 0)
    for e, _ in engines:
        e.close()


def test_sort_inside_the_push_or_before_it_is_decided_by_measurement(V, orc, L, monkeypatch):
    """Left to itself (no VPIC_HIP_SORT_IN_PUSH) a species that is due is sorted inside its push at first; before its push for
    the first time when the launch that sorted took more than 1.9 x the counting launch before it (a species whose particles
    have spread between sorts writes its new order in runs of two) or at the sixteenth sort at the latest; from then on
    whichever way took less time, the other tried again every eighth sort (engine.hip: sort_and_push).  Eighteen cycles of push,
    counting push, sort + push on one engine against sort_p + advance_p on another: the same particles bit for bit after every
    cycle whichever way was taken, both ways were taken, and the array ends in tile order."""
    monkeypatch.delenv("VPIC_HIP_SORT_IN_PUSH", raising=False)
    nx, ny, nz = 16, 12, 8
    rng = np.random.default_rng(37)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 40, vth=0.1)
    p["tag"] = 0
    engines = []
    for _ in range(2):
        e = V.Engine(g)
        e.set_sort_order("engine")
        e.set_interpolator(fi)
        sp = e.new_species(-1.0, 2 * len(p), 4096)
        e.set_particles(sp, p)
        e.sort_p(sp)
        engines.append((e, sp))
    (a, spa), (b, spb) = engines
    a.profile_enable(True)
    inside = []
    for cycle in range(18):
        for e, sp in engines:
            e.clear_accumulators()
            assert e.advance_p(sp) == 0
            V.lib().vpic_hip_species_sort_hint(e._h, sp)
            assert e.advance_p(sp) == 0
            if e is a:
                assert e.sort_advance_p(sp) == 0
            else:
                e.sort_p(sp)
                assert e.advance_p(sp) == 0
            e.sync()                                            # (the measurement of this cycle is back before the next decision)
        inside.append(a.profile_read_sorting()[1])
        assert np.array_equal(_records(a.get_particles(spa)), _records(b.get_particles(spb))), cycle
        acc_close(a.get_accumulator(), b.get_accumulator())
    assert inside[0] == 1                                       # inside the push first
    assert 1 <= inside[-1] <= 17                                # ... and before it at least once
    a.sort_p(spa)
    k = tile_key(a.get_particles(spa)["i"].astype(np.int64), nx, ny, nz)
    assert np.all(np.diff(k) >= 0)
    for e, _ in engines:
        e.close()


def test_the_window_follows_a_drifting_tile(V, orc, L, monkeypatch):
    """Between sorts a tile's window follows the tile's particles (PushParams::follow): a beam that drifts 0.3 cells per step is
    two cells off its tiles after seven steps.  Forced on (VPIC_HIP_FOLLOW=1), forced off (=0) and left to the device's own
    switch, seven pushes of the same particles through the same interpolator: bit for bit the same particles (the oracle's, pushed
    on the same arrays), accumulators equal to float-sum tolerance -- where a deposit is summed does not change what is summed."""
    nx, ny, nz = 16, 12, 8
    rng = np.random.default_rng(31)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5))
    fi = random_interpolator(orc, L, og, rng, amp=0.01)
    p = hot_particles(L, rng, nx, ny, nz, 40, vth=0.02)
    p["ux"] += np.float32(0.75)                                 # 0.6 c: 0.3 cells per step at dt = 0.5
    results = []
    for mode in ("1", "0", None):
        if mode is None:
            monkeypatch.delenv("VPIC_HIP_FOLLOW", raising=False)
        else:
            monkeypatch.setenv("VPIC_HIP_FOLLOW", mode)
        e = V.Engine(g)
        e.set_sort_order("engine")
        e.set_interpolator(fi)
        sp = e.new_species(-1.0, 2 * len(p), 4096)
        e.set_particles(sp, p)
        e.sort_p(sp)
        for step in range(7):
            e.clear_accumulators()
            assert e.advance_p(sp) == 0
        got = e.get_particles(sp)
        results.append((got[np.argsort(got["tag"], kind="stable")], e.get_accumulator()))
        e.close()
    monkeypatch.delenv("VPIC_HIP_FOLLOW", raising=False)
    ref = p.copy()
    pm = np.zeros(64, L.particle_mover_t)
    for step in range(7):
        ref_a = np.zeros(og.nv, L.accumulator_t)
        assert orc.advance_p(ref, len(ref), -1.0, pm, ref_a, fi, og) == 0
    for got, acc in results:
        assert bits_equal(got, ref)
        acc_close(acc, ref_a)


def test_a_phased_push_keeps_its_first_launch_s_decision(V, orc, L):
    """vpic_hip_advance_p_phase: the fullest tile's count (a pinned word the sort's kernels write while the host runs ahead)
    may land BETWEEN the two launches of one push and say 'too clumped for tiles'.  The second launch must still push the
    interior tiles the first one left (the boundary movers are on the wire by then); the switch takes effect with the
    next push.  Forced with the library's test hook; against a one-launch push of the same particles."""
    nx, ny, nz = 16, 8, 8
    rng = np.random.default_rng(41)
    kw = dict(pbc=[1, 0, 0, 1, 0, 0], fbc=[1, 0, 0, 1, 0, 0])              # the x faces belong to another domain (rank 1)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 24, vth=0.3)
    results = []
    for poke in (False, True):
        e = V.Engine(g)
        e.set_sort_order("engine")
        e.set_interpolator(fi)
        sp = e.new_species(-1.0, 2 * len(p), len(p))
        e.set_particles(sp, p)
        e.sort_p(sp)
        assert e.species_order(sp) == "tile"
        e.clear_accumulators()
        e.exchange_begin()
        e.advance_p_phase(sp, 1)
        if poke:
            assert V.lib().vpic_hip_debug_poke_tile_max(e._h, sp, 1 << 30) == 0
        e.advance_p_phase(sp, 2)                                           # (raised "phase 2 without phase 1" before the fix)
        e.sync()
        got = e.get_particles(sp)
        results.append((got[np.argsort(got["tag"], kind="stable")], e.get_accumulator()))
        if poke:                                                           # the next push honours the word: row windows, same particles
            e.clear_accumulators()
            e.exchange_begin()
            e.advance_p_phase(sp, 1)
            e.advance_p_phase(sp, 2)
            e.sync()
        e.close()
    assert bits_equal(results[0][0], results[1][0])
    acc_close(results[1][1], results[0][1])
    ref = p.copy()
    pm = np.zeros(len(p), L.particle_mover_t)
    ref_a = np.zeros(og.nv, L.accumulator_t)
    orc.advance_p(ref, len(ref), -1.0, pm, ref_a, fi, og)
    assert bits_equal(results[1][0], ref[np.argsort(ref["tag"], kind="stable")])


@pytest.mark.parametrize("stage", ["1", "0"])
@pytest.mark.parametrize("case", ["periodic", "reflecting_z", "absorbing_x", "charge_0_copy", "lukewarm"])
def test_positions_wait_for_their_crossers(V, orc, L, monkeypatch, case, stage):
    """advance_p of a species sorted by tile only, and of a charge-0 copy, keeps the positions of up to two passes in registers
    until the passes' cell-crossers have finished their moves and stores them once (push.hip, STAGE; VPIC_HIP_STAGE forces it
    on / off, the engine switches it on from a third of the particles crossing per step).  Either way: every particle bit
    for bit the oracle's, movers on absorbing faces included; with appended particles behind the sorted ones; over several
    steps between sorts (corner cutters that outlast their batch fall back to the late stores)."""
    monkeypatch.setenv("VPIC_HIP_WINDOW", "tile")
    monkeypatch.setenv("VPIC_HIP_TILE_COARSE", "1")
    monkeypatch.setenv("VPIC_HIP_STAGE", stage)
    nx, ny, nz = 10, 9, 7
    kw = {}
    if case == "reflecting_z":
        kw = dict(pbc=[0, 0, L.REFLECT_PARTICLES, 0, 0, L.REFLECT_PARTICLES])
    if case == "absorbing_x":
        kw = dict(pbc=[L.ABSORB_PARTICLES, 0, 0, L.ABSORB_PARTICLES, 0, 0])
    rng = np.random.default_rng(13)
    g = V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    og = orc.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), np.float32(0.5), **kw)
    fi = random_interpolator(orc, L, og, rng)
    p = hot_particles(L, rng, nx, ny, nz, 40, vth=0.15 if case == "lukewarm" else 0.6)
    if case == "charge_0_copy":
        p["q"] = 0
    extra = hot_particles(L, rng, nx, ny, nz, 3, vth=0.6)
    extra["tag"] += len(p)
    if case == "charge_0_copy":
        extra["q"] = 0
    e = V.Engine(g)
    e.set_interpolator(fi)
    sp = e.new_species(-1.0, 2 * len(p), len(p))
    e.set_particles(sp, p)
    e.sort_p(sp)
    e.append_particles(sp, extra)
    by_tag = lambda a: a[np.argsort(a["tag"], kind="stable")]
    ref = e.get_particles(sp)
    n = len(ref)
    for step in range(5):
        pm = np.zeros(n, L.particle_mover_t)
        ref_a = np.zeros(og.nv, L.accumulator_t)
        nm = orc.advance_p(ref, n, -1.0, pm, ref_a, fi, og)
        e.clear_accumulators()
        assert e.advance_p(sp) == nm
        got = e.get_particles(sp)
        assert bits_equal(by_tag(got), by_tag(ref[:n])), step
        acc_close(e.get_accumulator(), ref_a)
        if nm:                                               # the movers: the same particles with the same remaining displacement
            gm, rm = e.get_movers(sp), pm[:nm]
            gt, rt = got["tag"][gm["i"]], ref["tag"][rm["i"]]
            go, ro = np.argsort(gt), np.argsort(rt)
            assert np.array_equal(gt[go], rt[ro])
            for c in ("dispx", "dispy", "dispz"):
                assert np.array_equal(gm[c][go].view(np.uint32), rm[c][ro].view(np.uint32)), c
            e.boundary_p_pack()
            ref = e.get_particles(sp)
            n = len(ref)
        else:
            ref = got.copy()
    e.close()
