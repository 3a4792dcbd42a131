"""Replays the cleaning-enabled plumbing16 deck through the Engine API several times, keeping a
per-phase fingerprint (sum of |field| in double) for every step; prints where runs first disagree
beyond round-off."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deck16 as deck
V = importlib.import_module("old-vpic_amd")
COMPS = ("ex", "ey", "ez", "cbx", "cby", "cbz", "rhof", "rhob", "tcax", "jfx")

def fp(e):
    f = e.get_fields()
    return np.stack([f[c].astype(np.float64) for c in COMPS])

def run():
    n = deck.N
    e = V.Engine(V.make_grid(n, n, n, deck.LEN, deck.LEN, deck.LEN, deck.courant_dt()))
    e.set_vacuum()
    p = deck.load_particles()
    sp = e.new_species(-1.0, 2 * len(p), 4096)
    e.set_particles(sp, p)
    e.load_interpolator()
    # initialize.cxx:32-89
    e.synchronize_tang_e_norm_b(); e.compute_div_b_err(); e.compute_rms_div_b_err(); e.clean_div_b(); e.compute_curl_b()
    e.clear_rhof(); e.accumulate_rho_p(sp); e.synchronize_rho(); e.compute_rhob(); e.compute_div_e_err()
    if e.compute_rms_div_e_err() > 0: e.clean_div_e()
    e.synchronize_tang_e_norm_b(); e.load_interpolator(); e.uncenter_p(sp)
    log = []
    def mark(step, name): log.append((step, name, fp(e)))
    parts = {}
    for step in range(10):
        e.clear_accumulators()
        if step % deck.SORT_INTERVAL == 0: e.sort_p(sp)
        if step == 9: parts['before'] = e.get_particles(sp); parts['fi'] = e.get_interpolator(); parts['g'] = e
        e.advance_p(sp); e.reduce_accumulators()
        if step == 9: parts['after'] = e.get_particles(sp); parts['acc'] = e.get_accumulator()
        e.clear_jf(); e.unload_accumulator(); e.synchronize_jf(); mark(step, "jf")
        e.advance_b(0.5); e.advance_e(); e.advance_b(0.5); mark(step, "fields")
        if step % 10 == 0:
            e.clear_rhof(); e.accumulate_rho_p(sp); mark(step, "rho_p")
            e.synchronize_rho(); mark(step, "sync_rho")
            e.compute_div_e_err(); mark(step, "div_e")
            err = e.compute_rms_div_e_err()
            if err > 0:
                e.clean_div_e(); mark(step, "clean_e1")
                e.compute_div_e_err(); err = e.compute_rms_div_e_err()
                if err > 0: e.clean_div_e(); mark(step, "clean_e2")
            e.compute_div_b_err(); mark(step, "div_b")
            err = e.compute_rms_div_b_err()
            if err > 0:
                e.clean_div_b(); mark(step, "clean_b1")
                e.compute_div_b_err(); err = e.compute_rms_div_b_err()
                if err > 0: e.clean_div_b(); mark(step, "clean_b2")
                else: mark(step, "clean_b2_skipped")
            else: mark(step, "clean_b_skipped")
            e.synchronize_tang_e_norm_b(); mark(step, "sync_te")
        e.load_interpolator()
    return log, parts

allr = [run() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)]
runs = [a for a, _ in allr]
ref = runs[0]
from oracle import pyorc
L = importlib.import_module("old-vpic_amd.layout")
og = pyorc.make_grid(deck.N, deck.N, deck.N, deck.LEN, deck.LEN, deck.LEN, deck.courant_dt())
for k, (_, pr) in enumerate(allr):
    p = pr['before'].copy(); a = np.zeros(og.nv, L.accumulator_t); pm = np.zeros(4096, L.particle_mover_t)
    pyorc.advance_p(p, len(p), -1.0, pm, a, pr['fi'].copy(), og)
    ao = np.stack([a[c] for c in ('jx','jy','jz')]).astype(np.float64); ag = np.stack([pr['acc'][c] for c in ('jx','jy','jz')]).astype(np.float64)
    d = np.abs(ao - ag); w = np.unravel_index(d.argmax(), d.shape)
    pa = pr['after']
    same_p = all(np.array_equal(p[c], pa[c]) for c in ('dx','dy','dz','i','ux','uy','uz'))
    print('run', k, 'GPU vs oracle on its own inputs: acc max diff %.3e (of %.3e) at' % (d.max(), np.abs(ao).max()), w, 'particles bit-equal:', same_p)
    if d.max() > 1e-6:
        v = int(w[1]); sy = deck.N + 2; sz = sy * sy
        near = np.nonzero(np.isin(pr['before']['i'], [v, v - 1, v + 1, v - sy, v + sy, v - sz, v + sz]))[0]
        print('   voxel', v, '= (x,y,z)', v % sy, (v // sy) % sy, v // sz, ' oracle row', ao[:, v], ' gpu row', ag[:, v])
        for idx in near:
            b, af = pr['before'][idx], p[idx]
            if b['i'] != af['i'] or True:
                print('   idx', idx, 'i', b['i'], '->', af['i'], 'pos', b['dx'], b['dy'], b['dz'], '->', af['dx'], af['dy'], af['dz'], 'u', b['ux'], b['uy'], b['uz'])
canon = lambda q: q[np.argsort(q['tag'])]
for k, (_, pr) in enumerate(allr[1:], 1):
    p0, pk = allr[0][1], pr
    same_before = all(np.array_equal(canon(p0['before'])[c], canon(pk['before'])[c]) for c in ('dx','dy','dz','i','ux','uy','uz'))
    same_after = all(np.array_equal(canon(p0['after'])[c], canon(pk['after'])[c]) for c in ('dx','dy','dz','i','ux','uy','uz'))
    same_order = np.array_equal(p0['before']['tag'], pk['before']['tag'])
    a0 = np.stack([p0['acc'][c] for c in ('jx','jy','jz')]).astype(np.float64); ak = np.stack([pk['acc'][c] for c in ('jx','jy','jz')]).astype(np.float64)
    d = np.abs(a0 - ak); w = np.unravel_index(d.argmax(), d.shape)
    print('run', k, 'particles before step 9 equal:', same_before, 'same array order:', same_order, 'after:', same_after, 'acc max diff %.3e of max %.3e at' % (d.max(), np.abs(a0).max()), w, 'n voxels differing > 1e-6*max:', int((d.max(axis=(0, 2)) > 1e-6 * np.abs(a0).max()).sum()))
for k, r in enumerate(runs[1:], 1):
    names_r = [(s, n) for s, n, _ in r]; names_0 = [(s, n) for s, n, _ in ref]
    if names_r != names_0:
        first = next(i for i, (a, b) in enumerate(zip(names_r, names_0)) if a != b)
        print("run", k, "takes another branch at", names_r[first], "vs", names_0[first]); continue
    worst = None
    for (s, n, a), (_, _, b) in zip(r, ref):
        rel = np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-30)
        bad = [(COMPS[i], rel[i]) for i in range(len(COMPS)) if rel[i] > 2e-5 and np.abs(b[i]).max() > 1e-20]
        if bad:
            worst = (s, n, bad); break
    print("run", k, "first disagreement with run 0:", worst)
