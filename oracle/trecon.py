"""TEST INFRASTRUCTURE: the reference's production reconnection deck (decks/trecon-part/turbulence.cxx with its
tracer.cxx and energy.cxx, UNCHANGED; sizes from oracle/decks/trecon_small/config.h) run by the reference
executable on one and two ranks -> tests/golden/trecon.npz: the text outputs byte for byte, and what can be
compared statistically with a run that draws other random numbers (the HIP host's maxwellian_rand is not the
reference's ziggurat): particle counts and momentum moments per species from the particle dumps, field and
moment sums from the strided dumps, the energy spectra.  Needs /root/reference and MPI; run here:
    python oracle/trecon.py"""
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("old-vpic_amd.layout")
from oracle import dumpfmt as D  # noqa: E402

NX, NY, NZ, STEPS = 32, 8, 16, 40
TINY = (16, 4, 8)           # the replayed-normals runs (tests/golden/trecon_tiny.npz carries every normal they draw)
SPECIES = ("eT", "eB", "iT", "iB")
HYDRO = {"eT": "eTophydro", "eB": "eBothydro", "iT": "HTophydro", "iB": "HBothydro"}


def summarize(d, nranks, dims=None):
    """What two statistically equivalent runs must agree on, from the files in directory d."""
    NX, NY, NZ = dims or (globals()["NX"], globals()["NY"], globals()["NZ"])
    out = {}
    H = D.HEADER_V0 + 8
    for sp in SPECIES:
        parts = []
        for r in range(nranks):
            parts.append(np.fromfile(os.path.join(d, "particle", "T.%d" % STEPS, "%sparticle.%d.%d" % (sp, STEPS, r)), L.particle_t, offset=H + 4))
        p = np.concatenate(parts)
        out[sp + "_np"] = np.int64(len(p))
        out[sp + "_u2"] = np.array([np.mean(p[c].astype(np.float64) ** 2) for c in ("ux", "uy", "uz")])
        out[sp + "_q"] = np.float64(p["q"].astype(np.float64).sum())
        # hydro_dump (band: jx jy jz rho) at the last step: the charge the species carries, and its current
        tot = np.zeros(4)
        for r in range(nranks):
            raw = np.fromfile(os.path.join(d, "hydro", "T.%d" % STEPS, "%s.%d.%d" % (HYDRO[sp], STEPS, r)), np.uint8)
            nxl = NX // nranks
            n = (nxl + 2) * (NY + 2) * (NZ + 2)
            band = raw[H + 12:H + 12 + 4 * 4 * n].view(np.float32).reshape(4, NZ + 2, NY + 2, nxl + 2).astype(np.float64)
            tot += band[:, 1:-1, 1:-1, 1:-1].sum(axis=(1, 2, 3))
            if r == 0:
                out[sp + "_hydro_bytes"] = np.int64(len(raw))             # header + 4 bands + the 6 energy-band blocks energy.cxx appends
        out[sp + "_hydro_sum"] = tot
        spec = np.zeros(800)
        for r in range(nranks):
            spec += np.fromfile(os.path.join(d, "hydro", "T.%d" % STEPS, "spectrum-%s.%d.%d" % (HYDRO[sp], STEPS, r)), np.float32)
        out[sp + "_spectrum"] = spec
    fsum = np.zeros(6)
    for r in range(nranks):
        raw = np.fromfile(os.path.join(d, "fields", "T.%d" % STEPS, "fields.%d.%d" % (STEPS, r)), np.uint8)
        nxl = NX // nranks
        band = raw[H + 12:].view(np.float32).reshape(6, NZ + 2, NY + 2, nxl + 2).astype(np.float64)
        fsum += (band[:, 1:-1, 1:-1, 1:-1] ** 2).sum(axis=(1, 2, 3))
    out["field_sq_sum"] = fsum                                            # ex ey ez cbx cby cbz, squared and summed
    for name in ("info", "global.vpc", "rundata/species", "rundata/materials"):
        out["file_" + name] = np.fromfile(os.path.join(d, name), np.uint8)
    return out


def read_normals(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:8], np.int64)[0])
    return np.frombuffer(raw[8:8 + 8 * n], np.float64).copy(), np.frombuffer(raw[8 + 8 * n:8 + 9 * n], np.uint8).copy()


def write_normals(path, values, words):
    with open(path, "wb") as f:
        f.write(np.int64(len(values)).tobytes()); f.write(np.ascontiguousarray(values, np.float64).tobytes())
        f.write(np.ascontiguousarray(words, np.uint8).tobytes())


def tiny():
    """The deck at 16 x 4 x 8 cells: the reference run under oracle/normals_shim.c (LD_PRELOAD) records every normal it
    draws; the fixture carries them next to the run's outputs, so that the HIP host can load the very same particles
    (VPIC_HIP_NORMALS) and be held to tight tolerances."""
    ora = os.path.join(ROOT, "oracle")
    shim = os.path.join(ora, "_ref", "normals_shim.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-o", shim, os.path.join(ora, "normals_shim.c"), "-ldl"])
    extra = "-DVPIC_PARTICLE_X=%d -DVPIC_PARTICLE_Y=%d -DVPIC_PARTICLE_Z=%d" % TINY
    out = {}
    for nr in (1, 2):
        subprocess.check_call(["make", "-s", "-C", ora, "trecon", "TOPO=%d" % nr, "NAME=tiny%d" % nr, "EXTRA=" + extra], stdout=subprocess.DEVNULL)
        exe = os.path.join(ora, "_ref", "trecontiny%d.exe" % nr)
        with tempfile.TemporaryDirectory() as d:
            env = dict(os.environ, LD_PRELOAD=shim, VPIC_NORMALS_OUT=os.path.join(d, "normals"))
            cmd = [exe, "-tpp=1"] if nr == 1 else ["/opt/conda/bin/mpiexec", "-n", str(nr), exe, "-tpp=1"]
            subprocess.check_call(cmd, cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            for k, v in summarize(d, nr, TINY).items():
                out["n%d_%s" % (nr, k)] = v
            for r in range(nr):
                v, w = read_normals(os.path.join(d, "normals.%d" % r))
                out["n%d_normals_%d" % (nr, r)], out["n%d_words_%d" % (nr, r)] = v, w
            out["n%d_energies" % nr] = np.loadtxt(os.path.join(d, "rundata", "energies"), comments="%") if os.path.exists(os.path.join(d, "rundata", "energies")) else np.zeros(0)
    dst = os.path.join(ROOT, "tests", "golden", "trecon_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB;", len(out["n1_normals_0"]), "normals on one rank")


def main():
    if "--tiny" in sys.argv:
        return tiny()
    out = {}
    for nr in (1, 2):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "trecon", "TOPO=%d" % nr], stdout=subprocess.DEVNULL)
        exe = os.path.join(ROOT, "oracle", "_ref", "trecon%d.exe" % nr)
        with tempfile.TemporaryDirectory() as d:
            cmd = [exe, "-tpp=1"] if nr == 1 else ["/opt/conda/bin/mpiexec", "-n", str(nr), exe, "-tpp=1"]
            subprocess.check_call(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            for k, v in summarize(d, nr).items():
                out["n%d_%s" % (nr, k)] = v
    dst = os.path.join(ROOT, "tests", "golden", "trecon.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB")
    for sp in SPECIES:
        print(sp, out["n1_%s_np" % sp], out["n2_%s_np" % sp], out["n1_%s_u2" % sp], out["n2_%s_u2" % sp] / out["n1_%s_u2" % sp] - 1)
    print("fields", out["n1_field_sq_sum"], out["n2_field_sq_sum"] / out["n1_field_sq_sum"] - 1)


if __name__ == "__main__":
    main()
