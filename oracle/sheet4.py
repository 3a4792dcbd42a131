"""TEST INFRASTRUCTURE: golden outputs of THE REFERENCE for oracle/decks/sheet4.cxx (a small
force-free current sheet with tracer species advanced from the deck, conducting z walls, cleaning,
strided dumps) on one and on two ranks -> tests/golden/sheet4.npz.  Needs /root/reference and MPI;
run here, not on the GPU box:   python oracle/sheet4.py"""
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("old-vpic_amd.layout")
NX, NY, NZ, STEPS, DUMP_STEP = 32, 8, 16, 40, 20
SPECIES = ("iB", "iT", "eB", "eT")          # species_list order (newest first) = columns 7.. of energies4.txt


def read_tracers(path):
    """[(np, particles)] for eR... in list order: iR first (defined last)."""
    out, off = [], 0
    for _ in range(2):
        n = int(np.fromfile(path, np.int32, 1, offset=off)[0])
        out.append(np.fromfile(path, L.particle_t, n, offset=off + 4))
        off += 4 + n * L.particle_t.itemsize
    return out


def read_fields(path, nx):
    nv = (nx + 2) * (NY + 2) * (NZ + 2)
    f = np.fromfile(path, L.field_t, nv)
    counts = np.fromfile(path, np.int32, 4, offset=nv * L.field_t.itemsize)
    return f, counts


def run(nranks, workdir):
    exe = os.path.join(ROOT, "oracle", "_ref", "sheet4.exe")
    cmd = [exe, "-tpp=1"] if nranks == 1 else ["/opt/conda/bin/mpiexec", "-n", str(nranks), exe, "-tpp=1"]
    subprocess.check_call(cmd, cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)


def collect(out, d, nranks, key):
    out[key + "energies"] = np.loadtxt(os.path.join(d, "energies4.txt"))
    for name in ("global.vpc", "rundata/species", "rundata/materials", "rundata/energies"):
        out[key + name] = np.fromfile(os.path.join(d, name), np.uint8)
    for r in range(nranks):
        k = key + "r%d_" % r
        out[k + "grid"] = np.fromfile(os.path.join(d, "rundata/grid.%d" % r), np.uint8)
        out[k + "field_dump"] = np.fromfile(os.path.join(d, "fields/T.%d/fields.%d.%d" % (DUMP_STEP, DUMP_STEP, r)), np.uint8)
        out[k + "hydro_dump"] = np.fromfile(os.path.join(d, "hydro/T.%d/eThydro.%d.%d" % (DUMP_STEP, DUMP_STEP, r)), np.uint8)
        f, counts = read_fields(os.path.join(d, "fields4_rank%d.bin" % r), NX // nranks)
        for c in ("ex", "ey", "ez", "cbx", "cby", "cbz", "rhob", "rhof"):
            out[k + "f_" + c] = f[c]
        out[k + "np"] = counts
        for name, t in zip(("iR", "eR"), read_tracers(os.path.join(d, "tracers4_rank%d.bin" % r))):
            out[k + "tracers_" + name] = t[np.argsort(t["tag"])]


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "deck",
                           "DECK=" + os.path.join(ROOT, "oracle", "decks", "sheet4.cxx"), "OUT=sheet4"])
    out = {}
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        run(1, d1)
        run(2, d2)
        collect(out, d1, 1, "n1_")
        collect(out, d2, 2, "n2_")
    dst = os.path.join(ROOT, "tests", "golden", "sheet4.npz")
    np.savez_compressed(dst, **out)
    e1, e2 = out["n1_energies"], out["n2_energies"]
    print("wrote", dst, os.path.getsize(dst) // 1024, "KiB;", len(out["n1_r0_tracers_eR"]), "electron tracers,",
          len(out["n1_r0_tracers_iR"]), "ion tracers")
    print("1 vs 2 ranks (different particle loads), relative energy difference at the end:",
          np.abs(e1[-1, 1:] - e2[-1, 1:]) / np.abs(e1[-1, 1:]))


if __name__ == "__main__":
    main()
