"""RCCL carries this code's messages -- on the ONE GPU a box has.  A deck run on one rank with
VPIC_HIP_HOST_SELF_SEND=1 cuts its periodic axes into faces shared with the rank itself and sends every message of the
multi-domain path -- the per-species particle messages overlapped with the interior push, the later rounds, jf planes,
tangential-B ghosts, the cleaning family's planes -- to ITSELF, which is what the reference does on a periodic rank of its
own (src/grid/grid_comm.c:17-19,49: MPI_Issend to its own rank; mp_dmp.c:241-266).  With the default transport (rccl) that
is ncclSend / ncclRecv inside ncclGroupStart / End on the communication stream of a 1-rank communicator
(old-vpic_amd/csrc/transport.hip), ordered against the engine's stream with events; with VPIC_HIP_HOST_TRANSPORT=mpi the
same choreography with host staging.  Outputs are compared with the REFERENCE's own one-rank run of the same deck files
(tests/golden/deck16.npz, sheet4.npz), to the tolerances of the ordinary one-rank tests.  GPU box only."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, deck, out, defs=""):
    importlib.import_module("old-vpic_amd").lib()
    host = os.path.join(ROOT, "old-vpic_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host, "deck", "DECK=" + os.path.join(ROOT, "oracle", "decks", deck),
                           "OUT=" + str(tmp_path / out)] + (["DECK_DEFS=" + defs] if defs else []))
    return str(tmp_path / (out + ".hip.exe"))


def _run(exe, tmp_path, transport):
    env = dict(os.environ, VPIC_HIP_HOST_SELF_SEND="1", VPIC_HIP_HOST_TRANSPORT=transport)
    r = subprocess.run([exe, "-tpp=1"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stderr


def _check_plumbing16(tmp_path):
    gold = np.load(os.path.join(ROOT, "tests", "golden", "deck16.npz"))
    en = np.loadtxt(tmp_path / "energies16.txt")
    ref = gold["energies_1rank"]
    assert en.shape[0] == ref.shape[0] == 51
    np.testing.assert_allclose(en[:, 7], ref[:, 6], rtol=2e-7)           # kinetic energy
    np.testing.assert_allclose(en[1:, 1:7], ref[1:, :6], rtol=5e-4)      # field energies
    sys.path.insert(0, ROOT)
    from oracle import deck16
    _, f50, p50 = deck16.read_state(tmp_path / "state16_step50_rank0.bin")
    for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
        scale = max(np.abs(gold["f50_" + k]).max() for k in (("ex", "ey", "ez") if c[0] == "e" else ("cbx", "cby", "cbz")))
        assert np.abs(f50[c] - gold["f50_" + c]).max() <= 2e-4 * scale, c
    assert len(p50) == 16 * 16 * 16 * 8
    assert np.abs(np.bincount(p50["i"], minlength=len(f50)) - gold["p50_cell_count"]).sum() <= 4


@pytest.mark.parametrize("transport", ["rccl", "mpi"])
def test_plumbing_deck_sends_to_itself(tmp_path, transport):
    exe = _build(tmp_path, "plumbing16.cxx", "plumbing16s")
    err = _run(exe, tmp_path, transport)
    assert ("transport: " + transport) in err and "sending to itself" in err and "device-resident, overlapped with the push" in err, err[-2000:]
    m = re.search(r"(\d+) exchanges posted \((\d+) messages, ([0-9.]+) MB over RCCL\)", err)
    assert m, err[-2000:]
    if transport == "rccl":                                  # RCCL moved this code's messages: 50 steps x (particle rounds + jf + tang-B) x 6 faces
        assert int(m.group(2)) > 50 * 6 * 4 and float(m.group(3)) > 1.0
    else:
        assert int(m.group(2)) == 0
    _check_plumbing16(tmp_path)


def test_plumbing_deck_with_cleaning_sends_to_itself_over_rccl(tmp_path):
    """... and the divergence-cleaning family's plane messages (rho, normal E, div-B error, tangential E / normal B)."""
    exe = _build(tmp_path, "plumbing16.cxx", "plumbing16sc", "-DCLEAN_INTERVAL=10")
    err = _run(exe, tmp_path, "rccl")
    assert "transport: rccl" in err
    gold = np.load(os.path.join(ROOT, "tests", "golden", "deck16.npz"))
    en = np.loadtxt(tmp_path / "energies16.txt")
    ref = gold["clean_energies_1rank"]
    np.testing.assert_allclose(en[:, 7], ref[:, 6], rtol=1e-5)
    np.testing.assert_allclose(en[1:, 1:7], ref[1:, :6], rtol=5e-3)


def test_sheet_deck_sends_to_itself_over_rccl(tmp_path):
    """oracle/decks/sheet4.cxx: periodic x and y (cut into self-shared faces), conducting reflecting z walls, 4 species + 2
    tracer species the DECK advances through advance_p / boundary_p (served by the all-species rounds), cleaning."""
    from test_gpu_deck_host import _sheet4_check
    exe = _build(tmp_path, "sheet4.cxx", "sheet4s")
    err = _run(exe, tmp_path, "rccl")
    assert "transport: rccl" in err and "sending to itself" in err, err[-2000:]
    _sheet4_check(tmp_path, np.load(os.path.join(ROOT, "tests", "golden", "sheet4.npz")), "n1_", 1, migrating=True)


def test_python_driver_sends_to_itself_over_rccl(L):
    """The multi-GPU driver of bench.py (old-vpic_amd/domain.py) on ONE domain whose x and y axes are cut into faces shared with
    itself (deck key self_send): its overlapped particle exchange, jf planes and tangential-B ghosts travel through the SAME
    transport the C++ deck host uses (vpic_hip_comm_*: a 1-rank RCCL communicator) -- against vpic_hip_step on the same deck
    without any message.  Energies step by step, particle counts, and what RCCL carried."""
    import ctypes as C
    V = importlib.import_module("old-vpic_amd")
    domain = importlib.import_module("old-vpic_amd.domain")
    nx, ny, nz, ppc = 32, 16, 16, 16
    dt = np.float32(0.95 / np.sqrt(3.0))
    q = -float((0.2 / float(dt)) ** 2 / (2 * ppc))
    deck = dict(gx=nx, gy=ny, gz=nz, ppc=ppc, dt=dt, q=q, drift=0.2, vth=0.05, sort_interval=5, species=[(0.2, 0.0, 0.0), (-0.2, 0.0, 0.0)],
                self_send=[0, 1])
    dom = domain.SlabDomain(deck, 0, 1)
    assert dom.vcomm is not None and dom.transport.startswith("rccl (vpic_hip_comm"), dom.transport
    assert dom.axes == [0, 1] and len(dom.dirs) == 4
    e = V.Engine(V.make_grid(nx, ny, nz, float(nx), float(ny), float(nz), dt))
    e.set_vacuum()
    e.set_sort_order("engine")
    sps = []
    for k, u in enumerate(deck["species"]):
        sp = e.new_species(-1.0, 2 * nx * ny * nz * ppc, nx * ny * nz * ppc)
        e.load_maxwellian(sp, ppc, 1 + k, q, u, deck["vth"])               # (the seeds SlabDomain gives rank 0)
        sps.append(sp)
    e.load_interpolator()
    for step in range(12):
        dom.step(step)
        e.step(step, 5)
        a = np.concatenate([dom.engine.energy_f(), [dom.engine.energy_p(sp) for sp in dom.species]])
        b = np.concatenate([e.energy_f(), [e.energy_p(sp) for sp in sps]])
        np.testing.assert_allclose(a[6:], b[6:], rtol=2e-6)                                   # kinetic energy of either beam
        np.testing.assert_allclose(a[:6], b[:6], rtol=2e-4, atol=1e-7 * abs(b[6:]).max())     # field energies
        assert [dom.engine.np(sp) for sp in dom.species] == [e.np(sp) for sp in sps]
    msgs, nbytes = C.c_int64(), C.c_int64()
    assert V.lib().vpic_hip_comm_stats(dom.vcomm, C.byref(msgs), C.byref(nbytes)) == 0
    # per step: 2 species' particle messages + 2 later rounds + jf x 2 axes + tang-B, over 4 (or 2) faces each
    assert msgs.value >= 12 * 4 * 6 and nbytes.value > 1e6, (msgs.value, nbytes.value)
    assert dom.host_syncs_per_step() <= 2.0 + 1e-9
    e.close()
    dom.engine.close()
