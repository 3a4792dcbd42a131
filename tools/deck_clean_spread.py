"""Run the cleaning-enabled plumbing16 deck on the HIP host several times and print how far each
compared quantity lies from the reference's (float-atomic summation order varies from run to run)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import deck16
gold = np.load(os.path.join(ROOT, "tests", "golden", "deck16.npz"))
with tempfile.TemporaryDirectory() as d:
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "old-vpic_amd", "host"), "deck", "DECK=" + os.path.join(ROOT, "oracle", "decks", "plumbing16.cxx"),
                           "DECK_DEFS=-DCLEAN_INTERVAL=10", "OUT=" + os.path.join(d, "c")])
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        subprocess.check_call([os.path.join(d, "c.hip.exe"), "-tpp=1"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        en = np.loadtxt(os.path.join(d, "energies16.txt")); ref = gold["clean_energies_1rank"]
        ke = np.abs(en[:, 7] / ref[:, 6] - 1).max(); fe = np.abs(en[1:, 1:7] / ref[1:, :6] - 1).max()
        _, f0, _ = deck16.read_state(os.path.join(d, "state16_step0_rank0.bin"))
        _, f50, _ = deck16.read_state(os.path.join(d, "state16_step50_rank0.bin"))
        r0 = gold["clean_f0_rhob"]
        out = dict(ke=ke, fe=fe, rhob0=np.abs(f0["rhob"] - r0).max() / np.abs(r0).max(),
                   rhob50=np.abs(f50["rhob"] - gold["clean_f50_rhob"]).max() / np.abs(r0).max(),
                   rhof50=np.abs(f50["rhof"] - gold["clean_f50_rhof"]).max() / np.abs(gold["clean_f50_rhof"]).max(),
                   dive=np.abs(f50["div_e_err"]).max(), divb=np.abs(f50["div_b_err"]).max())
        for c in ("ex", "ey", "ez", "cbx", "cby", "cbz"):
            scale = max(np.abs(gold["clean_f50_" + k]).max() for k in (("ex", "ey", "ez") if c[0] == "e" else ("cbx", "cby", "cbz")))
            out[c] = np.abs(f50[c] - gold["clean_f50_" + c]).max() / scale
        print(" ".join("%s=%.2e" % kv for kv in out.items()))
