"""Find intermittent deviations: run the plumbing16 deck (with / without cleaning) N times on the HIP
host and report, per run, the first step whose energies leave a tight band around the reference's."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gold = np.load(os.path.join(ROOT, "tests", "golden", "deck16.npz"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for clean in (0, 10):
    ref = gold["clean_energies_1rank" if clean else "energies_1rank"]
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "old-vpic_amd", "host"), "deck", "DECK=" + os.path.join(ROOT, "oracle", "decks", "plumbing16.cxx"),
                               "DECK_DEFS=-DCLEAN_INTERVAL=%d" % clean, "OUT=" + os.path.join(d, "c")])
        for rep in range(N):
            subprocess.check_call([os.path.join(d, "c.hip.exe"), "-tpp=1"], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            en = np.loadtxt(os.path.join(d, "energies16.txt"))
            ke = np.abs(en[:, 7] / ref[:, 6] - 1)
            fe = np.abs(en[1:, 1:7] / ref[1:, :6] - 1).max(axis=1)
            bad_k = np.nonzero(ke > 2e-8)[0]; bad_f = np.nonzero(fe > 5e-6)[0] + 1
            print("clean=%d run %d: ke max %.1e (first step > 2e-8: %s)  fe max %.1e (first step > 5e-6: %s)" %
                  (clean, rep, ke.max(), bad_k[0] if len(bad_k) else None, fe.max(), bad_f[0] if len(bad_f) else None))
