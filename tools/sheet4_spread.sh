#!/bin/bash
# run-to-run spread of the 2-rank sheet4 deck's final fields against the reference's (why the tolerances are what they are)
cd "$(dirname "$0")/.."
for k in 1 2 3 4 5; do
  tools/run_sheet4.sh 2 > /dev/null 2>&1
  python - <<'PY'
import numpy as np, sys, importlib
sys.path.insert(0, '.')
from oracle import sheet4 as S
g = np.load('tests/golden/sheet4.npz')
out = []
for r in range(2):
    f, _ = S.read_fields('gpurun_out/sheet4_n2/fields4_rank%d.bin' % r, 16)
    b = max(np.abs(g['n2_r%d_f_%s' % (r, c)]).max() for c in ('cbx', 'cby', 'cbz'))
    out += ['%s%d %.1e' % (c, r, np.abs(f[c] - g['n2_r%d_f_%s' % (r, c)]).max() / b) for c in ('ex', 'ey', 'ez', 'cbx', 'cby', 'cbz')]
print(' '.join(out))
PY
done
