"""TEST INFRASTRUCTURE (oracle): numpy restatement of the payload layouts of the reference's
field_dump / hydro_dump (src/vpic/dump.cxx:1116-1364, 1366-1552) and of the V0 headers
(src/vpic/dumpmacros.h:10-48).  Pinned by oracle/deck16.py: the files the reference executable writes
for the -DWRITE_DUMPS deck equal gather() of the raw dump_fields / dump_hydro files of the same step."""
import numpy as np

HEADER_V0 = 5 + 2 + 4 + 4 + 8 + 4 * 2 + 4 * 4 + 4 * 10 + 4 * 2 + 4 * 2     # bytes of WRITE_HEADER_V0
BAND, INTERLEAVE, INTERLEAVE_INNER = 0, 1, 2


def offsets(n, s, unit, inner=False):
    """Source index of every output entry of one axis.  `unit`: all three strides are 1 (the
    reference's fast branch, plain indices); otherwise i*s-1 even on an axis whose own stride is 1
    (dump.cxx:1262-1273; :1523-1533 for `inner`, which has no far boundary entry)."""
    no = n // s
    if inner:
        return np.array([0 if i == 0 else i * s - 1 for i in range(no)])
    return np.array([0 if i == 0 else n + 1 if i == no + 1 else (i if unit else i * s - 1) for i in range(no + 2)])


def gather(records, nx, ny, nz, layout, words=(), strides=(1, 1, 1)):
    """records: structured array of nv field_t / hydro_t.  Returns uint32 words shaped like the file
    payload: band [len(words), Z, Y, X]; interleaved [Z, Y, X, W]."""
    W = records.dtype.itemsize // 4
    nv = (nx + 2) * (ny + 2) * (nz + 2)
    flat = np.concatenate([records.view(np.uint32).reshape(nv * W), np.zeros(4, np.uint32)])   # words 20-23 of the last record
    sy, sz = nx + 2, (nx + 2) * (ny + 2)
    if layout == INTERLEAVE_INNER and tuple(strides) == (1, 1, 1):
        return flat[:nx * ny * nz * W].reshape(nz, ny, nx, W).copy()                     # dump.cxx:1518-1519
    inner = layout == INTERLEAVE_INNER
    unit = tuple(strides) == (1, 1, 1)
    ox, oy, oz = (offsets(n, s, unit, inner) for n, s in zip((nx, ny, nz), strides))
    v = ox[None, None, :] + sy * oy[None, :, None] + sz * oz[:, None, None]
    if layout == BAND:
        return np.stack([flat[v * W + w] for w in words])                                # `fref[varlist[v]]`, dump.cxx:1200-1203
    return flat[(v * W)[..., None] + np.arange(W)]
