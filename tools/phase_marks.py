#!/usr/bin/env python3
"""Are the two workgroups of a CU in step?  Reads the raw stamp dump of a -DHWER_DIAG_STAMPS build (HWER_SCREEN_STAMPS=1
HWER_SCREEN_STAMPS_RAW=<file>): per wave the hardware id and the 100-MHz time of the MFMA-phase start of tiles 64 and 66 and of the
MFMA-phase end of tile 64; prints, per SIMD that held two waves, the phase difference of their tile loops as a fraction of the period.
    python tools/phase_marks.py RAW.bin"""
import sys
from collections import defaultdict

import numpy as np


def main():
    raw = open(sys.argv[1], "rb").read()
    g = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
    h = np.frombuffer(raw[4:], dtype=np.uint64)
    marks = h[g * 32:].reshape(g * 4, 4)
    ids = marks[:, 0]
    hw = (ids & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((ids >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    wave_id, simd, cu, sh, se, tg = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 15
    m0, n0, m2 = (marks[:, i].astype(np.int64) for i in (1, 2, 3))
    ok = m0 > 0
    period = (m2 - m0)[ok] / 2.0
    print("waves %d (with marks %d); period of a tile: median %.0f ticks of 10 ns (%.0f .. %.0f); MFMA phase of tile 64: median %.0f ticks"
          % (len(hw), ok.sum(), np.median(period), period.min(), period.max(), np.median((n0 - m0)[ok])))
    print("wave slot ids seen:", sorted(set(wave_id.tolist())), " threadgroup ids:", sorted(set(tg.tolist())))
    groups = defaultdict(list)
    for i in np.nonzero(ok)[0]:
        groups[(xcc[i], se[i], sh[i], cu[i], simd[i])].append(i)
    sizes = defaultdict(int)
    diffs = []
    for key, idx in groups.items():
        sizes[len(idx)] += 1
        if len(idx) == 2:
            a, b = idx
            p = 0.5 * ((m2[a] - m0[a]) / 2.0 + (m2[b] - m0[b]) / 2.0)
            d = ((m0[a] - m0[b]) % p) / p
            diffs.append(min(d, 1.0 - d))
    print("SIMDs by resident waves:", dict(sizes))
    if diffs:
        d = np.asarray(diffs)
        print("phase difference of the two waves of a SIMD, fraction of a period (0 = in step, 0.5 = alternating): median %.3f; histogram over [0, 0.5] in tenths: %s"
              % (np.median(d), np.histogram(d, bins=5, range=(0, 0.5))[0].tolist()))


if __name__ == "__main__":
    main()
