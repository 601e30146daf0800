#!/usr/bin/env python3
"""Round 5: where and when the plane sweep's workgroups run (a -DMD_CL_WGSTATS=1 -DMD_CL_WGPLACE=1 build: tools/ab_build.sh place ...;
records dumped by tools/bench_costvol.py with MD_CV_WGSTATS_DUMP).  usage: place.py <dump>_fwd.npy"""
import sys
import numpy as np

for f in sys.argv[1:]:
    a = np.load(f)
    life = a[:, 0]
    t0 = a[:, 2] - a[:, 2].min()
    t1 = a[:, 3] - a[:, 2].min()
    t1 = np.where(t1 < 0, t1 + (1 << 30), t1)
    hw = a[:, 6].astype(np.int64)
    xcc = a[:, 7].astype(np.int64) & 15
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    simd = (hw >> 4) & 3
    print("== %s: %d workgroups; wall clock in 10 ns ticks" % (f, len(a)))
    print("  start  : min 0, median %d, p90 %d, max %d ticks" % (np.median(t0), np.percentile(t0, 90), t0.max()))
    print("  end    : min %d, median %d, p90 %d, max %d ticks" % (t1.min(), np.median(t1), np.percentile(t1, 90), t1.max()))
    print("  length : mean %.0f, median %d, p90 %d, max %d ticks" % ((t1 - t0).mean(), np.median(t1 - t0), np.percentile(t1 - t0, 90), (t1 - t0).max()))
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print("  distinct (xcc, se, sh, cu): %d; workgroups per CU: %s" % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
    per = {k: c for k, c in zip(uniq, cnt)}
    n_on_cu = np.array([per[k] for k in key])
    for n in sorted(set(n_on_cu)):
        m = n_on_cu == n
        print("  workgroups on a CU with %d of them: %4d, length mean %.0f ticks, end mean %.0f, max %d" % (n, m.sum(), (t1 - t0)[m].mean(), t1[m].mean(), t1[m].max()))
    print("  by xcc : " + " ".join("%d:%d/%.0f" % (x, (xcc == x).sum(), (t1 - t0)[xcc == x].mean()) for x in sorted(set(xcc))))
    print("  by se  : " + " ".join("%d:%d/%.0f" % (x, (se == x).sum(), (t1 - t0)[se == x].mean()) for x in sorted(set(se))))
    print("  by cu  : " + " ".join("%d:%d/%.0f" % (x, (cu == x).sum(), (t1 - t0)[cu == x].mean()) for x in sorted(set(cu))))
    order = np.argsort(t0)
    print("  start order vs blockIdx: corr %.2f;  length vs start: corr %.2f" % (np.corrcoef(np.arange(len(a)), t0)[0, 1], np.corrcoef(t0, t1 - t0)[0, 1]))
    late = t1 >= np.percentile(t1, 95)
    print("  the last 5 %% to end: start mean %.0f, length mean %.0f, on CUs with %s workgroups, xcc %s" % (
        t0[late].mean(), (t1 - t0)[late].mean(), sorted(set(n_on_cu[late])), sorted(set(xcc[late]))))
