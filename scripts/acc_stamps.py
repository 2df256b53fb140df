"""Diagnostic: timeline of the accumulation workgroups (libbtf built with -DBTF_ACC_STAMPS; bench.py with
BTF_ACC_STAMPS_OUT=file.npz writes the stamps of one W and one V accumulation launch).  100 MHz wall clock."""
import sys
import numpy as np

d = np.load(sys.argv[1])
for name in d.files:
    s = d[name].astype(np.float64)
    side = s[4096:]
    side = side[side[:, 3] > 0]
    s = s[:4096]
    s = s[s[:, 3] > 0]
    if len(side):
        t00 = s[s[:, 3] > 0][:, 0].min() if (s[:, 3] > 0).any() else side[:, 0].min()
        for row in side[:8]:
            print("%s side workgroup: start %.1f  sums ready %.1f  end %.1f us (from the first streaming workgroup's start)"
                  % (name, (row[0] - t00) / 100, (row[1] - t00) / 100 if row[1] > 0 else float("nan"), (row[3] - t00) / 100))
        if side.shape[1] >= 8 and (side[:, 7] > 0).any():       # the fused W launch's owner workgroups (btf_fused.h)
            ow = side[side[:, 7] > 0]
            f = lambda c: "%.1f" % (np.median(ow[:, c] - t00) / 100)
            print("%s owner workgroups: %d; start %s, prepared %s, tile's counter seen %s, chunk sums in %s, end %s us (medians, from the first streaming workgroup's start)"
                  % (name, len(ow), f(0), f(4), f(5), f(6), f(7)))
        o = side[:, 0].min()
        print("%s side workgroups: %d; start %.1f..%.1f us, pass done +%.1f, barrier +%.1f, end +%.1f (medians, from their start); last end %.1f"
              % (name, len(side), 0.0, (side[:, 0].max() - o) / 100, np.median(side[:, 1] - side[:, 0]) / 100,
                 np.median(side[:, 2] - side[:, 0]) / 100, np.median(side[:, 3] - side[:, 0]) / 100, (side[:, 3].max() - o) / 100))
    if not len(s):
        continue
    t0 = s[:, 0].min()
    s = np.where(s > 0, (s - t0) / 100.0, 0.0)          # us (0: stamp not taken)
    n = len(s)
    print("%s accumulation: %d workgroups, span %.1f us" % (name, n, s[:, 3].max()))
    q = lambda a: "min %.1f  p10 %.1f  med %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print("  start           ", q(s[:, 0]))
    if (s[:, 1] > 0).all() and (s[:, 1] < s[:, 2]).all():
        print("  first U block in", q(s[:, 1] - s[:, 0]))
    elif (s[:, 1] > 0).all():
        print("  last wave - wave 0", q(s[:, 1] - s[:, 2]), " (all waves at the first barrier of the epilogue)")
    print("  stream (wave 0) ", q(s[:, 2] - s[:, 0]))
    print("  epilogue        ", q(s[:, 3] - s[:, 2]))
    print("  end             ", q(s[:, 3]))
    tl = s[s[:, 7] > 0] if s.shape[1] >= 8 else s[:0]       # workgroups that ran a fused tail (btf_fused.h): stamps 4..7
    if len(tl):
        print("  fused tails: %d workgroups" % len(tl))
        print("    ticket / entry  ", q(tl[:, 4] - tl[:, 3]))
        print("    phase 4 -> 5    ", q(tl[:, 5] - tl[:, 4]))
        print("    phase 5 -> 6    ", q(tl[:, 6] - tl[:, 5]))
        print("    phase 6 -> 7    ", q(tl[:, 7] - tl[:, 6]))
        print("    tail end        ", q(tl[:, 7]))
    # bandwidth-idle estimate: how many workgroups are streaming at each 1-us tick
    ticks = np.arange(0, s[:, 3].max() + 1, 2.0)
    live = [(int(((s[:, 0] <= t) & (s[:, 2] > t)).sum())) for t in ticks]
    print("  streaming workgroups every 2 us:", live)
