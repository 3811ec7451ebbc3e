#!/usr/bin/env python3
"""Turns the raw barrier stamps of tools/mb_stamps.bin (k_miller_x60, DBG == 4) into the per-role table and histograms of profiles/r6/.

For every sampled block and line step s the three waves record the s_memtime clock at: arrival at barrier A, release from A, arrival at B,
release from B.  From those:
  consumer  fold  = A_arrive[s+1] - B_release[s]   (six or seven line folds)
            waitA = A_release - A_arrive            (the producers' point step is late)
            sqr   = B_arrive - A_release            (squaring + publication; nothing on an addition step)
            waitB = B_release - B_arrive            (the producers are still storing their lines)
  producer  step  = A_arrive[s+1] - B_release[s]   (the G2 point step that makes the next lines)
            waitA = A_release - A_arrive            (the consumer is still folding the previous lines)
            store = B_arrive - A_release
            waitB = B_release - B_arrive            (the consumer is squaring)
usage: stamps_report.py <file.bin> [more.bin ...]
"""
import struct
import sys

import numpy as np


def load(fn):
    raw = open(fn, "rb").read()
    nsamp, dw, nsteps, every = struct.unpack("<4I", raw[:16])
    a = np.frombuffer(raw[16:], dtype=np.uint32).reshape(nsamp, 3, dw)
    return a, nsteps, every


def pct(x, qs=(5, 25, 50, 75, 95)):
    return " ".join("%7.0f" % v for v in np.percentile(x, qs))


def hist_line(x, edges):
    h, _ = np.histogram(x, bins=edges)
    tot = max(1, h.sum())
    return " ".join("%5.1f%%" % (100.0 * v / tot) for v in h)


def report(fn):
    a, nsteps_max, every = load(fn)
    nsamp = a.shape[0]
    t0 = a[:, :, 2].astype(np.uint64) | (a[:, :, 3].astype(np.uint64) << 32)
    t1 = a[:, :, 4].astype(np.uint64) | (a[:, :, 5].astype(np.uint64) << 32)
    ok = (t1 > t0).all(axis=1)
    role = a[:, :, 6]
    steps = a[:, :, 9]
    ns = int(np.median(steps[ok]))
    launch0, launch1 = t0[ok].min(), t1[ok].max()
    dur_us = (launch1 - launch0) / 100.0
    # clock of s_memtime relative to the 100 MHz s_memrealtime
    dm = (a[:, :, 8] - a[:, :, 7]).astype(np.uint32).astype(np.float64)
    dr = (t1 - t0).astype(np.float64)
    mhz = np.median(dm[ok] / dr[ok]) * 100.0
    print("== %s: %d sampled blocks (every %d-th), %d line steps, launch %.2f ms, s_memtime at %.0f MHz" % (fn, nsamp, every, ns, dur_us / 1000.0, mhz))
    # steady part: blocks that start after 10 % and end before 90 % of the launch
    lo = launch0 + np.uint64(0.10 * (launch1 - launch0))
    hi = launch0 + np.uint64(0.90 * (launch1 - launch0))
    steady = ok & (t0.min(axis=1) >= lo) & (t1.max(axis=1) <= hi) & (steps == ns).all(axis=1)
    print("   steady-state blocks used: %d; block lifetime %s us (5/25/50/75/95 %%)" % (steady.sum(), pct(((t1.max(axis=1) - t0.min(axis=1))[steady]) / 100.0)))
    st = a[:, :, 16:16 + 4 * ns].reshape(nsamp, 3, ns, 4)
    d = lambda x, y: (x - y).astype(np.uint32).astype(np.float64)       # wrap-safe difference of the low words
    rows = {}
    for name, rsel in (("consumer", lambda r: r == 2), ("producer", lambda r: r < 2)):
        sel = steady[:, None] & rsel(role)
        s = st[sel]                       # waves x steps x 4
        work = d(s[:, 1:, 0], s[:, :-1, 3])      # B release of step s -> A arrival of step s+1
        waitA = d(s[:, :, 1], s[:, :, 0])
        mid = d(s[:, :, 2], s[:, :, 1])
        waitB = d(s[:, :, 3], s[:, :, 2])
        period = d(s[:, 1:, 3], s[:, :-1, 3])
        rows[name] = (work, waitA, mid, waitB, period)
        tot = period.sum()
        print("   %s waves: %d" % (name, s.shape[0]))
        print("     per line step, clocks (5/25/50/75/95 %%)                               share of the wave's time")
        for label, v, denom in (("work (fold / point step)", work, work.sum()), ("wait at A", waitA[:, 1:], waitA[:, 1:].sum()), ("between A and B", mid[:, 1:], mid[:, 1:].sum()),
                                ("wait at B", waitB[:, 1:], waitB[:, 1:].sum()), ("period B -> B", period, tot)):
            print("     %-26s %s     %5.1f %%" % (label, pct(v.ravel()), 100.0 * denom / tot))
        edges = [0, 250, 1000, 4000, 16000, 64000, 1e9]
        print("     histogram of waits, clocks  [0,250) [250,1k) [1k,4k) [4k,16k) [16k,64k) [64k,..)")
        print("       wait at A                 %s" % hist_line(waitA[:, 1:].ravel(), edges))
        print("       wait at B                 %s" % hist_line(waitB[:, 1:].ravel(), edges))
    # who is last at A?  (consumer's wait at A close to zero: the consumer arrived last)
    cw = rows["consumer"][1][:, 1:].ravel()
    print("   consumer arrives LAST at A in %.1f %% of the steps (its wait there below 250 clocks)" % (100.0 * (cw < 250).mean()))
    per = rows["consumer"][4].ravel()
    print("   mean period %.0f clocks = %.2f us per line step; %d steps x %.2f rounds of resident blocks" % (per.mean(), per.mean() / mhz, ns, 0))


if __name__ == "__main__":
    for fn in sys.argv[1:]:
        report(fn)
