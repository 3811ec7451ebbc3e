#!/usr/bin/env python3
"""LDS bank-conflict model of k_miller_x60's consumer wave (bgls_amd/csrc/miller_x.hpp), used to choose the group layout.

Model (MI355X_MICROARCH.md, LDS): ds_read_b128 is serviced in four fixed 16-lane groups, bank = (addr / 4) mod 64, each lane
covers four banks; ds_read_b64 in two 32-lane groups, two banks per lane.  Identical addresses broadcast; every further
distinct address on a busy bank of a group costs one more LDS cycle.  Reports the conflict ratio extra / (base + extra) of one
Miller step of the consumer (squaring + six line folds), which is what SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measures.

usage: lds_conflicts.py [search]
"""
import itertools
import sys

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
               list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
B64_GROUPS = [list(range(0, 32)), list(range(32, 64))]
SQ_TAB = [0x5be2e900, 0xffe3ea88, 0x64eb0990, 0xffec9198, 0x6d1299a0, 0xff9aa1a8]


def cycles(addrs, width):
    """addrs: dword address per lane (64); width 4 (b128) or 2 (b64) dwords.  Returns (base cycles, extra cycles)."""
    groups = B128_GROUPS if width == 4 else B64_GROUPS
    base = extra = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = addrs[l]
            for k in range(width):
                per_bank.setdefault((a + k) % 64, set()).add(a)
        base += 1
        extra += max(len(v) for v in per_bank.values()) - 1
    return base, extra


def half_reads(nl, off):
    """(dword offset, width) of the reads that fetch NL limbs at dword offset `off` (16-byte aligned base)."""
    out = []
    o, left = off, nl
    if o % 4 == 2:
        out.append((o, 2)); o += 2; left -= 2
    while left >= 4:
        out.append((o, 4)); o += 4; left -= 4
    if left:
        out.append((o, 2))
    return out


class Layout:
    def __init__(self, nl, es, hs, group_dw, twist_d, swz=None):
        self.nl, self.es, self.hs, self.group_dw, self.twist_d, self.swz = nl, es, hs, group_dw, twist_d, swz

    def entry(self, g, e, h):          # dword address of half h of accumulator entry e of group g
        return g * self.group_dw + e * self.es + h * self.hs

    def line(self, g, e, h):
        return g * self.group_dw + 12 * self.es + e * self.es + h * self.hs


MAPPING = "gmajor"          # "gmajor": lane = 6 g + j (round 3);  "jmajor": lane = 10 j + g


def lanes():
    for lane in range(64):
        live = lane < 60
        if MAPPING == "gmajor":
            yield lane, (lane // 6 if live else 9), (lane % 6 if live else lane - 60)
        else:
            l = lane if live else lane - 12          # lanes 60..63 repeat lanes 48..51 (same 16-lane group of a b128 access)
            yield lane, l % 10, l // 10


def consumer_step(lay):
    base = extra = 0

    def issue(fn):
        nonlocal base, extra
        for h in (0, 1):
            for off, w in half_reads(lay.nl, 0):
                addrs = [0] * 64
                for lane, g, j in lanes():
                    addrs[lane] = fn(g, j, h) + off
                # offsets are relative to the half's own start; alignment of the half decides the split
                b, e = cycles(addrs, w)
                base += b; extra += e

    def issue_aligned(fn):
        # the half's start alignment may differ between halves (packed entries): recompute the split per half
        nonlocal base, extra
        for h in (0, 1):
            start = fn(0, 0, h) % 4
            for off, w in half_reads(lay.nl, start):
                addrs = [0] * 64
                for lane, g, j in lanes():
                    addrs[lane] = fn(g, j, h) - start + off
                b, e = cycles(addrs, w)
                base += b; extra += e

    sh = (lambda t: t + (1 if t == 2 else 0)) if lay.twist_d else (lambda t: t + (1 if t >= 1 else 0))
    # six folds: line operand (broadcast within the group) + accumulator operand
    for m in range(6):
        for t in range(3):
            issue_aligned(lambda g, j, h: lay.line(g, 3 * m + t, h))

            def acc(g, j, h, t=t):
                k = j - sh(t)
                wrap = 1 if k < 0 else 0
                k += 6 * wrap
                return lay.entry(g, 2 * k + wrap, h)
            issue_aligned(acc)
    # squaring: four table slots per lane
    for t in range(4):
        def a_op(g, j, h, t=t):
            e = (SQ_TAB[j] >> (8 * t)) & 0xFF
            i = 0 if (e & 7) == 7 else e & 7
            return lay.entry(g, 2 * i, h)

        def b_op(g, j, h, t=t):
            e = (SQ_TAB[j] >> (8 * t)) & 0xFF
            k2 = 0 if (e & 7) == 7 else 2 * ((e >> 3) & 7) + ((e >> 6) & 1)
            return lay.entry(g, k2, h)
        issue_aligned(a_op)
        issue_aligned(b_op)
    return base, extra


def report(name, lay):
    b, e = consumer_step(lay)
    print("%-44s ES %2d HS %2d GROUP_DW %4d  LDS cycles/step base %4d extra %4d  conflict ratio %.3f  block %6d B" %
          (name, lay.es, lay.hs, lay.group_dw, b, e, e / (b + e), 10 * lay.group_dw * 4))
    return e / (b + e)


if __name__ == "__main__":
    for nl, twist_d, cname in ((10, True, "alt-bn128"), (14, False, "BLS12-381")):
        hs = (nl + 3) & ~3
        report("%s round 3 (padded halves, +4)" % cname, Layout(nl, 2 * hs, hs, 30 * 2 * hs + 4, twist_d))
        if len(sys.argv) > 1 and sys.argv[1] == "search":
            for mapping in ("gmajor", "jmajor"):
                MAPPING = mapping
                best = []
                for es, hs2 in ((2 * nl, nl), (2 * hs, hs), (2 * hs + 4, hs), (2 * nl + 4, nl), (2 * nl + 2, nl), (2 * nl + 8, nl)):
                    for pad in range(0, 68, 4):
                        lay = Layout(nl, es, hs2, 30 * es + pad, twist_d)
                        b, e = consumer_step(lay)
                        best.append((e / (b + e), es, hs2, pad, 10 * lay.group_dw * 4))
                best.sort()
                print("  lane mapping", mapping)
                for r in best[:8]:
                    print("   ratio %.3f  ES %d HS %d pad %d  block %d B" % r)
            MAPPING = "gmajor"
        else:
            report("%s packed entries" % cname, Layout(nl, 2 * nl, nl, 30 * 2 * nl, twist_d))
