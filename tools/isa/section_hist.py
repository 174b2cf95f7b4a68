#!/usr/bin/env python3
"""Per-section instruction histogram of an ISA listing made with -DDIAL_ISA_MARKS (tools/isa/probe.sh <Dims> ... -DDIAL_ISA_MARKS).

The listing is walked in TEXT order: the instructions between mark a and the next mark b are booked to section a (the DIAL_MARK ids
of csrc/*.h: a mark closes the section it names in the cycle profiles, so the text after `mark a` belongs to the section that the NEXT
mark names -- both are printed).  Loops and branches are not unrolled: a count is static; the Go2's body is straight-line per Newton /
line-search iteration, so `static x trips` is the dynamic count.

usage: section_hist.py build/isa/DimsGo2_1_3_false.s [--classes]
"""
import collections
import re
import sys

CLASSES = [
    ("fma3", r"^v_fma_f32"), ("fmac", r"^v_fmac_f32(?!_dpp)"), ("fmac_dpp", r"^v_fmac_f32_dpp"), ("pk", r"^v_pk_"),
    ("mul", r"^v_mul_f32"), ("addsub", r"^v_(add|sub|subrev)_f32"), ("minmax", r"^v_(min|max|med3)_"), ("trans", r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_"),
    ("cndmask", r"^v_cndmask"), ("cmp", r"^v_cmp"), ("mov", r"^v_mov_b32(?!_dpp)|^v_accvgpr"), ("dpp", r"_dpp"), ("readlane", r"^v_read(first)?lane"),
    ("writelane", r"^v_writelane"), ("permlane", r"^v_permlane"), ("int", r"^v_(add|sub|lshl|lshr|and|or|xor|mul_lo|mul_hi|mad|bfe|ashr|lshlrev|lshrrev|not|bfi|cvt|mul_u32|add3|lshl_add|add_lshl|and_or|or3|xad|mbcnt|ldexp|frexp|fract|floor|rndne|trunc|ceil|alignbit|perm|sad|min_u|max_u|min_i|max_i|subrev_u|subrev_co|sub_co|add_co|addc|subb)"),
    ("valu_other", r"^v_"), ("s_nop", r"^s_nop"), ("s_waitcnt", r"^s_waitcnt"), ("s_branch", r"^s_(c?branch|setpc|call)"),
    ("salu", r"^s_"), ("ds", r"^ds_"), ("vmem", r"^(global|buffer|flat|scratch)_"),
]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other"


def main():
    path = sys.argv[1]
    sec = "entry"
    order = [sec]
    hist = collections.defaultdict(collections.Counter)
    for line in open(path):
        t = line.strip()
        m = re.match(r"; DIAL_MARK (\d+)", t)
        if m:
            sec = f"after mark {m.group(1)}"
            # several code regions may follow marks with the same id (the solver's loop): keep them apart
            k = 2
            base = sec
            while sec in hist:
                sec = f"{base} #{k}"
                k += 1
            order.append(sec)
            hist[sec]
            continue
        if not t or t.startswith((";", ".", "#")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
            continue
        hist[sec][classify(op)] += 1
    names = [c for c, _ in CLASSES]
    valu = set(names[:names.index("s_nop")])
    print(f"{'section':24s} {'total':>6s} {'VALU':>6s} " + " ".join(f"{n[:8]:>8s}" for n in names))
    tot = collections.Counter()
    for s in order:
        h = hist[s]
        n = sum(h.values())
        if n == 0:
            continue
        tot.update(h)
        print(f"{s:24s} {n:6d} {sum(v for k, v in h.items() if k in valu):6d} " + " ".join(f"{h.get(c, 0):8d}" for c in names))
    print(f"{'TOTAL':24s} {sum(tot.values()):6d} {sum(v for k, v in tot.items() if k in valu):6d} " + " ".join(f"{tot.get(c, 0):8d}" for c in names))


if __name__ == "__main__":
    main()
