#!/usr/bin/env python3
"""Static check of an ISA listing (tools/isa/probe.sh output) for the gfx9 DPP hazard the hand-written instructions of
csrc/wave.h (WaveH::fma_pick / rcp_pick: v_fmac_f32_dpp, v_rcp_f32_dpp from inline asm) are exposed to: a VALU instruction that
writes a VGPR must be followed by >= 2 wait states before a DPP instruction reads that VGPR as its DPP source.  The compiler's hazard
recogniser does not look inside inline asm, so the listing is checked instead: for every *_dpp instruction the two preceding issue
slots (s_nop N counts N + 1) must not hold a VALU write of its src0.  Exit status 1 and a report on any violation.
Reads compiler listings (probe.sh: `-S`) and disassemblies of the shipped library (tools/isa/disasm_lib.py).  Labels are treated
CONSERVATIVELY: control flow may join at a label from a block whose tail is not the text above it, so a DPP instruction within the
first two wait states after a label is reported unless its source was written -- and the hazard thereby covered -- inside the block.
usage: check_dpp_hazards.py file.s [kernel-name-substring]"""
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
REG = re.compile(r"v(\d+)|v\[(\d+):(\d+)\]")


def regs(tok):
    m = REG.fullmatch(tok.strip().lstrip("-|").rstrip("|"))
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


HAND_WRITTEN = ("v_fmac_f32_dpp", "v_rcp_f32_dpp", "v_mul_f32_dpp")   # csrc/wave.h: fma_pick / fnma_pick / rcp_pick / mul_pick (inline asm)
cur, bad, checked = None, [], 0
window = []   # (wait states this slot provides, set of VGPRs written by a VALU instruction in it, text)
UNKNOWN = (0, None, "<unknown predecessor at a label>")   # written-set None = "may have written anything"
for ln, line in enumerate(open(path), 1):
    m = re.match(r"^(_Z\w+|[A-Za-z]\w*_kernel\w*):", line)
    if m:
        cur, window = m.group(1), []
        continue
    if cur is None or want not in cur:
        continue
    t = line.split(";")[0].split("//")[0].strip()
    if not t or (t.startswith(".") and not t.endswith(":")):
        continue
    if t.endswith(":"):
        window = [UNKNOWN]       # a label: the other predecessor's tail is unknown -- anything may have been written just before
        continue
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")] if rest else []
    if op.endswith("_dpp"):
        checked += 1
        src0 = regs(ops[1].split()[0]) if len(ops) > 1 else set()
        need, k = 2, len(window) - 1
        while need > 0 and k >= 0:
            ws, written, text = window[k]
            # (the unknown tail of another predecessor matters for the HAND-WRITTEN forms only: what the compiler emits itself --
            #  v_mov_b32_dpp, v_add_f32_dpp from builtins -- went through its hazard recogniser with every predecessor in sight)
            if (written is None and op in HAND_WRITTEN) or (written is not None and (written & src0)):
                bad.append((cur, ln, t, text))
                break
            need -= ws
            k -= 1
    if op == "s_nop":
        window.append((int(ops[0]) + 1 if ops else 1, set(), t))
    elif op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
        w = regs(ops[0].split()[0]) if ops else set()
        if op.startswith("v_permlane") and len(ops) > 1:
            w |= regs(ops[1].split()[0])        # the swap writes both operands
        window.append((1, w, t))
    else:
        window.append((1, set(), t))
    window = window[-6:]
print(f"{path}: {checked} DPP instructions checked, {len(bad)} with a source written inside the two preceding wait states")
for k, ln, t, prev in bad[:20]:
    print(f"  line {ln}: `{t}`  <-  `{prev}`")
sys.exit(1 if bad else 0)
