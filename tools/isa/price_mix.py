#!/usr/bin/env python3
"""Price a rollout kernel's VALU instruction stream with the MEASURED issue costs of tools/ubench/issue.hip.

VERDICT r5 item 2: the issue ceiling `SQ_INSTS_VALU x cycles / (SIMDs x time x clock)` was priced at a hand-composed 16-instruction
"mix" row.  This tool prices the kernel that ships:

  * every VALU instruction of the kernel's disassembly (tools/isa/disasm_lib.py on libdialhip.so) is assigned the ubench row that
    matches its FORM -- what the microbenchmark shows to matter is not the opcode but the operand kinds: three distinct VGPR sources
    (4.1 cycles at two wavefronts per SIMD), an SGPR / VCC-as-data source or an SGPR destination (4.2-4.5), a DPP modifier (4.4), a
    transcendental or a lane swap (8.2); two-VGPR VOP1 / VOP2 forms issue at 2.25, VOP3 FMA with an inline constant at 2.5;
  * the static counts are grouped by the hardware's own dynamic classes (SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F32, INT32, CVT, the
    rest) and every class gets its static average cost;
  * with a PMC file (tools/pmc_to_json.py: per-class DYNAMIC counts of the launch) the classes are weighted by what actually
    executed: cycles per VALU instruction = sum_c (PMC share of class c) x (static average cost of class c).  Within a class the
    static distribution stands in for the dynamic one (there is no per-opcode counter); between classes the weights are measured.

usage: price_mix.py <ubench_issue.txt> <listing.s> <kernel-name-regex> [pmc.json] [out.json]
"""
import collections
import json
import re
import sys

ROW = {  # form -> label prefix of its row in the ubench table
    "fma_3v": "v_fma_f32 d, d, v, v", "fma_2v": "v_fma_f32 d, d, d, v", "fma_s": "v_fma_f32 d, d, s, v", "fma_k": "v_fma_f32 d, d, 2.0, v",
    "fmac_3v": "v_fmac_f32 d, v, v (VOP2, three", "fmac_2v": "v_fmac_f32 d, v, v (same", "fmac_s": "v_fmac_f32 d, s, v", "fmac_dpp": "v_fmac_f32_dpp",
    "vop2_vv": "v_mul_f32 d, d, v", "addsub": "v_add_f32 / v_sub_f32", "minmax": "v_max_f32 / v_min_f32", "mov_vv": "v_mov_b32 v, v", "sgpr_src": "v_mov_b32 v, s",
    "cnd_e64": "v_cndmask_b32 d, d, v, s[..]", "cnd_e32": "v_cndmask_b32 vcc alternating with v_mul_f32", "cmp_vcc": "v_cmp_lt_f32 -> vcc",
    "cmp_sgpr": "v_cmp_lt_f32 -> s[..]", "int": "int32 (add", "dpp": "DPP (v_mov_dpp", "readlane": "v_readlane_b32", "readfirstlane": "v_readfirstlane_b32",
    "writelane": "v_writelane_b32", "trans": "v_rcp_f32", "permlane": "v_permlane16_swap_b32", "pk_fma": "v_pk_fma_f32", "pk_mul": "v_pk_mul_f32", "pk_add": "v_pk_add_f32",
}
PMC_CLASSES = ("FMA_F32", "ADD_F32", "MUL_F32", "TRANS_F32", "INT32", "CVT", "OTHER")


def read_ubench(path):
    rows = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s{2,}(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s*$", line.rstrip())
        if m:
            rows[m.group(1).strip()] = [float(m.group(k)) for k in (2, 3, 4, 5)]
    table = {}
    for form, prefix in ROW.items():
        hit = [v for k, v in rows.items() if k.startswith(prefix)]
        if not hit:
            raise SystemExit(f"ubench table {path} has no row starting with `{prefix}`")
        table[form] = hit[0]
    return table


VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def classify(op, ops):
    """-> (ubench form, PMC class) of one VALU instruction (opcode without the _e32 / _e64 suffix logic: the suffix is kept in op)"""
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    srcs = ops[1:] if ops else []
    src_txt = " ".join(srcs)
    # SGPR-side sources: s<k>, s[a:b], vcc / exec / m0 used as DATA (not the implicit carry / mask of the VOP2 encodings)
    has_sgpr = bool(re.search(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|src_\w+)\b", src_txt))
    has_const = any(re.fullmatch(r"-?(0x[0-9a-f]+|\d+(\.\d+)?(e[+-]?\d+)?)", s.strip().lstrip("-|").rstrip("|")) is not None for s in srcs)
    vsrc = set()
    for s_ in srcs:
        for m in VREG.finditer(s_):
            vsrc.add(m.group(1) or m.group(2))
    pmc = "OTHER"
    if re.match(r"v_(fma|fmac|fmaak|fmamk|mad|mac)_(legacy_)?f32", base) or base.startswith("v_pk_fma_f32"):
        pmc = "FMA_F32"
    elif re.match(r"v_(add|sub|subrev)_f32", base) or base.startswith("v_pk_add_f32"):
        pmc = "ADD_F32"
    elif re.match(r"v_mul_(legacy_)?f32", base) or base.startswith("v_pk_mul_f32"):
        pmc = "MUL_F32"
    elif re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)(_iflag|_clamp|_legacy)?_f32", base):
        pmc = "TRANS_F32"
    elif base.startswith("v_cvt_"):
        pmc = "CVT"
    elif re.match(r"v_(add|sub|subrev|addc|subb|subbrev|mul_lo|mul_hi|mul_u32|mul_i32|mad_u|mad_i|add3|lshl_add|add_lshl|lshl_or|and_or|or3|xad|sad|min_[ui]|max_[ui]|med3_[ui]|mbcnt)", base) or \
            re.match(r"v_(lshlrev|lshrrev|ashrrev|and|or|xor|not|bfe|bfi|bfm|alignbit|perm|bitop3|xnor)_", base):
        pmc = "INT32"
    # ---- form
    if base.startswith("v_pk_fma"):
        return "pk_fma", pmc
    if base.startswith("v_pk_mul"):
        return "pk_mul", pmc
    if base.startswith("v_pk_add"):
        return "pk_add", pmc
    if pmc == "TRANS_F32":
        return "trans", pmc
    if base.startswith("v_permlane"):
        return "permlane", pmc
    if base.startswith("v_readlane"):
        return "readlane", pmc
    if base.startswith("v_readfirstlane"):
        return "readfirstlane", pmc
    if base.startswith("v_writelane"):
        return "writelane", pmc
    if base.endswith("_dpp"):
        return ("fmac_dpp" if base.startswith("v_fmac") else "dpp"), pmc
    if base.startswith("v_cmp"):
        dst = ops[0] if ops else "vcc"
        return ("cmp_vcc" if dst.startswith("vcc") or op.endswith("_e32") else "cmp_sgpr"), pmc
    if base.startswith("v_cndmask"):
        mask = ops[3] if len(ops) > 3 else "vcc"
        return ("cnd_e32" if op.endswith("_e32") else ("cnd_e32" if mask.startswith("vcc") and not has_const and False else "cnd_e64")), pmc
    if pmc == "FMA_F32":
        if base.startswith("v_fmac"):
            allv = set(vsrc) | ({m.group(1) or m.group(2) for m in VREG.finditer(ops[0])} if ops else set())
            return ("fmac_s" if has_sgpr else ("fmac_3v" if len(allv) >= 3 else "fmac_2v")), pmc
        if base.startswith(("v_fmaak", "v_fmamk")):
            return "fma_k", pmc
        return ("fma_s" if has_sgpr else ("fma_k" if has_const else ("fma_3v" if len(vsrc) >= 3 else "fma_2v"))), pmc
    if has_sgpr:
        return "sgpr_src", pmc
    if re.match(r"v_(min|max|med3)_", base):
        return "minmax", pmc
    if pmc == "INT32":
        return "int", pmc
    if pmc == "ADD_F32":
        return "addsub", pmc
    if base.startswith("v_mov_b32") or base.startswith("v_accvgpr"):
        return "mov_vv", pmc
    return "vop2_vv", pmc


def census(listing, want):
    """static VALU census of the kernels whose label contains `want`: Counter[(form, pmc class)], salu, other counts"""
    cur, on = None, False
    cnt = collections.Counter()
    extra = collections.Counter()
    for line in open(listing):
        m = re.match(r"^(\w[\w$.]*):", line)
        if m and not m.group(1).startswith((".L", "L")):
            cur = m.group(1)
            on = re.search(want, cur) is not None     # (`want`: a regular expression over the mangled kernel name)
            continue
        if not on:
            continue
        t = line.split(";")[0].split("//")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        if op.startswith("v_"):
            ops = [o.strip() for o in rest.split(",")] if rest else []
            cnt[classify(op, ops)] += 1
        elif op.startswith("s_nop"):
            extra["s_nop"] += 1
        elif op.startswith("s_waitcnt"):
            extra["s_waitcnt"] += 1
        elif op.startswith("s_"):
            extra["salu+branch"] += 1
        elif op.startswith("ds_"):
            extra["lds"] += 1
        else:
            extra["vmem/other"] += 1
    return cnt, extra


def main():
    ub, listing, want = sys.argv[1], sys.argv[2], sys.argv[3]
    pmc_path = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != "-" else None
    out_path = sys.argv[5] if len(sys.argv) > 5 else None
    table = read_ubench(ub)
    cnt, extra = census(listing, want)
    n_valu = sum(cnt.values())
    if n_valu == 0:
        raise SystemExit(f"no kernel matching `{want}` in {listing}")
    by_class = {c: collections.Counter() for c in PMC_CLASSES}
    for (form, pc), n in cnt.items():
        by_class[pc][form] += n
    static_share = {c: sum(by_class[c].values()) / n_valu for c in PMC_CLASSES}
    class_cost = {c: [sum(n * table[f][w] for f, n in by_class[c].items()) / max(1, sum(by_class[c].values())) for w in range(4)] for c in PMC_CLASSES}
    static_cost = [sum(static_share[c] * class_cost[c][w] for c in PMC_CLASSES) for w in range(4)]
    res = {"kernel": want, "listing": listing, "ubench": ub, "static_valu_instructions": n_valu, "static_other_instructions": dict(extra),
           "static_forms": {f: sum(n for (ff, _), n in cnt.items() if ff == f) for f in sorted({ff for ff, _ in cnt})},
           "cost_table_cycles_W1_W4": {f: table[f] for f in sorted({ff for ff, _ in cnt})},
           "class_static_share": static_share, "class_cost_cycles_W1_W4": class_cost, "cycles_per_valu_inst_static_W1_W4": static_cost}
    if pmc_path:
        pmc = json.load(open(pmc_path))
        c = pmc.get("counters", {})
        tot = c.get("SQ_INSTS_VALU")
        keys = {"FMA_F32": "SQ_INSTS_VALU_FMA_F32", "ADD_F32": "SQ_INSTS_VALU_ADD_F32", "MUL_F32": "SQ_INSTS_VALU_MUL_F32", "TRANS_F32": "SQ_INSTS_VALU_TRANS_F32",
                "INT32": "SQ_INSTS_VALU_INT32", "CVT": "SQ_INSTS_VALU_CVT"}
        if tot and all(k in c for k in keys.values()):
            dyn = {k: c[v] / tot for k, v in keys.items()}
            dyn["OTHER"] = 1.0 - sum(dyn.values())
            res["class_dynamic_share_pmc"] = dyn
            res["pmc"] = pmc_path
            res["cycles_per_valu_inst_pmc_weighted_W1_W4"] = [sum(dyn[k] * class_cost[k][w] for k in PMC_CLASSES) for w in range(4)]
    print(json.dumps(res, indent=1))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
