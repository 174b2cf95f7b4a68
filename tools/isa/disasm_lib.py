#!/usr/bin/env python3
"""Disassemble the gfx950 code objects of a SHIPPED library (libdialhip.so): what the GPU box executes, not a re-compilation.

The library's `.hip_fatbin` section holds one clang offload bundle per translation unit (magic `__CLANG_OFFLOAD_BUNDLE__`, a table of
(offset, size, triple) entries, the code objects themselves); every `hipv4-amdgcn-amd-amdhsa--gfx950` entry is written to
<outdir>/co<k>.o and disassembled with llvm-objdump into <outdir>/co<k>.s (one `<mangled name>:` label per kernel, one instruction per
line -- the format tools/isa/check_dpp_hazards.py, section_hist.py and price_mix.py read).  Also prints the resource notes of every
kernel (llvm-readelf --notes: VGPRs, spilled VGPRs / SGPRs, scratch bytes).

usage: disasm_lib.py <lib.so> <outdir> [--notes-only]
"""
import os
import re
import struct
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, outdir):
    """-> list of paths of the gfx950 code objects of `lib` (written into outdir)"""
    os.makedirs(outdir, exist_ok=True)
    fat = os.path.join(outdir, "fatbin.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(outdir, "_discard.so")])
    os.remove(os.path.join(outdir, "_discard.so"))
    blob = open(fat, "rb").read()
    os.remove(fat)
    out, pos, k = [], 0, 0
    while True:
        b0 = blob.find(MAGIC, pos)
        if b0 < 0:
            break
        p = b0 + len(MAGIC)
        (n,) = struct.unpack_from("<Q", blob, p)
        p += 8
        end = b0
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            end = max(end, b0 + off + size)
            if "gfx950" in triple and size > 0:
                path = os.path.join(outdir, f"co{k}.o")
                open(path, "wb").write(blob[b0 + off:b0 + off + size])
                out.append(path)
                k += 1
        pos = max(end, b0 + len(MAGIC))
    return out


def disassemble(co):
    """llvm-objdump -d -> an assembler-like listing: `<symbol>:` labels, one instruction per line"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", "--mcpu=gfx950", co],
                         capture_output=True, text=True, check=True).stdout
    lines = []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            # `<L12>` = a branch target (--symbolize-operands), anything else = a function symbol
            lines.append(f".{m.group(1)}:" if re.fullmatch(r"L\d+", m.group(1)) else f"{m.group(1)}:")
            continue
        t = line.split("//")[0].rstrip()
        if t.startswith(("\t", " ")) and t.strip():
            lines.append("\t" + t.strip())
    path = co[:-2] + ".s"
    open(path, "w").write("\n".join(lines) + "\n")
    return path


def kernel_notes(co):
    """robust variant: the metadata lists one map per kernel; fields of a kernel precede OR follow its .name -- parse per `- ` item"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    items, cur, in_kernels = [], None, False
    for line in txt.splitlines():
        if re.match(r"\s*amdhsa\.kernels:", line):
            in_kernels = True
            continue
        if in_kernels and re.match(r"\s*amdhsa\.\w+:", line):
            in_kernels = False
        if not in_kernels:
            continue
        if re.match(r"\s+- \.\w+", line) and not re.match(r"\s{6,}- ", line):   # a new kernel map starts at the list's own indentation
            cur = {}
            items.append(cur)
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", line)
        if m and cur is not None and m.group(1) in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                                                     "group_segment_fixed_size", "agpr_count"):
            cur.setdefault(m.group(1), m.group(2))     # (the first .name of a kernel map is the kernel's; its arguments' come later and are nested deeper)
    return [k for k in items if "vgpr_count" in k and "name" in k]


def demangle_short(name):
    """_Z14rollout_kernelI4DimsI..7TopoGo2..ELi1ELi3ELb0ELb0EEv... -> rollout_kernel<Go2, gen=0, nc=4; 1, 3, false, false>"""
    m = re.match(r"_Z\d+(rollout_kernel2?|env_step_kernel|env_reset_kernel)I4DimsI(.*)", name)
    if not m:
        return name[:100]
    base, rest = m.group(1), m.group(2)
    topo = re.search(r"\d+(Topo\w+?)(?=L[bi])", rest)
    dims = re.match(r"Lb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", rest)
    tail = re.search(r"EELi(\d+)ELi(\d+)ELb([01])ELb([01])EEv", rest)
    d = f"{topo.group(1)[4:] if topo else '?'}" + (f"(nq {dims.group(2)}, ncon {dims.group(9)})" if dims else "")
    if tail:
        return f"{base}<{d}; WPB {tail.group(1)}, OCC {tail.group(2)}, {'queue' if tail.group(3) == '1' else 'grid'}, {'trace/mean' if tail.group(4) == '1' else '-'}>"
    return f"{base}<{d}>"


def main():
    lib, outdir = sys.argv[1], sys.argv[2]
    cos = code_objects(lib, outdir)
    print(f"{lib}: {len(cos)} gfx950 code objects")
    print(f"{'kernel':110s} {'VGPR':>5s} {'SGPR':>5s} {'spilled V':>9s} {'spilled S':>9s} {'scratch B':>9s}")
    for co in cos:
        if "--notes-only" not in sys.argv:
            disassemble(co)
        for k in kernel_notes(co):
            print(f"{demangle_short(k['name']):110s} {k.get('vgpr_count', '?'):>5s} {k.get('sgpr_count', '?'):>5s} {k.get('vgpr_spill_count', '?'):>9s} "
                  f"{k.get('sgpr_spill_count', '?'):>9s} {k.get('private_segment_fixed_size', '?'):>9s}   [{os.path.basename(co)}: {k['name'][:60]}]")


if __name__ == "__main__":
    main()
