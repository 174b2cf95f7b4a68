#!/usr/bin/env python3
"""Per-kernel statistics of an ISA listing (tools/isa/probe.sh output): instruction count, code bytes, the opcode histogram's head."""
import collections
import re
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cur, kernels = None, collections.OrderedDict()
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        kernels[cur] = []
        continue
    if cur and re.match(r"^\s+(v_|s_|ds_|global_|buffer_|scratch_|flat_)", line):
        kernels[cur].append(line.split()[0])
    if "codeLenInByte" in line and cur:
        kernels[cur].append(("LEN", int(line.split("=")[1])))
for name, ins in kernels.items():
    length = [x[1] for x in ins if isinstance(x, tuple)]
    ops = [x for x in ins if not isinstance(x, tuple)]
    if not ops:
        continue
    h = collections.Counter(re.sub(r"_e(32|64)$", "", o) for o in ops)
    kind = "rollout_kernel2" if "rollout_kernel2" in name else "rollout_kernel"
    print(f"== {kind}  ({name[:60]}...)  instructions {len(ops)}  code bytes {length[0] if length else '?'}")
    valu = sum(v for k, v in h.items() if k.startswith("v_"))
    print(f"   VALU {valu}  SALU {sum(v for k, v in h.items() if k.startswith('s_'))}  LDS {sum(v for k, v in h.items() if k.startswith('ds_'))}"
          f"  v_readlane {h.get('v_readlane_b32', 0)}  v_writelane {h.get('v_writelane_b32', 0)}  dpp {sum(v for k, v in h.items() if 'dpp' in k)}"
          f"  permlane {sum(v for k, v in h.items() if 'permlane' in k)}  bpermute {h.get('ds_bpermute_b32', 0)}  s_nop {h.get('s_nop', 0)}")
    print("   " + "  ".join(f"{k}:{v}" for k, v in h.most_common(top)))
