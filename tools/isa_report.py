#!/usr/bin/env python
"""Resource usage (VGPR/SGPR/spills/LDS) and static instruction mix of the search kernels.
usage: python tools/isa_report.py [extra hipcc flags...]   (writes /tmp/granne_isa.s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/granne_isa.s"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-DGRANNE_HIP_USE_DPP=1", "-I",
       os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-Wno-unused-command-line-argument",
       os.path.join(ROOT, "granne_amd", "csrc", "granne_hip.hip"), "-o", OUT] + sys.argv[1:]
if not os.environ.get("ISA_REUSE"):
    subprocess.check_call(cmd)
txt = open(OUT).read()
names = {}
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?"
                     r"\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
    names[m.group(1)] = m.groups()[1:]
for n, (sg, ss, vg, vs) in names.items():
    if "fast_kernel" not in n and "search_kernel" not in n:
        continue
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    d = d.replace("granne_hip::", "").replace("(granne_hip::SearchParams)", "").replace("void ", "")
    body = re.search(r"^" + re.escape(n) + r":[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M)
    ins = re.findall(r"^\s+([a-z_0-9]+)", body.group(1), re.M) if body else []
    cnt = lambda pre: sum(1 for i in ins if i.startswith(pre))  # noqa: E731
    print("%-38s vgpr %3s sgpr %3s spill s%-3s v%-2s | static: %5d instr  valu %5d salu %5d ds %4d vmem %4d"
          % (d, vg, sg, ss, vs, len(ins), cnt("v_"), cnt("s_"), cnt("ds_"), cnt("global_") + cnt("flat_") + cnt("buffer_")))
