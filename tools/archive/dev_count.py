#!/usr/bin/env python
"""Static instruction counts of the walker's expansion loop in /tmp/dev/one.s (tools/dev_isa.sh): the loop nest that
holds the row loads, from its header to the last back-edge. usage: python tools/dev_count.py [file]"""
import re, sys
txt = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/dev/one.s").read()
for m in re.finditer(r"^(_ZN10granne_hip11fast_kernel\S+):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    # the LAST run of >= 4 row loads (dwordx4) lives in the expansion loop
    loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l]
    last = loads[-1]
    # the enclosing depth-2 loop header before it
    hdr = max(i for i, l in enumerate(body[:last]) if "Loop Header: Depth=2" in l)
    lab = None
    for i in range(hdr, 0, -1):
        mm = re.match(r"^(\.LBB\d+_\d+):", body[i])
        if mm:
            lab = mm.group(1); start = i; break
    end = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", l))
    ins = [l.split()[0] for l in body[start:end + 1] if re.match(r"^\s+[a-z]", l)]
    c = lambda p: sum(1 for i in ins if i.startswith(p))
    print("%s\n  loop %s: %d instr | valu %d (cmp %d, readlane %d, writelane %d, dpp-ish mov %d) salu %d (branch %d, nop %d, waitcnt %d) ds %d vmem %d"
          % (name[-40:], lab, len(ins), c("v_"), c("v_cmp"), c("v_readlane"), c("v_writelane"), c("v_mov_b32_dpp"), c("s_"),
             c("s_cbranch") + c("s_branch"), c("s_nop"), c("s_waitcnt"), c("ds_"), c("global_") + c("flat_")))
