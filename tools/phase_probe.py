#!/usr/bin/env python
"""Where does a walk spend its cycles?  Runs batches through the diagnostics build of the library
(-DGRANNE_HIP_PHASE_TIMERS=1: s_memtime stamps around the phases of an expansion, walk_fast.h) and prints
the mean cycles per expansion of every phase, for all walks of a batch and for its slowest walks (a launch
lasts as long as its slowest walk).  The stamps themselves cost ~10 % of the wave's cycles.

  GRANNE_HIP_LIB=granne_amd/lib/libgranne_hip_phase.so python tools/phase_probe.py --dtype f32 [--fast-build]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ["top of loop", "issue rows + y", "visited", "row wait", "distances", "decision + adj req", "merge", "layer setup",
         "adj wait", "make_room", "cache + filter", "rank loop", "flag", "-", "-", "-"]
SLOTS = 48


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--ef", type=int, default=50)
    ap.add_argument("--nq", type=int, default=1024)
    ap.add_argument("--inflight", type=int, default=1)
    ap.add_argument("--fast-build", action="store_true")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.dtype, args.n, args.dim = a.dtype, a.n, a.dim
    if a.fast_build:
        args.build_max_search, args.build_reinsert = 50, 0
    B = bench.Bench(args)
    L = B.lib
    if not hasattr(L, "granne_hip_debug_phases"):
        raise SystemExit("not the diagnostics build: set GRANNE_HIP_LIB to a library built with -DGRANNE_HIP_PHASE_TIMERS=1")
    L.granne_hip_debug_phases.restype = C.c_int
    L.granne_hip_debug_phases.argtypes = [C.c_void_p, C.c_uint32]
    elements = B.rows("uniform", bench.SEED, 0, a.n, a.dim, a.dtype)
    builder, index, tb = B.build_index(elements, a.dtype)
    print("build %.1fs" % tb, flush=True)
    esize = 4 if a.dtype == "f32" else 1
    steps, warmup = 6, 2
    queries = B.rows("uniform", bench.SEED + 1, 0, (steps + warmup) * a.nq, a.dim, a.dtype)
    m = B.measure(index, queries, a.dim, esize, a.nq, a.ef, 10, steps, warmup, 1, inflight=a.inflight)
    print("launch %.4f ms (min %.4f) with the stamps" % (m["launch_ms_mean"], m["launch_ms_min"]))
    # the clocks are those of the LAST launch, and measure() ends with its counting pass on the exact visited tables:
    # one more launch of the shipped form (or of the form GRANNE_HIP_VISITED names)
    torch = B.torch
    ids = torch.empty((a.nq, 10), dtype=torch.int64, device="cuda")
    ds = torch.empty((a.nq, 10), dtype=torch.float32, device="cuda")
    cnt = torch.empty((a.nq,), dtype=torch.int32, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    b = warmup + steps - 1
    index.search_batch_device(queries[b * a.nq:(b + 1) * a.nq].data_ptr(), a.nq, a.ef, 10, ids.data_ptr(), ds.data_ptr(),
                              cnt.data_ptr(), 0, status.data_ptr(), B.stream)
    torch.cuda.synchronize()
    out = np.zeros((a.nq, SLOTS), np.uint64)
    rc = L.granne_hip_debug_phases(out.ctypes.data_as(C.c_void_p), a.nq)
    assert rc == 0, rc
    ph = out.astype(np.float64)
    total = ph[:, 34]
    order = np.argsort(total)
    groups = [("all walks", order), ("slowest 2 %", order[-max(1, a.nq // 50):])]
    for name, idx in groups:
        nb, nu = ph[idx, 32].mean(), ph[idx, 33].mean()
        print("\n== %s: %.0f cycles per walk, %.1f bottom + %.1f upper expansions" % (name, total[idx].mean(), nb, nu))
        print("%-14s %12s %12s %14s %14s" % ("phase", "bottom/walk", "upper/walk", "bottom/expan.", "upper/expan."))
        sb = su = 0.0
        for i in range(13):
            b, u = ph[idx, i].mean(), ph[idx, 16 + i].mean()
            sb, su = sb + b, su + u
            print("%-14s %12.0f %12.0f %14.1f %14.1f" % (NAMES[i], b, u, b / max(nb, 1e-9), u / max(nu, 1e-9)))
        print("%-14s %12.0f %12.0f %14.1f %14.1f" % ("sum", sb, su, sb / max(nb, 1e-9), su / max(nu, 1e-9)))
        c = ph[idx, 36:44].mean(axis=0)
        print("bottom layer, per expansion: candidates inserted %.2f | expansions with 0: %.3f, 1-2: %.3f, 3-6: %.3f, >6: %.3f | "
              "a candidate is expanded next: %.3f | candidates that reach the rank loop %.2f"
              % (c[0] / nb, c[1] / nb, c[2] / nb, c[3] / nb, c[4] / nb, c[5] / nb, c[6] / nb))
    print("\nslowest walk %.0f cycles, mean %.0f, ratio %.2f" % (total.max(), total.mean(), total.max() / total.mean()))


if __name__ == "__main__":
    main()
