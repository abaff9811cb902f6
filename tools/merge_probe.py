#!/usr/bin/env python
"""merge_topk_kernel alone: ms per call for [n_shards][nq][k] sorted candidate lists (the brute-force scan's 64 x 16 and the
partitioned search's 8 x 10), against the numpy merge."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from granne_amd import _lib  # noqa: E402
from oracle.merge import merge_topk_numpy  # noqa: E402

lib = _lib.lib()
for G, k, nq in [(64, 16, 1024), (8, 10, 4096), (8, 10, 1024), (64, 16, 64)]:
    rng = np.random.default_rng(G * 100 + k)
    d = np.sort(rng.random((G, nq, k), dtype=np.float32), axis=2)
    ids = rng.integers(0, 1 << 20, (G, nq, k)).astype(np.uint64)
    cnt = np.full((G, nq), k, np.uint32)
    offs = (np.arange(G, dtype=np.uint64) << np.uint64(20))
    want = merge_topk_numpy(ids, d, cnt, list(offs), k)
    dd, di, dc = torch.from_numpy(d).cuda(), torch.from_numpy(ids.view(np.int64)).cuda(), torch.from_numpy(cnt.view(np.int32)).cuda()
    oi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    oc = torch.empty(nq, dtype=torch.int32, device="cuda")
    ho = (C.c_uint64 * G)(*[int(x) for x in offs])

    def run():
        _lib.check(lib.granne_hip_merge_topk_device(C.c_void_p(di.data_ptr()), C.c_void_p(dd.data_ptr()), C.c_void_p(dc.data_ptr()), ho, G, nq, k,
                                                    C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), C.c_void_p(oc.data_ptr()), 0, None))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ok = (oi.cpu().numpy().astype(np.uint64) == want[0]).all() and od.cpu().numpy().tobytes() == want[1].tobytes()
    print("%2d shards x k %2d x %4d queries: %.3f ms per merge, equal to the numpy merge: %s" % (G, k, nq, ms, ok))
