"""Not a pass/fail benchmark: a quick timing probe of the search kernel on a mid-size index built
by the oracle, printed for the log (bench.py is the real measurement)."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.environ.get("GRANNE_PROBE"), reason="set GRANNE_PROBE=<n_elements> to run")
def test_probe(oracle):
    import torch
    import granne_amd
    n = int(os.environ["GRANNE_PROBE"])
    dim, nq = 100, 1024
    t = time.time()
    el = oracle.synth_rows(0x6772616E6E65, 0, n, dim)
    import ctypes as C
    el = granne_amd.normalize(el)
    q = granne_amd.normalize(oracle.synth_rows(0x6772616E6E66, 0, nq, dim))
    print("\ngen %.1fs threads=%d" % (time.time() - t, oracle.lib().gro_max_threads()))
    t = time.time()
    oix = oracle.build_index(el, num_neighbors=30, max_search=int(os.environ.get("GRANNE_PROBE_MS", "50")),
                             reinsert_elements=False, n_threads=0)
    print("cpu build %.1fs layers=%s" % (time.time() - t, [l.shape[0] for l in oix.layers]))
    gix = granne_amd.Granne("angular", el, oix.layers)
    for ef in (50, 200):
        t = time.time()
        oi, od, oc, octr = oix.search_batch(q, ef, 10)
        cpu_s = time.time() - t
        ids, ds, cnt, st = gix.search_batch(q, ef, 10, stats=True)
        assert (ids == oi).all() and ds.tobytes() == od.tobytes()
        dq = torch.from_numpy(q).cuda()
        d_ids = torch.empty((nq, 10), dtype=torch.int64, device="cuda")
        d_ds = torch.empty((nq, 10), dtype=torch.float32, device="cuda")
        d_cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            gix.search_batch_device(dq.data_ptr(), nq, ef, 10, d_ids.data_ptr(), d_ds.data_ptr(), d_cnt.data_ptr(), 0, 0, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gix.search_batch_device(dq.data_ptr(), nq, ef, 10, d_ids.data_ptr(), d_ds.data_ptr(), d_cnt.data_ptr(), 0, 0, s)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nd = st[:, 0].astype(np.float64)
        alg = (nd * dim * 4 + st[:, 2] * 4 + dim * 4 + 80).sum()
        print("ef=%d: gpu %.3f ms/batch = %.0f qps; alg %.1f MB -> %.1f GB/s; cpu %.0f qps (%d thr); slow=%d n_dist=%.0f"
              % (ef, ms, nq / ms * 1e3, alg / 1e6, alg / ms / 1e6, nq / cpu_s, oracle.lib().gro_max_threads(),
                 gix.last_slow_count(), nd.mean()))
