#!/usr/bin/env python
"""Do two indexes searched on two streams overlap on one GPU? (the partitioned mode's shards)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
args = bench.parse()
args.build_max_search, args.build_reinsert = 50, 0
B = bench.Bench(args)
torch = B.torch
n, nq, k, ef = 5_000_000, 1024, 10, 50
idx = []
for g in range(2):
    el = B.rows("uniform", bench.SEED + 100 + g, 0, n, 100, "f32")
    b, ix, tb = B.build_index(el, "f32")
    idx.append((el, b, ix))
q = B.rows("uniform", bench.SEED + 1, 0, 8 * nq, 100, "f32")
ids = torch.empty((2, nq, k), dtype=torch.int64, device="cuda")
ds = torch.empty((2, nq, k), dtype=torch.float32, device="cuda")
cnt = torch.empty((2, nq), dtype=torch.int32, device="cuda")
st = torch.zeros((2, 4), dtype=torch.int32, device="cuda")
new_streams = [torch.cuda.Stream() for _ in range(3)]
for name, ss in (("bench streams 0,1", B.streams[:2]), ("fresh streams", new_streams[:2]), ("same stream", [B.streams[0], B.streams[0]])):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            for g in range(2):
                idx[g][2].search_batch_device(q[(i % 8) * nq:(i % 8 + 1) * nq].data_ptr(), nq, ef, k, ids[g].data_ptr(), ds[g].data_ptr(),
                                              cnt[g].data_ptr(), 0, st[g].data_ptr(), ss[g].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-20s %.3f ms per pair of shard searches" % (name, dt / 100 * 1e3), flush=True)

for name, ss in (("isolated pair, two streams", new_streams[:2]), ("isolated pair, one stream", [new_streams[0], new_streams[0]])):
    tot = 0.0
    for i in range(100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g in range(2):
            idx[g][2].search_batch_device(q[(i % 8) * nq:(i % 8 + 1) * nq].data_ptr(), nq, ef, k, ids[g].data_ptr(), ds[g].data_ptr(),
                                          cnt[g].data_ptr(), 0, st[g].data_ptr(), ss[g].cuda_stream)
        torch.cuda.synchronize()
        tot += time.perf_counter() - t0
    print("%-28s %.3f ms (host clock around enqueue + sync)" % (name, tot / 100 * 1e3), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cur = torch.cuda.current_stream()
acc = 0.0
for i in range(100):
    torch.cuda.synchronize()
    e0.record(cur)
    for g in range(2):
        new_streams[g].wait_event(e0)
        idx[g][2].search_batch_device(q[(i % 8) * nq:(i % 8 + 1) * nq].data_ptr(), nq, ef, k, ids[g].data_ptr(), ds[g].data_ptr(),
                                      cnt[g].data_ptr(), 0, st[g].data_ptr(), new_streams[g].cuda_stream)
        cur.wait_stream(new_streams[g])
    e1.record(cur)
    torch.cuda.synchronize()
    acc += e0.elapsed_time(e1)
print("isolated pair, fork/join by events: %.3f ms (HIP events on the caller's stream)" % (acc / 100), flush=True)

# the same two indexes through granne_amd.sharded (one rank): where does the pair's time go?
from granne_amd import sharded
sg = sharded.ShardedGranne([idx[0][2], idx[1][2]], [0, n])
qs = [q[(i % 8) * nq:(i % 8 + 1) * nq] for i in range(100)]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100):
        sg.search_batch(qs[i], ef, k, check_status=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("sharded.search_batch          %.3f ms per batch (host loop %.3f ms)" % (dt / 100 * 1e3, dt / 100 * 1e3), flush=True)
for depth in (1, 2, 3):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sg.search_batches(qs, ef, k, depth=depth, check_status=False)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("sharded.search_batches depth=%d %.3f ms per batch (enqueue %.3f ms per batch)" % (depth, dt / 100 * 1e3, t_enq / 100 * 1e3), flush=True)
