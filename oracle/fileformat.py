"""granne's on-disk formats, restated in plain Python (TEST INFRASTRUCTURE, like the rest of oracle/).

Independent of granne_amd/csrc/fileformat_host.h on purpose: tests/ write files with one and read
them with the other. Citations relative to /root/reference.
  index file    src/index/io.rs:11-113
  layer blob    src/slice_vector/set_vector.rs:164-222, src/slice_vector/offsets.rs:80-131,148-296
  node record   src/slice_vector/set_vector.rs:91-148 over stream-vbyte 0.3.2 (Scalar)
  elements file src/slice_vector/mod.rs:213-221, 460-466
"""
import json
import struct

import numpy as np

from . import oracle as orc

METADATA_LEN = 1024
OFFSETS_PER_CHUNK = 60
UNUSED = 0xFFFFFFFF


def write_elements(elements):
    e = np.ascontiguousarray(elements)
    return struct.pack("<Q", e.shape[1]) + e.tobytes()


def read_elements(buf, dtype):
    (dim,) = struct.unpack_from("<Q", buf, 0)
    data = np.frombuffer(buf, dtype, offset=8)
    assert dim > 0 and data.size % dim == 0
    return data.reshape(-1, dim)


def _layer_blob(rows):
    n = rows.shape[0]
    n_chunks = 1 + n // OFFSETS_PER_CHUNK
    records, offsets, total = [], [0], 0
    for r in rows:
        ids = np.sort(r[r != UNUSED]).astype(np.uint32)
        rec = orc.set_encode(ids)
        records.append(rec)
        total += len(rec)
        offsets.append(total)
    chunks = bytearray()
    for c in range(n_chunks):
        first = c * OFFSETS_PER_CHUNK
        initial = offsets[first] if first <= n else 0
        deltas, prev = [], initial
        for t in range(OFFSETS_PER_CHUNK):
            if first + t <= n:
                deltas.append(offsets[first + t] - prev)
                prev = offsets[first + t]
            else:
                deltas.append(0xFFFF)
        chunks += struct.pack("<Q60H", initial, *deltas)
    return struct.pack("<Q", len(chunks)) + bytes(chunks) + b"".join(records)


def write_index(layers):
    """io.rs:11-70 for fixed-width layers ([len][width] u32, UNUSED padded)."""
    blobs = [_layer_blob(np.asarray(l, np.uint32)) for l in layers]
    last = np.asarray(layers[-1]) if len(layers) else None
    meta = {
        "granne_version": "0.5.2", "version": 2,
        "num_elements": int(last.shape[0]) if last is not None else 0,
        "num_layers": len(layers),
        "num_neighbors": int((last[0] != UNUSED).sum()) if last is not None and last.shape[0] else 0,
        "layer_counts": [int(np.asarray(l).shape[0]) for l in layers],
        "layer_sizes": [len(b) for b in blobs],
        "compressed": True,
    }
    head = ("granne" + json.dumps(meta, sort_keys=True, separators=(",", ":"))).encode()
    assert len(head) <= METADATA_LEN
    return head.ljust(METADATA_LEN, b" ") + b"".join(blobs)


def read_index(buf):
    """io.rs:72-113: returns per layer a list of neighbor-id lists (ascending)."""
    assert buf[:6] == b"granne", "Library string missing"
    meta = json.loads(buf[6:METADATA_LEN].decode())
    assert meta["num_layers"] == len(meta["layer_counts"])
    layers, start = [], METADATA_LEN
    for size, count in zip(meta["layer_sizes"], meta["layer_counts"]):
        blob = buf[start:start + size]
        start += size
        (off_bytes,) = struct.unpack_from("<Q", blob, 0)
        chunks = blob[8:8 + off_bytes]
        data = blob[8 + off_bytes:]

        def offset(j):
            c = chunks[(j // OFFSETS_PER_CHUNK) * 128:(j // OFFSETS_PER_CHUNK + 1) * 128]
            initial, *deltas = struct.unpack("<Q60H", c)
            return initial + sum(deltas[: j % OFFSETS_PER_CHUNK + 1])

        nodes = []
        for j in range(count):
            nodes.append(orc.set_decode(data[offset(j):offset(j + 1)]).tolist())
        layers.append(nodes)
    return meta, layers
