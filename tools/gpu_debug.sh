#!/bin/bash
# one-off GPU debugging helper: per-file / per-test runs so that one abort does not hide the rest
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m granne_amd.build >/dev/null 2>&1
echo "== parity file"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -15
echo "== builder file, without the duplicates test"; timeout 900 python -m pytest tests/test_gpu_builder.py -m gpu -q -k "not duplicates" 2>&1 | tail -15
echo "== duplicates test with tracing"; GRANNE_HIP_DEBUG=1 timeout 300 python -m pytest tests/test_gpu_builder.py -m gpu -q -k duplicates -s 2>&1 | grep -v "^\s*File\|^$" | tail -40
for cmd in "$@"; do
  echo "== $cmd"
  timeout 900 bash -c "$cmd" 2>&1 | tail -30
done
