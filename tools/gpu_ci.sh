#!/bin/bash
# GPU-box driver: parity tests (falls back to the shuffle build of the wave primitives if the DPP
# build misbehaves), then whatever bench/probe commands are passed as arguments.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== host: $(nproc) cpus"; rocminfo 2>/dev/null | grep -m1 -E "gfx9" || true
python -m granne_amd.build >/dev/null 2>&1
if ! timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; then
  echo "== pytest failed with DPP build:"; tail -25 gpurun_out/pytest_gpu.log
  GRANNE_HIP_USE_DPP=0 python -m granne_amd.build --force >/dev/null 2>&1
  echo "== rerun with shuffle build"
  timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_shfl.log 2>&1; tail -25 gpurun_out/pytest_gpu_shfl.log
else
  tail -3 gpurun_out/pytest_gpu.log
fi
for cmd in "$@"; do
  echo "== $cmd"
  timeout 900 bash -c "$cmd" 2>&1 | tail -40
done
