#!/bin/bash
# Pins the oracle to the RUNNING reference: builds oracle/ref_fixtures (a small crate that depends on
# /root/reference by path, feature `singlethreaded`) with cargo and writes its fixtures to
# oracle/_ref/fixtures/. tests/test_ref_fixtures.py consumes them; without them it reports
# "parity unpinned" (skip). Called by __graft_entry__.build() when /root/reference exists.
#
# Needs: cargo + rustc (edition 2018; the reference was released against Rust 1.40-1.45) and the
# reference's dependencies (Cargo.toml:30-44: byteorder, flate2, fxhash, madvise, memmap, ordered-float 1.0,
# owning_ref, parking_lot 0.10, pbr, rayon 1.3, serde_json, stream-vbyte 0.3.2, rand 0.7) from a registry,
# a vendor directory (`cargo vendor`) or $CARGO_HOME. None of that exists in this image (no Rust
# toolchain, no network): then the script says so and exits 0 -- building the checker must not fail the
# product's build. Reference sources are never copied: the crate links them where they lie.
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d /root/reference ]; then
  echo "build_ref: /root/reference is absent: nothing to pin against (parity stays unpinned)"; exit 0
fi
# a toolchain installed by rustup lives outside PATH in non-login shells
if ! command -v cargo >/dev/null 2>&1 && [ -x "$HOME/.cargo/bin/cargo" ]; then export PATH="$HOME/.cargo/bin:$PATH"; fi
if ! command -v cargo >/dev/null 2>&1; then
  echo "build_ref: no cargo/rustc in this image: the reference cannot be executed here (parity stays unpinned)."
  echo "build_ref: on a box with a Rust toolchain run:  bash oracle/build_ref.sh  &&  python -m pytest tests/test_ref_fixtures.py"
  exit 0
fi
mkdir -p "$OUT"
export CARGO_TARGET_DIR="$OUT/target"
if ! cargo build --release --manifest-path "$HERE/ref_fixtures/Cargo.toml" ${CARGO_OFFLINE:+--offline}; then
  echo "build_ref: cargo could not build the fixture emitter (dependencies unavailable?): parity stays unpinned"; exit 0
fi
# the Rust side of the boundary (rust/granne-hip: GpuGranne, impl Index, ...) has only ever been checked by regular
# expressions (tests/test_abi.py): where cargo exists, let the compiler read it
if cargo check --manifest-path "$HERE/../rust/granne-hip/Cargo.toml" ${CARGO_OFFLINE:+--offline}; then
  echo "build_ref: rust/granne-hip: cargo check ok"
else
  echo "build_ref: rust/granne-hip: cargo check FAILED (see above): the binding needs a maintainer's eye"
fi
rm -rf "$OUT/fixtures.tmp"
if "$OUT/target/release/granne-ref-fixtures" "$OUT/fixtures.tmp"; then
  rm -rf "$OUT/fixtures" && mv "$OUT/fixtures.tmp" "$OUT/fixtures"
  echo "build_ref: reference fixtures in $OUT/fixtures"
else
  echo "build_ref: the fixture emitter failed: parity stays unpinned"
fi
exit 0
