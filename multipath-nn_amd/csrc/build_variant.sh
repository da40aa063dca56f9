#!/bin/bash
# A/B builds: the same sources with extra flags into their own object directory and library
#   bash build_variant.sh <tag> <flags...>   ->  ../libmpnn_hip_<tag>.so   (select with MPNN_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")"
TAG=$1; shift
ROOT=$(cd ../.. && pwd)
mkdir -p build_$TAG
rm -f build_$TAG/*.o                      # (a failed compile must not leave an older variant's object for the link)
SRCS=$(sed -n 's/^SRCS := //p' Makefile)
PIDS=()
for f in $SRCS; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=${ARCH:-gfx950} -I$ROOT/include -munsafe-fp-atomics -Wno-unused-result \
      -mllvm -amdgpu-kernarg-preload-count=16 "$@" -c $f -o build_$TAG/${f%.hip}.o &
  PIDS+=($!)
done
for pid in "${PIDS[@]}"; do
  wait $pid || { echo "build_variant: a compile failed" >&2; exit 1; }
done
hipcc -shared -fPIC --offload-arch=${ARCH:-gfx950} build_$TAG/*.o -o ../libmpnn_hip_$TAG.so
echo built ../libmpnn_hip_$TAG.so
