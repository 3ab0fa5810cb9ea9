#!/bin/bash
# A/B aid: build the product library of another git revision (default HEAD) as readsb_amd/csrc/libmodes_gpu_<name>.so;
# select it with MGPU_LIBRARY=libmodes_gpu_<name>.so (readsb_amd/binding.py).   usage: tools/build_variant.sh <name> [rev] [extra hipcc flags]
set -e
name=$1; rev=${2:-HEAD}; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" readsb_amd/csrc include | tar -x -C "$tmp"
make -s -C "$tmp/readsb_amd/csrc" libmodes_gpu.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $*"
cp "$tmp/readsb_amd/csrc/libmodes_gpu.so" "$root/readsb_amd/csrc/libmodes_gpu_$name.so"
rm -rf "$tmp"
echo "built readsb_amd/csrc/libmodes_gpu_$name.so from $rev"
