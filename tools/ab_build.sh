#!/bin/bash
# Build a second copy of the library from another git revision of csrc/ for same-box A/B runs:
#   tools/ab_build.sh <git-rev> [name]   ->  glue-factory-colon_amd/libgfc_amd_<name>.so   (git-ignored, travels with gpurun)
#   GFC_AMD_LIB=glue-factory-colon_amd/libgfc_amd_<name>.so python tools/bench_kernels.py ...
# Timings from different GPU boxes differ by up to 10 %: only two libraries timed in one gpurun call are comparable.
set -euo pipefail
rev=${1:?git revision}
name=${2:-prev}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/gfc_ab.XXXXXX)
mkdir -p "$tmp/pkg/csrc" "$tmp/include"
git -C "$root" archive "$rev" glue-factory-colon_amd/csrc include | tar -x -C "$tmp"
cd "$tmp/glue-factory-colon_amd/csrc"
objs=()
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -c "$f" -o "${f%.hip}.o" &
  objs+=("${f%.hip}.o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/glue-factory-colon_amd/libgfc_amd_${name}.so" "${objs[@]}"
echo "$root/glue-factory-colon_amd/libgfc_amd_${name}.so"
rm -rf "$tmp"
