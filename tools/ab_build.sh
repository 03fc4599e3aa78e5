#!/bin/bash
# Build a second copy of the library from another git revision of csrc/ for same-box A/B runs:
#   tools/ab_build.sh <git-rev> [name]   ->  tools/ab_libs/libgfc_amd_<name>.so   (git-ignored, travels with gpurun)
#   GFC_AMD_LIB=tools/ab_libs/libgfc_amd_<name>.so python tools/bench_kernels.py ...
# Timings from different GPU boxes differ by up to 10 %: only two libraries timed in one gpurun call are comparable.
set -euo pipefail
#   tools/ab_build.sh WORKTREE <name> "-DSOME_DIAGNOSTIC=1"   builds the working tree with extra compiler flags
rev=${1:?git revision or WORKTREE}
name=${2:-prev}
extra=${3:-}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/gfc_ab.XXXXXX)
if [ "$rev" = WORKTREE ]; then
  mkdir -p "$tmp/glue-factory-colon_amd"
  cp -r "$root/glue-factory-colon_amd/csrc" "$tmp/glue-factory-colon_amd/csrc"
  rm -rf "$tmp/glue-factory-colon_amd/csrc/build"
  cp -r "$root/include" "$tmp/include"
else
  git -C "$root" archive "$rev" glue-factory-colon_amd/csrc include | tar -x -C "$tmp"
fi
cd "$tmp/glue-factory-colon_amd/csrc"
objs=()
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -c "$f" -o "${f%.hip}.o" &
  objs+=("${f%.hip}.o")
done
wait
mkdir -p "$root/tools/ab_libs"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/ab_libs/libgfc_amd_${name}.so" "${objs[@]}"
echo "$root/tools/ab_libs/libgfc_amd_${name}.so"
rm -rf "$tmp"
