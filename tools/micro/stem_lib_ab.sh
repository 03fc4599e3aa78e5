# same-box A/B of the F(4x4,3x3) stem between builds of the library:  bash tools/micro/stem_lib_ab.sh <name> [<name> ...]
# (tools/ab_build.sh <rev|WORKTREE> <name> ["-DS4_DIAG=..."]; the in-tree library is always included as "worktree")
for i in 1 2; do for l in "" "$@"; do if [ -n "$l" ]; then export GFC_AMD_LIB=tools/ab_libs/libgfc_amd_$l.so; else unset GFC_AMD_LIB; fi; echo "lib ${l:-worktree}: $(python tools/micro/stem_ab.py 2>/dev/null | grep "^f43" | tail -1)"; done; done
