# same-box ablation timing of the second F(4x4,3x3) stem mapping (csrc/conv_wino43b.hip), libraries built with
#   for d in 1 2 4 8 16; do bash tools/ab_build.sh WORKTREE b4d$d "-DB4_DIAG=$d"; done
for l in "" b4d1 b4d2 b4d4 b4d8 b4d16; do if [ -n "$l" ]; then export GFC_AMD_LIB=tools/ab_libs/libgfc_amd_$l.so; else unset GFC_AMD_LIB; fi; echo "lib ${l:-full}: $(python tools/micro/stem_ab.py 2>/dev/null | grep "^f43b" | tail -1)"; done
