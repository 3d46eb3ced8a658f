# GPU box: the short-K GEMM microbench (Swin-T b256 stage 0 / 1 shapes) and the Swin-T bench, this tree against other builds
libs="$@"
for lib in "" $libs; do echo "== lib ${lib:-tree}"; IVIT_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} GB_M=802816 GB_SHAPES=sq0,sp0 python tools/gemm_bench.py 2>/dev/null | cut -c1-118; IVIT_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} GB_M=200704 GB_SHAPES=sq1,sp1,sf1 python tools/gemm_bench.py 2>/dev/null | cut -c1-118; done
bash tools/ab_multi.sh swin_tiny $libs
