cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p1; mkdir -p $O
ONLY=@2 timeout 600 python tools/gemm_pw_bench.py 2>&1 | tee $O/gemm_pw_2048.txt | cut -c1-260
