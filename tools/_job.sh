cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 300 ./tools/probe/hbm_rw > $O/hbm_rw.txt 2>&1; cat $O/hbm_rw.txt

