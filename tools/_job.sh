cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests/test_checkpoint_format_gpu.py -x -q > $O/t2.txt 2>&1
tail -40 $O/t2.txt
