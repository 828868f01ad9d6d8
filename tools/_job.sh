cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_blocks_gpu.py -x -q 2>&1 | tail -3
python tools/bn_bench.py > $O/bn_bench.txt 2>&1; cat $O/bn_bench.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python tools/measure_acq_traffic.py 2>&1 | tail -5
mkdir -p gpurun_out/prof_keep; cp profiles/acq_traffic.json gpurun_out/prof_keep/acq_traffic.json
