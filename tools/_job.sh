cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
python bench.py > gpurun_out/r3m/bench2.json 2> gpurun_out/r3m/bench2.err; tail -c 300 gpurun_out/r3m/bench2.json
