cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
timeout 2700 python -m pytest tests/ -q -m gpu > $O/tall.txt 2>&1; tail -5 $O/tall.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
