cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_x3_gpu.py -q 2>&1 | tail -2
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python tools/train_bench.py > $O/trace.log 2>&1
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $f --list > $O/timeline_x3b.txt 2>&1
rm -rf $O/trace
head -5 $O/timeline_x3b.txt
grep -n "x3\|wgrad_dma_kernel<128, 128>\|wgrad_dma_kernel<64, 128>\|ce_lowres_bwd\|adam" $O/timeline_x3b.txt | sed -n 1,60p
