cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o; mkdir -p $O
timeout 600 python -m pytest tests/test_networks_gpu.py -q -x -k "launch_plan_replay" > $O/t_replay.txt 2>&1; tail -2 $O/t_replay.txt
for k in 0 2 4 8 16 32 64 0; do
PIXELPICK_PLAN_SIDE_DELAY=$k python bench.py --mode train --replay on --no-cpu-baseline --no-other-configs --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('delay $k', d['value'], d['ms_per_step'], d['train']['host_enqueue_ms_per_step'])" >> $O/delay.txt
done
python bench.py --mode train --replay off --no-cpu-baseline --no-other-configs --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('eager', d['value'], d['ms_per_step'], d['train']['host_enqueue_ms_per_step'])" >> $O/delay.txt
cat $O/delay.txt
