cd $GRAFT_REPO_ROOT
run() { python bench.py --mode acq --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; y=r.get('read_only_yardstick') or {}
print('$*', '| kernel ms', r['kernel_ms_avg'], '| frac', r['frac'], '| acq/yard', y.get('acq_kernel_vs_yardstick'))"; }
run --classes 19 --height 1024 --width 2048 --strategy entropy --batch 8
run --classes 19 --height 1024 --width 2032 --strategy entropy --batch 8
run --classes 19 --height 1000 --width 2048 --strategy entropy --batch 8
run --classes 19 --height 512 --width 1024 --strategy entropy --batch 32
run --classes 19 --height 512 --width 1000 --strategy entropy --batch 32
run --classes 19 --height 256 --width 512 --strategy entropy --batch 128
run --classes 19 --height 256 --width 496 --strategy entropy --batch 128
