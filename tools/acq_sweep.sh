# acquisition tuning sweep (GPU box): bench.py --mode acq per configuration x (waves/SIMD, pixels/thread, block order)
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/acq_sweep}; mkdir -p $O
run() { # name, args...
  n=$1; shift
  python bench.py --mode acq --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; y=r['read_only_yardstick'] or {}
print('$n | %s | %.1f Mpix/s | kernel ms %.4f | frac %.4f | yard %.4f' % (' '.join(sys.argv[1:]), d['value'], r['kernel_ms_avg'], r['frac'], y.get('frac_of_peak',0)))" "$@" >> $O/sweep.txt
}
: > $O/sweep.txt
for t in "0 0" "3 8" "3 4" "259 4" "259 8" "515 4" "515 8" "514 4"; do set -- $t
  run cs1024 --classes 19 --height 1024 --width 2048 --strategy least_confidence --batch 8 --tune-occ $1 --tune-ppt $2
done
for t in "0 0" "2 8" "2 4" "3 4" "3 8" "4 4" "514 4" "515 4" "514 8"; do set -- $t
  run voc --classes 21 --height 320 --width 320 --strategy margin_sampling --batch 256 --tune-occ $1 --tune-ppt $2
done
for t in "0 0" "3 8" "3 4" "515 8" "515 4"; do set -- $t
  run camvid --classes 11 --height 360 --width 480 --strategy entropy --batch 128 --tune-occ $1 --tune-ppt $2
done
for t in "0 0" "3 8" "3 4" "515 8" "515 4" "2 8"; do set -- $t
  run cs256 --classes 19 --height 256 --width 512 --strategy entropy --batch 256 --tune-occ $1 --tune-ppt $2
done
for t in "0 0" "515 4" "3 4"; do set -- $t
  run cs512 --classes 19 --height 512 --width 1024 --strategy entropy --batch 32 --tune-occ $1 --tune-ppt $2
done
cat $O/sweep.txt
