cd $GRAFT_REPO_ROOT
python bench.py --mode acq --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['roofline'], indent=1))"
