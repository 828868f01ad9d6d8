cd $GRAFT_REPO_ROOT
for r in auto on off; do python bench.py --mode train --no-cpu-baseline --replay $r 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('$r', d['value'], d['ms_per_step'], t['replay'], t.get('replay_reason'), round(t['host_enqueue_ms_per_step'],2))"; done
