cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
python bench.py --no-other-configs > $O/bench_line.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5q/bench_line.json'))
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline_mfma'],indent=1)[:3000])
PY
cd /tmp && export TMPDIR=/tmp
FWD_ONLY=1 MODES=1 VARS=0,11 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o x -- python $GRAFT_REPO_ROOT/tools/x3_bench.py > /dev/null 2>&1
FWD_ONLY=1 MODES=1 VARS=0,11 timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -o x -- python $GRAFT_REPO_ROOT/tools/x3_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for d in ('pmc','pmc2'):
    cf=glob.glob(f'gpurun_out/r5q/{d}/**/*counter_collection.csv',recursive=True)
    kf=glob.glob(f'gpurun_out/r5q/{d}/**/*kernel_trace.csv',recursive=True)
    if not cf: print('no',d); continue
    dur={}
    for r in csv.DictReader(open(kf[0])):
        dur[r['Dispatch_Id']]=(r['Kernel_Name'],int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cf[0])):
        n,du=dur.get(r['Dispatch_Id'],(r['Kernel_Name'],0))
        if 'conv_x3' in n:
            agg[n[25:50]][r['Counter_Name']].append(float(r['Counter_Value'])); agg[n[25:50]]['ns'].append(du)
    for n,c in agg.items():
        print(d,n,{k:round(sum(v)/len(v)) for k,v in c.items()})
PY
