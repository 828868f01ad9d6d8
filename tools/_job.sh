cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o g -- python $GRAFT_REPO_ROOT/tools/lib_gemm_names.py > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/r6f/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = []
for r in rows:
    out.append((int(r['Start_Timestamp']), r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size','')), r.get('LDS_Block_Size', '')))
out.sort()
# collapse consecutive identical names
res = []
for t, n, d, g, w, l in out:
    if res and res[-1][0] == n and res[-1][3] == g:
        res[-1][1].append(d)
    else:
        res.append([n, [d], w, g, l])
with open('gpurun_out/r6f/names.txt', 'w') as fo:
    for n, ds, w, g, l in res:
        ds = sorted(ds)
        fo.write(f"{len(ds):3d} x med {ds[len(ds)//2]:8.1f} us  grid {g} wg {w} lds {l}  {n[:230]}\n")
P
cat gpurun_out/r6f/names.txt | cut -c1-300
