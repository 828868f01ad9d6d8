cd $GRAFT_REPO_ROOT
cp pixelpick_amd/libpixelpick_hip.so /tmp/new.so
for i in 1 2 3; do
cp tools/probe/lib_old.so pixelpick_amd/libpixelpick_hip.so; echo "old: $(STEPS=40 timeout 120 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-60)"
cp /tmp/new.so pixelpick_amd/libpixelpick_hip.so; echo "new: $(STEPS=40 timeout 120 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-60)"
done
