cd $GRAFT_REPO_ROOT
for s in 101 102 103 104 105; do timeout -k 10 300 python tools/fuzz_gpu.py $s 1000 2>&1 | tail -1; done
for s in 11 12 13 14 15 16 17 18; do timeout -k 10 300 python tools/fuzz_split.py $s 100 2>&1 | tail -1; done
timeout -k 10 300 python tools/stress_split.py 2>&1 | tail -3
