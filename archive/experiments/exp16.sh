# Timeline of a launch (diagnostic build -DCUKING_MFMA_TIMELINE).
cd $GRAFT_REPO_ROOT
tl() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass --no-check "$@" 2>&1 >/dev/null | grep timeline; }
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_TIMELINE $EXTRA" python -m cuking_amd.build --lib --force > /dev/null 2>&1
for n in ${SIZES:-10000}; do echo "== n=$n"; tl --samples $n --sites 100000 --kin-threshold 0.05 --steps 3 --warmup 1; done
python -m cuking_amd.build --lib --force > /dev/null 2>&1
