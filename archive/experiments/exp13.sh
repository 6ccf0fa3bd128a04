# Per-phase cycles of a k-step (diagnostic build -DCUKING_MFMA_STAMPS), configs[2].
cd $GRAFT_REPO_ROOT
CUKING_EXTRA_HIPFLAGS="-DCUKING_MFMA_STAMPS $EXTRA" python -m cuking_amd.build --lib --force > /dev/null 2>&1
python bench.py --config c2 --cpu-seconds 0 --extra-configs none --no-clock-pass --steps 2 --warmup 1 2>&1 >/dev/null | grep "mfma stamps"
python -m cuking_amd.build --lib --force > /dev/null 2>&1
