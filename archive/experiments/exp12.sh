cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-seconds 0 --extra-configs none --no-clock-pass "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],3), round(d['value']/1e9,3))"; }
for i in 1 2; do for sw in 1 2; do echo "c1 sw=$sw: $(run --steps 30 --warmup 5 --xcd-swizzle $sw)"; done; done
for sw in 1 2; do for br in 5 8; do echo "c2 sw=$sw br=$br: $(run --config c2 --steps 3 --warmup 1 --xcd-swizzle $sw --band-rows $br)"; done; done
echo "c1 sw=2 br=5: $(run --steps 30 --warmup 5 --xcd-swizzle 2 --band-rows 5)"
echo "30k sw=1: $(run --samples 30000 --steps 5 --warmup 2 --xcd-swizzle 1)"; echo "30k sw=2: $(run --samples 30000 --steps 5 --warmup 2 --xcd-swizzle 2)"
for sw in 1 2; do for br in 0 5 17; do echo "staged c2 sw=$sw br=$br: $(CUKING_AMD_XCD_SWIZZLE=$sw CUKING_AMD_BAND_ROWS=$br cuking_amd/bin/cuking --synthetic=100000,100000 --output_uri /tmp/o2 --kin_threshold=0.0884 --num_gpus=1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['exchange_and_compute_seconds'])")"; done; done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_order or staged" 2>&1 | tail -2
