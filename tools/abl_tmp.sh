set -e
python bench.py --variant 5 --steps 5 --warmup 2 --cpu-seconds 0 --samples 20000 > gpurun_out/abl_20k.log 2>&1
python -m cuking_amd.build --lib --tuning > gpurun_out/abl_build.log 2>&1
CUKING_MFMA_ABLATE=1 python bench.py --variant 5 --steps 10 --warmup 3 --cpu-seconds 0 --no-check > gpurun_out/abl_1.log 2>&1
CUKING_MFMA_ABLATE=2 python bench.py --variant 5 --steps 10 --warmup 3 --cpu-seconds 0 --no-check > gpurun_out/abl_2.log 2>&1
python bench.py --variant 5 --steps 10 --warmup 3 --cpu-seconds 0 > gpurun_out/abl_0.log 2>&1
for f in 20k 0 1 2; do python - "$f" <<'PY'
import json,sys
for l in open(f'gpurun_out/abl_{sys.argv[1]}.log'):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])
PY
done
