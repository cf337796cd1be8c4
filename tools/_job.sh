python -m pytest tests -x -q -m gpu > gpurun_out/r06_t7_full.log 2>&1; tail -3 gpurun_out/r06_t7_full.log | cut -c1-200
python bench.py --config 5 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/r06_c5_px200k.json 2>/dev/null
SAVSR_CLIP_BATCH_MAX_PX=70400 python bench.py --config 5 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/r06_c5_px70k.json 2>/dev/null
python bench.py --config 5 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/r06_c5_px200k_b.json 2>/dev/null
python bench.py --config 3 --steps 10 --warmup 2 > gpurun_out/r06_bench_config3.json 2>/dev/null
python bench.py --config 4 --steps 10 --warmup 2 > gpurun_out/r06_bench_config4.json 2>/dev/null
python - <<'PY'
import json
for f in ("c5_px200k","c5_px70k","c5_px200k_b"):
    d=json.loads(open(f"gpurun_out/r06_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d.get("clips_per_s"))
for f in ("bench_config3","bench_config4"):
    d=json.loads(open(f"gpurun_out/r06_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], [(c["scale"], c["ms_per_frame"], c["satu_frac"]) for c in d["per_case"]][::6])
PY
