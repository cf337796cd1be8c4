python3 tools/ab_conv.py --libs savsr_amd/csrc/libsavsr_hip.so savsr_amd/csrc/libsavsr_hip_exp_v1.so --shapes 1x64 6x64 2x192 > gpurun_out/r06_ab_conv_direct_fast_epilogue.log 2>&1; grep -v "probe windows" gpurun_out/r06_ab_conv_direct_fast_epilogue.log
python3 tools/ab_conv.py --libs savsr_amd/csrc/libsavsr_hip.so savsr_amd/csrc/libsavsr_hip_exp_v1.so --shapes 1x64 6x64 --h 64 --w 112 > gpurun_out/r06_ab_conv_direct_fast_epilogue_small.log 2>&1; grep -v "probe windows" gpurun_out/r06_ab_conv_direct_fast_epilogue_small.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_network.py -x -q -m gpu 2>&1 | tail -3
run() { name=$1; shift; "$@" > gpurun_out/r06_dfe_$name.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/r06_dfe_$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d.get("clips_per_s"), d.get("batch1_ms_per_frame"))
PY
}
run A1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run old env SAVSR_LIB_PATH=$PWD/savsr_amd/csrc/libsavsr_hip_exp_v1.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run A2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run c5_new python bench.py --config 5 --steps 8 --warmup 1 --no-cpu-baseline
run c5_old env SAVSR_LIB_PATH=$PWD/savsr_amd/csrc/libsavsr_hip_exp_v1.so python bench.py --config 5 --steps 8 --warmup 1 --no-cpu-baseline
run c5_new2 python bench.py --config 5 --steps 8 --warmup 1 --no-cpu-baseline
