mkdir -p gpurun_out/r05
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r05/gpu_tests.log
python tools/error_growth.py 7b 32 2> gpurun_out/r05/error_growth.err | tee gpurun_out/r05/error_growth_7b.txt
python tools/sweep_knob.py 13b 15 0,-1 2>&1 | tee gpurun_out/r05/sweep_hold_13b.log
python tools/sweep_knob.py mistral7b 15 0,-1 2>&1 | tee gpurun_out/r05/sweep_hold_mistral.log
