mkdir -p gpurun_out/r05
for i in 1 2 3 4 5 6; do Q4_LIB_OVERRIDE=$PWD/llama_cu_awq_amd/libllama2_q4_nofix.so Q4_PROFILING_BUILD=1 python -m pytest tests/prof_cases.py -x -q -m gpu -k "kv_rings and long16k" 2>&1 | tail -1; done | tee gpurun_out/r05/memset_race_nofix.log
for i in 1 2 3 4 5 6; do Q4_PROFILING_BUILD=1 python -m pytest tests/prof_cases.py -x -q -m gpu -k "kv_rings and long16k" 2>&1 | tail -1; done | tee gpurun_out/r05/memset_race_fix.log
python tools/sweep_attn_hold.py 7b 0,1 0,100,140,180,240,300 2 2>&1 | tee gpurun_out/r05/sweep_attn_hold.log
