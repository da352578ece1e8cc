cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/prof_r02
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_7b_2048 -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu --no-graphs --no-extra --model 7b --ntok 2048 > $out/trace_7b_2048.log 2>&1
python tools/summarize_profiles.py $out r02 > $out/summary2.log 2>&1; rm -rf $out/trace_7b_2048; grep -A9 "7b_2048" $out/summary2.log
