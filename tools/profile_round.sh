#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the eager decode (rocprofv3 crashes inside hipGraph capture, hence --no-graphs):
#      7B -n 256 (the headline), 13B -n 256, 7B -n 2048
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs per counter, never combined with traces) over every timed kernel
#      alone: 0 gate/up, 2 down, 3 qkv, 4 o-proj, 5 classifier, 6 attention (one-block and split-context)
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py writes the summaries that go to profiles/.
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
for cfg in "7b 256" "13b 256" "7b 2048"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$1_$2 -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu --no-graphs --no-extra --model $1 --ntok $2 > $out/trace_$1_$2.log 2>&1
done
for kid in 0 2 3 4 5 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_k${kid}_$c -o k -- python tools/prof_kernel.py $kid 32 > $out/pmc_k${kid}_$c.log 2>&1
  done
done
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/pmc_k0_sq -o k -- python tools/prof_kernel.py 0 32 > $out/pmc_k0_sq.log 2>&1
python tools/summarize_profiles.py $out $tag > $out/summary.log 2>&1
tail -40 $out/summary.log
