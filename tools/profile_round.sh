#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the eager bench (rocprofv3 crashes inside hipGraph capture, hence --no-graphs)
#   2. PMC passes over the dominant kernel alone (separate runs per counter, never combined with traces)
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu --no-graphs > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o ffn -- python tools/prof_kernel.py 0 32 > $out/pmc_$c.log 2>&1
done
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o ffn -- python tools/prof_kernel.py 0 32 > $out/pmc_sq.log 2>&1
find $out -name "*.csv" | head -20
grep -h "^{" $out/trace.log | tail -1 | cut -c1-400
