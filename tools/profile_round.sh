#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. kernel trace + stats of the eager decode (rocprofv3 crashes inside hipGraph capture, hence --no-graphs):
#      7B -n 256 (the headline), 13B -n 256, 7B -n 2048
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs per counter, never combined with traces) over the same eager decodes:
#      every kernel of the network as the product launches it (7B -n 256, 13B -n 128, 7B -n 2048)
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py writes the summaries that go to profiles/.
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/prof_$tag
mkdir -p $out
# a fresh box clocks up during its first seconds of work (the first trace of a call read 3 % slow once): warm it first
timeout 120 python tools/prof_kernel.py 0 4000 > /dev/null 2>&1
timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu --no-extra > /dev/null 2>&1
for cfg in "7b 256" "13b 256" "7b 2048"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$1_$2 -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu --no-graphs --no-extra --model $1 --ntok $2 > $out/trace_$1_$2.log 2>&1
done
# PMC passes over the product's own eager decode (every launch of the network, the attention -> o-proj launch in all its forms):
# one counter per pass, nothing else traced
for cfg in "7b 256" "13b 128" "7b 2048"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$1_$2_$c -o k -- python tools/prof_decode.py $1 $2 > $out/pmc_$1_$2_$c.log 2>&1
  done
done
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/pmc_k0_sq -o k -- python tools/prof_kernel.py ${SQ_KERNEL:-11} 32 > $out/pmc_k0_sq.log 2>&1
python tools/summarize_profiles.py $out $tag > $out/summary.log 2>&1
# the raw traces are hundreds of MB: only the summaries and the logs travel back
rm -rf $out/trace_*/ $out/pmc_*/
tail -40 $out/summary.log
