cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/prof_r04sq
mkdir -p $out
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/pmc_k0_sq -o k -- python tools/prof_kernel.py 0 32 > $out/pmc_k0_sq.log 2>&1
python tools/summarize_profiles.py $out r04 > $out/summary.log 2>&1
tail -3 $out/summary.log
rm -rf $out/pmc_k0_sq
