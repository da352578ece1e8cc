#!/bin/bash
# rocprofv3 kernel trace of the gate/up launch alone, wave-owned kernel vs LDS-DMA engine (with / without the next-slot prefetch),
# in ONE gpurun call (boxes differ by several per cent): tools/lab/profile_engine_ab.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/prof_engine_ab
mkdir -p $out
for rep in 1 2; do
for v in 0 1 5; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_$v -o k -- python tools/prof_kernel.py 0 2000 7b 11 $v > $out/t_$v.log 2>&1
  python - "$out/t_$v" $v <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemv_q4_kernel<2" in r["Kernel_Name"] or "ffn_engine_kernel" in r["Kernel_Name"]:
            rows.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = rows[32:]
rows.sort()
print("engine knob %s: %d launches, rocprofv3 kernel-trace avg %.3f us, median %.3f, min %.3f" % (sys.argv[2], len(rows), sum(rows) / len(rows), rows[len(rows) // 2], rows[0]))
PY
  rm -rf $out/t_$v
done
done
