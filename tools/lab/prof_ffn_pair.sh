cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_fp; mkdir -p $out
timeout 120 python tools/prof_decode.py 7b 64 4 > /dev/null 2>&1
for lvl in 3 4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$lvl -o k -- python tools/prof_decode.py 7b 200 $lvl > $out/t$lvl.log 2>&1
  f=$(find $out/t$lvl -name "*kernel_stats.csv" | head -1)
  echo "== level $lvl"; tail -2 $out/t$lvl.log
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    print("%-110s calls %6s avg %8.3f us  total %7.2f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  cp "$f" gpurun_out/fp_kernel_stats_level$lvl.csv
  rm -rf $out/t$lvl
done
