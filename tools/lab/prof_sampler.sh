# rocprofv3 kernel stats of an eager -n 256 generation with the CLI's default sampler (-t 0.5 -p 0.6): what the two sampler launches cost per token
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_smp; mkdir -p $out
cat > /tmp/smp_decode.py <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from llama_cu_awq_amd import api, synth
path = "/tmp/llama2_q4_synth_7b_seed20240229.bin"
if not os.path.exists(path): synth.write_model(path, "7b")
L = api.lib(); api.check(L.q4_set_device(0))
s = C.c_void_p(); api.check(L.q4_stream_create(C.byref(s))); L.q4_set_stream(s)
L.q4_set_use_graphs(2)
tr = api.Transformer(path, temperature=0.5, topp=0.6, seed=20240229)
print(tr.generate_ids([1, 2436, 385, 3686, 388, 1048, 22796, 118], 256)[1])
tr.close()
PY
timeout 120 python /tmp/smp_decode.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o k -- python /tmp/smp_decode.py > $out/t.log 2>&1
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-100s calls %6s avg %8.3f us min %8.3f max %8.3f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $out/t
