rocm-smi --showperflevel --showclocks --showpower --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30
echo "--- during bench"
(python bench.py --steps 12 --warmup 2 --no-cpu --no-extra > /tmp/b1.json 2>/dev/null &) 
sleep 8
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power" | head -8; sleep 1; done
wait; sleep 6
python -c "import json;d=json.load(open('/tmp/b1.json'));print('default', d['value'], d['tok_s_per_generation'])"
echo "--- setperflevel high"
rocm-smi --setperflevel high 2>&1 | tail -3
rocm-smi --showperflevel 2>&1 | grep -i perf
(python bench.py --steps 12 --warmup 2 --no-cpu --no-extra > /tmp/b2.json 2>/dev/null &)
sleep 8
for i in 1 2; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|power" | head -8; sleep 1; done
wait; sleep 8
python -c "import json;d=json.load(open('/tmp/b2.json'));print('high', d['value'], d['tok_s_per_generation'])"
rocm-smi --setperfdeterminism 2400 2>&1 | tail -2
python bench.py --steps 12 --warmup 2 --no-cpu --no-extra 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('determinism2400', d['value'], d['tok_s_per_generation'])"
rocm-smi --resetperfdeterminism 2>&1 | tail -1
rocm-smi --setperflevel auto 2>&1 | tail -1
