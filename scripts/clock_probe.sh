#!/bin/bash
# sample the shader clock while the reconstruction step runs back to back (DVFS evidence for DESIGN.md section 6)
python bench.py --steps 6000 --warmup 5 --no-cpu-baseline --no-edit --no-train > /tmp/bench_long.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 1 12); do rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|fclk|mclk" | tr '\n' ' '; echo; rocm-smi --showpower 2>/dev/null | grep -i -E "power" | head -2 | tr '\n' ' '; echo; sleep 0.4; done
wait $BP
tail -1 /tmp/bench_long.json | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms/step', r['ms_per_step'], 'tapgemm TF', r['roofline']['achieved'])"
echo "--- idle:"; rocm-smi --showclocks 2>/dev/null | grep -i sclk
scripts/ubench/mfma_peak | grep "random blocks/CU=1"
