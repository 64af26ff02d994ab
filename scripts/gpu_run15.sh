cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r05h
HL="--steps 50 --warmup 10 --no-cpu-baseline --no-edit --no-train --no-full-ian"
for i in 1 2; do
( timeout 300 python bench.py $HL ) > gpurun_out/r05h/def$i.json 2> gpurun_out/r05h/def$i.err
( IAN_DEBUG=1 IAN_OPTS=tg_fuse_tune=1,tg_fuse_max_m=100000 timeout 300 python bench.py $HL ) > gpurun_out/r05h/fuse$i.json 2> gpurun_out/r05h/fuse$i.err
done
for f in def1 fuse1 def2 fuse2; do python -c "import json; d=json.loads(open('gpurun_out/r05h/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), d['ms_per_step'], d['roofline']['frac'])"; done
grep "ian_autotune" gpurun_out/r05h/fuse1.err | grep "n=64" | cut -c1-150
