cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_train_kernels.py -m gpu -x -q -p no:cacheprovider -k "epilogue or colstats_step or fused_statistics" ) > gpurun_out/r05d/pytest_a.log 2>&1
tail -n 6 gpurun_out/r05d/pytest_a.log
cat > /tmp/ab.py <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from neural_photo_editor_amd import synthetic as O
from neural_photo_editor_amd.trainer import Trainer
B = 128
tr = Trainer(os.path.join(os.environ["GRAFT_REPO_ROOT"], "neural_photo_editor_amd", "configs", "IAN.py"), O.make_train_params(O.make_params("IAN", 1)), B)
rs = np.random.RandomState(0)
X = torch.from_numpy(O.make_images(B, seed=1)).cuda(); Z = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda(); eps = torch.from_numpy(rs.randn(B, 100).astype(np.float32)).cuda()
tr.autotune()
res = {0: [], 1: []}
for rep in range(3):
    for v in (0, 1):
        tr.set_option("fused_stats", v)
        for w in ("gen", "discrim"): tr.step(w, X, Z, eps, return_metrics=False)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            tr.step("gen", X, Z, eps, return_metrics=False); tr.step("discrim", X, Z, eps, return_metrics=False)
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t) / 3 * 1e3)
for v in (0, 1):
    print("fused_stats=%d: %s ms per G+D pair (median %.2f) -> %.4f of peak" % (v, " ".join("%.2f" % t for t in res[v]), float(np.median(res[v])), B * 79111800000.0 / (float(np.median(res[v])) * 1e-3) / 157.3e12))
PY
( timeout 600 python /tmp/ab.py ) > gpurun_out/r05d/ab_fused.log 2>&1
tail -n 3 gpurun_out/r05d/ab_fused.log
( time timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_reference_pinned.py -m gpu -x -q -p no:cacheprovider -k "train or step" ) > gpurun_out/r05d/pytest_b.log 2>&1
tail -n 6 gpurun_out/r05d/pytest_b.log
HL="--steps 50 --warmup 10 --no-cpu-baseline --no-edit --no-train --no-full-ian"
( timeout 300 python bench.py $HL ) > gpurun_out/r05d/bench_hl.json 2> gpurun_out/r05d/bench_hl.err
python -c "import json; d=json.loads(open('gpurun_out/r05d/bench_hl.json').read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])"
