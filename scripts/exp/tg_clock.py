"""Average shader clock of the tap-GEMM launches of the batch-64 reconstruction step (libian_ablation.so only: workgroup 0 of every
tapgemm launch accumulates s_memtime cycles and 100 MHz s_memrealtime ticks over its K loop, kernels_tapgemm.hip TG_CLK_*).
  IAN_LIB=neural_photo_editor_amd/libian_ablation.so python scripts/exp/tg_clock.py [arch] [batch]  ->  gpurun_out/r06_tg_clock.json"""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from neural_photo_editor_amd import IAN, synthetic as O
arch = sys.argv[1] if len(sys.argv) > 1 else "IAN_simple"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = IAN(os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py"), True, params=O.make_params(arch, 1))
h = m.handle
lib = h.lib
x = torch.from_numpy(O.make_images(B, seed=100)).cuda()
out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
step = lambda: h.call("ian_reconstruct", x, B, out, stream=st)
step(); h.autotune(B, 1, stream=st)
for _ in range(20): step()
torch.cuda.synchronize()
v = (C.c_ulonglong * 2)()
assert lib.ian_debug_tg_clock(v, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): step()
e1.record(); torch.cuda.synchronize()
assert lib.ian_debug_tg_clock(v, 0) == 0
cyc, ticks = int(v[0]), int(v[1])
# box probe (register-only MFMA loop) for comparison: its own achieved rate gives its clock = rate / spec * 2.4 GHz
tf, us = C.c_double(), C.c_double()
lib.ian_box_probe.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
lib.ian_box_probe(900, 250, C.byref(tf), C.byref(us), C.c_void_p(st))
res = {"arch": arch, "batch": B, "ms_per_step": e0.elapsed_time(e1) / 200, "tapgemm_wg0_shader_cycles": cyc, "tapgemm_wg0_100MHz_ticks": ticks,
       "tapgemm_avg_shader_clock_GHz": cyc / (ticks * 10e-9) / 1e9 if ticks else None,
       "box_probe_tflops": tf.value, "box_probe_implied_clock_GHz": tf.value / 157.3 * 2.4}
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r06_tg_clock.json"), "w"), indent=1)
