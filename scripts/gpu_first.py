"""First-contact GPU script: per-layer parity vs the oracle + rough timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from neural_photo_editor_amd import IAN
from oracle import ian_oracle as O
from oracle.torch_twin import TorchTwin

def rel(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

for arch in (sys.argv[1:] or ["IAN_simple", "IAN"]):
    cfg = os.path.join(ROOT, "neural_photo_editor_amd", "configs", arch + ".py")
    P = O.make_params(arch, 1)
    t = time.time(); m = IAN(cfg, True, params=P); print(arch, "init %.1fs" % (time.time() - t), flush=True)
    orc = O.Oracle(arch, P)
    n = 3
    x = O.make_images(n)
    z = m.encode_images(x)
    feats = orc.encoder_features(x)
    for i, f in enumerate(feats):
        print("  enc_conv%d" % (i + 1), rel(m.activation("enc_conv%d" % (i + 1), n), f))
    zr = orc.encode_images(x)
    print("  z", rel(z, zr))
    xh = m.sample_at(zr)
    for name, a in orc.decoder_activations(zr):
        nm = {"dec_fc2": "l_dec_fc2", "out": "l_out"}.get(name, name)
        if nm in m.lowered.slot_names:
            print("  ", name, rel(m.activation(nm, n), a))
    print("  xhat", rel(xh, orc.sample_at(zr)), flush=True)
    if arch == "IAN_simple":
        tw = TorchTwin(arch, P)
        z1 = O.make_latents(1)
        rgb = np.zeros((1, 3, 64, 64), np.float32); rgb[:, 0] = 1; rgb[:, 1:] = -1
        print("  gradRGB", rel(m.imgradRGB(26, 26, 30, 30, rgb, z1), tw.imgradRGB(26, 26, 30, 30, rgb, z1)))
        print("  gradL", rel(m.imgrad(10, 20, 30, 40, z1), tw.imgrad(10, 20, 30, 40, z1)))
    # timing, device-resident
    for B in (1, 64) if arch == "IAN_simple" else (1, 32):
        xd = torch.from_numpy(O.make_images(B)).cuda(); out = torch.empty_like(xd)
        for _ in range(3): m.handle.call("ian_reconstruct", xd, B, out)
        torch.cuda.synchronize(); t = time.time(); K = 20
        for _ in range(K): m.handle.call("ian_reconstruct", xd, B, out)
        torch.cuda.synchronize(); dt = (time.time() - t) / K
        m.handle.profile_enable(True)
        for _ in range(5): m.handle.call("ian_reconstruct", xd, B, out)
        pr = m.handle.profile_read(); m.handle.profile_enable(False)
        tf = pr["tapgemm_flops"] / (pr["tapgemm_ms"] * 1e-3) / 1e12
        print("  B=%d  %.3f ms/step  %.0f recon/s | tapgemm %.3f ms/step (%.1f TF/s) total(dev) %.3f ms" % (
            B, dt * 1e3, B / dt, pr["tapgemm_ms"] / 5, tf, pr["total_ms"] / 5), flush=True)
    m.close()
