"""Prototype: split the batch over two streams / two handles and let kernels of different layers overlap."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
from dino_amd.weights import synthetic_frames
cfg = ViTConfig(n_blocks=12); sd = procedural_state_dict(cfg)
def mk():
    m = DINOSeg(head="mlp", n_blocks=12, precision="bf16", arch=cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); return m.to("cuda:0")
m0, m1, m2 = mk(), mk(), mk()
fr = torch.from_numpy(synthetic_frames(32, 480, seed=1)).cuda()
def single():
    return m0.forward_frames(fr)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dual(nsplit=2):
    cur = torch.cuda.current_stream()
    outs = []
    for i, (m, s) in enumerate(((m1, s1), (m2, s2))):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(m.forward_frames(fr[i * 16:(i + 1) * 16]))
    cur.wait_stream(s1); cur.wait_stream(s2)
    return outs
for fn, name in ((single, "single B=32"), (dual, "2 streams x B=16"), (single, "single B=32"), (dual, "2 streams x B=16")):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"{name}: {t*1e3:.2f} ms/step  {32/t:.0f} fps", flush=True)
