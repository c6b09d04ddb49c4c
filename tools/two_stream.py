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
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
DELAY = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # GPU cycles the second stream spins before its half (phase offset)
m0 = mk()
ms = [mk() for _ in range(NS)]
fr = torch.from_numpy(synthetic_frames(32, 480, seed=1)).cuda()
def single():
    return m0.forward_frames(fr)
ss = [torch.cuda.Stream() for _ in range(NS)]
per = 32 // NS
def dual():
    cur = torch.cuda.current_stream()
    outs = []
    for i, (m, s) in enumerate(zip(ms, ss)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            if i > 0 and DELAY:
                torch.cuda._sleep(DELAY * i)
            outs.append(m.forward_frames(fr[i * per:(i + 1) * per]))
    for s in ss: cur.wait_stream(s)
    return outs
for fn, name in ((single, "single B=32"), (dual, f"{NS} streams x B={per}"), (single, "single B=32"), (dual, f"{NS} streams x B={per}")):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    print(f"{name}: {t*1e3:.2f} ms/step  {32/t:.0f} fps", flush=True)
