import sys, time
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
from dino_amd.weights import synthetic_frames
for L in (3, 12):
    for prec in ("bf16x3", "bf16"):
        cfg = ViTConfig(n_blocks=L); sd = procedural_state_dict(cfg)
        m = DINOSeg(head="mlp", n_blocks=L, precision=prec, arch=cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda:0")
        fr = synthetic_frames(1, 480, seed=1)[0]
        fdev = torch.from_numpy(fr[None]).cuda()
        for _ in range(5): m.forward_frames(fdev, want_logp=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): m.forward_frames(fdev, want_logp=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3): m.predict(fr)
        t2 = time.perf_counter()
        for _ in range(20): m.predict(fr)
        t3 = time.perf_counter()
        # graph capture
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m.forward_frames(fdev, want_logp=False)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = m.forward_frames(fdev, want_logp=False)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); t5 = time.perf_counter()
        print(f"L={L} {prec}: forward B=1 {1e3*(t1-t0)/50:.3f} ms eager, {1e3*(t5-t4)/50:.3f} ms graph; predict() {1e3*(t3-t2)/20:.3f} ms")
