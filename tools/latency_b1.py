"""GPU box: single-frame latency of the reference's documented use -- `model.predict(frame)` (README.md:26-36, pl_torch_modules.py:276-300)
-- next to the bare forward, with the host-side pieces of predict() timed one by one.

    python tools/latency_b1.py [precisions, default bf16x3,fp16]
"""
import sys
import time

sys.path.insert(0, '/root/repo')
import numpy as np
import torch

from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
from dino_amd.weights import synthetic_frames

precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16x3", "fp16"]


def t_us(fn, n=200, sync=True):
    fn()
    if sync:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for L in (3, 12):
    for prec in precs:
        cfg = ViTConfig(n_blocks=L)
        sd = procedural_state_dict(cfg)
        m = DINOSeg(head="mlp", n_blocks=L, precision=prec, arch=cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.to("cuda:0")
        fr = synthetic_frames(1, 480, seed=1)[0]
        fdev = torch.from_numpy(fr[None]).cuda()
        for _ in range(5):
            m.forward_frames(fdev, want_logp=False)
        fwd = t_us(lambda: m.forward_frames(fdev, want_logp=False), 100)          # back to back: GPU-bound throughput
        for _ in range(3):
            m.predict(fr)
        pred = t_us(lambda: m.predict(fr), 50, sync=False)
        # the pieces of predict(), each synchronised on its own
        sig = t_us(m._sync_weights, 500, sync=False)
        full_sig = t_us(m._param_signature, 100, sync=False)
        up = t_us(lambda: torch.from_numpy(fr).unsqueeze(0).to("cuda:0"), 100)
        amax = m.forward_frames(fdev, want_logp=False)[1]
        torch.cuda.synchronize()
        down = t_us(lambda: amax.cpu(), 200, sync=False)
        low = amax.cpu().numpy().astype(np.int64).reshape(60, 60)
        kron = t_us(lambda: np.repeat(np.repeat(low, 8, axis=0), 8, axis=1), 200, sync=False)

        def one_sync():
            m.forward_frames(fdev, want_logp=False)
            torch.cuda.synchronize()
        fwd_sync = t_us(one_sync, 100, sync=False)                                  # one forward, host waits for it: latency
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            m.forward_frames(fdev, want_logp=False)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                m.forward_frames(fdev, want_logp=False)
        torch.cuda.synchronize()
        graph = t_us(g.replay, 100)
        print(f"L={L} {prec}: forward B=1 {fwd / 1e3:.3f} ms back-to-back, {fwd_sync / 1e3:.3f} ms launch-to-done, {graph / 1e3:.3f} ms graph replay; "
              f"predict() {pred / 1e3:.3f} ms = forward + {(pred - fwd_sync) / 1e3:.3f} ms  [weight check {sig:.0f} us (full signature "
              f"{full_sig:.0f}), upload {up:.0f}, argmax download {down:.0f}, 8x8 upsample on the host {kron:.0f}]", flush=True)
