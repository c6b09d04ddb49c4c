import time, torch, sys
sys.path.insert(0, ".")
from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
from dino_amd.parallel import DataParallelFineTuner
from dino_amd.weights import synthetic_frames, synthetic_labels
dev = torch.device("cuda", 0)
cfg = ViTConfig(n_blocks=3)
sd = procedural_state_dict(cfg)
model = DINOSeg(head="mlp", n_blocks=3, precision="bf16", arch=cfg, optimizer=torch.optim.Adam, lr=1e-3)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
model.to(dev); model.unfreeze_bb()
frames = torch.from_numpy(synthetic_frames(8, 480, seed=7)).to(dev)
labels = torch.from_numpy(synthetic_labels(8, 3600, 7, seed=8)).to(dev)
tuner = DataParallelFineTuner(model, fused_optimizer=True)
for _ in range(5): tuner.step(frames, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): tuner.step(frames, labels)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue per step %.3f ms, total per step %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): tuner.step(frames, labels)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
