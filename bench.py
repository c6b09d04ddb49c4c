#!/usr/bin/env python3
"""Headline benchmark of the DINOSeg hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks 12] [--batch 32] [--res 480]
                    [--precision bf16|bf16x3] [--no-cpu-baseline] [--profile-all]

Metric (BASELINE.json): frames/sec of DINOSeg inference -- ViT-S/8 (12 blocks) + MLP head, 480x480 frames,
batch 32 per GPU, bf16 operands / fp32 accumulation -- whole job over all N GPUs.  One "step" = one forward
of the hot path (uint8 frames resident in HBM -> log-probs + argmax map) over one batch.  Frames are
independent, so N GPUs run N data-parallel replicas with no data-path collective ("weak" scaling); the
barrier + max-over-ranks timing follows the driver's contract.

The JSON line also carries
  roofline     : the dominant kernel (fused attention, 61 % of the FLOPs) -- algorithmic FLOPs per launch /
                 its mean launch duration measured with HIP events on the forward's stream inside the timed steps.
  cpu_baseline : the oracle (oracle/dinoseg_oracle.py = CPU fp32 restatement of the reference path, kind "port")
                 timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0}   # dense bf16 MFMA peak, MI355X_MICROARCH.md


def flops_per_frame(D, H, L, r, head="mlp", C=7):
    """Algorithmic FLOPs (2*MAC) per frame, SURVEY.md §8d: patch-embed + L blocks + head."""
    n = (r // 8) ** 2
    N = n + 1
    F = 4 * D
    patch = 2 * n * 192 * D
    qkv = 2 * N * D * 3 * D
    attn = 2 * (2 * N * N * D)
    proj = 2 * N * D * D
    mlp = 2 * (2 * N * D * F)
    hd = 2 * n * (D * 200 + 200 * 100 + 100 * C) if head == "mlp" else 2 * n * D * C
    return {"total": patch + L * (qkv + attn + proj + mlp) + hd, "attention": attn, "block": qkv + attn + proj + mlp}


def cpu_baseline(cfg, sd, r, budget_s=20.0):
    """Oracle forward on host cores, B=1 frames of the same workload, bounded to ~budget_s seconds."""
    import torch
    from dino_amd.weights import synthetic_frames
    from oracle import dinoseg_oracle as O
    # the box's CPU share, not the host's core count: a 1-GPU box owns 16 cores of a 256-core host
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("DINOSEG_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    W = O.to_torch(sd)
    frames = synthetic_frames(1, r, seed=0)
    times = []
    t_start = time.time()
    with torch.no_grad():
        x = O.preprocess(frames)
        while True:
            t0 = time.time()
            O.dinoseg_forward(x, W, cfg.num_heads)
            times.append(time.time() - t0)
            if time.time() - t_start > budget_s or len(times) >= 8:
                break
    best = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": round(1.0 / best, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 forward, B=1 {r}x{r} frame, ViT-S/8 L={cfg.n_blocks}, {len(times)} runs "
                      f"(first = warm-up), best {best * 1e3:.0f} ms, torch {torch.__version__} CPU threads={cores}"}


def bench_finetune(a, world, rank, dev):
    """Fine-tune step throughput: ViT-S/8 truncated to 3 blocks + MLP head, all 48 tensors trainable, Adam lr 1e-3
    (run_experiment.py:135-136), 480x480 frames, batch 8 per GPU (global 64 at 8 GPUs), parity precision (bf16x3)
    unless --precision bf16.  One step = forward + backward + gradient all-reduce + fused Adam."""
    import torch
    import torch.distributed as dist
    from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
    from dino_amd.parallel import DataParallelFineTuner
    from dino_amd.weights import synthetic_frames, synthetic_labels
    blocks = 3 if a.blocks == 12 else a.blocks
    per_gpu = 8 if a.batch == 32 else a.batch
    cfg = ViTConfig(n_blocks=blocks)
    sd = procedural_state_dict(cfg)
    model = DINOSeg(head="mlp", n_blocks=blocks, precision=a.precision, arch=cfg, optimizer=torch.optim.Adam, lr=1e-3)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev)
    model.unfreeze_bb()
    n = (a.res // 8) ** 2
    frames = torch.from_numpy(synthetic_frames(per_gpu * world, a.res, seed=7)).to(dev)
    labels = torch.from_numpy(synthetic_labels(per_gpu * world, n, 7, seed=8)).to(dev)
    tuner = DataParallelFineTuner(model, fused_optimizer=True)
    for _ in range(a.warmup):
        tuner.step(frames, labels)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tuner.step(frames, labels)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fl = flops_per_frame(cfg.embed_dim, cfg.num_heads, blocks, a.res)
    fps = per_gpu * world * a.steps / elapsed
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec (480x480, ViT-S/8 x3 blocks) DINOSeg fine-tune step", "value": round(fps, 2),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic", "final_loss": round(float(loss), 5),
            "config": {"workload": f"fine-tune step: ViT-S/8 x{blocks} blocks + MLP head unfrozen (48 tensors), fwd+bwd+"
                                   f"grad all-reduce+fused Adam, {a.res}x{a.res}, batch {per_gpu}/GPU", "blocks": blocks,
                       "batch_per_gpu": per_gpu, "global_batch": per_gpu * world, "resolution": a.res,
                       "precision": a.precision, "parallelism": f"dp{world} (RCCL gradient all-reduce, 22.1 MiB fp32)"},
            "model_mfma_frac": round(fps / world * 3 * fl["total"] / 1e12 / MFMA_PEAK_TFLOPS[a.precision], 4),
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=480)
    ap.add_argument("--arch", default="vit_small", choices=["vit_small", "vit_base"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class (adds event overhead)")
    ap.add_argument("--option", action="append", default=[], help="library tuning knob key=int (dinoseg_set_option)")
    ap.add_argument("--mode", default="infer", choices=["infer", "finetune"],
                    help="infer: the headline metric; finetune: BASELINE configs[3] (3-block unfrozen step, batch 8/GPU, "
                         "gradient all-reduce over RCCL, fused Adam)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
    from dino_amd.weights import VIT_B8, VIT_S8, synthetic_frames

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from dino_amd import capi
    for kv in a.option:
        key, val = kv.split("=")
        capi.check(capi.lib().dinoseg_set_option(key.encode(), int(val)))
    if a.mode == "finetune":
        return bench_finetune(a, world, rank, dev)
    base = VIT_S8 if a.arch == "vit_small" else VIT_B8
    cfg = ViTConfig(embed_dim=base.embed_dim, num_heads=base.num_heads, n_blocks=a.blocks)
    sd = procedural_state_dict(cfg)
    model = DINOSeg(head="mlp", n_blocks=a.blocks, precision=a.precision, arch=cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.to(dev)
    model.set_resolution(a.res)

    # synthetic frames, already r x r (resize = identity), resident in HBM before the timed region
    frames = torch.from_numpy(synthetic_frames(a.batch, a.res, seed=1000 + rank)).to(dev)
    torch.cuda.synchronize()

    def step():
        return model.forward_frames(frames, want_logp=True)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    model.profile(2 if a.profile_all else 1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        logp, amax = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = bool(torch.isfinite(logp).all().item()) and bool(torch.equal(amax.long(), logp.argmax(1)))
    fl = flops_per_frame(cfg.embed_dim, cfg.num_heads, a.blocks, a.res)
    fps = a.batch * a.steps * world / elapsed
    peak = MFMA_PEAK_TFLOPS[a.precision]
    att_ms, att_n = prof["attention"]
    att_avg_ms = att_ms / max(att_n, 1)
    att_flops = fl["attention"] * a.batch                      # algorithmic FLOPs of one attention launch
    achieved = att_flops / (att_avg_ms * 1e-3) / 1e12 if att_n else None

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "attention_traffic.json")
    if os.path.exists(tpath):
        # HBM bytes per attention launch from the committed rocprofv3 PMC passes of this same command
        # (tools/profile_bench.sh + tools/summarize_profile.py); only quoted when the workload matches
        tj = json.load(open(tpath))
        if (tj.get("batch"), tj.get("resolution"), tj.get("precision")) == (a.batch, a.res, a.precision) \
                and a.arch == "vit_small":
            traffic = tj["hbm_bytes_per_launch"]

    if rank == 0:
        out = {
            "metric": "frames/sec (480x480, ViT-S/8) DINOSeg inference",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "bf16x3 (bf16 hi+lo split, fp32 acc)",
            "data": "synthetic",
            "config": {"workload": f"DINOSeg predict path: ViT-{'S' if a.arch == 'vit_small' else 'B'}/8 x{a.blocks} blocks + MLP "
                                   f"head, {a.res}x{a.res} uint8 frames, batch {a.batch}/GPU, frames resident in HBM",
                       "blocks": a.blocks, "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                       "resolution": a.res, "tokens": (a.res // 8) ** 2 + 1, "precision": a.precision,
                       "parallelism": f"dp{world} (independent replicas, no data-path collective)"},
            "outputs_valid": ok,
            "model_gflop_per_frame": round(fl["total"] / 1e9, 2),
            "model_mfma_frac": round(fps / world * fl["total"] / 1e12 / peak, 4),
            "roofline": {"bound": "mfma", "kernel": "attn_fwd_kernel (fused QK^T-softmax-PV, head_dim 64)",
                         "achieved": None if achieved is None else round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": None if achieved is None else round(achieved / peak, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/attention_traffic.json)",
                         "algorithmic_bytes_per_launch": 4 * a.batch * cfg.num_heads * ((a.res // 8) ** 2 + 1) * 64 * 2
                         * (2 if a.precision == "bf16x3" else 1),
                         "launches_timed": att_n, "avg_launch_ms": round(att_avg_ms, 4),
                         "gflop_per_launch": round(att_flops / 1e9, 1)},
        }
        if a.profile_all:
            out["kernel_ms_per_step"] = {k: round(v[0] / a.steps, 4) for k, v in prof.items()}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, a.res)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
