#!/usr/bin/env python3
"""Headline benchmark of the DINOSeg hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks 12] [--batch 32] [--res 480]
                    [--precision fp16|bf16|bf16x3] [--no-cpu-baseline] [--dry-run]
                    [--config headline|parity|960|vitb|finetune]     (the other BASELINE.json configs, one JSON line each)

Metric (BASELINE.json): frames/sec of DINOSeg inference -- ViT-S/8 (12 blocks) + MLP head, 480x480 frames,
batch 32 per GPU, 16-bit MFMA operands / fp32 accumulation -- whole job over all N GPUs.  Default precision (round 4): fp16
operands for the linears and Q.K^T (v_mfma_f32_32x32x16_f16: the bf16 instruction's rate and the same 2.5 PFLOP/s dense peak),
bf16 for the probabilities and V -- same speed as the all-bf16 mode, ~6x closer to the fp32 reference; the line carries the all-bf16
mode's own frames/s and parity as `bf16_mode`, and `--precision bf16` times that mode as the headline.  One "step" = one forward
of the hot path (uint8 frames resident in HBM -> log-probs + argmax map) over one batch.  Frames are
independent, so N GPUs run N data-parallel replicas with no data-path collective ("weak" scaling); the
barrier + max-over-ranks timing follows the driver's contract.  `python bench.py --gpus N` with N > 1 and no
WORLD_SIZE in the environment starts the N ranks itself (a `python -m torch.distributed.run` child, before anything
touches the GPU); under torchrun it is one of the ranks.  The line records `ranks_seen` = an all-reduce of ones.

The JSON line also carries
  roofline     : the dominant kernel (fused attention, 61 % of the FLOPs) -- algorithmic FLOPs per launch /
                 its mean launch duration measured with HIP events on the forward's stream, in an untimed pass of the
                 same steps right after the timed loop (no event records inside the timed region).
  parity       : "mask argmax match vs ref": the golden fixture's frame through the timed precision (argmax_match,
                 max_abs_dlogp against the reference's log-probabilities); parity_mode = the same config in bf16x3
                 (the mode that meets argmax-identical / 1e-3) with its own frames/s and match; bf16_mode = the same in bf16.
  configs      : BASELINE.json configs [2] (@960 batch 8), [4] (ViT-B/8, batch 16 per GPU) and [3] (3-block fine-tune step, batch 8 per GPU)
                 as compact records (value, ms_per_step, dominant-kernel roofline, parity / gradient_parity), each measured by a child
                 `python bench.py --config ...` after the headline (N=1 only; --no-configs skips them).
  cpu_baseline : the oracle (oracle/dinoseg_oracle.py = CPU fp32 restatement of the reference path, kind "port")
                 timed on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only); --mode finetune:
                 the oracle's forward + loss + autograd backward of one frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0, "fp16": 2500.0, "fp16x3": 2500.0}   # dense bf16 / fp16 MFMA peak, MI355X_MICROARCH.md
DTYPE = {"fp16": "fp16 (linears, Q.K^T; P.V bf16; fp32 accumulate)", "bf16": "bf16", "bf16x3": "bf16x3 (bf16 hi+lo split, fp32 acc)",
         "fp16x3": "fp16x3 (fp16 hi+lo split, fp32 acc)"}


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


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(cfg, sd, r, warm=2, timed=10, budget_s=60.0):
    """the oracle (fp32 CPU restatement of the reference path, materialised attention included) on this
    box's host cores, B=1 frames of the same workload: `warm` warm-up + `timed` timed forwards, MEDIAN ms/frame, core count
    and CPU model stated.  Bounded: stops early (never below 3 timed runs) once budget_s seconds have gone."""
    import statistics

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
        for i in range(warm + timed):
            t0 = time.time()
            O.dinoseg_forward(x, W, cfg.num_heads)
            if i >= warm:
                times.append(time.time() - t0)
            if len(times) >= 3 and time.time() - t_start > budget_s:
                break
    med = statistics.median(times)
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": f"oracle fp32 forward (same unfused op order as the reference, materialised attention), B=1 {r}x{r} frame, "
                      f"ViT-S/8 L={cfg.n_blocks}, {warm} warm-up + {len(times)} timed runs, median {med * 1e3:.0f} ms "
                      f"(min {min(times) * 1e3:.0f}), torch {torch.__version__}, {cores} threads"}


def cpu_baseline_finetune(cfg, sd, r, warm=1, timed=5, budget_s=40.0):
    """The fine-tune step's CPU baseline: the oracle's forward + nll_loss + torch autograd backward over all 48 tensors (the reference's
    training_step, pl_torch_modules.py:258-268, without the optimizer update) on this box's host cores, ONE frame of the same
    workload; median of a bounded number of steps (kind "port")."""
    import statistics

    import torch
    from dino_amd.weights import synthetic_frames, synthetic_labels
    from oracle import dinoseg_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("DINOSEG_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    W = O.to_torch(sd, requires_grad=True)
    x = O.preprocess(synthetic_frames(1, r, seed=0))
    labels = torch.from_numpy(synthetic_labels(1, (r // 8) ** 2, cfg.n_classes, seed=1)).reshape(-1).long()
    times, t_start = [], time.time()
    for i in range(warm + timed):
        for t in W.values():
            t.grad = None
        t0 = time.time()
        O.nll_loss(O.dinoseg_forward(x, W, cfg.num_heads), labels).backward()
        if i >= warm:
            times.append(time.time() - t0)
        if len(times) >= 2 and time.time() - t_start > budget_s:
            break
    med = statistics.median(times)
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": f"oracle fp32 forward + nll_loss + autograd backward over all tensors (no optimizer update), B=1 {r}x{r} frame, "
                      f"ViT-S/8 L={cfg.n_blocks}, {warm} warm-up + {len(times)} timed steps, median {med * 1e3:.0f} ms, "
                      f"torch {torch.__version__}, {cores} threads"}


def golden_check(model, arch, blocks, res, batch=1):
    """"mask argmax match vs ref" half of BASELINE.json's metric: the frame of the committed golden fixture (captured from the
    reference, tests/golden/) through `model`; argmax_match = fraction of patches whose class equals the reference's,
    max_abs_dlogp over the fixture's log-probabilities.  The frame is run as a batch of `batch` copies, so that it goes through
    the kernels the timed loop uses (large-batch dispatch: fused MLP, persistent GEMMs, two streams), and every copy must give
    the same answer.  None when no fixture covers this configuration."""
    import numpy as np
    import torch
    from dino_amd.weights import synthetic_frames
    name = {("vit_small", 480): f"g3_vits8_L{blocks}_r480", ("vit_small", 960): f"g4_vits8_L{blocks}_r960",
            ("vit_base", 480): f"g7_vitb8_L{blocks}_r480"}.get((arch, res))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz") if name else None
    if not path or not os.path.exists(path):
        return None
    g = np.load(path)
    frame = torch.from_numpy(synthetic_frames(1, res, seed=int(g["frame_seed"]))).to(model.device)
    frames = frame.expand(batch, *frame.shape[1:]).contiguous()
    logp_all, amax_all = model.forward_frames(frames, want_logp=True)
    n = logp_all.shape[0] // batch
    logp_b = logp_all.float().reshape(batch, n, -1)
    amax_b = amax_all.reshape(batch, n)
    copies_identical = bool((logp_b == logp_b[:1]).all().item()) and bool((amax_b == amax_b[:1]).all().item())
    logp, amax = logp_b[-1].cpu(), amax_b[-1].cpu().long()        # the last copy: second half-batch when the forward is split
    ref_arg = torch.from_numpy(g["argmax"].astype(np.int64))
    if "logp" in g.files:
        err = float((logp - torch.from_numpy(g["logp"])).abs().max())
    else:
        err = float((logp[torch.from_numpy(g["rows"])] - torch.from_numpy(g["logp_rows"])).abs().max())
    flips = int((amax != ref_arg).sum())
    return {"fixture": name, "argmax_match": round(1.0 - flips / ref_arg.numel(), 6), "argmax_flips": flips,
            "patches": int(ref_arg.numel()), "max_abs_dlogp": float(f"{err:.3e}"), "batch": batch,
            "copies_identical": copies_identical}


def golden_grad_check(precision, dev):
    """The fine-tune step's gradients in `precision` against the reference ViT + torch autograd (fixture G12: 3 blocks, one frame @480,
    all 48 tensors): |loss - ref|, the worst |norm ratio - 1| over the tensors, and the worst RMS error over the fixture's 64 sampled
    entries relative to the tensor's RMS entry (~ ||g - g_ref|| / ||g_ref||).  The same quantities tests/test_train_gpu.py bounds."""
    import numpy as np
    import torch
    from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
    from dino_amd.weights import synthetic_frames, synthetic_labels
    path = os.path.join(ROOT, "tests", "golden", "g12_finetune_r480_ignore.npz")
    if not os.path.exists(path):
        return None
    g, tag = np.load(path), "vits8_L3_r480_B1"
    cfg = ViTConfig(n_blocks=3)
    m = DINOSeg(head="mlp", n_blocks=3, precision=precision, arch=cfg, optimizer=torch.optim.Adam, lr=1e-3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in procedural_state_dict(cfg).items()}, strict=True)
    m.to(dev)
    m.unfreeze_bb()
    out = m.fused_training_step((torch.from_numpy(synthetic_frames(1, 480, seed=121)).to(dev),
                                 torch.from_numpy(synthetic_labels(1, 3600, cfg.n_classes, seed=122)).to(dev)), 0)
    worst_norm = worst_rel = 0.0
    for k, p in m.named_parameters():
        gv = p.grad.detach().cpu().reshape(-1)
        gn = float(g[f"{tag}|gnorm|{k}"])
        idx, ref = torch.from_numpy(g[f"{tag}|gidx|{k}"]), torch.from_numpy(g[f"{tag}|gval|{k}"])
        worst_norm = max(worst_norm, abs(float(gv.norm()) / gn - 1.0))
        worst_rel = max(worst_rel, float((gv[idx] - ref).pow(2).mean().sqrt()) / (gn / gv.numel() ** 0.5))
    return {"fixture": "g12_finetune_r480_ignore/" + tag, "abs_dloss": float(f"{abs(float(out['loss']) - float(g[tag + '|loss'])):.3e}"),
            "worst_norm_ratio_error": float(f"{worst_norm:.3e}"), "worst_sampled_relative_error": float(f"{worst_rel:.3e}"), "tensors": 48}


def bench_finetune(a, world, rank, dev, rehearsal=False):
    """Fine-tune step throughput: ViT-S/8 truncated to 3 blocks + MLP head, all 48 tensors trainable, Adam lr 1e-3
    (run_experiment.py:135-136), 480x480 frames, batch 8 per GPU (global 64 at 8 GPUs), bf16 operands unless
    --precision bf16x3 (the parity mode; fp16 is inference-only).  One step = forward + backward + gradient all-reduce + fused
    Adam.  The bf16 step's gradients are bounded against the reference in tests/test_train_gpu.py (G6 / G12)."""
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
    tuner = DataParallelFineTuner(model, fused_optimizer=True, collective=a.collective)
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
    cdev = torch.device("cpu") if rehearsal else dev
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    seen = ranks_seen(dist, world, cdev)
    fl = flops_per_frame(cfg.embed_dim, cfg.num_heads, blocks, a.res)
    fps = per_gpu * world * a.steps / elapsed
    # the dominant kernel group of the step, the flash-style attention backward (prep + dQ + dK,dV kernels: 29 % of the step), timed with
    # HIP events by the library in an untimed pass of the same steps; algorithmic work = the five products of the backward
    # (S, dP, dV, dK, dQ: 10 N^2 d per head; the two kernels recompute S and dP each -- 7 products issued -- which is not counted)
    roof = None
    if not rehearsal:
        model.profile(2)
        psteps = max(2, a.steps // 3)
        for _ in range(psteps):
            tuner.step(frames, labels)
        torch.cuda.synchronize()
        prof = model.profile_read()
        model.profile(0)
        ms, n = prof["attention_bwd"]
        if n:
            ntok = (a.res // 8) ** 2 + 1
            gf = 10.0 * per_gpu * cfg.num_heads * ntok * ntok * 64 / 1e9
            ach = gf / (ms / n)          # GFLOP / ms = TFLOP/s
            peak = MFMA_PEAK_TFLOPS[a.precision]
            roof = {"bound": "mfma", "kernel": "flash attention backward (attention_bwd.hip: attn_bwd_prep + attn_bwd_dq + attn_bwd_dkv kernels, "
                                               "recompute from the forward's log-sum-exp)",
                    "achieved": round(ach, 1), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "gflop_per_launch": round(gf, 1), "avg_launch_ms": round(ms / n, 4),
                    "launches_timed": n, "traffic": None,
                    "note": "algorithmic FLOPs (5 products); the kernels issue 7 products' worth of MFMAs"
                            + (" x 3 (hi+lo planes)" if a.precision == "bf16x3" else "")}
    if rank == 0:
        grad_parity = None if rehearsal else golden_grad_check(a.precision, dev)
        print(json.dumps({
            "metric": "frames/sec (480x480, ViT-S/8 x3 blocks) DINOSeg fine-tune step", "value": round(fps, 2),
            "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic", "final_loss": round(float(loss), 5),
            "ranks_seen": seen, "rehearsal": rehearsal,
            "config": {"workload": f"fine-tune step: ViT-S/8 x{blocks} blocks + MLP head unfrozen (48 tensors), fwd+bwd+"
                                   f"grad all-reduce+fused Adam, {a.res}x{a.res}, batch {per_gpu}/GPU", "blocks": blocks,
                       "batch_per_gpu": per_gpu, "global_batch": per_gpu * world, "resolution": a.res,
                       "precision": a.precision, "collective": a.collective,
                       "parallelism": f"dp{world} (RCCL gradient {'all-reduce' if a.collective == 'allreduce' else 'reduce-scatter + all-gather'}, 22.1 MiB fp32)"},
            "model_mfma_frac": round(fps / world * 3 * fl["total"] / 1e12 / MFMA_PEAK_TFLOPS[a.precision], 4),
            "roofline": roof, "gradient_parity": grad_parity,
            "cpu_baseline": cpu_baseline_finetune(cfg, sd, a.res) if world == 1 and not rehearsal and not a.no_cpu_baseline else None,
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=480)
    ap.add_argument("--arch", default="vit_small", choices=["vit_small", "vit_base"])
    ap.add_argument("--precision", default=None, choices=["fp16", "bf16", "bf16x3", "fp16x3"],
                    help="default: fp16 for inference (see the module docstring), bf16 for --mode finetune (fp16 is inference-only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="(kept for compatibility: kernel_ms_per_step is always emitted)")
    ap.add_argument("--option", action="append", default=[], help="library tuning knob key=int (dinoseg_set_option)")
    ap.add_argument("--mode", default="infer", choices=["infer", "finetune"],
                    help="infer: the headline metric; finetune: BASELINE configs[3] (3-block unfrozen step, batch 8/GPU, "
                         "gradient all-reduce over RCCL, fused Adam)")
    ap.add_argument("--config", default=None, choices=["headline", "parity", "960", "vitb", "finetune"],
                    help="BASELINE.json configs: headline = [1] ViT-S/8 @480 batch 32 (default); parity = the same in a hi+lo mode (fp16x3 unless --precision bf16x3); "
                         "960 = [2] @960 batch 8; vitb = [4] ViT-B/8 @480 batch 16/GPU; finetune = [3] 3-block step, batch 8/GPU")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the bf16x3 and bf16 sub-records of the headline line")
    ap.add_argument("--no-two-stream", action="store_true", help="skip the one-stream sub-record")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` sub-records of the headline line (BASELINE.json configs [2], [3], [4], each in a child process)")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2],
                    help="0 = the library default (2: a batch of >= 8 frames runs as two half-batches on two streams); 1 / 2 force it")
    ap.add_argument("--collective", default="allreduce", choices=["allreduce", "rs_ag"],
                    help="--mode finetune: one all-reduce per gradient bucket, or reduce-scatter + all-gather (dino_amd/parallel.py)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: no model, no GPU (gloo); prints the JSON line with value 0")
    a = ap.parse_args(argv)
    if a.config == "parity":
        a.precision = a.precision or "fp16x3"
    elif a.config == "960":
        a.res, a.batch = 960, 8
    elif a.config == "vitb":
        a.arch, a.batch = "vit_base", 16
    elif a.config == "finetune":
        a.mode = "finetune"
    if a.precision is None:
        a.precision = "bf16" if a.mode == "finetune" else "fp16"
    if a.mode == "finetune" and a.precision in ("fp16", "fp16x3"):
        ap.error("the fp16 precisions are inference-only (fp16 gradients would need loss scaling): use bf16 or bf16x3")
    return a


def other_configs(a):
    """BASELINE.json configs [2] (@960 batch 8), [4] (ViT-B/8 batch 16 per GPU) and [3] (the 3-block fine-tune step, batch 8 per GPU) as
    compact sub-records of the headline line: each is this script run as a CHILD process with --config (one at a time, the parent idle
    meanwhile; started, never exec'ed), its own JSON line reduced to value / ms_per_step / the dominant kernel's roofline / parity."""
    import subprocess
    recs = {}
    for name in ("960", "vitb", "finetune"):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(max(5, a.steps // 2)), "--warmup", "3",
               "--no-cpu-baseline", "--no-parity-mode", "--no-two-stream", "--no-configs"]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}")
            d = json.loads(lines[-1])
            rec = {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype")}
            rec["workload"] = d["config"].get("workload")
            rl = d.get("roofline") or {}
            rec["roofline"] = {k: rl.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms")}
            for k in ("model_mfma_frac", "parity", "gradient_parity", "outputs_valid", "final_loss"):
                if k in d:
                    rec[k] = d[k]
        except Exception as e:          # a sub-record must never take the headline line with it
            rec = {"error": repr(e)[-400:]}
        rec["wall_s"] = round(time.time() - t0, 1)
        recs[name] = rec
    return recs


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpus():
    """GPUs this process would see, WITHOUT loading the HIP runtime: the KFD topology nodes that have SIMDs (CPU nodes have none),
    cut down by ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set; None when sysfs tells nothing."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(line.split(None, 1) for line in f if " " in line)
            n += int(props.get("simd_count", "0").strip()) > 0
        except (OSError, ValueError):
            return None
    if n == 0:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a `python -m torch.distributed.run` CHILD process and
    forward its exit code.  This parent never loads the HIP runtime: it counts GPUs from sysfs (`visible_gpus`; a box whose sysfs
    says nothing is treated as having enough, and the ranks fail loudly if it has not) -- and it is never replaced by exec.  Rank 0
    of the child prints the JSON line on the inherited stdout.  On a box with fewer than N GPUs the ranks rehearse the same code
    path on device 0 over gloo (DINOSEG_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device) and the line says so."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not a.dry_run:
        ngpu = visible_gpus()
        if ngpu is not None and ngpu < a.gpus and env.get("DINOSEG_BENCH_REHEARSAL") != "1":
            print(f"bench.py: {ngpu} GPU(s) visible for --gpus {a.gpus}: rehearsal mode (all ranks on device 0, gloo)", file=sys.stderr)
            env["DINOSEG_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def ranks_seen(dist, world, dev):
    """an all-reduce of ones over the job's process group (RCCL in a real multi-GPU run): proves every rank took part"""
    if world == 1:
        return 1
    import torch
    t = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return int(round(float(t.item())))


def main():
    a = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    if world != a.gpus:
        a.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if a.dry_run:
        # launcher / rendezvous / JSON plumbing without the model and without a GPU
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
        seen = ranks_seen(dist, world, torch.device("cpu"))
        if world > 1:
            dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "frames/sec (480x480, ViT-S/8) DINOSeg inference", "value": 0.0, "unit": "frames/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": 0.0, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "none", "dry_run": True,
                              "ranks_seen": seen, "config": {"workload": "dry run: launcher and rendezvous only", "mode": a.mode,
                                                             "collective": a.collective if a.mode == "finetune" else None}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from dino_amd import DINOSeg, ViTConfig, procedural_state_dict
    from dino_amd.weights import VIT_B8, VIT_S8, synthetic_frames

    # rehearsal of the N > 1 code path on a one-GPU box: DINOSEG_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo
    # (RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank over RCCL
    rehearsal = os.environ.get("DINOSEG_BENCH_REHEARSAL") == "1"
    if not rehearsal and world > 1 and torch.cuda.device_count() < world:
        # (sysfs may list GPUs this process cannot open; every rank sees the same count, so every rank takes the same decision)
        if rank == 0:
            print(f"bench.py: {torch.cuda.device_count()} usable GPU(s) for {world} ranks: rehearsal mode (all ranks on device 0, gloo)", file=sys.stderr)
        rehearsal = True
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from dino_amd import capi
    for kv in a.option:
        key, val = kv.split("=")
        capi.check(capi.lib().dinoseg_set_option(key.encode(), int(val)))
    if a.mode == "finetune":
        if rehearsal and world > 1 and not any(kv.startswith("train_streams=") for kv in a.option):
            # several processes on ONE device oversubscribe its hardware queues once each of them runs a side stream (measured:
            # 35 ms per step on one stream, 160-2000 ms with the side stream); one process per GPU -- the real layout -- keeps it
            capi.check(capi.lib().dinoseg_set_option(b"train_streams", 1))
        return bench_finetune(a, world, rank, dev, rehearsal)
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

    lib = capi.lib()
    streams = a.streams if a.streams else 2         # the library default: two half-batches on two streams from 8 frames on
    capi.check(lib.dinoseg_set_option(b"streams", streams))
    split = streams == 2 and a.batch >= 8

    def step():
        return model.forward_frames(frames, want_logp=True)

    def timed(nsteps):
        """barrier + synchronize on both sides, max over ranks (the driver's contract)"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            out = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if not rehearsal else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    # "mask argmax match vs ref" first: the fixture's frame through the timed precision and dispatch (a validation pass of the freshly
    # built model -- workspaces, packed weights and the resolution cache exist before the warm-up steps start)
    parity = golden_check(model, a.arch, a.blocks, a.res, a.batch) if rank == 0 else None
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    elapsed, (logp, amax) = timed(a.steps)
    seen = ranks_seen(dist, world, dev if not rehearsal else torch.device("cpu"))

    # per-kernel timing (HIP events on the forward's stream, recorded by the library) in a SEPARATE untimed pass right after
    # the timed one, on ONE stream: with two half-batches sharing the chip a launch's duration is not the kernel's own, and the
    # roofline object is defined on exclusive launches.  The timed loop has no event records in it.
    capi.check(lib.dinoseg_set_option(b"streams", 1))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    model.profile(2)
    prof_steps = max(3, a.steps // 2)
    for _ in range(prof_steps):
        lp1, am1 = step()
    torch.cuda.synchronize()
    prof = model.profile_read()
    model.profile(0)
    one_stream = None
    if split and not a.no_two_stream:
        # the same batch on one stream (what the roofline pass launches), timed for reference
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            lp1, am1 = step()
        torch.cuda.synchronize()
        tel = time.perf_counter() - t1
        one_stream = {"value": round(a.batch * prof_steps / tel, 2), "unit": "frames/s (this rank)", "steps": prof_steps,
                      "ms_per_step": round(tel / prof_steps * 1e3, 4),
                      "outputs_identical_to_two_streams": bool(torch.equal(lp1, logp) and torch.equal(am1, amax))}
    capi.check(lib.dinoseg_set_option(b"streams", streams))

    ok = bool(torch.isfinite(logp).all().item()) and bool(torch.equal(amax.long(), logp.argmax(1)))
    fl = flops_per_frame(cfg.embed_dim, cfg.num_heads, a.blocks, a.res)
    fps = a.batch * a.steps * world / elapsed
    peak = MFMA_PEAK_TFLOPS[a.precision]
    att_ms, att_n = prof["attention"]
    att_avg_ms = att_ms / max(att_n, 1)
    att_flops = fl["attention"] * a.batch           # algorithmic FLOPs of one attention launch (the whole batch: one stream)
    achieved = att_flops / (att_avg_ms * 1e-3) / 1e12 if att_n else None

    traffic = None
    clock = None
    measured_peak = None
    tpath = os.path.join(ROOT, "profiles", "attention_traffic.json")
    if os.path.exists(tpath):
        # HBM bytes per attention launch from the committed rocprofv3 PMC passes of this same command
        # (tools/profile_bench.sh + tools/summarize_profile.py); only quoted when the workload matches
        tj = json.load(open(tpath))
        if (tj.get("batch"), tj.get("resolution"), tj.get("precision")) == (a.batch, a.res, a.precision) \
                and a.arch == "vit_small":
            traffic = tj["hbm_bytes_per_launch"]
            clock = tj.get("clock_ghz_under_load")
        measured_peak = (tj.get("measured_mfma_peak_tflops") or {}).get("random_operands") if a.precision in ("bf16", "fp16") else None

    if rank == 0:
        # the instantiation the one-stream roofline pass launches (attention_z.hip's rule: 256-query workgroups from one round of
        # 128-query ones on)
        heads_ = 6 if a.arch == "vit_small" else 12
        wgs4 = (a.batch * heads_ + 7) // 8 * 8 * (((a.res // 8) ** 2 + 1 + 127) // 128)
        nw = 8 if wgs4 >= 4 * torch.cuda.get_device_properties(dev).multi_processor_count else 4
        # (kernels.h Options::attn_variant: from four rounds of workgroups on, bit 10 = the assembly tile loop, bit 16 = 64 queries per wave)
        av = next((int(kv.split("=")[1]) for kv in a.option if kv.startswith("attn_variant=")), 11 | 1024 | 65536)
        fmt_note = "; fp16 Q.K^T, bf16 P.V" if a.precision == "fp16" else ""
        f16 = 1 if a.precision in ("fp16", "fp16x3") else 0
        ncu_ = torch.cuda.get_device_properties(dev).multi_processor_count
        if nw == 8 and (av & 1024) and not (av & 64):
            z_symbol = (f"dseg::attn_fwd_za_kernel<{f16}, {2 if av & 65536 else 1}, 0, false> (attention_za.hip: fused "
                        f"QK^T-softmax-PV, head_dim 64, zero-reference softmax, tile loop = generated assembly pipeline, "
                        f"{64 if av & 65536 else 32} queries per wave{fmt_note})")
        else:
            z_symbol = (f"dseg::attn_fwd_z_kernel<1, 4, {nw}, {f16}> (attention_z.hip: fused QK^T-softmax-PV, "
                        f"head_dim 64, zero-reference softmax{fmt_note})")
        # hi + lo planes (attention_z.hip: attention_x3_za): from two rounds of 256-query workgroups on, the assembly body with three
        # MFMAs per product (attention_za.hip, X3); below that the reference-based kernel of attention.hip
        wgs256 = (a.batch * heads_ + 7) // 8 * 8 * (((a.res // 8) ** 2 + 1 + 255) // 256)
        if (av & 1024) and not (av & 64) and ((av & 2048) or wgs256 >= 2 * ncu_):
            x3_symbol = (f"dseg::attn_fwd_za_kernel<{f16}, 1, 0, true> (attention_za.hip: fused QK^T-softmax-PV, head_dim 64, hi+lo planes = "
                         f"three MFMAs per product, zero-reference softmax, tile loop = generated assembly pipeline"
                         f"{'; Q / K / ctx fp16 hi+lo, P and V bf16 hi+lo' if f16 else ''})")
        else:
            x3_symbol = (f"dseg::attn_fwd_kernel<2, 4, false, 3, {f16}> (attention.hip: fused QK^T-softmax-PV, head_dim 64, hi+lo planes)")
        attn_symbol = z_symbol if a.precision in ("bf16", "fp16") else x3_symbol
        metric_text = (f"frames/sec ({a.res}x{a.res}, ViT-{'S' if a.arch == 'vit_small' else 'B'}/8"
                       f"{'' if a.blocks == 12 else f' x{a.blocks} blocks'}) DINOSeg inference")
        out = {
            "metric": metric_text,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE[a.precision],
            "data": "synthetic", "ranks_seen": seen, "rehearsal": rehearsal,
            "config": {"workload": f"DINOSeg predict path: ViT-{'S' if a.arch == 'vit_small' else 'B'}/8 x{a.blocks} blocks + MLP "
                                   f"head, {a.res}x{a.res} uint8 frames, batch {a.batch}/GPU, frames resident in HBM",
                       "blocks": a.blocks, "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                       "resolution": a.res, "tokens": (a.res // 8) ** 2 + 1, "precision": a.precision,
                       "streams": 2 if split else 1,
                       "parallelism": f"dp{world} (independent replicas, no data-path collective)"},
            "outputs_valid": ok,
            "model_gflop_per_frame": round(fl["total"] / 1e9, 2),
            "model_mfma_frac": round(fps / world * fl["total"] / 1e12 / peak, 4),
            "roofline": {"bound": "mfma", "kernel": attn_symbol,
                         "achieved": None if achieved is None else round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": None if achieved is None else round(achieved / peak, 4),
                         "measured_how": "HIP events around every launch on the forward's stream, in an untimed pass of the same steps on "
                                         "ONE stream (exclusive launches) right after the timed loop; `value` above is timed with the "
                                         "library default (two half-batches on two streams from 8 frames on)",
                         "peak_note": "peak = dense bf16 / fp16 MFMA at the nominal 2.4 GHz (MI355X_MICROARCH.md: one rate).  clock_ghz_under_load = what "
                                      "this kernel was measured to hold (rocprofv3 PMC GRBM_GUI_ACTIVE / 8 / duration, committed in "
                                      "profiles/attention_traffic.json); peak_at_measured_clock scales the peak by it",
                         "clock_ghz_under_load": clock,
                         "peak_at_measured_clock": None if clock is None else round(peak * clock / 2.4, 1),
                         "frac_at_measured_clock": None if (clock is None or achieved is None) else round(achieved / (peak * clock / 2.4), 4),
                         # what a register-only MFMA loop sustains on this chip with non-trivial operand bits (tools/mfma_peak.py):
                         # the power-limited ceiling of ANY bf16 MFMA kernel here, 79 % of the nominal peak
                         "measured_mfma_peak": measured_peak,
                         "frac_of_measured_mfma_peak": None if (measured_peak is None or achieved is None) else round(achieved / measured_peak, 4),
                         "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/attention_traffic.json)",
                         "algorithmic_bytes_per_launch": 4 * a.batch * cfg.num_heads * ((a.res // 8) ** 2 + 1) * 64 * 2
                         * (2 if a.precision in ("bf16x3", "fp16x3") else 1),
                         "launches_timed": att_n, "avg_launch_ms": round(att_avg_ms, 4),
                         "gflop_per_launch": round(att_flops / 1e9, 1)},
            # every kernel class of one forward, same untimed one-stream pass (fc1_gemm holds the fused MLP launch when it applies)
            "kernel_ms_per_step": {k: round(v[0] / prof_steps, 4) for k, v in prof.items()},
        }
        if one_stream is not None:
            out["one_stream"] = one_stream
        out["parity"] = parity      # the timed precision / dispatch against the reference fixture (computed before the timed loop)
        def sub_mode(prec):
            """the same configuration in another precision: its own short timing (library defaults) and its parity record"""
            pm = DINOSeg(head="mlp", n_blocks=a.blocks, precision=prec, arch=cfg)
            pm.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            pm.to(dev)
            pm.set_resolution(a.res)
            for _ in range(2):
                pm.forward_frames(frames, want_logp=True)
            torch.cuda.synchronize()
            psteps = max(3, a.steps // 2)
            t1 = time.perf_counter()
            for _ in range(psteps):
                pm.forward_frames(frames, want_logp=True)
            torch.cuda.synchronize()
            pel = time.perf_counter() - t1
            pfps = a.batch * psteps / pel
            rec = {"precision": prec, "value": round(pfps, 2), "unit": "frames/s (this rank)", "steps": psteps,
                   "ms_per_step": round(pel / psteps * 1e3, 4), "streams": 2 if split else 1,
                   "parity": golden_check(pm, a.arch, a.blocks, a.res, a.batch)}
            del pm
            return rec, pfps
        if a.precision in ("bf16", "fp16") and not a.no_parity_mode:
            # north_star's bar (argmax identical, |dlogp| <= 1e-3) is met by the hi+lo modes: same config, own timing.  fp16x3 =
            # fp16 hi+lo planes (~22 bits, round 4): the parity mode with margin; bf16x3 = rounds 1-3's (~16 bits)
            for key, prec, text in (("parity_mode", "fp16x3", "fp16x3 (fp16 hi+lo operand planes, 3 MFMAs per product, fp32 accumulate)"),
                                    ("parity_mode_bf16x3", "bf16x3", "bf16x3 (bf16 hi+lo operand planes, 3 MFMAs per product, fp32 accumulate)")):
                rec, pfps = sub_mode(prec)
                rec["precision"] = text
                rec["mfma_issue_frac"] = round(3 * pfps * fl["total"] / 1e12 / peak, 4)
                out[key] = rec
        if a.precision == "fp16" and not a.no_parity_mode:
            # BASELINE.json names bf16: the all-bf16 mode's own numbers next to the fp16-operand headline (same kernels, same rate)
            out["bf16_mode"] = sub_mode("bf16")[0]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, a.res)
        # (never under a profiler: the children would inherit its preload, write into the same output names and mix their kernels
        #  into the statistics this line's roofline is read from -- ADVICE r5)
        profiled = bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or "rocprofiler" in os.environ.get("LD_PRELOAD", "") \
            or bool(os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD")) or bool(os.environ.get("ROCPROF_OUTPUT_PATH"))
        if world == 1 and not a.no_configs and not profiled and a.config in (None, "headline") and a.arch == "vit_small" and a.res == 480 and a.batch == 32:
            del model
            torch.cuda.empty_cache()
            out["configs"] = other_configs(a)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
