"""CPU tests of the data-parallel fine-tune plumbing (gloo, world_size 2): contiguous batch sharding + bucketed
gradient all-reduce give the full-batch gradient.  Per-rank gradients come from the oracle's autograd (the native
step needs a GPU); what is under test is dino_amd/parallel.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dino_amd.parallel import allreduce_gradients, make_buckets, shard_batch, shard_bounds
from dino_amd.weights import ViTConfig, procedural_state_dict, synthetic_frames, synthetic_labels

CFG = ViTConfig(embed_dim=128, num_heads=2, n_blocks=1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_grads(frames, labels):
    from oracle import dinoseg_oracle as O
    W = O.to_torch(procedural_state_dict(CFG), requires_grad=True)
    loss = O.nll_loss(O.dinoseg_forward(O.preprocess(frames), W, CFG.num_heads), torch.from_numpy(labels))
    loss.backward()
    return float(loss), {k: v.grad.clone() for k, v in W.items()}


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    frames = synthetic_frames(4, 32, seed=3)
    labels = synthetic_labels(4, 16, 7, seed=4)
    x, y = shard_batch(torch.from_numpy(frames), torch.from_numpy(labels), rank, world)
    _, grads = _oracle_grads(x.numpy(), y.numpy())
    n = allreduce_gradients(list(grads.items()), world, bucket_bytes=64 << 10)
    if rank == 0:
        torch.save({"grads": grads, "n_coll": n}, os.path.join(tmp, "dp.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_gradient_equals_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(os.path.join(tmp_path, "dp.pt"))
    frames = synthetic_frames(4, 32, seed=3)
    labels = synthetic_labels(4, 16, 7, seed=4)
    _, full = _oracle_grads(frames, labels)
    assert got["n_coll"] >= 2                                        # several buckets, not one collective per tensor
    assert got["n_coll"] < len(full)
    for k, g in full.items():
        assert float((got["grads"][k] - g).abs().max()) <= 1e-6 * (float(g.abs().max()) + 1e-6) + 1e-9, k


def test_shard_bounds_and_buckets():
    assert [shard_bounds(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        shard_bounds(10, 0, 4)
    sd = procedural_state_dict(ViTConfig(n_blocks=3))
    named = [(k, torch.zeros(v.shape)) for k, v in sd.items()]
    buckets = make_buckets(named, 8 << 20)
    assert sum(len(b) for b in buckets) == 48 and 3 <= len(buckets) <= 6        # 22.1 MiB in <= 8 MiB buckets
    assert buckets[0][0][0] == "clf.layer_3.bias" and buckets[-1][-1][0] == "dino.cls_token"   # reverse order
    assert sum(g.numel() for b in buckets for _, g in b) == 5797903


# ---------------------------------------------------------------------------------------------------------------------
# DataParallelFineTuner.step itself (world 2, gloo), with a stub model whose fused_training_step is the oracle + torch
# autograd writing into the flat gradient buckets: after 2 steps every parameter equals the single-process run on the
# whole batch.  What is under test: sharding, bucket views as .grad, one collective per bucket (both forms), the 1/world
# scale, the optimiser hand-off.
class _OracleStubModel:
    def __init__(self):
        from oracle import dinoseg_oracle as O
        self.O = O
        self.W = O.to_torch(procedural_state_dict(CFG), requires_grad=True)
        self._bk = None
        self.waited = []

    def named_parameters(self):
        return list(self.W.items())

    def grad_buckets(self, bucket_bytes=8 << 20):
        from dino_amd import DINOSeg
        from dino_amd.parallel import make_flat_buckets
        if self._bk is None:
            self._bk = make_flat_buckets(self.named_parameters(), lambda n: DINOSeg.grad_stage(n, CFG.n_blocks), bucket_bytes)
            for n, p in self.W.items():
                p.grad = self._bk["views"][n]
        return self._bk["buckets"]

    def stream_wait_grad_stage(self, stage, stream):
        self.waited.append(stage)

    def fused_training_step(self, batch, batch_idx=0):
        x, y = batch
        self.grad_buckets(64 << 10)
        for b in self._bk["buckets"]:
            b["flat"].zero_()
        loss = self.O.nll_loss(self.O.dinoseg_forward(self.O.preprocess(x.numpy()), self.W, CFG.num_heads), y)
        grads = torch.autograd.grad(loss, list(self.W.values()))
        for (n, p), g in zip(self.W.items(), grads):
            self._bk["views"][n].copy_(g)
            assert p.grad.data_ptr() == self._bk["views"][n].data_ptr()
        return {"loss": loss.detach()}

    def configure_optimizers(self):
        return torch.optim.SGD(list(self.W.values()), lr=0.05)


def _tuner_worker(rank, world, port, tmp, collective):
    from dino_amd.parallel import DataParallelFineTuner
    if world > 1:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    frames = torch.from_numpy(synthetic_frames(4, 32, seed=3))
    labels = torch.from_numpy(synthetic_labels(4, 16, 7, seed=4))
    model = _OracleStubModel()
    tuner = DataParallelFineTuner(model, fused_optimizer=False, bucket_bytes=64 << 10, collective=collective)
    losses = [float(tuner.step(frames, labels)) for _ in range(2)]
    if rank == 0:
        torch.save({"params": {k: v.detach().clone() for k, v in model.W.items()}, "losses": losses,
                    "n_coll": tuner.last_collectives, "n_buckets": len(model.grad_buckets(64 << 10))},
                   os.path.join(tmp, f"tuner_{world}_{collective}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("collective", ["allreduce", "rs_ag"])
def test_fine_tuner_step_world2_equals_single_process(tmp_path, collective):
    _tuner_worker(0, 1, 0, str(tmp_path), collective)
    mp.spawn(_tuner_worker, args=(2, _free_port(), str(tmp_path), collective), nprocs=2, join=True)
    one = torch.load(os.path.join(tmp_path, f"tuner_1_{collective}.pt"))
    two = torch.load(os.path.join(tmp_path, f"tuner_2_{collective}.pt"))
    assert two["n_coll"] == two["n_buckets"] >= 2 and one["n_coll"] == 0
    for a, b in zip(one["losses"], two["losses"]):
        assert abs(a - b) <= 1e-6 * abs(a) + 1e-7
    for k, v in one["params"].items():
        assert float((two["params"][k] - v).abs().max()) <= 2e-6 * (float(v.abs().max()) + 1e-6), k


def test_flat_buckets_are_views_in_backward_order():
    from dino_amd import DINOSeg
    from dino_amd.parallel import make_flat_buckets
    cfg = ViTConfig(n_blocks=3)
    named = [(k, torch.zeros(v.shape)) for k, v in procedural_state_dict(cfg).items()]
    bk = make_flat_buckets(named, lambda n: DINOSeg.grad_stage(n, 3), 8 << 20)
    stages = [b["stage"] for b in bk["buckets"]]
    assert stages == sorted(stages) and stages[-1] == 4 and 3 <= len(stages) <= 6
    assert bk["buckets"][0]["names"][0] == "clf.layer_3.bias" and bk["buckets"][-1]["names"][-1] == "dino.cls_token"
    for b in bk["buckets"]:
        lo, hi = b["flat"].data_ptr(), b["flat"].data_ptr() + b["flat"].numel() * 4
        assert b["flat"].numel() * 4 <= (8 << 20) + 256
        for n in b["names"]:
            v = bk["views"][n]
            assert lo <= v.data_ptr() < hi and v.data_ptr() % 256 == lo % 256 and v.is_contiguous()
    assert DINOSeg.grad_stage("clf.layer_1.weight", 3) == 0 and DINOSeg.grad_stage("dino.blocks.2.mlp.fc1.bias", 3) == 1
    assert DINOSeg.grad_stage("dino.blocks.0.norm1.weight", 3) == 3 and DINOSeg.grad_stage("dino.pos_embed", 3) == 4


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus 2` with no launcher starts its own two ranks (a torch.distributed.run child, before anything touches
    a GPU), rank 0 prints the one JSON line, every rank is seen by the all-reduce of ones.  --dry-run: no model, gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for extra in ([], ["--config", "finetune"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"] + extra, env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        j = json.loads(lines[0])
        assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["dry_run"] is True
    # a failing rank makes the launcher exit non-zero
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--no-such-flag"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_scale_script_dry_run(tmp_path):
    """tools/scale.sh --dry-run: the one command that yields the 1 -> 8 GPU curve on an 8-GPU lease (headline replicas; the fine-tune
    step with both collective forms), here with N = 1, 2 through the real launcher on gloo -- one tagged JSON line per run."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "scale.jsonl"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SCALE_GPUS"] = "1 2"
    r = subprocess.run(["bash", os.path.join(root, "tools", "scale.sh"), "--dry-run", str(out)], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    rows = [json.loads(ln) for ln in open(out)]
    assert [x["scale_label"] for x in rows] == ["headline N=1", "headline N=2", "finetune allreduce N=1", "finetune allreduce N=2",
                                                "finetune rs_ag N=1", "finetune rs_ag N=2"]
    for x in rows:
        assert x["dry_run"] is True and x["ranks_seen"] == x["n_gpus"]
    assert [x["config"]["collective"] for x in rows] == [None, None, "allreduce", "allreduce", "rs_ag", "rs_ag"]


def test_finetuner_fixes_the_bucket_size_before_the_first_step():
    """with a non-default bucket_bytes the tuner must make the model build (and bind) buckets of THAT size before the
    first step, and refuse to reduce buckets the step did not write into."""
    import torch
    from dino_amd.parallel import DataParallelFineTuner, make_flat_buckets

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.zeros(1000))
            self.b = torch.nn.Parameter(torch.zeros(3000))
            self.asked = []
            self._bk = None

        def grad_buckets(self, bucket_bytes=8 << 20):
            self.asked.append(bucket_bytes)
            if self._bk is None or self._bk[0] != bucket_bytes:
                self._bk = (bucket_bytes, make_flat_buckets(list(self.named_parameters()), None, bucket_bytes))
            return self._bk[1]["buckets"]

        def bind(self):
            for n, p in self.named_parameters():
                p.grad = self._bk[1]["views"][n]

    m = Model()
    t = DataParallelFineTuner(m, fused_optimizer=True, bucket_bytes=8 << 10)
    assert m.asked == [8 << 10] and len(m.grad_buckets(8 << 10)) == 2
    m.bind()
    t.check_bound(m.grad_buckets(8 << 10))
    stale = make_flat_buckets(list(m.named_parameters()), None, 8 << 20)["buckets"]      # a second, unbound set
    with pytest.raises(RuntimeError):
        t.check_bound(stale)
