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
