"""CPU, 2 processes, gloo: the data-parallel step (flat buffer, staged all-reduce, averaging)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vit_pytorch_amd import engine as E
from vit_pytorch_amd.parallel import DataParallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _StagedFn(torch.autograd.Function):
    """Stand-in for the fused engine on CPU: writes its parameter gradient into the sink's buffer and
    announces the stage, exactly like engine.TransformerFn / PatchEmbedFn do on the GPU."""

    @staticmethod
    def forward(ctx, x, w, stage):
        ctx.save_for_backward(x, w)
        ctx.stage = stage
        return x * w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        buf = E._grad_buf(w)
        buf.copy_((g * x).sum(0))
        if E._sink() is not None:
            E._sink().stage_done(ctx.stage)
        return g * w, E._ret(buf), None


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.to_patch_embedding = torch.nn.Linear(6, 8)           # "late" stage by name
        self.transformer = torch.nn.Module()
        self.transformer.layers = torch.nn.ModuleList([torch.nn.ModuleList([torch.nn.Linear(8, 8)]) for _ in range(2)])
        self.transformer.scale = torch.nn.Parameter(torch.ones(8))
        self.mlp_head = torch.nn.Linear(8, 3)

    def forward(self, x):
        x = self.to_patch_embedding(x)
        for (lin,) in self.transformer.layers:
            x = torch.tanh(lin(x))
        x = _StagedFn.apply(x, self.transformer.scale, "transformer")
        return self.mlp_head(x)


def _worker(rank, world, port, staged):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(123 + rank)                               # different init per rank: broadcast must fix it
        model = _Toy() if staged else torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
        dp = DataParallel(model)
        ref = [p.detach().clone() for p in model.parameters()]
        gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
        dist.all_gather(gathered, ref[0])
        assert all(torch.equal(g, gathered[0]) for g in gathered)  # parameters identical after broadcast
        torch.manual_seed(1000 + rank)
        x = torch.randn(5, 6)
        loss = dp(x).square().mean()
        dp.backward(loss)
        # expected: average over ranks of the local gradients (recomputed without the wrapper)
        model.zero_grad()
        E.set_grad_sink(None)
        model(x).square().mean().backward()
        for p in model.parameters():
            local = p.grad.detach().clone()
            dist.all_reduce(local)
            local /= world
            p._expected = local
        dp.backward(dp(x).square().mean())
        for i, p in enumerate(dp.sink.params):
            assert p.grad.data_ptr() == dp.sink.views[i].data_ptr()           # .grad is a view of the flat buffer
            assert torch.allclose(p.grad, p._expected, atol=1e-6), i
        if staged:
            assert dp.sink._early_launched                                    # the overlapped (staged) path ran
        assert dp.sink.flat.numel() == dp.sink.total
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("staged", [False, True])
def test_data_parallel_two_ranks_gloo(staged):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, staged), nprocs=2, join=True)
