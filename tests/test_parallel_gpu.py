"""GPU: the data-parallel step with the REAL engine (gradients written by the backward kernels into the flat
buffer, all-reduce launched from inside backward on the side stream), two ranks sharing the one visible GPU over
gloo (gloo moves device tensors through the host; NCCL refuses two ranks on one device)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vit_pytorch_amd import ViT
        from vit_pytorch_amd.parallel import DataParallel
        cfg = dict(image_size=32, patch_size=8, num_classes=10, dim=64, depth=2, heads=2, mlp_dim=128)
        torch.manual_seed(100 + rank)                                   # different init: broadcast must align the ranks
        model = ViT(**cfg).to("cuda", dtype=torch.bfloat16)
        dp = DataParallel(model, layers_per_chunk=1)                       # one chunk per layer: exercises the chunked path
        torch.manual_seed(200 + rank)
        x = torch.randn(4, 3, 32, 32, device="cuda").to(torch.bfloat16)
        # local gradients without the wrapper
        model.zero_grad(set_to_none=True)
        model(x).float().square().mean().backward()
        local = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
        dp.backward(dp(x).float().square().mean())
        assert dp.sink._early_launched and dp.sink._late_launched           # all collectives left from inside backward
        assert dp.sink._cursor == dp.sink.boundary
        assert len(dp.sink._filled) == len([p for p in dp.sink.params if p.numel()])  # every gradient went straight to the sink
        worst = 0.0
        for i, (n, p) in enumerate(model.named_parameters()):
            if not p.numel():
                continue
            assert p.grad.data_ptr() in dp.sink._view_ptrs
            avg = local[n].clone()
            dist.all_reduce(avg)
            avg /= world
            err = ((p.grad.float() - avg).norm() / (avg.norm() + 1e-12)).item()
            worst = max(worst, err)
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


def test_engine_data_parallel_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(2)]
    assert all(w < 2e-2 for _, w in res), res   # bf16 average of two bf16 gradients
