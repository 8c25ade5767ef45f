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


def _nccl_worker(port, q):
    """One rank, backend nccl (= RCCL): the collectives the 8-GPU run issues -- AVG all-reduce of bf16 slices of the flat
    buffer, launched from inside backward on the side stream -- run for real on the device.  With one rank the average
    is the identity, so the gradients must equal the unwrapped model's bit for bit; `world` is forced to 2 on the sink
    so that the chunked in-backward launches (a no-op at world 1) are taken."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from vit_pytorch_amd import ViT
        from vit_pytorch_amd.parallel import DataParallel
        cfg = dict(image_size=64, patch_size=8, num_classes=10, dim=128, depth=6, heads=2, mlp_dim=256)
        torch.manual_seed(1)
        model = ViT(**cfg).to("cuda", dtype=torch.bfloat16)
        x = torch.randn(8, 3, 64, 64, device="cuda").to(torch.bfloat16)
        model.zero_grad(set_to_none=True)
        model(x).float().square().mean().backward()
        local = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        dp = DataParallel(model)                     # default: one all-reduce per 3 layers
        dp.sink.world = 2                            # take the multi-rank code path on the 1-rank communicator
        for _ in range(2):                           # twice: the second step reuses streams / events / buffers
            dp.backward(dp(x).float().square().mean())
            torch.cuda.synchronize()
            assert dp.sink._early_launched and dp.sink._late_launched and dp.sink._cursor == dp.sink.boundary
            for n, p in model.named_parameters():
                if p.numel():
                    assert torch.equal(p.grad, local[n]), n
        q.put("ok")
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_in_backward_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"


@pytest.mark.gpu
def test_native_comm_entry_points_world_size_one():
    """vitk_comm_* (include/vitk.h): RCCL opened by the library itself, one rank -- the all-reduce is the identity for sum and
    for average, in f32 and in the library's 16-bit type, on the caller's stream.  (Two ranks need two GPUs: the gpurun boxes
    have one; the multi-rank exchange is covered through torch.distributed by the tests above and the gloo suite.)"""
    from vit_pytorch_amd.comm import NativeComm
    uid = NativeComm.unique_id()
    assert len(uid) == 128 and any(uid)
    c = NativeComm(0, 1, uid)
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(1 << 20, device="cuda").to(dtype)
        ref = x.clone()
        c.all_reduce(x, average=True)
        c.all_reduce(x, average=False)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = torch.ones(4096, device="cuda")
        c.all_reduce(y)
    s.synchronize()
    assert torch.equal(y, torch.ones_like(y))
    c.close()


# ---- torch's own DistributedDataParallel around the drop-in (what accelerate gives the reference: train_vit_decorr.py:74-78) ----------
def _ddp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        from vit_pytorch_amd import SimpleViT, ViT
        worst = 0.0
        for cls, cfg in ((ViT, dict(image_size=112, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, pool="mean")),   # M = 6 x 196 = 1176
                         (SimpleViT, dict(image_size=32, patch_size=8, num_classes=10, dim=64, depth=2, heads=2, mlp_dim=128))):
            torch.manual_seed(100 + rank)                               # DDP broadcasts rank 0's weights at construction
            model = cls(**cfg).to("cuda", dtype=torch.bfloat16)
            ddp = DDP(model, device_ids=[0])
            torch.manual_seed(200 + rank)
            x = torch.randn(6, 3, cfg["image_size"], cfg["image_size"], device="cuda").to(torch.bfloat16)
            for step in range(2):                                       # the second step runs on DDP's rebuilt buckets
                ddp.zero_grad(set_to_none=True)
                ddp(x).float().square().mean().backward()
            got = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.numel()}
            model.zero_grad(set_to_none=True)
            model(x).float().square().mean().backward()                 # local gradients, no wrapper
            for n, p in model.named_parameters():
                if not p.numel():
                    continue
                avg = p.grad.detach().float().clone()
                dist.all_reduce(avg)
                avg /= world
                worst = max(worst, ((got[n] - avg).norm() / (avg.norm() + 1e-12)).item())
        q.put((rank, worst))
    finally:
        dist.destroy_process_group()


def test_torch_ddp_wraps_the_drop_in_two_ranks_one_gpu():
    """torch.nn.parallel.DistributedDataParallel is what the reference's training script ends up with; the drop-in's fused autograd
    Functions must feed its per-parameter hooks like any module (a (0, dim) cls_token under pool='mean' included)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get(timeout=5) for _ in range(2)]
    assert all(w < 2e-2 for _, w in res), res
