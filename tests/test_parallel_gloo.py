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


# ---- world size 8, ViT-L depth: the chunk plan ---------------------------------------------------------------------------
class _SinkFn(torch.autograd.Function):
    """y = tanh(x * w) with the parameter gradient written into the sink's buffer and the stage announced, like the fused engine's
    Functions do (head -> final norm -> layers depth-1 .. 0, then ("transformer")).  late = True: the gradient is returned to
    autograd instead (a foreign module inside the stack): it reaches the flat buffer only in finish_step."""

    @staticmethod
    def forward(ctx, x, w, li, late):
        ctx.save_for_backward(x, w)
        ctx.li, ctx.late = li, late
        return torch.tanh(x * w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        y = torch.tanh(x * w)
        gz = g * (1 - y * y)
        gw = (gz * x).sum(0)
        sink = E._sink()
        if ctx.late:
            ret = gw
        else:
            buf = E._grad_buf(w)
            buf.copy_(gw)
            ret = E._ret(buf)
        if sink is not None and ctx.li >= 0:
            if sink.wants_layer(ctx.li):
                sink.stage_done("layer", ctx.li)
            if ctx.li == 0:
                sink.stage_done("transformer")
        return gz * w, ret, None, None


class _DeepToy(torch.nn.Module):
    """Parameter names laid out like ViT-L/16: 24 layers, final norm, head, and the late patch-embedding stage."""

    def __init__(self, depth=24, width=13, late_layer=None):          # width 13: slices of 13 elements, padded to 16 in the flat buffer
        super().__init__()
        self.to_patch_embedding = torch.nn.Linear(6, width)
        self.transformer = torch.nn.Module()
        self.transformer.layers = torch.nn.ModuleList()
        for _ in range(depth):
            blk = torch.nn.Module()
            blk.w = torch.nn.Parameter(1 + 0.1 * torch.randn(width))
            self.transformer.layers.append(blk)
        self.transformer.norm = torch.nn.Module()
        self.transformer.norm.weight = torch.nn.Parameter(1 + 0.1 * torch.randn(width))
        self.mlp_head = torch.nn.Module()
        self.mlp_head.weight = torch.nn.Parameter(1 + 0.1 * torch.randn(width))
        self.late_layer = late_layer

    def forward(self, x):
        x = self.to_patch_embedding(x)
        for li, blk in enumerate(self.transformer.layers):
            x = _SinkFn.apply(x, blk.w, li, li == self.late_layer)
        x = _SinkFn.apply(x, self.transformer.norm.weight, -1, False)
        return _SinkFn.apply(x, self.mlp_head.weight, -1, False)


def _worker8(rank, world, port, lpc, late_layer):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7)
        model = _DeepToy(late_layer=late_layer)
        dp = DataParallel(model, layers_per_chunk=lpc)
        sink = dp.sink
        plan = sink.chunk_plan()
        # the plan tiles the flat buffer exactly once, in order, on 8-element (16-byte) boundaries
        cur = 0
        for off, n, _ in plan:
            assert off == cur and n > 0 and off % 8 == 0
            cur += n
        assert cur == sink.total and sink.total == (24 + 2) * 16 + sink.total - sink.boundary
        in_backward = [c for c in plan if c[2].startswith("layers")]
        assert len(in_backward) == ((24 + lpc - 1) // lpc if lpc > 0 else 0)
        if lpc == 3:        # head + norm + layers 23, 22, 21 in the first message, then three layers each
            assert [n for _, n, _ in in_backward] == [5 * 16] + [3 * 16] * 7
        torch.manual_seed(100 + rank)
        x = torch.randn(4, 6)
        sink.log = []
        dp.backward(dp(x).square().mean())
        # every element is reduced exactly once -- except a gradient that arrived late: its slot went out with its chunk (stale
        # content) and is reduced again on its own from finish_step, which is the value that counts
        covered = torch.zeros(sink.total, dtype=torch.int32)
        for off, n in sink.log:
            covered[off:off + n] += 1
        expect = torch.ones(sink.total, dtype=torch.int32)
        if late_layer is not None:
            i = next(k for k, p in enumerate(sink.params) if p is model.transformer.layers[late_layer].w)
            expect[sink.offsets[i]:sink.offsets[i] + 13] += 1
        assert torch.equal(covered, expect), (late_layer, sorted(sink.log))
        assert [tuple(c[:2]) for c in plan] == sorted(sink.log)[:len(plan)] or late_layer is not None
        # result = average over ranks of the local gradients
        sink.log = None
        E.set_grad_sink(None)
        for p in model.parameters():
            p.grad = None
        model(x).square().mean().backward()
        exp = []
        for p in sink.params:
            g = p.grad.detach().clone()
            dist.all_reduce(g); g /= world
            exp.append(g)
        dp.backward(dp(x).square().mean())
        for i, p in enumerate(sink.params):
            assert torch.allclose(p.grad, exp[i], atol=1e-6), (i, late_layer)
        model_ms = sink.comm_model(world=8)
        assert len(model_ms) == len(plan) and all(m["ring_ms"] >= m["direct_ms"] > 0 for m in model_ms)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lpc,late_layer", [(3, None), (4, 13), (0, None), (24, 5)])
def test_chunk_plan_world_size_8_depth_24(lpc, late_layer):
    """ViT-L depth on 8 ranks (BASELINE config 3's topology): offsets of the in-backward chunks, every slice reduced exactly once,
    a gradient that arrives outside the sink after its chunk has gone out (late_layer) is reduced again on its own, result = average."""
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, lpc, late_layer), nprocs=8, join=True)


# ---- the REAL fused Transformer stage under DataParallel, 2 ranks (kernels replaced by the CPU doubles) --------------------------------
class _EngineModel(torch.nn.Module):
    """Named like the drop-in ViT (to_patch_embedding / transformer / mlp_head) so that the sink orders its segments as it does there;
    the transformer IS vit_pytorch_amd.vit.Transformer running engine.TransformerFn."""

    def __init__(self):
        super().__init__()
        from vit_pytorch_amd.vit import Transformer
        self.to_patch_embedding = torch.nn.Linear(24, 64)
        self.transformer = Transformer(dim=64, depth=4, heads=2, dim_head=32, mlp_dim=128)
        self.mlp_head = torch.nn.Linear(64, 5)

    def forward(self, x):
        return self.mlp_head(self.transformer(self.to_patch_embedding(x)).mean(dim=1))


def _engine_worker(rank, world, port):
    import _kernel_doubles as KD
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with KD.installed():
            torch.manual_seed(7)
            full = torch.randn(world * 3, 16, 24)                       # the global batch; rank r owns rows 3r .. 3r + 2
            torch.manual_seed(50 + rank)                                # different init per rank: the broadcast must fix it
            model = _EngineModel().to(torch.bfloat16)
            dp = DataParallel(model, layers_per_chunk=3)
            dp.sink.log = []
            x = full[3 * rank:3 * rank + 3].to(torch.bfloat16)
            dp.backward(dp(x).float().square().mean())
            # the stack's gradients left in two in-backward chunks (after layers 3 and 0) + the rest: several collectives, not one
            assert len(dp.sink.log) >= 3, dp.sink.log
            # ... which tile the flat buffer exactly once; the head here is a plain nn.Linear (autograd hands its gradients over after the
            # segment that holds its slots has gone out), so finish_step re-reduces exactly those two slots on their own
            tiles = [seg for seg in dp.sink.log if seg[1] > 320]
            assert [o for o, _ in tiles] == [0] + [o + n for o, n in tiles][:-1] and sum(n for _, n in tiles) == dp.sink.total, dp.sink.log
            assert sorted(n for _, n in dp.sink.log if n <= 320) == [5, 320], dp.sink.log
            got = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.sink.params, dp.sink.views))
            # expected: the same model on the WHOLE batch in one process (mean loss over 2x the samples = average of the rank losses)
            E.set_grad_sink(None)
            model.zero_grad(set_to_none=True)
            model(full.to(torch.bfloat16)).float().square().mean().backward()
            for n, p in model.named_parameters():
                ref = p.grad.detach().float()
                err = (got[n] - ref).norm() / ref.norm().clamp_min(1e-12)
                assert err < 3e-2, (n, float(err))                      # 16-bit gradients summed in a different order
    finally:
        dist.destroy_process_group()


def test_real_engine_stage_under_data_parallel_two_ranks_gloo():
    """engine.TransformerFn (host logic: gradient buffers from the sink, layer ticks, per-chunk stage announcements) with the kernels
    replaced by tests/_kernel_doubles.py, wrapped in DataParallel on 2 gloo ranks: gradients equal the single-process full-batch run."""
    port = _free_port()
    mp.spawn(_engine_worker, args=(2, port), nprocs=2, join=True)


class _WideEngineModel(torch.nn.Module):
    """As _EngineModel, at the smallest extents the fp8 GEMM kernels serve (dim 256, 1,024 token rows per rank)."""

    def __init__(self):
        super().__init__()
        from vit_pytorch_amd.vit import Transformer
        self.to_patch_embedding = torch.nn.Linear(24, 256)
        self.transformer = Transformer(dim=256, depth=2, heads=4, dim_head=64, mlp_dim=512)
        self.mlp_head = torch.nn.Linear(256, 5)

    def forward(self, x):
        return self.mlp_head(self.transformer(self.to_patch_embedding(x)).mean(dim=1))


def _fp8_engine_worker(rank, world, port):
    import _kernel_doubles as KD
    from vit_pytorch_amd.fp8 import enable_fp8
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with KD.installed() as calls:
            torch.manual_seed(9)
            full = torch.randn(world * 8, 128, 24)
            torch.manual_seed(70 + rank)
            model = _WideEngineModel().to(torch.bfloat16)
            dp = DataParallel(model, layers_per_chunk=1)
            enable_fp8(model)
            x = full[8 * rank:8 * rank + 8].to(torch.bfloat16)
            for _ in range(3):                                          # step 1 records, steps 2-3 run the twelve GEMMs of a layer on fp8 operands
                del calls[:]
                dp.backward(dp(x).float().square().mean())
            names = [c[0] for c in calls]
            assert names.count("gemm_nt_fp8_v2") == 8 * 2 and names.count("gemm_tn_fp8") == 4 * 2 and "gemm_tn_bf16" not in names
            st = model.transformer._fp8
            assert st.ready and st.bwd_ready
            got = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
            # every rank ends with the same (averaged) gradients although its scales come from its own shard's amax
            for n in ("transformer.layers.0.1.net.1.weight", "transformer.layers.1.0.to_qkv.weight"):
                gathered = [torch.zeros_like(got[n]) for _ in range(world)]
                dist.all_gather(gathered, got[n])
                assert all(torch.equal(t, gathered[0]) for t in gathered)
            # and they are the full-batch 16-bit gradients to fp8 accuracy
            E.set_grad_sink(None)
            enable_fp8(model, enabled=False)
            model.zero_grad(set_to_none=True)
            model(full.to(torch.bfloat16)).float().square().mean().backward()
            worst = 0.0
            for n, p in model.named_parameters():
                ref = p.grad.detach().float()
                worst = max(worst, float((got[n] - ref).norm() / ref.norm().clamp_min(1e-12)))
            assert worst < 1.5e-1, worst
    finally:
        dist.destroy_process_group()


def test_fp8_engine_stage_under_data_parallel_two_ranks_gloo():
    """The fp8 state machine (per-rank delayed scales, lean saving, fp8 weight gradients written into the sink's flat buffer) under
    DataParallel on 2 gloo ranks, kernels replaced by the CPU doubles."""
    port = _free_port()
    mp.spawn(_fp8_engine_worker, args=(2, port), nprocs=2, join=True)


# ---- a model called TWICE before one backward (siamese / DINO-style student passes, dino.py:283-290) under DataParallel ----------------
def _twice_worker(rank, world, port):
    import _kernel_doubles as KD
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with KD.installed():
            torch.manual_seed(17)
            full_a = torch.randn(world * 3, 16, 24); full_b = torch.randn(world * 3, 16, 24)
            torch.manual_seed(80 + rank)
            model = _EngineModel().to(torch.bfloat16)
            dp = DataParallel(model, layers_per_chunk=3)
            dp.sink.log = []
            xa = full_a[3 * rank:3 * rank + 3].to(torch.bfloat16); xb = full_b[3 * rank:3 * rank + 3].to(torch.bfloat16)
            loss = dp(xa).float().square().mean() + (dp(xb).float() - 1).square().mean()
            dp.backward(loss)
            # every fused parameter met the sink twice: the second gradients went through autograd and were reduced on their own
            fused = [i for i, p in enumerate(dp.sink.params) if any(p is q for q in model.transformer.parameters())]
            assert set(fused) <= dp.sink._multi and fused
            got = {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(dp.sink.params, dp.sink.views))
            E.set_grad_sink(None)
            model.zero_grad(set_to_none=True)
            (model(full_a.to(torch.bfloat16)).float().square().mean() + (model(full_b.to(torch.bfloat16)).float() - 1).square().mean()).backward()
            for n, p in model.named_parameters():
                ref = p.grad.detach().float()
                err = (got[n] - ref).norm() / ref.norm().clamp_min(1e-12)
                assert err < 3e-2, (n, float(err))
    finally:
        dist.destroy_process_group()


def test_model_called_twice_per_step_under_data_parallel_two_ranks_gloo():
    """The flat gradient buffer holds ONE slot per parameter and the backward kernels overwrite it: a second use of the model in
    the same step must not replace the first gradient (it did before round 4's fix: only the last call's gradient survived)."""
    port = _free_port()
    mp.spawn(_twice_worker, args=(2, port), nprocs=2, join=True)


# ---- torch's own DistributedDataParallel around the fused stage (what accelerate gives the reference), 2 ranks, CPU doubles ------------
def _torch_ddp_worker(rank, world, port):
    import _kernel_doubles as KD
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with KD.installed():
            torch.manual_seed(23)
            full = torch.randn(world * 3, 16, 24)
            torch.manual_seed(90 + rank)                                # DDP broadcasts rank 0's parameters at construction
            model = _EngineModel()
            ddp = DDP(model)
            x = full[3 * rank:3 * rank + 3]
            for _ in range(2):                                          # second step: DDP's rebuilt buckets
                ddp.zero_grad(set_to_none=True)
                ddp(x).float().square().mean().backward()
            got = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
            model.zero_grad(set_to_none=True)
            model(full).float().square().mean().backward()              # the whole batch in one process = the average of the rank losses
            for n, p in model.named_parameters():
                err = (got[n] - p.grad).norm() / p.grad.norm().clamp_min(1e-12)
                assert err < 1e-4, (n, float(err))
    finally:
        dist.destroy_process_group()


def test_torch_ddp_around_the_fused_stage_two_ranks_gloo():
    """DistributedDataParallel's per-parameter autograd hooks are fed by the fused Functions like by any module: gradients equal the
    single-process full-batch run (float32, doubles)."""
    port = _free_port()
    mp.spawn(_torch_ddp_worker, args=(2, port), nprocs=2, join=True)


def test_sink_defaults_follow_the_one_environment_variable(monkeypatch):
    """VITK_DP="chunk=…,prio=…,reserve=…,layers=…" (any subset) sets the data-parallel sink's defaults for runs that cannot pass arguments --
    the driver's `bench.py --gpus N`; explicit constructor arguments win.  (Round 6: one variable instead of four VITK_DP_* names.)"""
    from vit_pytorch_amd.parallel import FlatGradSink
    m = _EngineModel()
    monkeypatch.delenv("VITK_DP", raising=False)
    s = FlatGradSink(m)
    assert (s.layers_per_chunk, s.comm_priority, s.cu_reserve, s.reserve_layers) == (3, -1, 32, 1)
    monkeypatch.setenv("VITK_DP", "chunk=0, reserve=48")
    s = FlatGradSink(m)
    assert (s.layers_per_chunk, s.comm_priority, s.cu_reserve, s.reserve_layers) == (0, -1, 48, 1)
    monkeypatch.setenv("VITK_DP", "chunk=2,prio=0,reserve=16,layers=2")
    s = FlatGradSink(m, layers_per_chunk=5, cu_reserve=0)
    assert (s.layers_per_chunk, s.comm_priority, s.cu_reserve, s.reserve_layers) == (5, 0, 0, 2)
