"""Data-parallel training step: one process per GPU, one flat gradient buffer, RCCL over xGMI.

The reference has no distributed code; its only training script gets data parallelism from
accelerate -> torch DDP (train_vit_decorr.py:74-78,109: bucketed all-reduce, 25 MiB buckets,
hooks per parameter).  This is the MI355X-first replacement for that step:

  * every parameter gradient is written by the backward kernels DIRECTLY into one flat buffer
    (engine.set_grad_sink), laid out in reverse-readiness order: head, final norm, layers
    depth-1 .. 0, then the patch-embedding stage (cls, pos, patch Linear, LayerNorms);
  * during the transformer backward, every `layers_per_chunk` layers the contiguous slice finished so
    far is all-reduced on a side stream behind an event (3-4 large messages for ViT-B, overlapping
    the backward of the layers below); the remainder of the big segment goes out when the transformer
    stage is done and overlaps the patch-embedding backward (north_star); the small patch-embedding
    segment follows as the last collective when its gradients are enqueued;
  * xGMI is point-to-point (7 links/GPU): a few large collectives beat many 25 MiB buckets, so
    there is no bucketing at all -- 173 MB (ViT-B bf16) / 609 MB (ViT-L) go out as one message
    and RCCL picks its algorithm for the fully connected 8-GPU node.

`torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" on CPU for tests) is the transport.
Models that are not vit_pytorch_amd modules work too (no overlap: flatten after backward).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import engine as E


def _ordered_params(model: torch.nn.Module):
    """Parameters in reverse-readiness order, the index where the late (patch-embed) stage starts, and for every
    transformer layer the index one past its last parameter (layers are laid out depth-1 .. 0)."""
    params = [p for p in model.parameters() if p.requires_grad]
    late_names = ("to_patch_embedding", "cls_token", "pos_embedding")
    named = list(model.named_parameters())
    if not any(n.startswith(late_names) for n, _ in named):
        return params, len(params), {}
    early, late = [], []
    for n, p in named:
        if not p.requires_grad:
            continue
        (late if n.startswith(late_names) else early).append((n, p))

    def key(item):
        n = item[0]
        if n.startswith(("mlp_head", "linear_head")):
            return (0, 0)
        if n.startswith("transformer.norm"):
            return (1, 0)
        if n.startswith("transformer.layers."):
            return (2, -int(n.split(".")[2]))
        return (3, 0)

    early.sort(key=key)
    layer_end = {}
    for i, (n, _) in enumerate(early):
        if n.startswith("transformer.layers."):
            layer_end[int(n.split(".")[2])] = i + 1
    return [p for _, p in early] + [p for _, p in late], len(early), layer_end


class FlatGradSink:
    """Owns the flat gradient buffer; handed to the engine as the gradient sink."""

    def __init__(self, model: torch.nn.Module, process_group=None, average: bool = True, layers_per_chunk: Optional[int] = None,
                 comm_priority: Optional[int] = None, cu_reserve: Optional[int] = None, reserve_layers: Optional[int] = None):
        """cu_reserve / reserve_layers: while a chunk's all-reduce is (probably) in flight -- from its launch until `reserve_layers`
        more transformer layers have been enqueued -- the weight-gradient GEMMs are planned for 256 - cu_reserve CUs
        (vitk_set_cu_reserve), so that their workgroups fit BESIDE the collective's resident kernel instead of queueing behind it
        as a second round ([measured on one GPU with a stand-in kernel, tools/cu_contention.py] 32 CUs held: 1.62x -> 1.02-1.10x per
        layer; the persistent NT GEMMs adapt too: the four-wave kernel is launched on 256 - cu_reserve workgroups, the 8-wave kernel draws
        tile tickets).  Defaults 32 / 1; 0 turns it off.
        layers_per_chunk: transformer layers per in-backward all-reduce (0 = one message for the whole stack when its backward
        ends); default 3.  comm_priority: HIP stream priority of the side stream that carries the collectives (lower = more urgent;
        default -1 so that RCCL's kernels are scheduled ahead of the GEMM workgroups that would otherwise hold every CU).
        Arguments left at None take the defaults, or the fields of ONE environment variable for runs that cannot pass arguments (the
        driver's `bench.py --gpus N`):  VITK_DP="chunk=3,prio=-1,reserve=32,layers=1"  (any subset)."""
        import os
        env = dict(kv.split("=", 1) for kv in os.environ.get("VITK_DP", "").replace(" ", "").split(",") if "=" in kv)
        self.model = model
        self.group = process_group
        self.average = average
        self.params, self.n_early, layer_end = _ordered_params(model)
        self.layers_per_chunk = int(env.get("chunk", "3")) if layers_per_chunk is None else layers_per_chunk
        self.comm_priority = int(env.get("prio", "-1")) if comm_priority is None else comm_priority
        self.cu_reserve = int(env.get("reserve", "32")) if cu_reserve is None else cu_reserve
        self.reserve_layers = int(env.get("layers", "1")) if reserve_layers is None else reserve_layers
        self._reserve_left = 0
        p0 = self.params[0]
        self.dtype, self.device = p0.dtype, p0.device
        assert all(p.dtype == self.dtype and p.device == self.device for p in self.params), \
            "one dtype/device for all parameters"
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8  # keep every slice 16-byte aligned for the kernels
        self.offsets = offs
        self.total = total
        self.boundary = offs[self.n_early] if self.n_early < len(self.params) else total
        # flat offset one past layer li's gradients (layers are stored depth-1 .. 0)
        self.layer_end_off = {li: (offs[i] if i < len(offs) else total) for li, i in layer_end.items()}
        self._cursor = 0
        self.flat = torch.zeros(total, dtype=self.dtype, device=self.device)
        self.views = [self.flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)]
        self._by_ptr: Dict[int, int] = {p.data_ptr(): i for i, p in enumerate(self.params) if p.numel()}
        self._view_ptrs = {v.data_ptr() for v, p in zip(self.views, self.params) if p.numel()}
        self._filled = set()
        self._multi = set()
        self.side = torch.cuda.Stream(device=self.device, priority=self.comm_priority) if self.device.type == "cuda" else None
        self.log = None      # tests: list collecting (offset, length) of every all-reduce this rank issues
        self._early_launched = False
        self._late_launched = False
        self.world = dist.get_world_size(self.group) if dist.is_initialized() else 1

    def chunk_plan(self):
        """[(offset, elements, label)] of the collectives one step issues, in issue order: the in-backward chunks of the transformer
        stack (every `layers_per_chunk` layers, counted from layer 0 upwards, so the first chunk also holds the head and the final
        norm), the rest of the early segment, and the patch-embedding segment.  Together they tile [0, total) exactly once."""
        plan, cur = [], 0
        depth = len(self.layer_end_off)
        if self.layers_per_chunk > 0:
            for li in range(depth - 1, -1, -1):
                end = self.layer_end_off[li]
                if li % self.layers_per_chunk == 0 and end > cur:
                    plan.append((cur, end - cur, f"layers {li}.. (chunk ending at layer {li})"))
                    cur = end
        if self.boundary > cur:
            plan.append((cur, self.boundary - cur, "rest of the early segment"))
            cur = self.boundary
        if self.total > cur:
            plan.append((cur, self.total - cur, "patch embedding, cls, pos"))
        return plan

    def comm_model(self, world: int = 8, link_gbs: float = 153.0, links: int = 7, step_ms: Optional[float] = None):
        """Bytes and modelled time of every collective of chunk_plan() on one xGMI node (a MODEL: no multi-GPU run backs it).
        All-reduce of S bytes over `world` fully connected GPUs: a ring moves 2 (w-1)/w S per GPU over ONE link per direction
        (link_gbs); the direct (all-to-all reduce-scatter + all-gather) schedule spreads the same bytes over min(links, w-1)
        links.  Returns a list of dicts."""
        esz = torch.empty((), dtype=self.dtype).element_size()
        out = []
        for off, n, label in self.chunk_plan():
            S = n * esz
            moved = 2.0 * (world - 1) / world * S
            out.append({"label": label, "offset": off, "bytes": S, "ring_ms": moved / (link_gbs * 1e9) * 1e3,
                        "direct_ms": moved / (link_gbs * 1e9 * min(links, world - 1)) * 1e3})
        return out

    def refresh_param_ptrs(self):
        """Re-index the parameters after their storage moved (optim.Adam re-homes them into one flat buffer)."""
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params) if p.numel()}

    # ---- engine-facing ------------------------------------------------------------------------
    def buffer_for(self, param: torch.Tensor) -> Optional[torch.Tensor]:
        i = self._by_ptr.get(param.data_ptr())
        if i is None:
            return None
        if i in self._filled:
            # the parameter's SECOND gradient of this step (the model was called twice before backward: siamese / DINO-style
            # student passes, dino.py:283-290): the slot already holds -- and may already have sent -- the first one, so this one
            # goes through autograd's own accumulation into p.grad and finish_step() reduces and adds it
            self._multi.add(i)
            return None
        self._filled.add(i)
        return self.views[i]

    def wants_layer(self, layer: int) -> bool:
        return self.world > 1 and self.layers_per_chunk > 0 and layer % self.layers_per_chunk == 0 and layer in self.layer_end_off

    def layer_tick(self):
        """Called by the engine after every transformer layer's backward has been enqueued."""
        if self._reserve_left > 0:
            self._reserve_left -= 1
            if self._reserve_left == 0:
                self._set_reserve(0)

    def owns(self, t: torch.Tensor) -> bool:
        return t.data_ptr() in self._view_ptrs

    def stage_done(self, stage: str, layer: int = -1):
        """Called from inside backward (autograd thread) when a stage's gradients are enqueued.

        "layer" (every transformer layer, depth-1 .. 0): every `layers_per_chunk` layers the contiguous slice of the
        flat buffer finished so far goes out as one all-reduce on the side stream, overlapping the backward of the
        layers below it.  "transformer": whatever is left of the early segment.  "patch_embed": the late segment."""
        if self.world == 1:
            return
        if stage == "layer":
            end = self.layer_end_off.get(layer)
            if end is None or self.layers_per_chunk <= 0:
                return
            if layer % self.layers_per_chunk == 0 and end > self._cursor:
                self._launch(self.flat[self._cursor:end])
                self._cursor = end
        elif stage == "transformer" and not self._early_launched and self.boundary > 0:
            if self.boundary > self._cursor:
                self._launch(self.flat[self._cursor:self.boundary])
                self._cursor = self.boundary
            self._early_launched = True
        elif stage == "patch_embed" and not self._late_launched and self.boundary < self.total:
            self._launch(self.flat[self.boundary:])
            self._late_launched = True

    def _set_reserve(self, cus: int):
        if self.device.type == "cuda":
            from . import kernels as K
            K.set_cu_reserve(cus, self.dtype)

    def _launch(self, seg: torch.Tensor):
        if self.cu_reserve > 0 and self.reserve_layers > 0:
            self._set_reserve(self.cu_reserve)
            self._reserve_left = self.reserve_layers
        if self.side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                self._allreduce(seg)
        else:
            self._allreduce(seg)

    # ---- instrumentation (bench.py --gpus N: what the collective costs, and how much of it the step waits for) ----
    def enable_timing(self, on: bool = True):
        """Record, per collective of a step, its bytes and a HIP-event pair on the stream it is enqueued on, and around finish_step's
        wait for the communication stream an event pair on the main stream (the EXPOSED part).  timing_stats() reads them."""
        self._timing = [] if (on and self.device.type == "cuda") else None
        self._exposed = None

    def timing_stats(self):
        """{"world", "collectives": [{"bytes", "ms"}...], "comm_ms", "exposed_ms", "cu_reserve", "layers_per_chunk"} of the LAST step
        (synchronises the device); None when timing is off."""
        if getattr(self, "_timing", None) is None:
            return None
        torch.cuda.synchronize(self.device)
        cols = [{"bytes": b, "ms": round(e0.elapsed_time(e1), 4)} for b, e0, e1 in self._timing]
        exposed = round(self._exposed[0].elapsed_time(self._exposed[1]), 4) if self._exposed else 0.0
        return {"world": self.world, "collectives": cols, "comm_ms": round(sum(c["ms"] for c in cols), 4), "exposed_ms": exposed,
                "cu_reserve": self.cu_reserve, "layers_per_chunk": self.layers_per_chunk}

    def _allreduce(self, seg: torch.Tensor):
        if self.log is not None:
            self.log.append(((seg.data_ptr() - self.flat.data_ptr()) // seg.element_size(), seg.numel()))
        timing = getattr(self, "_timing", None)
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
            try:
                self._allreduce_raw(seg)
            finally:
                e1.record(torch.cuda.current_stream(self.device))
                timing.append((seg.numel() * seg.element_size(), e0, e1))
            return
        self._allreduce_raw(seg)

    def _allreduce_raw(self, seg: torch.Tensor):
        if self.average and dist.get_backend(self.group) == "nccl":
            dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                seg.div_(self.world)

    # ---- step-facing ----------------------------------------------------------------------------
    def begin_step(self):
        if getattr(self, "_timing", None) is not None:
            self._timing = []
            self._exposed = None
        self._filled.clear()
        self._multi.clear()
        self._cursor = 0
        self._early_launched = self._late_launched = False
        for p in self.params:
            p.grad = None
        E.set_grad_sink(self)

    def finish_step(self):
        """After loss.backward(): make every p.grad the (reduced) flat view."""
        E.set_grad_sink(None)
        if self._reserve_left > 0:
            self._reserve_left = 0
            self._set_reserve(0)
        # gradients autograd produced outside the sink (foreign modules / non-fused paths): copy them in
        missing = [i for i in range(len(self.params)) if i not in self._filled]
        if missing and self.world > 1 and self.side is not None:
            # a segment covering one of these slots may still be in flight on the side stream (e.g. the head ran op by op while
            # the fused transformer already handed the early segment to the collective): writing the slot now would race with it
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        for i in missing:
            p = self.params[i]
            if p.grad is not None and p.grad.data_ptr() != self.views[i].data_ptr():
                self.views[i].copy_(p.grad)
            elif p.grad is None:
                self.views[i].zero_()
        if self.world > 1:
            early_was, late_was = self._early_launched, self._late_launched
            if self.side is not None:
                if getattr(self, "_timing", None) is not None:       # how long the main stream stands still for the communication stream
                    x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    x0.record(torch.cuda.current_stream(self.device))
                    torch.cuda.current_stream(self.device).wait_stream(self.side)
                    x1.record(torch.cuda.current_stream(self.device))
                    self._exposed = (x0, x1)
                else:
                    torch.cuda.current_stream(self.device).wait_stream(self.side)
            if not early_was and not late_was and self._cursor == 0:
                self._allreduce(self.flat)                       # nothing went out during backward: one collective
            else:
                if not early_was and self.boundary > self._cursor:
                    self._allreduce(self.flat[self._cursor:self.boundary])
                if not late_was and self.boundary < self.total:
                    self._allreduce(self.flat[self.boundary:])
                for i in missing:                                # arrived after their segment had already gone out
                    if (early_was if i < self.n_early else late_was) and self.params[i].numel():
                        self._allreduce(self.views[i])
        if self._multi:
            if self.world > 1 and self.side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self.side)
            for i in sorted(self._multi):           # further gradients of a parameter used more than once: autograd summed them in p.grad
                extra = self.params[i].grad
                if extra is not None and extra.data_ptr() != self.views[i].data_ptr():
                    if self.world > 1:
                        self._allreduce(extra)
                    self.views[i].add_(extra)
        for p, v in zip(self.params, self.views):
            p.grad = v


class DataParallel(torch.nn.Module):
    """Wrap a model for one-process-per-GPU data parallelism.

        dp = DataParallel(model)          # after dist.init_process_group(...)
        loss = loss_fn(dp(x), y)
        dp.backward(loss)                 # backward + overlapped all-reduce; p.grad are views of dp.sink.flat
        optimizer.step()
    """

    def __init__(self, model: torch.nn.Module, process_group=None, average: bool = True, broadcast: bool = True,
                 layers_per_chunk: Optional[int] = None, comm_priority: Optional[int] = None):
        super().__init__()
        self.module = model
        self.sink = FlatGradSink(model, process_group, average, layers_per_chunk, comm_priority)
        if broadcast and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            for p in model.parameters():
                dist.broadcast(p.data, src=0, group=process_group)
            from ._epoch import invalidate_weight_caches
            invalidate_weight_caches()      # `.data` writes do not bump p._version: K-blocked / transposed copies made by an earlier forward are stale

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def backward(self, loss: torch.Tensor):
        self.sink.begin_step()
        try:
            loss.backward()
        finally:
            self.sink.finish_step()
