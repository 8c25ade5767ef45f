"""HIP-graph capture of the forward (serving) and of forward+backward (training) for fixed shapes.

Every op of this package is a kernel enqueued on the CURRENT stream through the C-ABI, with all workspaces taken from
torch's caching allocator -- nothing synchronises, nothing allocates with hipMalloc on the path -- so a whole model call
is capturable as ONE graph with torch's stream capture (`torch.cuda.CUDAGraph` is hipGraph on ROCm).  At the headline
batch the GPU is the bottleneck and a graph buys nothing; at serving batches (1-16 images) a ViT-B forward is ~170 kernel
launches of a few microseconds each and the Python/ctypes launch path (~10 us per launch) is what bounds latency: replaying
the graph removes it.  The side stream that runs the weight-gradient GEMMs forks from and joins the capturing stream inside
the capture, so the training graph keeps that concurrency.

    fwd = GraphedForward(model, example_images)          # model.eval(); fixed shape / dtype
    logits = fwd(images)                                 # copy-in, one graph launch, static output buffer

    step = GraphedForwardBackward(model, loss_fn, example_images, example_labels)
    loss = step(images, labels)                          # p.grad of every parameter is updated in place; then optimizer.step()

Restrictions: fixed input shape; dropout must be inactive (its counter-RNG seed/offset are kernel arguments and would be
frozen into the graph); single process (no collectives are captured); outputs live in static buffers that the next call
overwrites.
"""
from __future__ import annotations

from typing import Callable

import torch

from ._lib import VitkError


def _active_dropout(model: torch.nn.Module) -> bool:
    return model.training and any(isinstance(m, torch.nn.Dropout) and m.p > 0.0 for m in model.modules())


def _warm(fn: Callable[[], None], iters: int):
    """Run `fn` a few times on a side stream first: lazy one-time work (LDS opt-ins, allocator growth, autograd's
    buffers) must not happen inside the capture."""
    cur = torch.cuda.current_stream()
    s = torch.cuda.Stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        for _ in range(iters):
            fn()
    cur.wait_stream(s)
    torch.cuda.synchronize()


class GraphedForward:
    def __init__(self, model: torch.nn.Module, example: torch.Tensor, warmup: int = 3):
        if not example.is_cuda:
            raise VitkError("GraphedForward: the example input must be on the GPU")
        if _active_dropout(model):
            raise VitkError("GraphedForward: active dropout cannot be captured (call model.eval() or use p = 0)")
        self.model = model
        self.static_in = example.detach().clone()

        def run():
            with torch.no_grad():
                return model(self.static_in)

        _warm(run, warmup)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = run()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise VitkError(f"GraphedForward was captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, got {tuple(x.shape)} {x.dtype}")
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out


class GraphedForwardBackward:
    """One graph = zero grads, forward, loss, backward.  Gradients land in the same `.grad` tensors at every replay."""

    def __init__(self, model: torch.nn.Module, loss_fn: Callable[[torch.Tensor, torch.Tensor], torch.Tensor],
                 example_x: torch.Tensor, example_y: torch.Tensor, warmup: int = 3):
        if _active_dropout(model):
            raise VitkError("GraphedForwardBackward: active dropout cannot be captured")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise VitkError("GraphedForwardBackward: single process only (collectives are not captured)")
        self.model = model
        self.static_x = example_x.detach().clone()
        self.static_y = example_y.detach().clone()
        params = [p for p in model.parameters() if p.requires_grad]

        def run():
            for p in params:
                if p.grad is not None:
                    p.grad.zero_()                 # in place: the graph must keep writing the same tensors
            loss = loss_fn(model(self.static_x), self.static_y)
            loss.backward()
            return loss.detach()

        _warm(run, warmup)                          # also materialises every .grad
        self.params = params
        self.grads = [p.grad for p in params]       # the graph writes THESE tensors: keep them alive and attached
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = run()

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        self.static_x.copy_(x)
        self.static_y.copy_(y)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):   # survive a zero_grad(set_to_none=True) between steps
            p.grad = g
        return self.static_loss
