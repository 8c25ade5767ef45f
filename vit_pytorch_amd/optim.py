"""Fused Adam / AdamW over the flat buffers of the data-parallel sink.

The reference's training step is `Adam(vit.parameters(), lr)` + `optim.step()` (train_vit_decorr.py:68-70,110): a
per-tensor loop in torch.  Here the gradients already live in ONE flat buffer (`parallel.FlatGradSink`, the buffer the
RCCL all-reduce runs on); this optimizer re-homes the parameters into a second flat buffer with the same offsets, keeps
f32 moments (and f32 master weights for bf16 parameters) in two / three more, and one step is one `vitk_adam_step`
launch over the whole model -- HBM-bound, 2+2+8 B read and 2+8 B written per parameter (+8 with master weights).

    dp  = DataParallel(model)                      # world size 1 is fine
    opt = Adam(dp, lr=3e-4)                        # or AdamW(dp, lr=..., weight_decay=0.05)
    loss = F.cross_entropy(dp(x).float(), y); dp.backward(loss); opt.step()

Update rule = torch.optim.Adam / AdamW (bias-corrected, eps outside the square root); tested against them step by step.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import kernels as K
from ._lib import VitkError
from ._epoch import bump_weights_epoch
from .parallel import DataParallel, FlatGradSink


class Adam:
    decoupled = False

    def __init__(self, dp, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0., master_weights: bool = True):
        sink = dp.sink if isinstance(dp, DataParallel) else dp
        if not isinstance(sink, FlatGradSink):
            raise TypeError("Adam/AdamW take a vit_pytorch_amd.parallel.DataParallel (or its FlatGradSink): the step runs on its flat buffers")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameters")
        if sink.device.type != "cuda":
            raise VitkError("vit_pytorch_amd.optim: parameters must live on the GPU (there is no CPU path)")
        self.sink = sink
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.t = 0
        n = sink.total
        # parameters -> one flat buffer with the gradient buffer's offsets; p.data become views of it
        self.flat_p = torch.zeros(n, dtype=sink.dtype, device=sink.device)
        with torch.no_grad():
            for p, o in zip(sink.params, sink.offsets):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        sink.refresh_param_ptrs()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=sink.device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=sink.device)
        self.master = self.flat_p.float() if (master_weights and sink.dtype != torch.float32) else None

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        """One update from the gradients of the last `dp.backward` (already averaged over ranks by the sink)."""
        self.t += 1
        K.adam_step(self.flat_p, self.sink.flat, self.exp_avg, self.exp_avg_sq, self.master, self.sink.total, self.lr,
                    self.betas[0], self.betas[1], self.eps, self.weight_decay, self.decoupled, self.t, grad_scale)
        bump_weights_epoch()      # the kernel wrote the parameters through raw pointers: caches derived from them are stale

    def zero_grad(self, set_to_none: bool = False):
        """Kept for drop-in use after `optim.step()` (train_vit_decorr.py:111); `dp.backward` zeroes the buffer anyway."""
        self.sink.flat.zero_()

    # ---- checkpoint / resume --------------------------------------------------------------------------------------
    def state_dict(self) -> Dict:
        return {"step": self.t, "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "master": None if self.master is None else self.master.clone()}

    def load_state_dict(self, sd: Dict):
        if sd["exp_avg"].numel() != self.exp_avg.numel():
            raise ValueError("optimizer state belongs to a different model (flat size mismatch)")
        self.t, self.lr, self.betas, self.eps, self.weight_decay = int(sd["step"]), float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"]), float(sd["weight_decay"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        if self.master is not None:
            if sd.get("master") is not None:
                self.master.copy_(sd["master"])
            else:
                self.master.copy_(self.flat_p.float())


class AdamW(Adam):
    """Decoupled weight decay (torch.optim.AdamW): p *= 1 - lr * wd before the Adam update."""
    decoupled = True

    def __init__(self, dp, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, master_weights: bool = True):
        super().__init__(dp, lr, betas, eps, weight_decay, master_weights)
