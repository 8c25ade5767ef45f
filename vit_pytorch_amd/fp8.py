"""Opt-in fp8 (OCP e4m3) operands for the forward GEMMs of the fused ViT engine (BASELINE config 5).

    model = ViT(...).cuda().bfloat16()
    vit_pytorch_amd.fp8.enable_fp8_forward(model)          # forward QKV / FF1 / FF2 GEMMs read e4m3 operands
    ...train as usual: backward stays in the 16-bit dtype on the saved 16-bit activations...

What runs: the NT GEMM kernel with 1-byte operands (`vitk_gemm_nt_fp8_ex`: the same tiles, twice the k per K-step, f32
accumulation).  Scaling is per tensor and DELAYED: the producers that already hold an activation in registers -- LayerNorm
forward for the QKV / FF1 inputs, the GELU epilogue of FF1 for the FF2 input -- write its e4m3 copy themselves under the
scale decided from the PREVIOUS step's amax while recording this step's amax (64 atomicMax words per tensor), so no extra
pass over any activation exists and the step never waits for its own statistics; `vitk_fp8_update_scales` folds the
records after every forward.  The first forward (no scales yet) runs the 16-bit GEMMs and only records.  Weights are
quantised with their current amax whenever their values change (torch version counter, or the epoch this package's fused
optimizer bumps; once per optimizer step).  The out-projection
(8 % of the layer's GEMM FLOPs; its input comes out of the attention kernel) stays 16-bit.

Values beyond the delayed scale's range saturate at +-448 * 1/scale; accumulation is f32; everything the backward reads
(saved activations, weights) is the 16-bit original.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import kernels as K
from ._lib import VitkError
from ._epoch import weight_key

SLOTS_PER_LAYER = 3      # LN1 output (QKV input), LN2 output (FF1 input), GELU output (FF2 input)


class Fp8State:
    """Delayed-scaling state of one Transformer: scales / amax records per (layer, tensor) and the e4m3 weight cache."""

    def __init__(self, depth: int, device):
        self.depth = depth
        n = depth * SLOTS_PER_LAYER
        self.scales = torch.zeros(n, 2, dtype=torch.float32, device=device)        # {scale, 1/scale}
        self.amax = torch.zeros(n, 64, dtype=torch.int32, device=device)           # float bit patterns
        self.ready = False                                                         # True once one forward has recorded amax
        self._w: Dict[int, Tuple[int, torch.Tensor, torch.Tensor]] = {}            # id(param) -> (version, e4m3 bytes, {scale, 1/scale})

    def slot(self, layer: int, k: int):
        i = layer * SLOTS_PER_LAYER + k
        return self.scales[i], self.amax[i]

    def weight(self, w: torch.Tensor):
        """e4m3 copy of a (N, K) weight and its scale pair; re-quantised when the parameter changed."""
        key = weight_key(w)       # data_ptr, torch version counter AND the epoch the fused optimizer bumps (it writes through raw pointers)
        ent = self._w.get(id(w))
        # while a HIP graph is captured the quantisation must be part of the graph (replays then see the current weights)
        if ent is None or ent[0] != key or ent[1].device != w.device or torch.cuda.is_current_stream_capturing():
            sc = torch.empty(2, dtype=torch.float32, device=w.device)
            w8 = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
            K.fp8_amax_scale(w, sc)
            K.quantize_fp8(w, w8, scale_dev=sc)
            ent = (key, w8, sc)
            self._w[id(w)] = ent
        return ent[1], ent[2]

    def end_of_forward(self):
        K.fp8_update_scales(self.amax, self.scales, self.depth * SLOTS_PER_LAYER)
        self.ready = True


def enable_fp8_forward(model: torch.nn.Module, enabled: bool = True):
    """Switch the fused Transformer stack(s) inside `model` to e4m3 forward GEMM operands (see module docstring)."""
    from .simple_vit import Transformer as SimpleTransformer
    from .vit import Transformer
    found = False
    for m in model.modules():
        if isinstance(m, (Transformer, SimpleTransformer)):
            p = next(m.parameters())
            if enabled and p.dtype not in (torch.bfloat16, torch.float16):
                raise VitkError("enable_fp8_forward: the model must be bfloat16 or float16 (fp8 replaces the 16-bit forward GEMM operands)")
            m._fp8 = Fp8State(len(m.layers), p.device) if enabled else None
            found = True
    if not found:
        raise VitkError("enable_fp8_forward: no vit_pytorch_amd.vit.Transformer inside this model")
    return model
