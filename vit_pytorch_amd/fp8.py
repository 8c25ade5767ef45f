"""Opt-in fp8 (OCP e4m3 / e5m2) operands for the Linear layers of the fused ViT engine (BASELINE config 5).

    model = ViT(...).cuda().bfloat16()
    vit_pytorch_amd.fp8.enable_fp8(model)                  # forward + backward (dX) GEMMs on fp8 operands
    vit_pytorch_amd.fp8.enable_fp8_forward(model)          # round-2 behaviour: the forward GEMMs only

What runs on fp8 once the scales exist (the reference's four nn.Linear per layer, vit.py:20,23,44,47, and their autograd):

    forward   QKV, out-projection, FF1, FF2      e4m3 activations x e4m3 weights   (v_mfma_f32_16x16x32_fp8_fp8)
    backward  dX of FF2 (+ GELU'), FF1, out-projection, QKV
                                                 e5m2 gradients x e4m3 weights^T   (v_mfma_f32_16x16x32_fp8_bf8)
              dW of the same four layers         e5m2 gradients^T x e4m3 activations (gemm_tn_fp8.hip: ds_read_b64_tr_b8 fragments,
                                                 v_mfma_f32_16x16x32_bf8_fp8; `wgrad=False` keeps them 16-bit)
    16-bit    attention, LayerNorm, the residual streams.
    saved     DEFAULT ("lean", engine.FP8_LEAN, engine.TransformerFn `lean8`): when this step's dW GEMMs will run on fp8 operands, the e4m3
              copies the forward GEMMs just consumed ARE their activation operands -- they are kept (1 B / element, made under the
              PREVIOUS step's scale, snapshotted with it) INSTEAD of the 16-bit LayerNorm / GELU outputs; the backward raises if the fp8
              state no longer allows fp8 weight gradients when it runs.  engine.FP8_LEAN = False: the 16-bit activations are saved and the e4m3
              operands re-made from them under this step's scales, one pass each (slightly different numerics: no stale-scale saturation).

Every fp8 GEMM whose reduction extent is a multiple of 128 (every dW GEMM: tokens are zero-padded) runs on
v_mfma_f32_16x16x128_f8f6f4 with unit block scales (`VITK_FP8_K128=0`: the K = 32 forms) -- the one fp8 form above the bf16 matrix rate on gfx950
(MI355X_MICROARCH.md: 2x) -- by pairing the fragments of two consecutive K-steps in registers (gemm_bf16.hip) / four
transposing reads per operand (gemm_tn_fp8.hip); per-tensor scales stay outside the instruction.

Scaling is per tensor and DELAYED by one step.  Producers that already hold an activation in registers -- LayerNorm forward for
the QKV / FF1 inputs, the GELU epilogue of FF1 for the FF2 input -- write its e4m3 copy themselves under the scale decided from
the PREVIOUS step's amax while recording this step's amax (64 atomicMax words per tensor).  The attention output and the four
gradient tensors of a layer (grad of the FF output, of the GELU pre-activation, of the attention branch output, of qkv) are
written by kernels without such a side output; they take ONE extra pass each (`vitk_quantize_fp8_delayed`: read 2 B, write 1 B
per element, amax recorded on the way).  Gradient slots keep a factor 2 of headroom below the e5m2 maximum (57344) because
their scale is one step old.  `vitk_fp8_update_scales_fmt` folds all records after every forward.  The first forward and the
first backward (no scales yet) run the 16-bit GEMMs and only record.  Weights are quantised with their current amax whenever
their values change (torch version counter, or the epoch this package's fused optimizer bumps; once per optimizer step); the
backward's e4m3 W^T shares the scale of W.

Values beyond a delayed scale's range saturate at +-max / scale; accumulation is f32.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

from . import kernels as K
from . import ops
from ._lib import VitkError
from ._epoch import weight_key

# slots of one layer: activations (e4m3) ...
S_A1, S_A2, S_ACT, S_O = 0, 1, 2, 3       # LN1 output (QKV input), LN2 output (FF1 input), GELU output (FF2 input), attention output
# ... and gradients (e5m2)
S_G3, S_DPRE, S_G2, S_DQKV = 4, 5, 6, 7   # grad at the FF2 output, at the GELU pre-activation, at the out-projection output, of qkv
SLOTS_PER_LAYER = 8
E4M3_MAX = 448.0
E5M2_MAX_USED = 28672.0                   # 57344 / 2: one-step-old gradient scales keep a factor 2 of headroom


def _identity(w: torch.Tensor):
    """(cache slot, staleness key) of a weight.  A 16-bit copy functional.autocast_aware made of a float32 master parameter for ONE forward
    carries the master's identity (`_vitk_master`): the e4m3 copies then live as long as the master keeps its value (gradient
    accumulation: one quantisation per optimizer step), and no entry is left behind for a tensor that died with its step."""
    m = getattr(w, "_vitk_master", None)
    return m if m is not None else (id(w), weight_key(w))


class Fp8State:
    """Delayed-scaling state of one Transformer: scales / amax records per (layer, tensor) and the e4m3 weight caches."""

    def __init__(self, depth: int, device, backward: bool = True, wgrad: bool = True):
        self.depth = depth
        self.backward = bool(backward)
        self.wgrad = bool(backward and wgrad)
        self.k128 = os.environ.get("VITK_FP8_K128", "1") not in ("0", "")     # [measured] ViT-H/14 batch 256: 696 -> 630 ms / step
        n = depth * SLOTS_PER_LAYER
        self.scales = torch.zeros(n, 2, dtype=torch.float32, device=device)        # {scale, 1/scale}
        self.amax = torch.zeros(n, 64, dtype=torch.int32, device=device)           # float bit patterns
        fmax = torch.full((depth, SLOTS_PER_LAYER), E4M3_MAX, dtype=torch.float32)
        fmax[:, S_G3:] = E5M2_MAX_USED
        self.fmax = fmax.reshape(n).to(device)
        self.ready = False             # True once one forward has recorded amax (activation slots have scales)
        self.bwd_ready = False         # True once one backward has recorded amax AND a later forward folded it
        self._bwd_recorded = False
        self._w: Dict[int, Tuple[tuple, torch.Tensor, torch.Tensor]] = {}          # id(param) -> (key, e4m3 bytes (N, K), {scale, 1/scale})
        self._wt: Dict[int, Tuple[tuple, torch.Tensor]] = {}                       # id(param) -> (key, e4m3 bytes of W^T (K, N))

    def slot(self, layer: int, k: int):
        i = layer * SLOTS_PER_LAYER + k
        return self.scales[i], self.amax[i]

    def weight(self, w: torch.Tensor):
        """e4m3 copy of a (N, K) weight and its scale pair; re-quantised when the parameter changed."""
        ident, key = _identity(w)       # data_ptr, torch version counter AND the epoch the fused optimizer bumps (it writes through raw pointers)
        ent = self._w.get(ident)
        # while a HIP graph is captured the quantisation must be part of the graph (replays then see the current weights)
        if ent is None or ent[0] != key or ent[1].device != w.device or torch.cuda.is_current_stream_capturing():
            sc = torch.empty(2, dtype=torch.float32, device=w.device)
            w8 = torch.empty(w.shape, dtype=torch.uint8, device=w.device)
            K.fp8_amax_scale(w, sc)
            K.quantize_fp8(w, w8, scale_dev=sc)
            ent = (key, w8, sc)
            self._w[ident] = ent
        return ent[1], ent[2]

    def weight_t(self, w: torch.Tensor):
        """e4m3 copy of W^T (K, N) -- the "W" operand that makes dX = dY . W an NT GEMM -- under the SAME scale as `weight(w)`."""
        w8, sc = self.weight(w)
        ident, key = _identity(w)
        ent = self._wt.get(ident)
        if ent is None or ent[0] != key or ent[1].device != w.device or torch.cuda.is_current_stream_capturing():
            wt = ops.transpose_weight(w)                       # 16-bit (K, N), cached per parameter value
            w8t = torch.empty(wt.shape, dtype=torch.uint8, device=w.device)
            K.quantize_fp8(wt, w8t, scale_dev=sc)
            ent = (key, w8t)
            self._wt[ident] = ent
        return ent[1], sc

    def backward_will_be_fp8(self) -> bool:
        """Asked DURING a forward: will the backward of this step find gradient scales (a backward has recorded, and the fold at the
        end of this forward turns the records into scales)?"""
        return self.bwd_ready or self._bwd_recorded

    def end_of_forward(self):
        K.fp8_update_scales_fmt(self.amax, self.scales, self.depth * SLOTS_PER_LAYER, self.fmax)
        self.ready = True
        if self._bwd_recorded:
            self.bwd_ready = True

    def end_of_backward(self):
        self._bwd_recorded = True


def enable_fp8(model: torch.nn.Module, enabled: bool = True, backward: bool = True, wgrad: bool = True):
    """Switch the fused Transformer stack(s) inside `model` to fp8 GEMM operands (see module docstring).  `backward=False`: the
    forward GEMMs only (QKV, out-projection, FF1, FF2); the backward then runs entirely on the saved 16-bit tensors.
    `wgrad=False`: forward and dX GEMMs on fp8, the weight-gradient GEMMs in 16 bit."""
    from .simple_vit import Transformer as SimpleTransformer
    from .vit import Transformer
    found = False
    for m in model.modules():
        if isinstance(m, (Transformer, SimpleTransformer)):
            p = next(m.parameters())
            # a float32 model: fp8 takes effect where its forward runs on 16-bit operands, i.e. inside torch.autocast (the reference's
            # accelerate mixed precision: train_vit_decorr.py:74-78); outside autocast the forward raises (engine.TransformerFn)
            if enabled and p.dtype not in (torch.bfloat16, torch.float16, torch.float32):
                raise VitkError("enable_fp8: the model must be bfloat16, float16, or float32 run under torch.autocast (fp8 replaces the 16-bit GEMM operands)")
            m._fp8 = Fp8State(len(m.layers), p.device, backward, wgrad) if enabled else None
            found = True
    if not found:
        raise VitkError("enable_fp8: no vit_pytorch_amd.vit.Transformer inside this model")
    return model


def enable_fp8_forward(model: torch.nn.Module, enabled: bool = True):
    """e4m3 operands for the forward GEMMs only (the round-2 entry point; `enable_fp8(model, backward=False)`)."""
    return enable_fp8(model, enabled, backward=False)
