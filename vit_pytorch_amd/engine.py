"""Fused forward/backward of the ViT encoder as three torch.autograd.Functions.

    PatchEmbedFn   img -> x0        vit.py:99-104 + :122-128   (patchify, LN, Linear, LN, cls, +pos)
    TransformerFn  x0  -> tokens    vit.py:78-83               (depth x [attn + x ; ff + x], final LN)
    HeadFn         tokens -> logits vit.py:135-138             (pool, Linear)

Each Function runs the whole stage on libvitk kernels (vit_pytorch_amd.ops), keeps what backward
needs, and its backward produces the gradients of every parameter of the stage.  Backward is
entered from autograd's device thread; all launches go to that thread's current stream.

Data flow per transformer layer (T = model dtype, residual stream x in f32):

    a1  = LN1(x)                        T   (M, D)      stats1
    qkv = a1 Wqkv^T                     T   (M, 3I)     merged layout, never split
    o   = softmax(scale q k^T) v        T   (M, I)      lse (B,H,N)
    x2  = x + o Wout^T + bout           f32 (M, D)      fused residual epilogue
    a2  = LN2(x2)                       T               stats2
    pre = a2 W1^T + b1 ; act = gelu     T   (M, F) x2   fused bias+GELU epilogue
    x3  = x2 + act W2^T + b2            f32             fused residual epilogue

With active dropout (training, p > 0) the same kernels take the four dropouts of a layer in place: o uses the dropped
attention matrix, the two residual epilogues drop the Linear output before the add, act = dropout(gelu(pre)); the keep
decisions are a counter hash that the backward kernels regenerate (see TransformerFn).

A GradSink (see parallel.py) may be installed to place parameter gradients into one flat
buffer and to be told when the transformer's gradients are complete (data-parallel overlap).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import kernels as K
from . import ops
from . import _lib as L
from ._lib import RowMap, VitkError
from ._epoch import caller_grad_mode, reset_grad_mode

Tensor = torch.Tensor
F32 = torch.float32

_sink_global = [None]


def set_grad_sink(sink):
    """Install (or clear with None) the object that receives parameter-gradient buffers."""
    _sink_global[0] = sink


def _sink():
    return _sink_global[0]


def _grad_buf(param: Tensor) -> Tensor:
    s = _sink()
    if s is not None:
        buf = s.buffer_for(param)
        if buf is not None:
            return buf
    return torch.empty_like(param, memory_format=torch.contiguous_format)


def _ret(g: Optional[Tensor]) -> Optional[Tensor]:
    """What backward hands to autograd for a parameter gradient.  A gradient that already sits in the
    sink's flat buffer is NOT returned (autograd would clone the view into a fresh .grad tensor: one
    device copy per parameter per step); the sink assigns p.grad itself in finish_step()."""
    s = _sink()
    if g is not None and s is not None and s.owns(g):
        return None
    return g


_side_streams = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = (device.type, device.index)
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


class _Fork:
    """Run weight-gradient GEMMs on a side stream, concurrently with the dX chain of the main stream.

    Within one layer's backward the dW GEMMs (TN) depend on tensors the chain has already produced but
    nothing on the chain depends on them, so they can fill the CUs the chain leaves idle (the tail of
    every 591-tile N=768 GEMM on 256 CUs, the HBM-bound LayerNorm/attention phases).  `run` orders the
    side stream behind everything enqueued so far; `join` makes the main stream wait for the side work.

    Lifetime of the side stream's inputs: NOT `record_stream`.  A block recorded on another stream can only be recycled once an
    event on that stream has completed, and the host runs far ahead of the GPU: every big temporary the backward frees (1.5 GB
    each at ViT-H/14, batch 256) would be replaced by a fresh allocation, the reserved pool grows until the device is full
    (measured: 146 GiB allocated, 287 GiB reserved, allocator retries, 4 s instead of 0.85 s per step).  Instead the inputs of
    the last LAG launches are held here, and before a launch's inputs are let go the MAIN stream waits for that launch's event:
    whatever main-stream kernel reuses the memory is ordered behind the side-stream reader by the streams themselves."""
    LAG = 6                 # launches (a layer and a half) whose inputs stay referenced
    serialize = False       # bench.py's per-class kernel timing sets it: a launch timed beside a concurrent GEMM measures the contention

    def __init__(self, device, fp8: bool = False):
        # fp8 weight gradients ONLY (gemm_tn256_f8_kernel, eight waves, 22 % of the ViT-H/14 step): that kernel leaves the CU resources a
        # second resident workgroup needs, and serializing it costs the fp8 step 7 % [measured, profiles/r04e_h14_dw_stream_ab.log].  The
        # 16-bit weight-gradient kernel needs a whole CU per workgroup (136 KB of LDS, 512 registers per wave): beside it a side stream cost
        # the ViT-B/16 step 0.3-0.46 ms in every A/B of rounds 4 and 5 (profiles/r04_dw_stream_ab.log, r05k2_switch_sweep.log), so since
        # round 6 the 16-bit path has no side stream and no switch for one.
        self.enabled = bool(fp8) and device.type == "cuda" and not _Fork.serialize
        self._held = []
        if self.enabled:
            self.main = torch.cuda.current_stream(device)
            self.side = _side_stream(device)

    def run(self, fn, *tensors):
        if not self.enabled:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()
        done = torch.cuda.Event()
        done.record(self.side)
        self._held.append((done, tensors))
        while len(self._held) > self.LAG:
            old, _ = self._held.pop(0)
            self.main.wait_event(old)

    def join(self):
        if self.enabled:
            self.main.wait_stream(self.side)
            self._held.clear()


def _check_dims(D: int, what: str):
    if D % 4 != 0:
        raise VitkError(f"{what}: feature dimension {D} must be a multiple of 4 for the HIP kernels")


def pack_layer_params(attn, ff) -> List[Optional[Tensor]]:
    """Fixed argument order of one layer for TransformerFn (None where the module has no tensor)."""
    to_out = attn.to_out
    if isinstance(to_out, torch.nn.Identity):
        wout, bout = None, None
    elif isinstance(to_out, torch.nn.Linear):      # SimpleViT: bias-free Linear (simple_vit.py:48)
        wout, bout = to_out.weight, to_out.bias
    else:                                          # ViT: Sequential(Linear, Dropout) (vit.py:46-49)
        wout, bout = to_out[0].weight, to_out[0].bias
    lins = [m for m in ff.net if isinstance(m, torch.nn.Linear)]
    return [attn.norm.weight, attn.norm.bias, attn.to_qkv.weight, wout, bout,
            ff.net[0].weight, ff.net[0].bias, lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias]


NLP = 11  # tensors per layer in pack_layer_params
FP8_LEAN = True     # fp8 with fp8 weight gradients: keep the e4m3 copies the forward GEMMs consumed INSTEAD of the 16-bit LayerNorm / GELU outputs (tests flip it)


def _hash32(x: int) -> int:
    """The device's hash32 (common.h): decorrelates the per-site dropout seeds derived from one base seed."""
    x &= 0xffffffff
    x ^= x >> 16; x = (x * 0x21f0aaad) & 0xffffffff
    x ^= x >> 15; x = (x * 0x735a2d97) & 0xffffffff
    x ^= x >> 15
    return x


def dropout_fusable(T, B: int, N: int, D: int, heads: int, dim_head: int, F: int) -> bool:
    """Active dropout (vit.py:22,24,42,48) runs inside the fused engine when the four Linear layers are served by the
    256-row GEMM kernel (their epilogues carry the fused dropout) and attention by the fixed-length flash kernel."""
    M, I = B * N, heads * dim_head
    return (T in ops.HALF and ops.attn_fast_ok(T, N, dim_head) and ops.fused_dropout_ok(T, M, D, I)
            and ops.fused_dropout_ok(T, M, F, D) and ops.fused_dropout_ok(T, M, D, F))


def _recompute_policy(saved_bytes: int, device, frac: float = 0.45) -> bool:
    """Keep every activation of the stack (the default: 1.4 GB / layer at ViT-B/16 batch 256, nothing but the attention
    probabilities is ever recomputed) unless that would not fit: when the activations a forward pass is about to save exceed
    45 % of the device memory -- BASELINE config 5, ViT-H/14 at 336 px and batch 256 / GPU: 6.8 GB x 32 layers = 217 GB of 288 --
    the three cheapest-to-rebuild tensors of a layer are dropped and rebuilt in its backward: the GELU output (one elementwise pass
    over the saved pre-activation) and the two LayerNorm outputs (one LayerNorm forward each from the saved residual stream):
    2.3 of the 6.8 GB.  VITK_RECOMPUTE=0 / 1 overrides.  The fp8 path (round 6): its lean saving (e4m3 copies instead of the 16-bit
    LayerNorm / GELU outputs: 5.9 GB a layer at config 5, 187 GiB in all) is kept while it fits -- `frac` = 0.70 of the device for that
    question -- and gives way to the same recompute (4.5 GB a layer, 145 GiB; the e4m3 operands of the weight-gradient GEMMs are then
    re-made from the rebuilt 16-bit tensors) beyond it or under VITK_RECOMPUTE=1."""
    import os
    v = os.environ.get("VITK_RECOMPUTE")
    if v is not None:
        return v not in ("0", "")
    try:
        total = torch.cuda.get_device_properties(device).total_memory
    except Exception:
        return False
    return saved_bytes > frac * total


def forward_stream_is_16bit(T, M: int, D: int, I: int, Fh: int, depth: int, drop_p: float, fp8, has_out: bool, has_b1: bool) -> bool:
    """Whether TransformerFn runs its forward residual stream in the parameter dtype (ops.fwd_stream_16: the default for bfloat16;
    no fp8, no active dropout, both residual GEMMs of a layer on the persistent kernel).  One predicate for the stage itself and for
    the patch embedding in front of it (PatchEmbedFn out16)."""
    return bool(T in ops.HALF and depth > 0 and drop_p == 0.0 and fp8 is None and has_out and has_b1
                and ops.fwd_stream_16(T) and ops.stream16_ok(M, D, I, Fh))


class TransformerFn(torch.autograd.Function):
    """Transformer.forward (vit.py:78-83 / simple_vit.py:74-78).  drop_p > 0 (training): the four dropouts of a layer --
    attention matrix (vit.py:60), after to_out (:48), after the GELU (:22), after the second FeedForward Linear (:24) --
    are taken inside the attention kernel and the GEMM epilogues with counter-hash keep decisions under per-site seeds
    derived from drop_seed; the backward kernels regenerate them, no mask tensor exists."""

    @staticmethod
    def forward(ctx, x, heads: int, dim_head: int, drop_p: float, drop_seed: int, fp8, norm_w, norm_b, *lp):
        K.require_device(x, norm_w)
        depth = len(lp) // NLP
        T = norm_w.dtype
        B, N, D = x.shape
        M = B * N
        _check_dims(D, "Transformer")
        I = heads * dim_head
        scale = dim_head ** -0.5
        x = x.contiguous()
        # the forward residual stream: float32, or (ops.fwd_stream_16: default for bfloat16 parameters; no fp8, no active dropout, both residual
        # GEMMs on the persistent kernel) the parameter dtype -- every consumer below takes either
        depth_, I_ = len(lp) // NLP, heads * dim_head
        s16f = forward_stream_is_16bit(T, M, D, I_, lp[7].shape[0] if depth_ else 0, depth_, drop_p, fp8, depth_ > 0 and lp[3] is not None,
                                       depth_ > 0 and lp[8] is not None)
        SD = T if s16f else F32
        if x.dtype == SD:
            xs = x
        else:
            xs = ops.empty((B, N, D), SD, x)
            K.cast(x, xs)
        saved = []
        keep = any(ctx.needs_input_grad) and caller_grad_mode()      # no_grad / eval: drop each layer's activations as soon as the layer is done
        # (needs_input_grad mirrors requires_grad even under torch.no_grad(): the module-level caller records the real mode, _epoch.py)
        esz = 4 if T == F32 else 2
        Fh0 = lp[7].shape[0] if depth else 0
        per_layer = M * (2 * D * 4 + (2 * D + 4 * I + 2 * Fh0) * esz)          # xs, x2 (f32) + a1, a2, qkv, o, pre, act
        recompute = bool(keep and depth and drop_p == 0.0 and _recompute_policy(per_layer * depth, xs.device))
        if drop_p > 0.0 and (lp[3] is None or lp[8] is None or not dropout_fusable(T, B, N, D, heads, dim_head, lp[7].shape[0])):
            raise VitkError("TransformerFn: this shape does not take the fused dropout path (caller must check dropout_fusable)")
        site = (lambda li, k: (drop_p, _hash32(drop_seed + 4 * li + k))) if drop_p > 0.0 else (lambda li, k: None)
        # fp8 (fp8.py): e4m3 operands for QKV / out-projection / FF1 / FF2 once the delayed scales exist; the first call only records amax
        if fp8 is not None and T not in ops.HALF:
            raise VitkError("enable_fp8: fp8 replaces 16-bit GEMM operands -- call this float32 model inside torch.autocast('cuda', dtype=torch.bfloat16) "
                            "(or convert it with .bfloat16()); a float32 forward would silently ignore the switch")
        use8 = fp8 is not None and T in ops.HALF and drop_p == 0.0 and depth > 0 and lp[8] is not None and ops.fp8_gemm_ok(M, D, I, lp[7].shape[0])
        go8 = use8 and fp8.ready
        out8 = use8 and lp[3] is not None and ops.fp8_out_ok(M, D, I)       # the out-projection takes an e4m3 copy of the attention output
        bwd8 = bool(use8 and out8 and fp8.backward and ops.fp8_bwd_ok(M, D, I, lp[7].shape[0]))
        # "lean" saving: when THIS step's weight-gradient GEMMs will run on fp8 operands, the e4m3 copies the forward GEMMs just
        # consumed ARE their activation operands -- they are kept (1 B / element) INSTEAD of the 16-bit LayerNorm / GELU outputs
        # (2 B / element), which nothing else in the backward reads: no re-quantisation pass, no recompute pass, less memory.
        # The scales they were made under are snapshotted (the fold at the end of this forward overwrites the live ones).
        lean8 = bool(go8 and keep and bwd8 and fp8.wgrad and fp8.backward_will_be_fp8() and FP8_LEAN
                     and all(ops.fp8_tn_ok(M, n_, k_) for n_, k_ in ((D, lp[7].shape[0]), (lp[7].shape[0], D), (D, I), (3 * I, D))))
        if lean8:       # ... while it fits (see _recompute_policy): xs, x2 (f32) + three e4m3 activations + the e4m3 attention output + qkv, o, pre (16 bit)
            lean_layer = M * (10 * D + Fh0 + I + (4 * I + Fh0) * esz)
            if _recompute_policy(lean_layer * depth, xs.device, frac=0.70):
                lean8, recompute = False, True
        scales_used = fp8.scales.clone() if lean8 else None
        # the FeedForward GEMM stores the gelu' factor instead of the pre-activation (ops.gelu_dg_ok) when the backward will want only
        # that of it: not under recompute (the GELU output is rebuilt from the pre-activation), fp8 or active dropout
        dg_mode = bool(keep and depth and not recompute and drop_p == 0.0 and fp8 is None and lp[8] is not None
                       and ops.gelu_dg_ok(T, M, lp[7].shape[0], D))
        ctx.dg_mode = dg_mode
        ctx.grad16 = ops.grad_stream_16()       # the backward's stream dtype is decided HERE (a per-call policy -- autocast -- is gone by backward time)

        def gemm8(a8, sc_a, w, out, Nn, Kd, epi, **kw):       # out (M, Nn) = a8 (M, Kd) e4m3 . e4m3(w)^T under the two per-tensor scales
            w8, wsc = fp8.weight(w)
            K.gemm_nt_fp8_v2(a8, Kd, w8, Kd, out, Nn, M, Nn, Kd, epi, a_kind=K.A_E4M3, alpha_a=sc_a[1:], alpha_w=wsc[1:],
                             k128=fp8.k128 and Kd % 128 == 0, **kw)
        if T in ops.HALF and fp8 is None:       # every K-blocked weight copy this step will ask for, in one launch (after an optimizer step: all of them)
            ops.prepack_weights([lp[li * NLP + j] for li in range(depth) for j in (2, 3, 7, 9)], M)
        for li in range(depth):
            ln1w, ln1b, wqkv, wout, bout, ln2w, ln2b, w1, b1, w2, b2 = lp[li * NLP:(li + 1) * NLP]
            a1 = ops.empty((M, D), T, xs)
            if use8:
                sc1, am1 = fp8.slot(li, 0)
                a1_8 = torch.empty((M, D), dtype=torch.uint8, device=xs.device) if go8 else None
                st1 = ops.ln_fwd(xs, ln1w, ln1b, M, D, a1, f8=(a1_8, sc1, am1))
            else:
                st1 = ops.ln_fwd(xs, ln1w, ln1b, M, D, a1)
            if go8:
                qkv = ops.empty((M, 3 * I), T, xs)
                gemm8(a1_8, sc1, wqkv, qkv, 3 * I, D, L.EPI_NONE)
            else:
                qkv = ops.linear_fwd(a1, wqkv, None, M)
            o, att_saved = ops.attn_fwd(qkv, B, N, heads, dim_head, scale, drop=site(li, 0))
            o_8 = act_8 = None
            if out8:
                sc4, am4 = fp8.slot(li, 3)
                if go8:         # one pass over o: its e4m3 copy under last step's scale + this step's amax
                    o_8 = torch.empty((M, I), dtype=torch.uint8, device=xs.device)
                    K.quantize_fp8_delayed(o, o_8, sc4, am4, K.FMT_E4M3)
                    x2 = ops.empty((M, D), F32, xs)
                    gemm8(o_8, sc4, wout, x2, D, I, L.EPI_RESID, bias=bout, resid=xs)
                else:
                    K.quantize_fp8_delayed(o, None, None, am4, K.FMT_E4M3)
                    x2 = ops.linear_fwd(o, wout, bout, M, resid=xs)
            elif wout is not None:
                x2 = ops.linear_fwd(o, wout, bout, M, resid=xs, drop=site(li, 1))
            else:  # to_out = Identity (heads == 1 and dim_head == dim, vit.py:34,49)
                x2 = ops.empty((M, D), F32, xs)
                K.add_rows(xs, o, None, x2, M, D)
            a2 = ops.empty((M, D), T, xs)
            if use8:
                sc2, am2 = fp8.slot(li, 1)
                sc3, am3 = fp8.slot(li, 2)
                Fh = w1.shape[0]
                a2_8 = torch.empty((M, D), dtype=torch.uint8, device=xs.device) if go8 else None
                st2 = ops.ln_fwd(x2, ln2w, ln2b, M, D, a2, f8=(a2_8, sc2, am2))
                pre = ops.empty((M, Fh), T, xs)
                act = None if lean8 else ops.empty((M, Fh), T, xs)      # lean saving: nothing reads the 16-bit GELU output -- FF2 and dW2 take the e4m3 copy
                if go8:
                    act_8 = torch.empty((M, Fh), dtype=torch.uint8, device=xs.device)
                    gemm8(a2_8, sc2, w1, act, Fh, D, L.EPI_BIAS_GELU, bias=b1, aux=pre, c8=act_8, c8_scale=sc3, c8_amax64=am3)
                    x3 = ops.empty((M, D), F32, xs)
                    gemm8(act_8, sc3, w2, x3, D, Fh, L.EPI_RESID, bias=b2, resid=x2)
                else:       # recording pass: 16-bit operands, amax of the GELU output collected by the same epilogue
                    K.gemm_nt_fp8_ex(a2, D, w1, D, act, Fh, M, Fh, D, L.EPI_BIAS_GELU, a_is_fp8=False, bias=b1, aux=pre, c8_amax64=am3)
                    x3 = ops.linear_fwd(act, w2, b2, M, resid=x2)
            else:
                st2 = ops.ln_fwd(x2, ln2w, ln2b, M, D, a2)
                act, pre = ops.linear_fwd(a2, w1, b1, M, gelu=True, drop=site(li, 2), save_dg=dg_mode)      # dg_mode: `pre` holds gelu'(pre)
                x3 = ops.linear_fwd(act, w2, b2, M, resid=x2, drop=site(li, 3))
            if keep:
                if lean8:
                    saved.append((xs, None, st1, qkv, o, att_saved, x2, None, st2, pre, None, (a1_8, a2_8, act_8, o_8)))
                else:
                    saved.append((xs, None, st1, qkv, o, att_saved, x2, None, st2, pre, None, None) if recompute
                                 else (xs, a1, st1, qkv, o, att_saved, x2, a2, st2, pre, act, None))
            xs = x3
        if use8:
            fp8.end_of_forward()
        y = ops.empty((B, N, D), T, xs)
        stf = ops.ln_fwd(xs, norm_w, norm_b, M, D, y)
        ctx.saved = saved
        reset_grad_mode()       # the caller's note was for THIS forward only (_epoch.py)
        ctx.x_last = xs
        ctx.stf = stf
        ctx.meta = (heads, dim_head, depth, B, N, D, x.dtype)
        ctx.drop = (drop_p, drop_seed)
        # (the backward re-quantises the saved activations under the FORWARD slots' scales: all four of them must be live -> out8)
        ctx.fp8 = fp8 if bwd8 else None
        ctx.f8_scales = scales_used
        ctx.save_for_backward(norm_w, norm_b, *[t for t in lp if t is not None])
        ctx.lp_mask = [t is not None for t in lp]
        return y

    @staticmethod
    def backward(ctx, dy):
        with ops.deferred_folds():          # the LayerNorm finalizes and bias-gradient column sums of a layer go out as one launch
            return TransformerFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        heads, dim_head, depth, B, N, D, in_dtype = ctx.meta
        drop_p, drop_seed = ctx.drop
        site = (lambda li, k: (drop_p, _hash32(drop_seed + 4 * li + k))) if drop_p > 0.0 else (lambda li, k: None)
        sv = list(ctx.saved_tensors)
        norm_w, norm_b = sv[0], sv[1]
        it = iter(sv[2:])
        lp = [next(it) if m else None for m in ctx.lp_mask]
        T = norm_w.dtype
        bf = T != F32
        M = B * N
        scale = dim_head ** -0.5
        if ctx.x_last is None:
            raise RuntimeError("vit_pytorch_amd: backward through this fused stage a second time -- its saved activations were released during the first backward (retain_graph is not supported by the fused engine)")
        dy = dy.contiguous()
        grads: List[Optional[Tensor]] = [None] * len(lp)

        # the residual stream of the backward: float32, or (16-bit parameters, no active dropout: ops.grad_stream_16) the parameter
        # dtype -- then the LayerNorm backward's 16-bit output IS the stream and the f32 tensors below do not exist
        s16 = bf and drop_p == 0.0 and D % 4 == 0 and ctx.grad16

        def newg():
            g32 = None if s16 else ops.empty((M, D), F32, dy)
            return g32, (ops.empty((M, D), T, dy) if bf else None)

        # fp8 backward (fp8.py): dX = dY . W of the four Linear layers on e5m2 gradients x e4m3 weights; dY takes one pass
        # (vitk_quantize_fp8_delayed: e5m2 copy under last step's scale + this step's amax).  The first backward only records.
        f8 = ctx.fp8
        I = heads * dim_head
        ctx_scales = ctx.f8_scales.view(depth, -1, 2) if ctx.f8_scales is not None else None      # scales the kept e4m3 copies were made under

        def q5(li, slot, t):
            """(e5m2 copy, scale pair) of a gradient tensor under last step's scale, this step's amax recorded on the way; None (record
            only) while the gradient scales do not exist yet."""
            sc, am = f8.slot(li, slot)
            if not f8.bwd_ready:
                K.quantize_fp8_delayed(t, None, None, am, K.FMT_E5M2)
                return None
            t8 = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
            K.quantize_fp8_delayed(t, t8, sc, am, K.FMT_E5M2)
            return t8, sc

        def dx8(q, W, epi=L.EPI_NONE, pre=None, db=None, c8=None):
            """dX (M, Kd) = dY (M, Nw) . W (Nw, Kd) [* gelu'(pre), column sums -> db] from q = q5(dY); c8 = (bytes, scale, amax words):
            the GELU' epilogue also writes the e5m2 copy of dX (and records its amax) for the two GEMMs that read it next."""
            dy8, sc = q
            Nw, Kd = W.shape
            w8t, wsc = f8.weight_t(W)
            dx = ops.empty((M, Kd), T, dy8)
            part = None
            if db is not None:
                R = K.gemm_nt_fp8_colsum_rows(M, Kd, Nw, Kd)
                part = ops.empty((R * Kd,), F32, dy8)
            c8b, c8s, c8a = c8 if c8 is not None else (None, None, None)
            K.gemm_nt_fp8_v2(dy8, Nw, w8t, Nw, dx, Kd, M, Kd, Nw, epi, a_kind=K.A_E5M2, aux=pre, partials=part,
                             alpha_a=sc[1:], alpha_w=wsc[1:], c8=c8b, c8_scale=c8s, c8_amax64=c8a, k128=f8.k128 and Nw % 128 == 0)
            if db is not None:
                ops.fold(part, R, Kd, Kd, db)
            return dx

        def dw(li, q, dyT, x, xslot, dW, db=None, x8=None):
            """dW (Nw, Kd) = dY^T X (+ db = colsum dY): on fp8 operands -- the e5m2 copy q already holds and the e4m3 activation: the
            copy the forward kept (x8, lean saving; its scale from the snapshot) or one re-made from the saved 16-bit tensor under
            the live forward scale -- when the state asks for it and the shape is served, else the 16-bit GEMM."""
            Nw, Kd = dW.shape
            if q is None or not f8.wgrad or not ops.fp8_tn_ok(M, Nw, Kd):
                if x is None:
                    raise RuntimeError("vit_pytorch_amd: fp8 lean saving kept only the e4m3 activations, but this backward cannot run its weight-gradient GEMMs on fp8 (fp8 state changed between forward and backward?)")
                ops.linear_dw(dyT, x, M, dW, db)
                return
            dy8, sc = q
            if x8 is not None:
                scx = ctx_scales[li, xslot]
            else:
                scx, _ = f8.slot(li, xslot)
                x8 = torch.empty((M, Kd), dtype=torch.uint8, device=x.device)
                K.quantize_fp8_delayed(x, x8, scx, None, K.FMT_E4M3)
            k128 = f8.k128
            splits = K.gemm_tn_fp8_splits(M, Nw, Kd, k128)
            ws = ops.empty((splits * Nw * Kd,), F32, x8)
            K.gemm_tn_fp8(dy8, Nw, x8, Kd, dW, Kd, M, Nw, Kd, ws, splits, alpha_y=sc[1:], alpha_x=scx[1:], k128=k128)
            if db is not None:
                ops.colsum(dyT, M, Nw, db)

        fork = _Fork(dy.device, fp8=f8 is not None)
        # final LayerNorm (vit.py:83)
        g, gb = newg()
        dnw, dnb = _grad_buf(norm_w), _grad_buf(norm_b)
        # colsum of g == bias gradient of the last layer's second FF Linear: the LayerNorm backward's finalize writes it straight into
        # that gradient's buffer (no float32 temporary, no cast launch)
        def bias_target(param):
            return _grad_buf(param) if param is not None else ops.empty((D,), F32, dy)
        dcol = bias_target(lp[(depth - 1) * NLP + 10] if depth else None)
        # (with dropout: gb and dcol carry the keep decisions of the LAST layer's post-FF2 dropout; g, the stream gradient, does not)
        ops.ln_bwd(dy, ctx.x_last, norm_w, ctx.stf[0], ctx.stf[1], M, D, dx_f32=g, dx_t=gb, dw=dnw, db=dnb, dcol=dcol,
                   drop=site(depth - 1, 3) if depth else None)
        ctx.x_last = None
        for li in reversed(range(depth)):
            ln1w, ln1b, wqkv, wout, bout, ln2w, ln2b, w1, b1, w2, b2 = lp[li * NLP:(li + 1) * NLP]
            xs, a1, st1, qkv, o, att_saved, x2, a2, st2, pre, act, kept8 = ctx.saved[li]
            ctx.saved[li] = None
            a1_8, a2_8, act_8, o_8 = kept8 if kept8 is not None else (None, None, None, None)
            del kept8
            if act is None and act_8 is None:         # recompute mode (_recompute_policy): rebuild the GELU output and the LayerNorm outputs of this layer
                act = ops.empty(pre.shape, T, pre)
                K.gelu_fwd(pre, act)
                a2 = ops.empty((M, D), T, x2)
                ops.ln_fwd(x2, ln2w, ln2b, M, D, a2)
                a1 = ops.empty((M, D), T, xs)
                ops.ln_fwd(xs, ln1w, ln1b, M, D, a1)
            gT = gb if bf else g
            base = li * NLP
            # ---- feed-forward branch (vit.py:18-25) ----
            q3 = q5(li, 4, gT) if f8 is not None else None
            dw2 = _grad_buf(w2)
            if f8 is not None:
                fork.run(lambda: dw(li, q3, gT, act, 2, dw2, x8=act_8), gT, act, dw2, q3, act_8)
            else:
                fork.run(lambda: ops.linear_dw(gT, act, M, dw2), gT, act, dw2)
            grads[base + 9] = dw2
            if b2 is not None:
                grads[base + 10] = dcol          # written by the LayerNorm backward above this layer (bias_target)
            dw1 = _grad_buf(w1)
            db1 = _grad_buf(b1) if b1 is not None else None
            qd = None
            if q3 is not None:      # the GELU' epilogue emits the e5m2 copy of dpre itself (no quantisation pass over the widest gradient)
                sc5, am5 = f8.slot(li, 5)
                dpre_8 = torch.empty((M, w2.shape[1]), dtype=torch.uint8, device=gT.device)
                dpre, db_done = dx8(q3, w2, L.EPI_GELU_BWD, pre, db1, c8=(dpre_8, sc5, am5)), True
                qd = (dpre_8, sc5)
                del dpre_8
            elif ctx.dg_mode:                      # `pre` is the gelu' factor the forward stored: the epilogue multiplies
                dpre, db_done = ops.linear_dx(gT, w2, M, gelu_dg=pre, db=db1)
            elif db1 is not None:
                dpre, db_done = ops.linear_dx(gT, w2, M, gelu_pre=pre, db=db1, drop=site(li, 2))   # b1's gradient out of the GEMM epilogue
            else:
                dpre, db_done = ops.linear_dx(gT, w2, M, gelu_pre=pre, drop=site(li, 2)), True
            del q3
            db_todo = None if db_done else db1
            if qd is None and f8 is not None:
                qd = q5(li, 5, dpre)          # recording pass (no scales yet): amax only
            if f8 is not None:
                fork.run(lambda: dw(li, qd, dpre, a2, 1, dw1, db_todo, x8=a2_8), dpre, a2, dw1, db1, qd, a2_8)
            else:
                fork.run(lambda: ops.linear_dw(dpre, a2, M, dw1, db_todo), dpre, a2, dw1, db1)
            grads[base + 7], grads[base + 8] = dw1, db1
            da2 = dx8(qd, w1) if qd is not None else ops.linear_dx(dpre, w1, M)
            del dpre, pre, act, qd, act_8, a2_8
            g2, g2b = newg()
            dl2w, dl2b = _grad_buf(ln2w), _grad_buf(ln2b)
            dcol2 = bias_target(bout if wout is not None else None)
            ops.ln_bwd(da2, x2, ln2w, st2[0], st2[1], M, D, gin=gb if s16 else g, dx_f32=g2, dx_t=g2b, dw=dl2w, db=dl2b, dcol=dcol2,
                       drop=site(li, 1))      # g2b / dcol2: gradient at to_out's output, behind its dropout
            grads[base + 5], grads[base + 6] = dl2w, dl2b
            del da2, g, gb
            g2T = g2b if bf else g2
            # ---- attention branch (vit.py:51-64) ----
            if wout is not None:
                q2 = q5(li, 6, g2T) if f8 is not None else None
                dwo = _grad_buf(wout)
                pair_out = None
                if f8 is not None:
                    fork.run(lambda: dw(li, q2, g2T, o, 3, dwo, x8=o_8), g2T, o, dwo, q2, o_8)
                elif not fork.enabled and T in ops.HALF and K.gemm_tn_pair_splits(M, wqkv.shape[0], wqkv.shape[1], wout.shape[0], wout.shape[1], T) > 0:
                    # issued together with to_qkv's weight gradient below, ONE launch (ops.linear_dw_pair: 36 tiles x 7 splits instead of
                    # 27 x 9 and 9 x 28 -- half the f32 slabs, a launch and a fold less: 31.55 -> 31.29 ms per step, same box).  Only without
                    # the side stream: beside the dX chain the deferred gradient loses more overlap than the pair saves (+0.5 ms measured).
                    pair_out = (g2T, o, dwo)
                else:
                    fork.run(lambda: ops.linear_dw(g2T, o, M, dwo), g2T, o, dwo)
                grads[base + 3] = dwo
                if bout is not None:
                    grads[base + 4] = dcol2
                do = dx8(q2, wout) if q2 is not None else ops.linear_dx(g2T, wout, M)
                del q2
            else:
                do = g2T
                pair_out = None
            dqkv = ops.attn_bwd(qkv, o, do, att_saved, B, N, heads, dim_head, scale, drop=site(li, 0))
            qq = q5(li, 7, dqkv) if f8 is not None else None
            dwq = _grad_buf(wqkv)
            if f8 is not None:
                fork.run(lambda: dw(li, qq, dqkv, a1, 0, dwq, x8=a1_8), dqkv, a1, dwq, qq, a1_8)
            elif pair_out is not None:
                po = pair_out
                fork.run(lambda: ops.linear_dw_pair(dqkv, a1, dwq, po[0], po[1], po[2], M), dqkv, a1, dwq, *po)
            else:
                fork.run(lambda: ops.linear_dw(dqkv, a1, M, dwq), dqkv, a1, dwq)
            pair_out = None
            grads[base + 2] = dwq
            da1 = dx8(qq, wqkv) if qq is not None else ops.linear_dx(dqkv, wqkv, M)
            del dqkv, do, qkv, o, qq, o_8, a1_8
            g1, g1b = newg()
            dl1w, dl1b = _grad_buf(ln1w), _grad_buf(ln1b)
            dcol = bias_target(lp[(li - 1) * NLP + 10] if li > 0 else None)
            ops.ln_bwd(da1, xs, ln1w, st1[0], st1[1], M, D, gin=g2b if s16 else g2, dx_f32=g1, dx_t=g1b, dw=dl1w, db=dl1b, dcol=dcol,
                       drop=site(li - 1, 3) if li > 0 else None)   # feeds the layer below: behind ITS post-FF2 dropout
            grads[base + 0], grads[base + 1] = dl1w, dl1b
            g, gb = g1, g1b
            del g2, g2b, da1
            ops.flush_folds()                    # this layer's parameter-gradient folds: one launch, before the sink may send them
            sk = _sink()
            if sk is not None:
                tick = getattr(sk, "layer_tick", None)
                if tick is not None:
                    tick()                       # counts down the CU reserve a collective's launch opened (parallel.FlatGradSink)
            if sk is not None and sk.wants_layer(li):
                fork.join()                      # this layer's weight gradients (side stream) are part of the chunk
                sk.stage_done("layer", li)
        ops.flush_folds()
        fork.join()
        if f8 is not None:
            f8.end_of_backward()
        s = _sink()
        if s is not None:
            s.stage_done("transformer")
        if in_dtype == F32:
            if g is None:                        # 16-bit stream, float32 input (the embedding stage's f32 stream): widen once
                g = ops.empty((M, D), F32, dy)
                K.cast(gb, g)
            dx = g.view(B, N, D)
        else:
            dx = (gb if gb is not None else g).view(B, N, D)
        return (dx, None, None, None, None, None, _ret(dnw), _ret(dnb), *[_ret(t) for t in grads])


class PatchEmbedFn(torch.autograd.Function):
    """to_patch_embedding + cls token + positional embedding (vit.py:99-104, :122-128;
    simple_vit.py:90-95, :113-114).  Output: float32 residual stream (B, N, D)."""

    @staticmethod
    def forward(ctx, img, p1: int, p2: int, ln1w, ln1b, w, b, ln2w, ln2b, cls, pos, out16: bool = False):
        """out16: write the stream in the parameter dtype (the caller knows that the transformer stage behind it runs the 16-bit
        forward stream, forward_stream_is_16bit): saves the float32 write here and the cast there."""
        K.require_device(img, w)
        T = w.dtype
        if img.dtype != T:
            raise VitkError(f"input dtype {img.dtype} != parameter dtype {T} (same rule as the reference nn.Linear)")
        img = img.contiguous()
        B, C, H, W = img.shape
        if H % p1 or W % p2:
            raise VitkError("Image dimensions must be divisible by the patch size.")
        Np = (H // p1) * (W // p2)
        P = C * p1 * p2
        D = w.shape[0]
        _check_dims(D, "patch embedding (dim)")
        _check_dims(P, "patch embedding (patch_dim)")
        ncls = cls.shape[0] if cls is not None else 0
        N = Np + ncls
        if pos is not None and pos.shape[0] < N:
            raise VitkError(f"sequence of {N} tokens exceeds the positional table ({pos.shape[0]} rows)")
        Mp = B * Np
        pn = ops.empty((Mp, P), T, img)
        # Rearrange + LayerNorm(patch_dim) (vit.py:100-101).  16-bit images of 3 channels with 16 x 16 patches: ONE kernel gathers the patch
        # straight from the NCHW image inside the LayerNorm's load (no `patches` tensor, no patchify pass; the backward re-gathers from the
        # image, which is kept instead); everything else: patchify, then LayerNorm.
        gather = T in ops.HALF and img.data_ptr() % 16 == 0 and K.patch_ln_serves(img, C, H, W, p1, p2)      # (a misaligned view: patchify + LayerNorm)
        if gather:
            patches = None
            st1 = (ops.empty((Mp,), F32, img), ops.empty((Mp,), F32, img))
            K.patch_ln_fwd(img, ln1w, ln1b, pn, st1[0], st1[1], B, C, H, W, p1, p2, ops.LN_EPS)
        else:
            patches = ops.empty((Mp, P), T, img)
            K.patchify(img, patches, B, C, H, W, p1, p2)
            st1 = ops.ln_fwd(patches, ln1w, ln1b, Mp, P, pn)
        # patch_dim that is not a multiple of 32 (ViT-H/14: 3*14*14 = 588): zero-pad the contraction to the MFMA K-step
        Pp = (P + 31) // 32 * 32 if (T != F32 and P % 32) else P
        if Pp != P:
            pn = ops.pad_cols(pn, Mp, P, Pp)
            w_mm = ops.pad_cols(w, D, P, Pp)
        else:
            w_mm = w
        y = ops.linear_fwd(pn, w_mm, b, Mp)
        x0 = ops.empty((B, N, D), T if (out16 and T in ops.HALF) else F32, img)
        st2 = ops.ln_fwd(y, ln2w, ln2b, Mp, D, x0, omap=RowMap(Np, N, ncls), add=pos, add_group=Np, add_off=ncls)
        if ncls:
            K.write_cls_rows(x0, cls, pos, B, N, D, ncls)
        ctx.save_for_backward(ln1w, ln1b, w, ln2w, ln2b, *([b] if b is not None else []))
        ctx.inter = (patches if not gather else img, st1, pn, y, st2)
        ctx.gather = (B, C, H, W, p1, p2) if gather else None
        ctx.img_version = img._version if gather else None       # the backward re-gathers from the caller's image
        ctx.geom = (B, C, H, W, p1, p2)
        ctx.pad = (Pp, w_mm if Pp != P else None)
        ctx.cls_pos = (cls, pos)
        ctx.meta = (B, Np, N, P, D, ncls, b is not None, cls is not None, pos is not None and pos.requires_grad,
                    pos.shape if pos is not None else None)
        return x0

    @staticmethod
    def backward(ctx, g):
        sv = ctx.saved_tensors
        ln1w, ln1b, w, ln2w, ln2b = sv[:5]
        bparam = sv[5] if len(sv) > 5 else None
        if ctx.inter is None:
            raise RuntimeError("vit_pytorch_amd: backward through this fused stage a second time -- its saved activations were released during the first backward (retain_graph is not supported by the fused engine)")
        patches, st1, pn, y, st2 = ctx.inter
        ctx.inter = None
        B, Np, N, P, D, ncls, has_b, has_cls, pos_grad, pos_shape = ctx.meta
        T = w.dtype
        Mp = B * Np
        g = g.contiguous()
        if g.dtype != F32 and not (g.dtype == T and T in ops.HALF):        # the 16-bit stream's gradient is consumed as it is
            g32 = ops.empty((B, N, D), F32, g)
            K.cast(g, g32)
            g = g32
        dpos = dcls = None
        cls_p, pos_p = ctx.cls_pos
        if pos_grad or has_cls:
            # d pos[n] = sum_b g[b, n] ; d cls = d pos[0:ncls]   (x[b,n] = tok[b,n] + pos[n]; x[b,0] = cls + pos[0])
            gsum = ops.empty((N, D), T, g)
            ops.colsum(g, B, N * D, gsum)
            if pos_grad:
                dpos = _grad_buf(pos_p)
                if pos_shape[0] != N:
                    dpos[N:].zero_()  # rows of the table this (smaller) input never touched (vit.py:125-127)
                K.cast(gsum, dpos[:N])
            if has_cls:
                dcls = _grad_buf(cls_p)
                if ncls:
                    K.cast(gsum[:ncls], dcls)
        # LN(dim) backward: dy rows are the patch rows of g (behind the cls slot)
        dyp = ops.empty((Mp, D), T, g)
        dl2w, dl2b = _grad_buf(ln2w), _grad_buf(ln2b)
        ops.ln_bwd(g, y, ln2w, st2[0], st2[1], Mp, D, dx_t=dyp if T != F32 else None, dx_f32=dyp if T == F32 else None,
                   dw=dl2w, db=dl2b, dymap=RowMap(Np, N, ncls))
        dw = _grad_buf(w)
        db = _grad_buf(bparam) if has_b else None
        Pp, w_pad = ctx.pad
        if Pp != P:
            dw_pad = ops.empty((D, Pp), T, g)
            ops.linear_dw(dyp, pn, Mp, dw_pad, db)
            ops.unpad_cols(dw_pad, D, Pp, P, out=dw)
            dpn = ops.unpad_cols(ops.linear_dx(dyp, w_pad, Mp), Mp, Pp, P)
        else:
            ops.linear_dw(dyp, pn, Mp, dw, db)
            dpn = ops.linear_dx(dyp, w, Mp)
        dl1w, dl1b = _grad_buf(ln1w), _grad_buf(ln1b)
        dimg = None
        if ctx.gather is not None and patches._version != ctx.img_version:
            # what autograd's saved-tensor check would have said: the backward re-gathers the patches from the caller's image
            raise RuntimeError("vit_pytorch_amd: the input image was modified in place between forward and backward (the fused patch "
                               "embedding re-reads it in backward); pass a copy, or modify it after backward")
        if ctx.needs_input_grad[0]:
            # the INPUT requires a gradient (saliency maps, adversarial inputs: the reference differentiates through Rearrange + LayerNorm,
            # vit.py:100-101): LayerNorm(patch_dim) backward with its dx, then the inverse scatter of the Rearrange
            gB, gC, gH, gW, gp1, gp2 = ctx.geom
            if ctx.gather is not None:                        # the gather path kept the image instead of the patches: re-form them
                img_saved = patches
                patches = ops.empty((Mp, P), T, dpn)
                K.patchify(img_saved, patches, gB, gC, gH, gW, gp1, gp2)
            dpatch = ops.empty((Mp, P), T, dpn)
            ops.ln_bwd(dpn, patches, ln1w, st1[0], st1[1], Mp, P, dx_t=dpatch if T != F32 else None, dx_f32=dpatch if T == F32 else None,
                       dw=dl1w, db=dl1b)
            dimg = ops.empty((gB, gC, gH, gW), T, dpn)
            K.unpatchify(dpatch, dimg, gB, gC, gH, gW, gp1, gp2)
        elif ctx.gather is not None:     # the image needs no gradient: only dgamma / dbeta, xhat re-formed from the image by the same gather
            gB, gC, gH, gW, gp1, gp2 = ctx.gather
            nblk = K.patch_ln_bwd_blocks(Mp)
            part = ops.empty((2 * nblk * P,), F32, dpn)
            K.patch_ln_bwd_params(dpn, patches, st1[0], st1[1], part, gB, gC, gH, gW, gp1, gp2)       # `patches` is the image here
            K.colsum_partials(part, nblk, P, P, dl1w)
            K.colsum_partials(part[nblk * P:], nblk, P, P, dl1b)
        else:
            ops.ln_bwd(dpn, patches, ln1w, st1[0], st1[1], Mp, P, dw=dl1w, db=dl1b)  # the image needs no gradient
        s = _sink()
        if s is not None:
            s.stage_done("patch_embed")
        return (dimg, None, None, _ret(dl1w), _ret(dl1b), _ret(dw), _ret(db), _ret(dl2w), _ret(dl2b), _ret(dcls), _ret(dpos), None)


def _head_dx(dl, w, out, ldo, B, D, C):
    """d(pooled) = dl . W  (B x C by C x D).  16-bit: an NT GEMM on the MFMA kernel against the cached W^T, the class dimension
    zero-padded to a multiple of 32 (C = 1000 -> 1024); it used to run on the VALU coverage kernel (0.2 ms of the ViT-B step)."""
    T = w.dtype
    if T in ops.HALF and D % 4 == 0 and ldo % 4 == 0:
        Cp = (C + 31) // 32 * 32
        wt = ops.transpose_weight(w, pad_to=Cp)                         # (D, Cp)
        dlp = dl if Cp == C else ops.pad_cols(dl, B, C, Cp)
        K.gemm_nt_bf16(dlp, Cp, wt, Cp, out, ldo, B, D, Cp)
    else:
        K.gemm_generic(K.mat(dl, C, 1), K.mat(w, D, 1), K.mat(out, ldo, 1), B, D, C)


class HeadFn(torch.autograd.Function):
    """pool ('cls' -> row 0, 'mean' -> mean over tokens) + Linear head (vit.py:135-138)."""

    @staticmethod
    def forward(ctx, y, pool_mean: bool, w, b):
        K.require_device(y, w)
        B, N, D = y.shape
        T = w.dtype
        C = w.shape[0]
        y = y.contiguous()
        logits = ops.empty((B, C), T, y)
        fast = T in ops.HALF and D % 32 == 0 and C % 8 == 0          # MFMA kernels (row strides are arguments)
        if pool_mean:
            pooled = ops.empty((B, D), T, y)
            K.mean_pool_fwd(y, pooled, B, N, D)
            src, ld = pooled, D
            ctx.pooled = pooled
        else:  # row 0 of every image read in place through the row stride N*D
            src, ld = y, N * D
            ctx.pooled = None
        if fast:
            K.gemm_nt_bf16(src, ld, w, D, logits, C, B, C, D, L.EPI_BIAS if b is not None else L.EPI_NONE, bias=b)
        else:
            K.gemm_generic(K.mat(src, ld, 1), K.mat(w, 1, D), K.mat(logits, C, 1), B, C, D, bias=b)
        ctx.save_for_backward(y, w, *([b] if b is not None else []))
        ctx.meta = (B, N, D, C, pool_mean, b is not None)
        return logits

    @staticmethod
    def backward(ctx, dl):
        y, w = ctx.saved_tensors[:2]
        B, N, D, C, pool_mean, has_b = ctx.meta
        T = w.dtype
        dl = dl.contiguous()
        dw = _grad_buf(w)
        db = _grad_buf(ctx.saved_tensors[2]) if has_b else None
        dy = torch.zeros((B, N, D), dtype=T, device=dl.device) if not pool_mean else ops.empty((B, N, D), T, dl)
        src, ld = (ctx.pooled, D) if pool_mean else (y, N * D)
        ops.linear_dw(dl, src, B, dw, db if has_b else None, ldx=ld)       # dW = dl^T . pooled rows (+ db = colsum(dl))
        if pool_mean:
            dpooled = ops.empty((B, D), T, dl)
            _head_dx(dl, w, dpooled, D, B, D, C)
            K.mean_pool_bwd(dpooled, dy, B, N, D)
        else:
            _head_dx(dl, w, dy, N * D, B, D, C)      # writes row 0 of each image
        s = _sink()
        if s is not None:
            s.stage_done("head")
        return dy, None, _ret(dw), _ret(db)


# ==================================================================================================
# NaViT transformer stack (na_vit.py:171-193) on packed tokens -- the same fusion plan as TransformerFn
# (f32 residual stream, GEMM epilogues for bias+GELU / residual, LayerNorm backward producing the bias
# column sums, weight gradients on the side stream) with NaViT's block: LayerNorms without bias, separate
# to_q / to_kv weights (concatenated once per call so q|k|v is ONE GEMM into a merged buffer), RMSNorm on
# q and k per head, variable-length attention over per-image token ranges (scale = 1), bias-free to_out.
# ==================================================================================================
NLP_NAVIT = 11  # ln1_g, wq, wkv, gq, gk, wout, ln2_g, w1, b1, w2, b2


def pack_navit_layer_params(attn, ff) -> List[Tensor]:
    return [attn.norm.gamma, attn.to_q.weight, attn.to_kv.weight, attn.q_norm.gamma, attn.k_norm.gamma,
            attn.to_out[0].weight, ff[0].gamma, ff[1].weight, ff[1].bias, ff[4].weight, ff[4].bias]


_CAT_CACHE = {}     # (id(a), id(b)) -> (weakref a, weakref b, weight_key a, weight_key b, concatenated Parameter)


def _cat_rows(a: Tensor, b: Tensor) -> Tensor:
    """to_q | to_kv as ONE (3I, D) weight (na_vit.py:125-126 keeps them apart; one GEMM serves both).  Cached per parameter VALUE and
    returned as a (frozen) Parameter, so that ops.nt_weight gives it the K-blocked copies and the cached transpose every other
    Linear weight gets -- it used to be rebuilt by two cast launches per layer per forward, and, not being a Parameter, fed the
    QKV / dX GEMMs from the slower row-major layout and re-transposed every step.  Never cached during HIP-graph capture."""
    import weakref
    from ._epoch import weight_key
    capturing = torch.cuda.is_current_stream_capturing()
    cacheable = not capturing and ops.is_weight(a) and ops.is_weight(b)
    k = (id(a), id(b))
    if cacheable:
        ka, kb = weight_key(a), weight_key(b)
        ent = _CAT_CACHE.get(k)
        if ent is not None and ent[0]() is a and ent[1]() is b and ent[2] == ka and ent[3] == kb:
            return ent[4]
    out = torch.empty((a.shape[0] + b.shape[0], a.shape[1]), dtype=a.dtype, device=a.device)
    K.cast(a, out[:a.shape[0]])
    K.cast(b, out[a.shape[0]:])
    if not cacheable:
        return out
    outp = torch.nn.Parameter(out, requires_grad=False)
    drop = lambda _r, k=k: _CAT_CACHE.pop(k, None)
    _CAT_CACHE[k] = (weakref.ref(a, drop), weakref.ref(b, drop), ka, kb, outp)
    return outp


def packed_dropout_fusable(T, Tn: int, D: int, heads: int, dim_head: int, F: int) -> bool:
    """Active dropout (na_vit.py:100-103,163,171-175) runs inside the packed engine when the three Linear layers that carry one are
    served by the 256-row GEMM kernel (its epilogues hold the fused dropout); the packed attention kernels always take theirs."""
    I = heads * dim_head
    return T in ops.HALF and ops.fused_dropout_ok(T, Tn, D, I) and ops.fused_dropout_ok(T, Tn, F, D) and ops.fused_dropout_ok(T, Tn, D, F)


class PackedTransformerFn(torch.autograd.Function):
    """na_vit.Transformer.forward (na_vit.py:196-214) on packed tokens.  drop_p > 0 (training): the four dropouts of a layer --
    inside scaled_dot_product_attention (na_vit.py:163), after to_out (:172), after the GELU (:101), after the second FeedForward
    Linear (:103) -- are taken inside the packed attention kernels and the GEMM epilogues, exactly as TransformerFn does (per-site
    seeds from drop_seed, keep decisions regenerated by the backward kernels, no mask tensor)."""

    @staticmethod
    def forward(ctx, x, segs, heads: int, dim_head: int, drop_p: float, drop_seed: int, norm_g, *lp):
        K.require_device(x, norm_g)
        depth = len(lp) // NLP_NAVIT
        T = norm_g.dtype
        Tn, D = x.shape
        _check_dims(D, "NaViT Transformer")
        I = heads * dim_head
        d = dim_head
        if not ops.attn_varlen_ok(T, d):
            raise VitkError("PackedTransformerFn: fused NaViT stack needs 16-bit parameters and dim_head 32 / 48 / 64 / 80 / 96")
        x = x.contiguous()
        if x.dtype == F32:
            xs = x
        else:
            xs = ops.empty((Tn, D), F32, x)
            K.cast(x, xs)
        saved = []
        keep = any(ctx.needs_input_grad) and caller_grad_mode()      # no_grad / eval: drop each layer's activations as soon as the layer is done
        # (needs_input_grad mirrors requires_grad even under torch.no_grad(): the module-level caller records the real mode, _epoch.py)
        if drop_p > 0.0 and depth and not packed_dropout_fusable(T, Tn, D, heads, dim_head, lp[7].shape[0]):
            raise VitkError("PackedTransformerFn: this shape does not take the fused dropout path (caller must check packed_dropout_fusable)")
        site = (lambda li, k: (drop_p, _hash32(drop_seed + 4 * li + k))) if drop_p > 0.0 else (lambda li, k: None)
        att_drop = lambda li: site(li, 0) or (0.0, 0)
        dg_mode = bool(keep and depth and drop_p == 0.0 and lp[8] is not None and ops.gelu_dg_ok(T, Tn, lp[7].shape[0], D))     # see TransformerFn
        ctx.dg_mode = dg_mode
        ctx.grad16 = ops.grad_stream_16()       # the backward's stream dtype is decided HERE (a per-call policy -- autocast -- is gone by backward time)
        if T in ops.HALF:       # the K-blocked weight copies of the step in one launch (see TransformerFn)
            ops.prepack_weights([lp[li * NLP_NAVIT + j] for li in range(depth) for j in (5, 7, 9)], Tn)       # (q | kv go through _cat_rows: their concatenation is the GEMM operand)
        for li in range(depth):
            ln1g, wq, wkv, gq, gk, wout, ln2g, w1, b1, w2, b2 = lp[li * NLP_NAVIT:(li + 1) * NLP_NAVIT]
            a1 = ops.empty((Tn, D), T, xs)
            st1 = ops.ln_fwd(xs, ln1g, None, Tn, D, a1)
            wcat = _cat_rows(wq, wkv)                                   # (3I, D): q | k | v in one GEMM
            qkv = ops.linear_fwd(a1, wcat, None, Tn)
            gqf, gkf = gq.reshape(heads, d).contiguous(), gk.reshape(heads, d).contiguous()
            qn = ops.empty((Tn, I), T, xs); kn = ops.empty((Tn, I), T, xs)
            rq = ops.empty((Tn * heads,), F32, xs); rk = ops.empty((Tn * heads,), F32, xs)
            K.rmsnorm_heads_fwd(qkv, 3 * I, gqf, qn, I, rq, Tn, heads, d)
            K.rmsnorm_heads_fwd(qkv, 3 * I, gkf, kn, I, rk, Tn, heads, d, x_off=I)
            o = ops.empty((Tn, I), T, xs)
            lse = ops.empty((heads, Tn), F32, xs)
            K.attn_varlen_fwd_bf16(K.hnd(qn, d, I), K.hnd(kn, d, I), K.hnd(qkv, d, 3 * I, offset=2 * I), K.hnd(o, d, I), lse,
                                   segs.cu_q, segs.cu_k, segs.qblk_seg, segs.qblk_r0, segs.nqblk, Tn, heads, d, 1.0, *att_drop(li))
            x2 = ops.linear_fwd(o, wout, None, Tn, resid=xs, drop=site(li, 1))
            a2 = ops.empty((Tn, D), T, xs)
            st2 = ops.ln_fwd(x2, ln2g, None, Tn, D, a2)
            act, pre = ops.linear_fwd(a2, w1, b1, Tn, gelu=True, drop=site(li, 2), save_dg=dg_mode)      # dg_mode: `pre` holds gelu'(pre) (TransformerFn)
            x3 = ops.linear_fwd(act, w2, b2, Tn, resid=x2, drop=site(li, 3))
            if keep:
                saved.append((xs, a1, st1, wcat, qkv, gqf, gkf, qn, kn, rq, rk, o, lse, x2, a2, st2, pre, act))
            xs = x3
        y = ops.empty((Tn, D), T, xs)
        stf = ops.ln_fwd(xs, norm_g, None, Tn, D, y)
        ctx.saved = saved
        reset_grad_mode()       # the caller's note was for THIS forward only (_epoch.py)
        ctx.x_last, ctx.stf = xs, stf
        ctx.meta = (segs, heads, dim_head, depth, Tn, D, x.dtype)
        ctx.drop = (drop_p, drop_seed)
        ctx.save_for_backward(norm_g, *lp)
        return y

    @staticmethod
    def backward(ctx, dy):
        segs, heads, d, depth, Tn, D, in_dtype = ctx.meta
        drop_p, drop_seed = ctx.drop
        site = (lambda li, k: (drop_p, _hash32(drop_seed + 4 * li + k))) if drop_p > 0.0 else (lambda li, k: None)
        att_drop = lambda li: site(li, 0) or (0.0, 0)
        sv = list(ctx.saved_tensors)
        norm_g, lp = sv[0], sv[1:]
        T = norm_g.dtype
        I = heads * d
        dy = dy.contiguous()
        if ctx.x_last is None:
            raise RuntimeError("vit_pytorch_amd: backward through this fused stage a second time -- its saved activations were released during the first backward (retain_graph is not supported by the fused engine)")
        grads: List[Optional[Tensor]] = [None] * len(lp)
        fork = _Fork(dy.device)

        s16 = T in ops.HALF and drop_p == 0.0 and D % 4 == 0 and ctx.grad16        # the backward's residual stream in the parameter dtype (same guards as TransformerFn.backward)

        def newg():
            return (None if s16 else ops.empty((Tn, D), F32, dy)), ops.empty((Tn, D), T, dy)

        g, gb = newg()
        dng = _grad_buf(norm_g)
        dcol = ops.empty((D,), F32, dy)
        # (with dropout: gb and dcol carry the keep decisions of the LAST layer's post-FF2 dropout; g, the stream gradient, does not)
        ops.ln_bwd(dy, ctx.x_last, norm_g, ctx.stf[0], ctx.stf[1], Tn, D, dx_f32=g, dx_t=gb, dw=dng, dcol=dcol,
                   drop=site(depth - 1, 3) if depth else None)
        ctx.x_last = None
        for li in reversed(range(depth)):
            ln1g, wq, wkv, gq, gk, wout, ln2g, w1, b1, w2, b2 = lp[li * NLP_NAVIT:(li + 1) * NLP_NAVIT]
            xs, a1, st1, wcat, qkv, gqf, gkf, qn, kn, rq, rk, o, lse, x2, a2, st2, pre, act = ctx.saved[li]
            ctx.saved[li] = None
            base = li * NLP_NAVIT
            # ---- feed-forward ----
            dw2 = _grad_buf(w2)
            fork.run(lambda: ops.linear_dw(gb, act, Tn, dw2), gb, act, dw2)
            db2 = _grad_buf(b2)
            K.cast(dcol, db2)
            grads[base + 9], grads[base + 10] = dw2, db2
            dw1, db1 = _grad_buf(w1), _grad_buf(b1)
            if ctx.dg_mode:
                dpre, db_done = ops.linear_dx(gb, w2, Tn, gelu_dg=pre, db=db1)
            else:
                dpre, db_done = ops.linear_dx(gb, w2, Tn, gelu_pre=pre, db=db1, drop=site(li, 2))
            db_todo = None if db_done else db1
            fork.run(lambda: ops.linear_dw(dpre, a2, Tn, dw1, db_todo), dpre, a2, dw1, db1)
            grads[base + 7], grads[base + 8] = dw1, db1
            da2 = ops.linear_dx(dpre, w1, Tn)
            del dpre, pre, act
            g2, g2b = newg()
            dl2 = _grad_buf(ln2g)
            ops.ln_bwd(da2, x2, ln2g, st2[0], st2[1], Tn, D, gin=gb if s16 else g, dx_f32=g2, dx_t=g2b, dw=dl2,
                       drop=site(li, 1))      # g2b: gradient at to_out's output, behind its dropout
            grads[base + 6] = dl2
            del da2, g, gb
            # ---- attention ----
            dwo = _grad_buf(wout)
            fork.run(lambda: ops.linear_dw(g2b, o, Tn, dwo), g2b, o, dwo)
            grads[base + 5] = dwo
            do = ops.linear_dx(g2b, wout, Tn)
            dqn = ops.empty((Tn, I), T, dy); dkn = ops.empty((Tn, I), T, dy)
            dqkv = ops.empty((Tn, 3 * I), T, dy)
            delta = ops.empty((heads, Tn), F32, dy)
            K.attn_varlen_bwd_bf16(K.hnd(qn, d, I), K.hnd(kn, d, I), K.hnd(qkv, d, 3 * I, offset=2 * I), K.hnd(o, d, I),
                                   K.hnd(do, d, I), lse, delta, K.hnd(dqn, d, I), K.hnd(dkn, d, I),
                                   K.hnd(dqkv, d, 3 * I, offset=2 * I), segs.cu_q, segs.cu_k, segs.qblk_seg, segs.qblk_r0,
                                   segs.nqblk, segs.kblk_seg, segs.kblk_r0, segs.nkblk, Tn, heads, d, 1.0, *att_drop(li))
            dgq = torch.empty_like(gqf); dgk = torch.empty_like(gkf)
            part = ops.empty((K.rmsnorm_heads_partials(Tn, heads, d),), F32, dy)
            K.rmsnorm_heads_bwd(dqn, I, qkv, 3 * I, gqf, rq, dqkv, 3 * I, dgq, part, Tn, heads, d)
            K.rmsnorm_heads_bwd(dkn, I, qkv, 3 * I, gkf, rk, dqkv, 3 * I, dgk, part, Tn, heads, d, x_off=I, dx_off=I)
            grads[base + 3], grads[base + 4] = dgq.view(gq.shape), dgk.view(gk.shape)
            dwq, dwkv = _grad_buf(wq), _grad_buf(wkv)
            fork.run(lambda: (ops.linear_dw(dqkv, a1, Tn, dwq, ldy=3 * I), ops.linear_dw(dqkv[:, I:], a1, Tn, dwkv, ldy=3 * I)),
                     dqkv, a1, dwq, dwkv)
            grads[base + 1], grads[base + 2] = dwq, dwkv
            da1 = ops.linear_dx(dqkv, wcat, Tn)
            del dqkv, do, qkv, o, dqn, dkn
            g1, g1b = newg()
            dl1 = _grad_buf(ln1g)
            dcol = ops.empty((D,), F32, dy)
            ops.ln_bwd(da1, xs, ln1g, st1[0], st1[1], Tn, D, gin=g2b if s16 else g2, dx_f32=g1, dx_t=g1b, dw=dl1, dcol=dcol,
                       drop=site(li - 1, 3) if li > 0 else None)   # feeds the layer below: behind ITS post-FF2 dropout
            grads[base + 0] = dl1
            g, gb = g1, g1b
            del g2, g2b, da1
        fork.join()
        if in_dtype == F32 and g is None:        # 16-bit stream, float32 input: widen once
            g = ops.empty((Tn, D), F32, dy)
            K.cast(gb, g)
        dx = g if in_dtype == F32 else gb
        return (dx, None, None, None, None, None, _ret(dng), *[_ret(t) for t in grads])
