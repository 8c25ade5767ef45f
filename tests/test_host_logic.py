"""CPU: drop-in contract of the modules (constructor, module tree, state_dict) and host-side logic."""
import pytest
import torch

import bench
from oracle import vit_oracle as O
from oracle.params import CASES, make_params, param_shapes
from vit_pytorch_amd import SimpleViT, ViT
from vit_pytorch_amd import simple_vit as SV
from vit_pytorch_amd import vit as V
from vit_pytorch_amd._lib import VitkError
from vit_pytorch_amd.parallel import FlatGradSink, _ordered_params


@pytest.mark.parametrize("name", list(CASES))
def test_state_dict_contract(name):
    c = CASES[name]
    m = (ViT if c["kind"] == "vit" else SimpleViT)(**c["cfg"])
    sd = m.state_dict()
    exp = param_shapes(c["kind"], c["cfg"])
    assert list(sd.keys()) == list(exp.keys())          # names AND registration order
    for k in sd:
        assert tuple(sd[k].shape) == exp[k], k
    m.load_state_dict(make_params(c["kind"], c["cfg"], 0), strict=True)


def test_module_tree_matches_reference_shape():
    v = ViT(image_size=32, patch_size=8, num_classes=10, dim=32, depth=2, heads=2, mlp_dim=64)
    assert [n for n, _ in v.named_children()] == ["to_patch_embedding", "dropout", "transformer", "to_latent", "mlp_head"]
    s = SimpleViT(image_size=32, patch_size=8, num_classes=10, dim=32, depth=2, heads=2, mlp_dim=64)
    assert [n for n, _ in s.named_children()] == ["to_patch_embedding", "transformer", "to_latent", "linear_head"]
    # consumers index into these (mae.py:28-31, recorder.py:26-29, dino.py:138-140, accept_video_wrapper.py:72-74)
    assert len(v.to_patch_embedding) == 4 and isinstance(v.to_patch_embedding[1], torch.nn.LayerNorm)
    assert isinstance(v.to_patch_embedding[2], torch.nn.Linear) and v.to_patch_embedding[2].weight.shape == (32, 192)
    assert v.patch_size == (8, 8) and v.pool == "cls"
    attn, ff = v.transformer.layers[0]
    assert isinstance(attn.attend, torch.nn.Softmax) and isinstance(attn.to_out, torch.nn.Sequential)
    assert attn.to_qkv.bias is None and isinstance(ff.net[2], torch.nn.GELU)
    assert list(v.children())[-2] is v.to_latent
    assert "pos_embedding" not in s.state_dict() and s.pos_embedding.shape == (16, 32)   # plain tensor attribute
    # to_out collapses to Identity iff heads == 1 and dim_head == dim (vit.py:34,49)
    a = V.Attention(16, heads=1, dim_head=16)
    assert isinstance(a.to_out, torch.nn.Identity)
    t = V.Transformer(16, 1, 2, 8, 32)                   # vit.Transformer(dim, depth, heads, dim_head, mlp_dim)
    assert len(t.layers) == 1


def test_constructor_assertions_and_tuple_sizes():
    with pytest.raises(AssertionError, match="divisible"):
        ViT(image_size=30, patch_size=8, num_classes=2, dim=8, depth=1, heads=1, mlp_dim=8)
    with pytest.raises(AssertionError, match="pool type"):
        ViT(image_size=32, patch_size=8, num_classes=2, dim=8, depth=1, heads=1, mlp_dim=8, pool="max")
    with pytest.raises(AssertionError, match="multiple of 4"):
        SimpleViT(image_size=32, patch_size=8, num_classes=2, dim=10, depth=1, heads=1, mlp_dim=8)
    m = ViT(image_size=(24, 32), patch_size=(4, 8), num_classes=0, dim=8, depth=1, heads=1, mlp_dim=8, pool="mean")
    assert m.mlp_head is None and m.cls_token.shape == (0, 8) and m.pos_embedding.shape == (24, 8)
    assert V.pair(3) == (3, 3) and V.pair((2, 5)) == (2, 5)


def test_sincos_table_equals_oracle():
    assert torch.equal(SV.posemb_sincos_2d(3, 5, 16), O.posemb_sincos_2d(3, 5, 16))


def test_cpu_tensors_raise_never_fall_back():
    m = SimpleViT(image_size=32, patch_size=8, num_classes=10, dim=32, depth=1, heads=2, mlp_dim=64)
    with pytest.raises(VitkError, match="HIP"):
        m(torch.randn(2, 3, 32, 32))
    with pytest.raises(VitkError, match="HIP"):
        m.transformer(torch.randn(2, 16, 32))
    with pytest.raises(VitkError, match="HIP"):
        m.to_patch_embedding(torch.randn(2, 3, 32, 32))


def test_flop_model_matches_survey_table():
    assert abs(3 * bench.fwd_gflop_per_image(bench.CONFIGS["vit_b16"][0]) - 105.383) < 1e-3
    assert abs(3 * bench.fwd_gflop_per_image(bench.CONFIGS["vit_l16"][0]) - 369.328) < 1e-3


def test_flat_gradient_layout_is_reverse_readiness():
    m = ViT(image_size=32, patch_size=8, num_classes=10, dim=32, depth=3, heads=2, mlp_dim=64)
    names = {id(p): n for n, p in m.named_parameters()}
    params, n_early, layer_end = _ordered_params(m)
    assert sorted(layer_end) == [0, 1, 2] and layer_end[2] < layer_end[1] < layer_end[0] == n_early
    order = [names[id(p)] for p in params]
    assert order[0].startswith("mlp_head") and order[2].startswith("transformer.norm")
    layer_of = [int(n.split(".")[2]) for n in order if n.startswith("transformer.layers.")]
    assert layer_of == sorted(layer_of, reverse=True)                     # depth-1 ... 0
    late = order[n_early:]
    assert late and all(n.startswith(("to_patch_embedding", "cls_token", "pos_embedding")) for n in late)
    sink = FlatGradSink(m)
    assert sink.total % 8 == 0 and all(o % 8 == 0 for o in sink.offsets)  # 16-byte aligned slices for the kernels
    assert sink.boundary == sink.offsets[n_early]
    w = m.transformer.layers[1][1].net[1].weight
    buf = sink.buffer_for(w)
    idx = [i for i, q in enumerate(params) if q is w][0]
    assert buf.shape == w.shape and buf.data_ptr() == sink.views[idx].data_ptr()
    assert sink.buffer_for(torch.zeros(3)) is None


def test_navit_state_dict_and_packing_contract():
    from oracle.params import NAVIT_CASES, make_navit_params, navit_param_shapes
    from vit_pytorch_amd.na_vit import NaViT, Segments, group_images_by_max_seq_len
    case = NAVIT_CASES["navit_two_packs"]
    m = NaViT(**case["cfg"])
    sd = m.state_dict()
    exp = navit_param_shapes(case["cfg"])
    assert list(sd.keys()) == list(exp.keys())
    assert all(tuple(sd[k].shape) == exp[k] for k in sd)
    m.load_state_dict(make_navit_params(case["cfg"], 0), strict=True)
    # greedy packing identical to na_vit.py:38-77
    imgs = [torch.zeros(3, 32, 32), torch.zeros(3, 64, 64), torch.zeros(3, 16, 48), torch.zeros(3, 64, 32)]   # 16, 64, 12, 32 tokens at p=8
    groups = group_images_by_max_seq_len(imgs, patch_size=8, max_seq_len=80)
    assert [len(g) for g in groups] == [2, 2]
    with pytest.raises(AssertionError, match="exceeds maximum sequence length"):
        group_images_by_max_seq_len(imgs, patch_size=8, max_seq_len=40)
    # block tables of the varlen kernels
    s = Segments([130, 5, 256], [130, 5, 256], torch.device("cpu"))
    assert s.tq == 391 and s.cu_q.tolist() == [0, 130, 135, 391]
    assert s.qblk_seg.tolist() == [0, 0, 1, 2, 2] and s.qblk_r0.tolist() == [0, 128, 0, 0, 128]
    with pytest.raises(VitkError, match="HIP"):
        m([torch.randn(3, 16, 16)])


def test_dropout_routing_decisions_are_host_side():
    """Which path active dropout takes is decided from shapes and module state on the host (no GPU needed): the fused
    engine when the 256-row GEMM kernel and the fixed-length attention kernel serve the block, op by op otherwise."""
    from vit_pytorch_amd import engine as E
    from vit_pytorch_amd.vit import Transformer
    blk = Transformer(768, 2, 12, 64, 3072, dropout=0.1).to(torch.bfloat16)
    blk.train()
    assert blk._dropout_p() == 0.1
    big = torch.empty(8, 197, 768, dtype=torch.bfloat16)            # M = 1576 rows: served
    small = torch.empty(2, 197, 768, dtype=torch.bfloat16)          # M = 394 rows: 128-row kernels, no fused dropout
    long_seq = torch.empty(4, 577, 768, dtype=torch.bfloat16)       # N > 480: chunked attention kernels, no fused dropout
    assert blk._fusable(big) and not blk._fusable(small) and not blk._fusable(long_seq)
    assert E.dropout_fusable(torch.bfloat16, 8, 197, 768, 12, 64, 3072) and E.dropout_fusable(torch.float16, 8, 197, 768, 12, 64, 3072)
    assert not E.dropout_fusable(torch.float32, 8, 197, 768, 12, 64, 3072)
    blk.layers[1][1].net[3].p = 0.2                                  # a user edited one Dropout: no common p -> op by op
    assert blk._dropout_p() is None and not blk._fusable(big)
    blk.layers[1][1].net[3].p = 0.1
    blk.eval()
    assert blk._dropout_p() == 0.0 and blk._fusable(small)           # inactive dropout: every shape is fusable
    # the host's copy of the device hash that derives per-site seeds (lowbias32)
    def ref(x):
        x &= 0xffffffff; x ^= x >> 16; x = (x * 0x21f0aaad) & 0xffffffff; x ^= x >> 15; x = (x * 0x735a2d97) & 0xffffffff; x ^= x >> 15
        return x
    assert all(E._hash32(v) == ref(v) for v in (0, 1, 12345, 0xffffffff, 0x9E3779B1 * 7))
    assert len({E._hash32(1000 + k) for k in range(48)}) == 48       # 12 layers x 4 sites: distinct seeds


def test_gpu_only_helpers_refuse_cpu_models():
    from vit_pytorch_amd import ViT
    from vit_pytorch_amd._lib import VitkError
    from vit_pytorch_amd.fp8 import enable_fp8_forward
    from vit_pytorch_amd.graphs import GraphedForward
    from vit_pytorch_amd.optim import Adam
    from vit_pytorch_amd.parallel import DataParallel
    m = ViT(image_size=32, patch_size=8, num_classes=10, dim=64, depth=1, heads=2, mlp_dim=128)
    import copy
    with pytest.raises(VitkError):
        enable_fp8_forward(copy.deepcopy(m).double())               # fp8 replaces 16-bit operands: bf16 / f16 models, or f32 ones run under autocast (round 6)
    with pytest.raises(VitkError):
        Adam(DataParallel(m))                                        # parameters on the CPU
    with pytest.raises(VitkError):
        GraphedForward(m, torch.zeros(1, 3, 32, 32))
    with pytest.raises(TypeError):
        Adam(m)


def test_dropout_seed_state_survives_deepcopy_and_old_pickles():
    """the per-instance dropout sequence state is plain instance state -- a deepcopy must not share its source's salt (an
    EMA / teacher copy would draw the same masks) and a module unpickled without the attributes must not raise."""
    import copy
    from vit_pytorch_amd.vit import Transformer
    t = Transformer(64, 2, 2, 32, 128, dropout=0.1)
    t2 = copy.deepcopy(t)
    assert t2._drop_state()[1] != t._drop_state()[1]
    assert [k for k, _ in t2.state_dict().items()] == [k for k, _ in t.state_dict().items()]
    del t.__dict__["_drop_salt"], t.__dict__["_drop_calls"]            # what an old pickle looks like
    calls, salt = t._drop_state()
    assert calls == 0 and salt not in (t2._drop_state()[1],)


def test_caller_grad_mode_is_thread_local_and_defaults_to_on():
    import threading
    from vit_pytorch_amd import _epoch as E
    E.note_grad_mode(False)
    seen = []
    th = threading.Thread(target=lambda: seen.append(E.caller_grad_mode()))
    th.start(); th.join()
    assert seen == [True] and E.caller_grad_mode() is False
    E.note_grad_mode(True)


def test_weight_cache_switches(monkeypatch):
    import torch
    from vit_pytorch_amd import _epoch as E, invalidate_weight_caches
    w = torch.nn.Parameter(torch.zeros(4, 4))
    k0 = E.weight_key(w)
    assert E.weight_key(w) == k0
    w.data.add_(1)                       # invisible to torch's version counter ...
    assert E.weight_key(w) == k0
    invalidate_weight_caches()           # ... hence the explicit call
    assert E.weight_key(w) != k0
    monkeypatch.setenv("VITK_WEIGHT_CACHE", "0")
    assert E.weight_key(w) != E.weight_key(w)


def test_weight_key_changes_across_every_torch_optimizer_step():
    """torch's fused optimizers (torch._fused_adamw_ & co.) update parameters without bumping `_version` [measured, torch 2.10]; the
    derived-weight caches must still see the step (the global optimizer-step post hook of _epoch.py)."""
    import torch
    from vit_pytorch_amd import _epoch as E
    for make in (lambda p: torch.optim.SGD([p], lr=0.1), lambda p: torch.optim.AdamW([p], lr=0.1, foreach=True),
                 lambda p: torch.optim.AdamW([p], lr=0.1, fused=True), lambda p: torch.optim.Adam([p], lr=0.1, fused=True)):
        w = torch.nn.Parameter(torch.ones(4, 4))
        opt = make(w)
        w.grad = torch.ones(4, 4)
        k0, v0 = E.weight_key(w), w.detach().clone()
        opt.step()
        assert not torch.equal(w.detach(), v0)
        assert E.weight_key(w) != k0, type(opt).__name__
