"""GPU: the caching allocator must be able to recycle the backward's temporaries although the weight-gradient GEMMs read them on
a second stream (engine._Fork keeps them alive by references ordered with stream events, not by record_stream).  With
record_stream the reserved pool grew with every layer the host ran ahead of the GPU: 2x the allocated bytes at ViT-H/14."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reserved_memory_stays_close_to_allocated_memory():
    from vit_pytorch_amd import ViT
    torch.manual_seed(0)
    m = ViT(image_size=224, patch_size=16, num_classes=100, dim=768, depth=8, heads=12, mlp_dim=3072).to("cuda", dtype=torch.bfloat16)
    img = torch.randn(96, 3, 224, 224, device="cuda", dtype=torch.bfloat16)
    lab = torch.randint(0, 100, (96,), device="cuda")

    def step():
        for p in m.parameters():
            p.grad = None
        torch.nn.functional.cross_entropy(m(img).float(), lab).backward()

    step()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    for _ in range(4):
        step()                      # no synchronisation in between: the host runs ahead as it does in training
    torch.cuda.synchronize()
    alloc, reserved = torch.cuda.max_memory_allocated(), torch.cuda.max_memory_reserved()
    assert reserved <= 1.25 * alloc + (1 << 30), (alloc / 2 ** 30, reserved / 2 ** 30)
    assert torch.cuda.memory_stats().get("num_alloc_retries", 0) == 0


def test_no_grad_forward_drops_activations_and_data_writes_need_invalidation(monkeypatch):
    """(1) under torch.no_grad() the fused stack must free each layer's activations (needs_input_grad alone cannot tell);
    (2) a raw `.data` write is invisible to the K-blocked weight cache until invalidate_weight_caches()."""
    import vit_pytorch_amd
    from vit_pytorch_amd import ViT
    torch.manual_seed(0)
    m = ViT(image_size=224, patch_size=16, num_classes=10, dim=768, depth=6, heads=12, mlp_dim=3072).to("cuda", dtype=torch.bfloat16)
    x = torch.randn(32, 3, 224, 224, device="cuda").to(torch.bfloat16)

    def peak(fn):
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = fn()
        torch.cuda.synchronize()
        return torch.cuda.max_memory_allocated() - base, out

    p_grad, y1 = peak(lambda: m(x))
    del y1
    with torch.no_grad():
        p_nograd, y0 = peak(lambda: m(x))
    assert p_nograd < 0.45 * p_grad, (p_nograd, p_grad)            # six layers kept vs one layer's worth alive at a time
    w = m.transformer.layers[0][1].net[1].weight
    w.data.mul_(0.5)
    with torch.no_grad():
        stale = m(x)
    assert torch.equal(stale, y0)                                   # the cached K-blocked copy still holds the old values ...
    vit_pytorch_amd.invalidate_weight_caches()
    with torch.no_grad():
        fresh = m(x)
    assert not torch.equal(fresh, y0)                               # ... until the documented call
