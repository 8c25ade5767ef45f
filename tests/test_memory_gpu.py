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
