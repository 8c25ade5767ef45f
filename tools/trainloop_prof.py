import torch
from vit_pytorch_amd import ViT, invalidate_weight_caches
torch.manual_seed(0)
m = ViT(image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072).cuda().bfloat16()
x = torch.randn(256, 3, 224, 224, device="cuda").bfloat16(); y = torch.randint(0, 1000, (256,), device="cuda")
for i in range(8):
    invalidate_weight_caches()
    m.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(x).float(), y).backward()
torch.cuda.synchronize()
