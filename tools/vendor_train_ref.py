"""Context only (NOT used by the product, not the reference): the stock PyTorch-ROCm stack on the same GPU and workload - HuggingFace
ViT-B/16 (dropout 0.1, SDPA attention), torch.autocast(bf16), fused torch.optim.AdamW, batch 512, synthetic data - images/s of its
train step (no augmentation stage, inputs already normalised float tensors)."""
import sys
import time

import torch
from transformers import ViTConfig, ViTForImageClassification

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = ViTConfig(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                attention_probs_dropout_prob=0.1, layer_norm_eps=1e-6, image_size=224, patch_size=16, num_labels=1000, attn_implementation="sdpa")
model = ViTForImageClassification(cfg).cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05, fused=True)
x = torch.randn(B, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (B,), device="cuda")


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(model(pixel_values=x).logits.float(), y)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 8
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("stock PyTorch-ROCm stack: ViT-B/16 train step, batch %d, autocast bf16, fused AdamW: %.1f ms/step = %.0f images/s (loss %.3f)"
      % (B, 1e3 * dt, B / dt, float(loss)), flush=True)
