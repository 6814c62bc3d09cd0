"""Fused attention kernels (bf16): time per launch and effective rates at the benchmark shapes.
PASSL_ATTN_WAVES=4|8 forces the workgroup size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from passl_amd.hip import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

print('waves:', os.environ.get('PASSL_ATTN_WAVES', 'auto'))
for name, B, T, H, DH, causal in (('clip16 vision', 256, 197, 12, 64, False), ('clip text', 256, 77, 8, 64, True),
                                  ('mae encoder', 256, 50, 12, 64, False), ('mae decoder', 256, 197, 16, 32, False),
                                  ('clip32 vision', 128, 50, 12, 64, False)):
    qkv = torch.randn(B * T, 3 * H * DH, device='cuda').bfloat16()
    scale = DH ** -0.5
    out, lse = ops.attention_fwd(qkv, B, T, H, DH, scale, causal)
    dout = torch.randn_like(out)
    tf = timeit(lambda: ops.attention_fwd(qkv, B, T, H, DH, scale, causal))
    tb = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, T, H, DH, scale, causal))
    fl = 4.0 * B * H * T * T * DH * (0.5 if causal else 1.0)
    mb = qkv.numel() * 2 / 1e6
    print('%-14s B%d T%d H%d d%d  fwd %7.1f us (%6.1f TF, qkv %5.0f MB -> %5.2f TB/s)   bwd %7.1f us (%6.1f TF)' % (
        name, B, T, H, DH, tf, fl / tf / 1e6, mb, (mb + mb / 3) / tf, tb, 2.5 * fl / tb / 1e6))
