import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import starflashattention_amd as sfa
dev = torch.device("cuda:0")
B, H, Hkv, Sk, D = 256, 32, 4, 8192, 128
kc = torch.randn((B, 1, Sk, Hkv, D), device=dev, dtype=torch.float32).bfloat16()
vc = torch.randn((B, 1, Sk, Hkv, D), device=dev, dtype=torch.float32).bfloat16()
qkv = torch.randn((B, H + 2 * Hkv, D), device=dev).bfloat16()
o = torch.empty((B, H, D), dtype=torch.bfloat16, device=dev)
sl = torch.full((B,), Sk - 1, dtype=torch.int32, device=dev)
z = torch.zeros(0, dtype=torch.bfloat16, device=dev)
nbytes = 2.0 * B * Sk * Hkv * D * 2 + (H + 2 * Hkv + H) * B * D * 2
for rep in range(2):
    for S in (0, 1, 2, 3, 4, 8):
        run = lambda: sfa.flash_decode(qkv, z, z, z, kc, vc, sl, o, B, Sk, H, D, D, Sk, 1, 0, num_heads_kv=Hkv, num_splits=S)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"num_splits {S}: {ms:.4f} ms  {nbytes/ms/1e6:.0f} GB/s", flush=True)
