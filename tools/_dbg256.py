import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import starflashattention_amd as sfa
from oracle import decode_ref, round_to
dev = torch.device("cuda:0")
B, H, D, L, M = 1, 1, 256, 1, 64
rng = np.random.default_rng(0)
for case in ("onehot0", "onehot200", "rand"):
    qkv = np.zeros((B, 3, H, D), np.float32)
    if case == "rand": qkv = round_to(rng.standard_normal((B, 3, H, D)), "fp16")
    elif case == "onehot0": qkv[:, 0, :, 0] = 4.0
    else: qkv[:, 0, :, 200] = 4.0
    kc = round_to(rng.standard_normal((B, L, M, H, D)), "fp16"); vc = round_to(rng.standard_normal((B, L, M, H, D)), "fp16")
    lens = [16]
    ref = decode_ref(qkv, kc.copy(), vc.copy(), lens, 0, 0, dtype="fp16")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).half().to(dev)
    o = torch.empty((B, H, D), dtype=torch.float16, device=dev)
    z = torch.zeros(0, dtype=torch.float16, device=dev)
    sfa.flash_decode(t(qkv), z, z, z, t(kc), t(vc), torch.tensor(lens, dtype=torch.int32, device=dev), o, B, M, H, D, 0, M, L, 0, num_splits=1)
    torch.cuda.synchronize()
    err = np.abs(o.float().cpu().numpy() - ref["o"])[0, 0]
    print(case, "max err per 32-dim block:", [float(err[i:i+32].max().round(4)) for i in range(0, D, 32)])
