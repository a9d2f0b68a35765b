import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from summertts_amd import engine
rng = np.random.default_rng(1)
for (ci, co, k, dil, L) in ((128,128,11,5,4000),(256,256,7,3,2000),(128,128,3,1,4000),(64,64,11,1,6000),(512,512,3,1,700)):
    x = (rng.standard_normal((ci, L)) * rng.uniform(0.05, 3.0, (ci, 1))).astype(np.float32)
    w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    pad = dil*(k-1)//2
    ref = F.conv1d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double().permute(0,2,1).contiguous(), torch.from_numpy(b).double(), padding=pad, dilation=dil)[0].numpy()
    r = {}
    for name, mode in (("f32", 6), ("bf16x3", 13), ("f16x2", 50)):
        y, ms = engine.debug_conv1d(x, w, b, pad, dil, 0, False, mode=mode, iters=20)
        e = y - ref
        r[name] = (np.sqrt((e*e).mean()), np.abs(e).max(), ms)
    print((ci,co,k,dil,L), " ".join("%s rmse %.3e max %.2e %.1f us |" % (n, *v[:2], v[2]*1e3) for n, v in r.items()), flush=True)
    for sx, sw in ((2.0**-20, 1.0), (1.0, 2.0**-20), (2.0**10, 1.0), (2.0**-12, 2.0**9)):
        zero = np.zeros_like(b)
        base = F.conv1d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double().permute(0,2,1).contiguous(), None, padding=pad, dilation=dil)[0].numpy()
        for name, mode in (("bf16x3", 13), ("f16x2", 50)):
            y = engine.debug_conv1d((x*np.float32(sx)).astype(np.float32), (w*np.float32(sw)).astype(np.float32), zero, pad, dil, 0, False, mode=mode)
            rel = np.sqrt(np.mean((y/(sx*sw) - base)**2)) / np.sqrt(np.mean(base**2))
            print("   scale x %.1e w %.1e %s rel rmse %.3e" % (sx, sw, name, rel))
