import sys, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mantis_amd import hip_ops as K
def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, Kk in [("qwen vit qkv", 25024, 3840, 1280), ("qwen vit proj", 25024, 1280, 1280), ("qwen vit fc1", 25024, 5120, 1280),
                       ("qwen vit fc2", 25024, 1280, 5120), ("siglip qkv", 4608, 3456, 1152), ("siglip out", 4608, 1152, 1152), ("siglip fc1", 4608, 4304, 1152),
                       ("siglip fc2", 4608, 1152, 4304), ("projector 1", 4608, 4096, 1152), ("projector 2", 4608, 4096, 4096), ("lm_head fwd qwen", 1504, 152064, 3584), ("lm_head fwd llama", 512, 128256, 4096)]:
    a = torch.randn(M, Kk, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(N, Kk, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    line = f"{name:18s} {M:6d}x{N:6d}x{Kk:5d}:"
    for v in (0, 12, 13, 14, 2, 1):
        try:
            t = timeit(lambda: K.gemm_nt(a, b, bias=bias, variant=v))
            line += f"  v{v}: {1e3*t:7.1f} us {2.0*M*N*Kk/t/1e9:5.0f} TF"
        except Exception as e:
            line += f"  v{v}: n/a"
    print(line, flush=True)
