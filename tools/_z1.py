import sys, torch
sys.path.insert(0, '/root/repo')
from mantis_amd import hip_ops as K
for M,N,Kd in ((8192,8192,8192),):
    a=torch.zeros(M,Kd,device="cuda",dtype=torch.bfloat16); b=torch.zeros(N,Kd,device="cuda",dtype=torch.bfloat16); out=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
    for _ in range(3): K.gemm_nt(a,b,out=out,variant=12)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.gemm_nt(a,b,out=out,variant=12)
    e1.record(); torch.cuda.synchronize()
    print(f"zeros {2.0*M*N*Kd/(e0.elapsed_time(e1)/10*1e-3)/1e12:6.0f} TF")
