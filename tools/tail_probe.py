import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from u2tokenizer_amd import ops
import time
dev='cuda:0'; bf=torch.bfloat16
def rnd(*s, seed=0):
    g=torch.Generator().manual_seed(seed); return (torch.randn(*s, generator=g)).to(bf)
def timeit(f, iters=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters
scratch=torch.empty(48<<20,dtype=torch.uint8,device=dev)
for (M,N,K,kw) in [(8,2304,768,{}),(8,768,768,dict(bias=True,residual=True)),(8,768,3072,dict(bias=True,residual=True)),(8,3072,768,dict(bias=True,gelu=True)),
                   (2048,768,1024,dict(bias=True,residual=True)),(256,1792,512,{}),(256,512,1792,{})]:
    a,b=rnd(M,K,seed=1).to(dev),rnd(N,K,seed=2).to(dev)
    bias=rnd(N,seed=3).to(dev) if kw.get('bias') else None
    res=rnd(M,N,seed=4).to(dev) if kw.get('residual') else None
    line=f"{M}x{N}x{K}"
    for sc in (None, scratch):
        ops.set_gemm_scratch(sc)
        ms=timeit(lambda: ops.gemm(a,b,bias=bias,residual=res,gelu=bool(kw.get('gelu'))))
        line+=f" | {'split' if sc is not None else 'plain'} {ms*1e3:6.1f} us"
    print(line, flush=True)
ops.set_gemm_scratch(None)
