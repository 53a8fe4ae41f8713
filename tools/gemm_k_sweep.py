import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
lib = L.lib(); stream = torch.cuda.current_stream().cuda_stream
def run(M,N,K,epi,odt,label):
    a=(torch.randn(M,K,device="cuda")*0.5).bfloat16(); w=(torch.randn(N,K,device="cuda")*K**-0.5).bfloat16(); b=torch.randn(N,device="cuda")
    out=torch.zeros(M,N,device="cuda",dtype=odt)
    for _ in range(2): ops.gemm(a,w,b,epilogue=epi,out=out)
    torch.cuda.synchronize(); lib.vtc_prof_begin()
    for _ in range(5): ops.gemm(a,w,b,epilogue=epi,out=out)
    n=len(L.PROF_CLASSES); ms,cnt,work=(C.c_double*n)(),(C.c_longlong*n)(),(C.c_double*n)()
    lib.vtc_prof_end(stream,ms,cnt,work); t=ms[0]/5
    print(f"{label:22s} M={M} N={N} K={K:5d} {t*1e3:8.1f} us {2.0*M*N*K/t/1e9:7.1f} TF/s",flush=True)
for K in (512,1024,2048,4096):
    run(118272,1536,K,L.EPI_STORE,torch.bfloat16,"store bf16")
for K in (512,2048):
    run(118272,1536,K,L.EPI_STORE,torch.float32,"store f32")
    run(118272,1536,K,L.EPI_GELU,torch.bfloat16,"gelu bf16")
    run(118272,1536,K,L.EPI_RESID,torch.float32,"resid f32")
