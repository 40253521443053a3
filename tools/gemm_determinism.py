import sys; sys.path[:0]=['/root/repo/open-diffusiongs_amd']
import torch
from dgs_amd import _native
from dgs_amd.dit import DitOps
ops=DitOps(); DEV='cuda:0'
g=torch.Generator(device=DEV).manual_seed(0)
for (M,N,K,valid,rpb) in ((1024,1024,1024,258,512),(1024,1024,4096,258,512),(4352,1024,4096,4098,4352),(1024,4096,1024,258,512),(1024,3072,1024,258,512)):
    A=torch.randn(M,K,generator=g,device=DEV).to(torch.bfloat16); W=(torch.randn(N,K,generator=g,device=DEV)*0.05).to(torch.bfloat16)
    bias=torch.randn(N,generator=g,device=DEV)
    ref=A.float()@W.float().t()+bias
    live=(torch.arange(M,device=DEV)%rpb)<valid
    for algo in (0,1,4,6):
        outs=[]
        for it in range(4):
            o=torch.zeros(M,N,device=DEV)
            ops.gemm(A,W,bias,_native.EPI_F32,out=o,rows_per_batch=rpb,valid_rows=valid,algo=algo)
            outs.append(o[live].clone())
        err=float((outs[0]-ref[live]).norm()/ref[live].norm())
        same=all(torch.equal(outs[0],x) for x in outs[1:])
        print(M,N,K,valid,'algo',algo,'relerr %.2e'%err,'deterministic',same)
