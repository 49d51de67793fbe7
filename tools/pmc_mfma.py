import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
dev='cuda'
M,N,K=4096,4096,4096
a=torch.randn(M,K,device=dev); b=torch.randn(N,K,device=dev); out=torch.empty(M,N,device=dev)
for _ in range(3): _hip.gemm(a,b,False,True,out=out)
x=torch.randn(6,148,148,256,device=dev); wt=torch.randn(9,256,256,device=dev)*0.05; bias=torch.randn(256,device=dev)
for _ in range(3): _hip.conv3x3_nhwc(x,wt,bias,1)
torch.cuda.synchronize()
