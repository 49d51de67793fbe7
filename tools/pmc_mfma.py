import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
dev='cuda'
M,N,K=1536,4096,25088
a=torch.randn(M,K,device=dev); b=torch.randn(N,K,device=dev); out=torch.empty(M,N,device=dev)
for _ in range(3): _hip.gemm(a,b,False,True,out=out)
x=torch.randn(6,296,296,128,device=dev); wt=_hip.conv3x3_pack_weight(torch.randn(128,128,3,3,device=dev)*0.05); bias=torch.randn(128,device=dev)
for _ in range(3): _hip.conv3x3_nhwc(x,wt,bias,1)
torch.cuda.synchronize()
