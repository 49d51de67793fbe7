import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
from lib import _hip
torch.manual_seed(0)
for (M, N, K) in [(4096, 25088, 25), (4096, 25088, 13), (4096, 25088, 40), (4096, 4096, 25), (512, 4096, 25), (4096, 25088, 24)]:
    for mode in ('randn', 'relu_big'):
        dy = torch.randn(K, M, device='cuda') * 1e-3
        x = torch.randn(K, N, device='cuda')
        if mode == 'relu_big':
            x = torch.relu(x) * 300.0
            dy = dy * (torch.rand_like(dy) > 0.5)
        ref = dy.double().t() @ x.double()
        out = _hip.gemm(dy, x, True, False)
        err = (out.double() - ref).abs()
        i = err.argmax().item()
        print(M, N, K, mode, 'max err %.3e  max|ref| %.3e at (%d,%d) got %.6e ref %.6e' % (
            err.max().item(), ref.abs().max().item(), i // N, i % N, out.flatten()[i].item(), ref.flatten()[i].item()))
