import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'neural-motifs_amd'))
import torch.nn.functional as F
from lib import _hip
from lib.resnet import ResNet101Trunk
from oracle import model as OM
torch.manual_seed(3)
net = ResNet101Trunk()
g = torch.Generator().manual_seed(5)
for n, p in net.named_parameters():
    p.requires_grad = False
    if n.endswith('bn3.weight') or 'downsample.1.weight' in n: p.data.fill_(0.5)
sd = {'f.' + k: v.detach().clone() for k, v in net.state_dict().items()}
net.cuda()
x = torch.randn(2, 3, 192, 256, generator=g)
for training in (True, False):
    net.train(training)
    sdc = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        y = net.conv1(_hip.nchw_to_nhwc(x.cuda()))
        r = F.conv2d(x, sdc['f.conv1.weight'], None, stride=2, padding=3)
        def cmp(tag, a, b):
            a = a.permute(0, 3, 1, 2).cpu().double(); b = b.double()
            print('%-22s max|ref| %8.3f  max err %.3e  rel-rms %.3e' % (tag, b.abs().max(), (a - b).abs().max(), ((a - b).pow(2).mean() / b.pow(2).mean()).sqrt()))
        cmp('conv1', y, r)
        mean, invstd = net.bn1.stats(y)
        y, _ = _hip.bn_pool_fwd(y, mean, invstd, net.bn1.weight, net.bn1.bias); y = torch.relu_(y)
        r = F.max_pool2d(F.relu(OM._bn(sdc, r, 'f.bn1.', training)), 3, 2, 1)
        cmp('stem', y, r)
        for lname, blocks, stride in OM.RESNET_LAYERS:
            layer = getattr(net, lname)
            for b in range(blocks):
                p = 'f.%s.%d.' % (lname, b); s_ = stride if b == 0 else 1
                y = layer[b](y)
                out = F.relu(OM._bn(sdc, F.conv2d(r, sdc[p + 'conv1.weight']), p + 'bn1.', training))
                out = F.relu(OM._bn(sdc, F.conv2d(out, sdc[p + 'conv2.weight'], None, stride=s_, padding=1), p + 'bn2.', training))
                out = OM._bn(sdc, F.conv2d(out, sdc[p + 'conv3.weight']), p + 'bn3.', training)
                if p + 'downsample.0.weight' in sdc:
                    r = OM._bn(sdc, F.conv2d(r, sdc[p + 'downsample.0.weight'], None, stride=s_), p + 'downsample.1.', training)
                r = F.relu(out + r)
                if b in (0, 1, 2, 3, 10, 22) : cmp('%s.%d (train=%s)' % (lname, b, training), y, r)
