import sys, torch, numpy as np, torch.nn as nn
sys.path.insert(0, '.')
from drn_amd import functional as DF, ops
DEV='cuda:0'
def rnd(*shape, seed=0): return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
def mk(Cin, Cout, k, seed):
    conv = nn.Conv1d(Cin, Cout, k, padding=(k-1)//2, bias=False).to(DEV); bn = nn.BatchNorm1d(Cout).to(DEV)
    with torch.no_grad(): conv.weight.copy_(rnd(Cout, Cin, k, seed=seed)/np.sqrt(Cin*k))
    return conv, bn
import time
for B in (4, 8, 16, 24, 28, 32):
    for chain in (True, False):
        dt = torch.bfloat16
        Ls, Cins, N = (256,128,64), (256,512,1024), 512
        xs = [rnd(B, L, Ci, seed=5+i).to(DEV, dt) for i,(L,Ci) in enumerate(zip(Ls,Cins))]
        blocks = [mk(Ci, N, 1, 30+i) for i,Ci in enumerate(Cins)]
        ops.kernel_timer = []
        torch.cuda.synchronize(); t0=time.time()
        outs = DF.multi_conv_block(xs, blocks, True, dt, chain_up=chain)
        torch.cuda.synchronize(); t1=time.time()
        tags=[t[0] for t in ops.kernel_timer]; ops.kernel_timer=None
        print(B, chain, 'wgs', B*(256+128+64)//128*4, '%.3f s'%(t1-t0), 'timeouts', ops.conv_bn_train_timeouts(), tags, flush=True)
