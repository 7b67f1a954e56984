import sys, os, torch
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'latent-nerf-test_amd'))
from src.latent_nerf.models import encoding as E
dev=torch.device('cuda:0')
enc = E.GridEncoder(scatter_variant=3).to(dev)
levels = enc.levels
for M in (150000, 6000, 777, 64, 200000):
    g = torch.Generator().manual_seed(4)
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 0.999).to(dev)
    dfeat = torch.randn(16, M, 2, generator=g).to(dev)
    m_dev = torch.tensor([M], dtype=torch.int32, device=dev)
    ref0 = torch.zeros(levels.n_rows, 2, device=dev)
    E.grid_encode_backward(x, 1.0, dfeat, levels, M, m_dev, M, ref0, variant=0)
    ws = E.scatter_workspace(levels, M, dev)
    for v in (2,3):
        outs=[]
        for poison in (0x7F, 0x00, 0x7F):
            ws.fill_(poison)
            d = torch.zeros(levels.n_rows, 2, device=dev)
            E.grid_encode_backward(x, 1.0, dfeat, levels, M, m_dev, M, d, variant=v)
            outs.append(d)
        per=[float((outs[0]-ref0)[levels.offsets[l]:levels.offsets[l+1]].abs().max()) for l in range(16)]
        print('M',M,'v',v,'eq01',bool(torch.equal(outs[0],outs[1])),'eq02',bool(torch.equal(outs[0],outs[2])),'refmax',float(ref0.abs().max()))
        print('   per-level err vs atomics', ['%.1e'%e for e in per])
