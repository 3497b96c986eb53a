import sys, importlib, torch
sys.path[:0] = ['/root/repo', '/root/repo/tests']
import pairs
pkg = importlib.import_module('atm-vfi_amd')
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (576, 960)
for v, cls in (('base', pkg.NetworkBase),):
    sd = pkg.synthetic_state_dict(v, seed=1)
    n1 = cls(); n1.load_state_dict(sd); n1.to(dev).eval()
    n2 = cls(); n2.load_state_dict(sd); n2.to(dev).eval()
    a0, a1 = pairs.smooth_pair(1, H, W, seed=51)
    b0, b1 = pairs.random_pair(1, H, W, seed=52)
    o1 = n1(a0.to(dev), a1.to(dev))
    o2 = n2(torch.cat([a0, b0]).to(dev), torch.cat([a1, b1]).to(dev))
    torch.cuda.synchronize()
    print('I_t diff', (o2['I_t'][0] - o1['I_t'][0]).abs().max().item())
    # compare workspace buffers by name: B=1 buffer [F or B, ...] vs B=2
    b1map = {k[0]: (k, t) for k, t in n1._bufs.items()}
    for k2, t2 in n2._bufs.items():
        name = k2[0]
        if name not in b1map: continue
        k1, t1 = b1map[name]
        if t1.dim() < 2: continue
        f1, f2 = t1.shape[0], t2.shape[0]
        if f2 == 2 * f1 and t1.shape[1:] == t2.shape[1:]:
            if f1 == 1:
                d = (t2[0] - t1[0]).abs()
            elif f1 == 2:   # frames stacked [im0 batch, im1 batch]
                d = torch.stack([(t2[0] - t1[0]).abs().max(), (t2[2] - t1[1]).abs().max()])
            else:
                continue
            d = torch.nan_to_num(d, nan=-1.0)
            print(f'{name:12s} {tuple(t1.shape)} maxdiff {d.max().item():.3e}')
