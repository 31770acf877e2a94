import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch, numpy as np
from util import O, amd_model
import nerf_amd as amd
from nerf_amd import ops
DEV='cuda'
P=1000
m = amd_model(O.init_params(0), 'bf16')
pts = torch.randn(P,3,device=DEV); vd = torch.nn.functional.normalize(torch.randn(P,3,device=DEV),dim=-1)
raw = m.forward_rays(pts, vd, torch.zeros(P,1,device=DEV))
saved = raw.grad_fn.saved_acts.cpu().numpy()
n_tiles = (P+255)//256; n_wt = n_tiles*8
units = saved[:n_wt*158*1024].reshape(-1,1024)
masks = saved[n_wt*158*1024:].view(np.uint32).reshape(n_wt, 9, 4, 64)   # wt, tensor, pair, lane
def mask_bit(par,r): return 8*par + 16*(r&1) + (r>>1)
bad=0
for t in range(8):
    ks0 = 4 + 16*t
    for wt in range(n_wt):
        H = units[ks0*n_wt + wt*16 : ks0*n_wt + wt*16+16].view(np.uint16).reshape(16, 32, 2, 8)  # ks, sample, h, j
        for mt in range(8):
            p, par = mt>>1, mt&1
            for lane in (0, 5, 37, 63):
                c, h = lane&31, lane>>5
                mw = int(masks[wt,t,p,lane])
                for r in range(16):
                    ks = 2*mt + (r>>3); j = r&7
                    pos = H[ks, c, h, j] != 0
                    bit = (mw >> mask_bit(par,r)) & 1
                    if pos != bit: bad+=1
print("mismatch", bad, "nonzero mask words", (masks!=0).mean())
t=0; wt=0; ks0=4
H = units[ks0*n_wt + wt*16 : ks0*n_wt + wt*16+16].view(np.uint16).reshape(16, 32, 2, 8)
for lane in (0,37):
    c,h = lane&31, lane>>5
    for p in range(2):
        mw = int(masks[wt,t,p,lane])
        print("lane",lane,"pair",p,"mask %08x"%mw)
        for mt in (2*p,2*p+1):
            print("  tile",mt,"signs", "".join("1" if H[2*mt+(r>>3),c,h,r&7]!=0 else "0" for r in range(16)))
