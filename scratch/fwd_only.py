import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from util import O, amd_model
DEV='cuda'
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
N,S = 65536,192
m = amd_model(O.init_params(0), prec)
ro = torch.randn(N,3,device=DEV); rd = torch.nn.functional.normalize(torch.randn(N,3,device=DEV),dim=-1)
z = torch.linspace(2,6,S,device=DEV).expand(N,S).contiguous()
with torch.no_grad():
    for _ in range(4): m.forward_rays(ro,rd,z)
torch.cuda.synchronize()
