import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch, time
from util import O, amd_model
import nerf_amd as amd
DEV='cuda'
for prec in ('bf16','fp16'):
    m = amd_model(O.init_params(0), prec)
    for N,S in ((1024,64),(1024,192),(4096,192),(65536,192)):
        ro = torch.randn(N,3,device=DEV); rd = torch.nn.functional.normalize(torch.randn(N,3,device=DEV),dim=-1)
        z = torch.linspace(2,6,S,device=DEV).expand(N,S).contiguous()
        with torch.no_grad():
            for _ in range(3): m.forward_rays(ro,rd,z)
            torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            it=10
            for _ in range(it): m.forward_rays(ro,rd,z)
            e1.record(); torch.cuda.synchronize()
            ms=e0.elapsed_time(e1)/it
        fl=N*S*1186816
        print(f"{prec} N={N} S={S}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  {N*S/ms/1e3:.2f} Msamples/s")
