import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from util import O, T, amd_model, maxabs
import nerf_amd as amd
g = dict(np.load('tests/golden/f8_e2e_render.npz'))
DEV='cuda'
ro, rd = T(g["rays_o"], DEV), T(g["rays_d"], DEV)
for prec in ('bf16','fp16'):
    mc, mf = amd_model(O.init_params(0), prec), amd_model(O.init_params(1), prec)
    pe, pv = amd.PositionalEncoder(3, 10).to(DEV), amd.PositionalEncoder(3, 4).to(DEV)
    with torch.no_grad():
        c, f = amd.render_hierarchical(mc, mf, pe, pv, ro, rd, 2.0, 6.0, 64, 128, rand=False, det=True)
        print(prec, 'coarse weights diff', maxabs(c['weights'], T(g['c_weights_l0w0'])), 'raw_sigma', maxabs(c['raw_sigma'], T(g['c_raw_sigma_l0w0'])))
        zc = T(g['z_comb_l0w0'])
        print(' z_comb diff max', maxabs(f['z_vals'], zc), 'mean', (f['z_vals'].cpu()-zc).abs().mean().item())
        f2 = amd.render_rays(mf, pe, pv, ro, rd, 2.0, 6.0, 192, z_vals_override=zc.to(DEV))
        print(' fine on golden z: raw_sigma', maxabs(f2['raw_sigma'], T(g['f_raw_sigma_l0w0'])))
        ok = torch.from_numpy(np.abs(g['f_sigma_last_l0w0'])>3e-3)
        for k in ('rgb_map','depth_map','acc_map'):
            print('  ', k, maxabs(f2[k].cpu()[ok], T(g[f'f_{k}_l0w0'])[ok]), 'own z:', maxabs(f[k].cpu()[ok], T(g[f'f_{k}_l0w0'])[ok]))
        # oracle on my z
        of = O.render_rays(O.init_params(1), ro.cpu(), rd.cpu(), 2.0, 6.0, 192, z_vals_override=f['z_vals'].cpu())
        for k in ('rgb_map','depth_map','acc_map'):
            print('   oracle(fp32) on my z vs golden', k, maxabs(of[k][ok], T(g[f'f_{k}_l0w0'])[ok]), ' mine vs oracle-on-my-z', maxabs(f[k].cpu()[ok], of[k][ok]))
