#!/bin/bash
# cfg5-shape bf16 runs (V = 3, 480x640 sources, 16 384 rays): fp32 feature maps against bf16 feature maps, table and direct gather
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY'
import time, numpy as np, torch
from thesis_clip_nerf_amd import ops
from thesis_clip_nerf_amd.synthetic import make_scene
DEV='cuda:0'
for (v,h,w,r) in ((3,480,640,16384),(1,64,64,4096)):
    sc = make_scene(seed=0, batch=1, n_views=v, height=h, width=w, n_rays=r if (h,w)!=(64,64) else None, with_features=False)
    d = {k: torch.from_numpy(np.ascontiguousarray(sc[k])).to(DEV) for k in ['rays_o','rays_d','images','intrinsics','extrinsics_inv','u_coarse','u_fine','coarse','fine']}
    g = torch.Generator(device=DEV).manual_seed(0)
    f32 = torch.randn((1,v,h,w,256), device=DEV, generator=g).mul_(0.5)
    f16 = f32.to(torch.bfloat16).contiguous()
    pc,pf = ops.pack_net(d['coarse']), ops.pack_net(d['fine'])
    pc16,pf16 = ops.pack_net_bf16(d['coarse']), ops.pack_net_bf16(d['fine'])
    n_rays = d['rays_o'].shape[1]
    for name, feats in (('fp32 maps', f32), ('bf16 maps', f16)):
        for tables in ('auto', None):
            tab = torch.empty((2,1,v,h,w,128), device=DEV) if tables == 'auto' else None
            fn = lambda: ops.render_fwd_bf16(d['rays_o'], d['rays_d'], d['images'], feats, d['intrinsics'], d['extrinsics_inv'], pc, pf, pc16, pf16, d['u_coarse'], d['u_fine'], sc['near'], sc['far'], texel_tables=tab)
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0=time.perf_counter()
            n=10
            for _ in range(n): fn()
            torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/n
            # the projection alone
            if tab is not None:
                for _ in range(2): ops.project_texels_bf16(feats, pc16, out=tab, packed16_b=pf16)
                torch.cuda.synchronize(); t1=time.perf_counter()
                for _ in range(n): ops.project_texels_bf16(feats, pc16, out=tab, packed16_b=pf16)
                torch.cuda.synchronize(); dp=(time.perf_counter()-t1)/n
                extra = '  projection %.3f ms = %.2f TB/s of feature-map reads' % (1e3*dp, feats.numel()*feats.element_size()/dp/1e12)
            else:
                extra = ''
            print('V=%d %dx%d R=%d  %s  %s: %.3f ms/step = %.2f M rays/s%s' % (v,h,w,n_rays,name,'texel table' if tab is not None else 'direct gather',1e3*dt,n_rays/dt/1e6,extra))
PY
