import os, sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
from ccsd_amd import loader, solver
from tests.helpers import load_ckpt_np, make_flags
from ccsd_amd.plan import rank2_dim
dev='cuda:0'
name="ccsd_community_small_CC"
meta, parts = load_ckpt_np(name); cfg=meta["config"]
N, Fd = cfg["data"]["max_node_num"], cfg["data"]["max_feat_num"]
d_min, d_max = cfg["data"]["d_min"], cfg["data"]["d_max"]
B=512
flags = make_flags(B, N, [20, 12, 16, 18, 14, 20]).to(dev)
names=["x","adj","rank2"]
kw = dict(shape_x=(B, N, Fd), shape_adj=(B, N, N), predictor="Euler", corrector="Langevin", snr=0.05, scale_eps=0.7, n_steps=1,
          probability_flow=False, continuous=True, denoise=True, eps=1e-4, is_cc=True, shape_rank2=(B, *rank2_dim(N, d_min, d_max)), d_min=d_min, d_max=d_max)
sd = [loader.load_sde(cfg["sde"][p]) for p in names]
ms = [loader.load_model_from_ckpt(meta[f"params_{p}"], parts[p], dev) for p in names]
outs={}
for mode in ("0","3","6"):
    if mode=="0": os.environ.pop("CCSD_SPLIT_BF16",None)
    else: os.environ["CCSD_SPLIT_BF16"]=mode
    res={}
    for steps in (1, 10, 50):
        fn = solver.get_pc_sampler(device=dev, rng="philox", seed=11, max_steps=steps, sde_x=sd[0], sde_adj=sd[1], sde_rank2=sd[2], **kw)
        r = fn(*ms, flags)
        res[steps]=[t.clone() for t in r[:3]]
    eng=fn.engine(); st=eng.alloc_state(B); eng.init_state(flags, st, None, 3, 0)
    res['score']=eng.score(2, st[0], st[1], st[2], flags).clone()
    outs[mode]=res
for mode in ("3","6"):
    for k in (1,10,50):
        for i,nm in enumerate(names):
            a=outs["0"][k][i]; b=outs[mode][k][i]
            print(mode, 'steps',k, nm, 'max rel', float((a-b).abs().max()/a.abs().max()))
    a=outs["0"]['score']; b=outs[mode]['score']; print(mode,'score rank2 max rel', float((a-b).abs().max()/a.abs().max()))
