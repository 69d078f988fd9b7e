import copy, sys, torch
sys.path.insert(0, '.')
from oracle.step import OracleTrainer
from seghiero_amd.synthetic import make_batch
from seghiero_amd.train_step import SegHieroTrainer
DEV='cuda:0'
def relerr(a,b):
    a,b=a.detach().cpu().double(),b.detach().cpu().double()
    return float((a-b).norm()/b.norm().clamp_min(1e-30))
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw = dict(depth=18, n_fine=4, coarse_to_fine_map=[[0,1],[2,3]], lr=0.01)
ref = OracleTrainer(**kw); mine = SegHieroTrainer(device=DEV, **kw)
mine.load_state_dicts(ref.state_dicts())
ref64 = OracleTrainer(**kw)
for k,m in ref64.modules().items():
    m.load_state_dict(ref.modules()[k].state_dict()); m.double()
ref64.params = [p for m in ref64.modules().values() for p in m.parameters()]
ref.train(); mine.train(); ref64.train()
img, lab = make_batch(B, 128, 4, seed=0)
l,lm_,la_,_ = ref.forward_loss(img, lab, 0); l.backward()
l64,_,_,_ = ref64.forward_loss(img.double(), lab, 0); l64.backward()
lo,lmm,lam,_ = mine.forward_loss(img.to(DEV), lab.to(DEV), 0); lo.backward()
print("loss", float(l), float(l64), float(lo), "main", float(lm_), float(lmm), "aux", float(la_), float(lam))
rows=[]
for name in ref.modules():
    g32=dict(ref.modules()[name].named_parameters()); g64=dict(ref64.modules()[name].named_parameters()); gm=dict(mine.modules()[name].named_parameters())
    for k in g32:
        if gm[k].grad is None: print("NO GRAD", name, k); continue
        rows.append((relerr(gm[k].grad,g64[k].grad), relerr(g32[k].grad,g64[k].grad), name+"."+k, float(g64[k].grad.norm())))
rows.sort(key=lambda r:-r[0]/max(r[1],1e-9))
for r in rows[:25]: print("%.3e %.3e %s |g|=%.3e"%r)
