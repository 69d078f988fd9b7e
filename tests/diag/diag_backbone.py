import copy, sys, torch
sys.path.insert(0, '.')
from oracle import nets
from seghiero_amd.backbone import ResNetBackbone
DEV='cuda:0'
def relerr(a,b):
    a,b=a.detach().cpu().double(),b.detach().cpu().double()
    return float((a-b).norm()/b.norm().clamp_min(1e-30))
for depth,size in [(50,64),(50,128)]:
    torch.manual_seed(depth)
    ref = nets.ResNetBackbone(depth, pretrained=False).train()
    ref64 = copy.deepcopy(ref).double()
    mine = ResNetBackbone(depth, pretrained=False)
    mine.load_state_dict({k:v.clone() for k,v in ref.state_dict().items()})
    mine.to(DEV).train()
    x = torch.randn(2,3,size,size+32)
    outs_r = ref(x); gs=[torch.randn(o.shape) for o in outs_r]
    sum((o*g).sum() for o,g in zip(outs_r,gs)).backward()
    outs_64 = ref64(x.double()); sum((o*g.double()).sum() for o,g in zip(outs_64,gs)).backward()
    outs_m = mine(x.to(DEV)); sum((o*g.to(DEV)).sum() for o,g in zip(outs_m,gs)).backward()
    print("size",size,[ (round(relerr(a,t),8), round(relerr(b,t),8)) for a,b,t in zip(outs_m,outs_r,outs_64)])
    gm,g64=dict(mine.named_parameters()),dict(ref64.named_parameters())
    rows=[]
    for k,p in ref.named_parameters():
        rows.append((relerr(gm[k].grad,g64[k].grad)/max(relerr(p.grad,g64[k].grad),1e-9),k,relerr(gm[k].grad,g64[k].grad),relerr(p.grad,g64[k].grad)))
    rows.sort(reverse=True)
    for r in rows[:12]: print("%.1f %s %.3e %.3e"%r)
    import statistics
    print("median ratio", statistics.median(r[0] for r in rows))
