import torch, sys
sys.path.insert(0, "/root/repo")
from audiossl_amd.engine import AtstEngine
B=int(sys.argv[1]) if len(sys.argv)>1 else 256
eng = AtstEngine("small", ncrops=2, drop_path_rate=0.0); eng.init_weights(seed=4)
g = torch.Generator().manual_seed(13)
mels = [torch.randn(B,1,64,1001,generator=g).clamp_(-1,1) for _ in range(2)]
lens = [torch.full((B,),1001), torch.randint(400,1002,(B,),generator=g)]
outs=[]
for it in range(3):
    loss,_,_ = eng.forward(mels,lens); s_out = eng.last_outputs[0].clone()
    eng.backward(); outs.append((float(loss), s_out, eng.g32.clone()))
for it in (1,2):
    print("loss", outs[0][0], outs[it][0], "s_out maxdiff", float((outs[it][1]-outs[0][1]).abs().max()), "grad rel", float((outs[it][2]-outs[0][2]).norm()/outs[0][2].norm()))
# per-tensor
for name,(off,shape) in eng.layout.entries.items():
    n=1
    for d in shape: n*=d
    a=outs[0][2][off:off+n]; b=outs[1][2][off:off+n]
    r=float((a-b).norm()/(a.norm()+1e-30))
    if r>1e-4: print(name, r)
