import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audiossl_amd.engine import AtstEngine
B = 256
eng = AtstEngine("small", ncrops=2, drop_path_rate=0.0); eng.init_weights(seed=4)
g = torch.Generator().manual_seed(13)
mels = [torch.randn(B, 1, 64, 1001, generator=g).clamp_(-1, 1) for _ in range(2)]
lens = [torch.full((B,), 1001), torch.randint(400, 1002, (B,), generator=g)]
eng.forward(mels, lens)
res = []
sv = {k: h.saved for k, h in eng.heads.items()}
for it in range(2):
    for k, h in eng.heads.items(): h.saved = sv[k]
    eng.g32.zero_()
    dz = eng.heads["student.predictor"].backward(eng._ds).clone()
    df = eng.heads["student.projector"].backward(dz).clone()
    res.append((dz, df, eng.g32.clone()))
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
print("dz", rel(res[1][0], res[0][0]), "df", rel(res[1][1], res[0][1]), "head grads", rel(res[1][2], res[0][2]))
# encoder backward alone, same dout twice
ep, rows = eng._student_groups[0]
outs = []
for it in range(2):
    eng.g32.zero_(); ep.dout.zero_()
    from audiossl_amd import hip
    src = res[0][1][:rows.numel()].contiguous()
    hip.call("atst_scatter_rows_bf16", hip.ptr(src), hip.ptr(rows), rows.numel(), 384, hip.ptr(ep.dout), hip.stream())
    ep.backward()
    outs.append(eng.g32.clone())
print("encoder grads", rel(outs[1], outs[0]))
for name, (off, shape) in eng.layout.entries.items():
    n = 1
    for d in shape: n *= d
    r = rel(outs[1][off:off + n], outs[0][off:off + n])
    if r > 1e-5 and ("blocks.11" in name or "norm." in name or "blocks.0." in name): print("  ", name, r)
