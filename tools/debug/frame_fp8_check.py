import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from audiossl_amd.engine import AtstEngine
from oracle import atst_oracle as O
for arch, depth in (("small", 12), ("small", 2), ("base", 2)):
    d = 768 if arch == "base" else 384
    S = 6
    W = O.recipe_weights(arch, depth=depth, frame=True, seed=101)
    eng = AtstEngine(arch, depth=depth, frame=True, fp8=True); eng.load_weights(W)
    ep = eng._pass("student", S, 1001, True, 0)
    mel = O.recipe_mel(S, 1001, seed=103); length = torch.tensor([1001, 1001, 702, 941, 523, 1001])
    valid = eng._valid(length, 0, ep.n_tok)
    rs = np.random.RandomState(107)
    mk = torch.from_numpy(np.stack([O.block_mask(250, 0.65, 5, rng=rs) for _ in range(S)]))
    rows, rowflag = eng._frame_rows(mk.bool(), valid, ep.RS, True)
    keep = torch.ones(depth, 2, S)
    out = ep.forward(mel.cuda(), valid, rowflag, eng.drop_path_scales(S, keep))
    y_h = out.float()[rows.long()].cpu()
    res = {}
    for name, ctxs in (("fp32", ()), ("bf16emu", (O.emulate_bf16(),)), ("fp8emu", (O.emulate_bf16(), O.emulate_fp8()))):
        import contextlib
        with contextlib.ExitStack() as st:
            for c in ctxs: st.enter_context(c)
            with torch.enable_grad():
                Wl = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and k.startswith("student.encoder.") else v) for k, v in W.items()}
                y = O.encoder_forward(Wl, "student.encoder.", mel, length, arch, depth=depth, use_cls=False, mask_index=mk, mask_input=True, keep=keep, drop_path_rate=0.1)
        res[name] = y.detach()
    r = lambda a, b: float((a - b).norm() / b.norm())
    print(f"{arch} depth {depth} frame student rows: HIP fp8 vs fp32 oracle {r(y_h, res['fp32']):.3e} | vs fp8-emulating oracle {r(y_h, res['fp8emu']):.3e} | fp8-emulating oracle vs fp32 {r(res['fp8emu'], res['fp32']):.3e}")
