#!/usr/bin/env python3
"""Diagnosis of the short-crop ATST-Frame case: encoder features of the teacher pass row by row against the oracle, packed and padded layouts."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from audiossl_amd.engine import AtstEngine
from oracle import atst_oracle as O
width, B = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(width)
n_tok = (width - width % 4) // 4
W = O.recipe_weights("small", frame=True, seed=17)
mels = [O.recipe_mel(B, width, seed=41), O.recipe_mel(B, width, seed=42)]
ln = torch.from_numpy(rng.integers(width // 2, width + 1, size=B)); ln[0] = width
lens = [ln, ln]
mask = torch.from_numpy(rng.random((B, n_tok)) < 0.6)
masks = [mask, mask]
keep_t = [torch.from_numpy((rng.random((12, 2, 2 * B)) < 0.95).astype(np.float32))]
keep_s = [torch.from_numpy((rng.random((12, 2, 2 * B)) < 0.95).astype(np.float32))]
keep_t[0][0] = 1.0; keep_s[0][0] = 1.0
eng = AtstEngine("small", frame=True)
eng.load_weights(W)
loss, _, _ = eng.forward(mels, lens, masks, keep_t, keep_s)
ep_s = eng._student_groups[0][0]
print("RS", ep_s.RS, "NP", ep_s.NP, "M", ep_s.M, "pack env", os.environ.get("ATST_PACK"))
tf, t_out = eng._teacher_keep
with torch.no_grad():
    f_ref = O.encoder_forward(W, "teacher.encoder.", torch.cat(mels), torch.cat(lens), "small", use_cls=False, mask_index=torch.cat(masks), mask_input=False,
                              keep=keep_t[0], drop_path_rate=0.1)
    t_ref = O.frame_net_forward(W, "teacher.", mels, lens, masks, False, "small", False, keep_t, None, 0.1)
print("rows", tf.shape, f_ref.shape)
d = (tf.cpu() - f_ref).norm(dim=1) / f_ref.norm(dim=1)
print("encoder feature rows: rel err median %.3e max %.3e ; rows > 0.05: %d of %d" % (d.median(), d.max(), int((d > 0.05).sum()), d.numel()))
bad = (d > 0.05).nonzero().flatten()[:20]
print("bad rows", bad.tolist(), [round(float(x), 3) for x in d[bad]])
d2 = (t_out.cpu() - t_ref).norm(dim=1) / t_ref.norm(dim=1)
print("head out rows: rel err median %.3e max %.3e" % (d2.median(), d2.max()), " overall", float((t_out.cpu() - t_ref).norm() / t_ref.norm()))
print("loss", float(loss))
