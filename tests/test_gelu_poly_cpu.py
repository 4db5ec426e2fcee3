"""The GELU epilogue of the fc1 GEMM evaluates the Gaussian tail Q(t) = 0.5 erfc(t / sqrt 2) as exp2 of a degree-6 polynomial
(audiossl_amd/csrc/common.h: gelu_tail / gelu_f, fitted by tools/gelu_fit.py).  This test reads the SHIPPED coefficients out of the
source and checks, in fp32 arithmetic, that gelu(x) = max(x, 0) - t Q(t) stays within 1e-6 of nn.GELU (erf form, ref:
audiossl/modules/transformer.py:70-92) over the whole range -- far below the bf16 rounding of the stored activation."""
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shipped_coefficients():
    src = open(os.path.join(ROOT, "audiossl_amd", "csrc", "common.h")).read()
    body = src[src.index("DEVFN float gelu_tail(float t) {"):]
    body = body[:body.index("return")]
    lead = float(re.search(r"float r = ([-+0-9.e]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"fmaf\(r, t, ([-+0-9.e]+)f\)", body)]
    assert len(rest) == 6
    return [lead] + rest                       # highest degree first (Horner order)


def gelu_shipped(x):
    c = [np.float32(v) for v in shipped_coefficients()]
    x = x.astype(np.float32)
    t = np.minimum(np.abs(x), np.float32(5.5))
    r = np.full_like(t, c[0])
    for v in c[1:]:
        r = r * t + v
    return np.maximum(x, np.float32(0)) - t * np.exp2(r).astype(np.float32)


def test_gelu_polynomial_matches_erf_gelu():
    x = np.concatenate([np.linspace(-12, 12, 400001), np.linspace(-1e-3, 1e-3, 2001)])
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    got = gelu_shipped(x).astype(np.float64)
    err = np.abs(got - want)
    assert err.max() < 1e-6, err.max()
    # the tail is cut at t = 5.5: Q(5.5) = 1.9e-8, i.e. gelu(-5.5) = -1e-7 and gelu(-12) must not blow up
    assert abs(gelu_shipped(np.array([-12.0]))[0]) < 2e-7 and gelu_shipped(np.array([12.0]))[0] == np.float32(12.0)
