"""The GELU epilogues of the fc1 / fc2-dgrad GEMMs (audiossl_amd/csrc/common.h, fitted by tools/gelu_fit.py).  These tests read
the SHIPPED coefficients out of the source and check them in fp32 arithmetic against nn.GELU (erf form, ref:
audiossl/modules/transformer.py:70-92) and its derivative:
  * gelu_f (round-3 reference form, degree-6 tail clamped at 5.5): |err| < 1e-6 everywhere;
  * gelu_bf16dst / gelu_grad_bf16dst (what the epilogues compute; results are rounded to bf16 = 2^-9 on the spot): relative error
    below 2^-11 wherever the value is not negligible, tiny absolute error elsewhere, correct limits, NaN propagates."""
import os
import re

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "audiossl_amd", "csrc", "common.h")).read()


def coefficients(fn, n_fma):
    body = SRC[SRC.index(fn):]
    body = body[:body.index("return")]
    lead = float(re.search(r"float r = ([-+0-9.e]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"fmaf\(r, t, ([-+0-9.e]+)f\)", body)]
    assert len(rest) == n_fma, (fn, len(rest))
    return [np.float32(v) for v in [lead] + rest]          # highest degree first (Horner order)


def horner(c, t):
    r = np.full_like(t, c[0])
    for v in c[1:]:
        r = (r * t + v).astype(np.float32)
    return r


def gelu_shipped(x):
    c = coefficients("DEVFN float gelu_tail(float t) {", 6)
    x = x.astype(np.float32)
    t = np.minimum(np.abs(x), np.float32(5.5))
    return np.maximum(x, np.float32(0)) - t * np.exp2(horner(c, t)).astype(np.float32)


def gelu_fast(x):
    c = coefficients("DEVFN float gelu_tail_bf16dst(float t) {", 5)
    x = x.astype(np.float32)
    t = np.abs(x)
    with np.errstate(invalid="ignore", over="ignore"):
        return np.maximum(x, np.float32(0)) - t * np.exp2(horner(c, t)).astype(np.float32)


def gelu_grad_fast(x):
    body = SRC[SRC.index("DEVFN float gelu_grad_bf16dst(float x) {"):]
    body = body[body.index("#else"):]
    c = coefficients("DEVFN float gelu_grad_bf16dst(float x) {", 6) if False else None
    lead = float(re.search(r"float r = ([-+0-9.e]+)f;", body).group(1))
    rest = [float(v) for v in re.findall(r"fmaf\(r, t, ([-+0-9.e]+)f\)", body[:body.index("const float w")])]
    assert len(rest) == 6
    c = [np.float32(v) for v in [lead] + rest]
    k2, k0 = (np.float32(float(v)) for v in re.search(r"fmaf\(x \* x, ([-+0-9.e]+)f, ([-+0-9.e]+)f\)", body).groups())
    x = x.astype(np.float32)
    t = np.abs(x)
    e = np.exp2((x * x * k2 + k0).astype(np.float32)).astype(np.float32)
    w = (e * horner(c, t) + np.float32(0.5)).astype(np.float32)
    return np.float32(0.5) + np.copysign(w, x)


def test_gelu_polynomial_matches_erf_gelu():
    x = np.concatenate([np.linspace(-12, 12, 400001), np.linspace(-1e-3, 1e-3, 2001)])
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    got = gelu_shipped(x).astype(np.float64)
    err = np.abs(got - want)
    assert err.max() < 1e-6, err.max()
    # the tail is cut at t = 5.5: Q(5.5) = 1.9e-8, i.e. gelu(-5.5) = -1e-7 and gelu(-12) must not blow up
    assert abs(gelu_shipped(np.array([-12.0]))[0]) < 2e-7 and gelu_shipped(np.array([12.0]))[0] == np.float32(12.0)


def test_bf16_destination_gelu_is_within_a_quarter_of_a_bf16_ulp():
    x = np.concatenate([np.linspace(-40, 40, 800001), np.linspace(-1e-3, 1e-3, 2001)]).astype(np.float32).astype(np.float64)
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    got = gelu_fast(x).astype(np.float64)
    err = np.abs(got - want)
    inner = np.abs(x) <= 4
    assert (err[inner] / np.maximum(np.abs(want[inner]), 1e-30)).max() < 2.0 ** -11      # 1/4 of the bf16 half-ulp 2^-9
    assert err[~inner].max() < 1e-6
    assert gelu_fast(np.array([100.0]))[0] == np.float32(100.0) and gelu_fast(np.array([-100.0]))[0] == 0.0   # no clamp needed: exp2 underflows
    assert np.isnan(gelu_fast(np.array([np.nan]))[0])                                   # NaN in, NaN out (round-3 ADVICE: was -1e-7)
    # after the bf16 rounding the epilogue applies, the fast and the 3e-7 forms agree on all but a sliver of inputs, and never by more than one ulp
    xs = np.clip(np.random.default_rng(0).normal(0, 1.5, 1 << 20), -4.0, 4.0)     # beyond: |gelu| < 1.3e-4 on the negative side, ulps are meaningless
    a = torch.from_numpy(gelu_fast(xs)).bfloat16(); b = torch.from_numpy(gelu_shipped(xs)).bfloat16()
    diff = (a.view(torch.int16).int() - b.view(torch.int16).int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.08


def test_bf16_destination_gelu_derivative():
    x = np.linspace(-40, 40, 800001).astype(np.float32).astype(np.float64)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    torch.nn.functional.gelu(xt).sum().backward()
    want = xt.grad.numpy()
    got = gelu_grad_fast(x).astype(np.float64)
    err = np.abs(got - want)
    assert err.max() < 2e-5
    big = np.abs(want) > 0.05
    assert (err[big] / np.abs(want[big])).max() < 2.0 ** -11
    assert gelu_grad_fast(np.array([50.0]))[0] == 1.0 and gelu_grad_fast(np.array([-50.0]))[0] == 0.0
    assert np.isnan(gelu_grad_fast(np.array([np.nan]))[0])
