"""Fits behind the GELU epilogues (audiossl_amd/csrc/common.h).  Build container, CPU only.

    python tools/gelu_fit.py          # round-3 forms: log2 of the Gaussian tail Q(t), degree 6..9, clamp at T (3e-7 class)
    python tools/gelu_fit.py --fast   # round-4 bf16-destination forms: degree-5 tail without clamp ; Mills-ratio form of gelu'

Reference function: erf-GELU (nn.GELU default), audiossl/modules/transformer.py:70-92."""
import sys
import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C, polynomial as P


def Q(t):
    return 0.5 * erfc(t / np.sqrt(2.0))


def minimax(tt, y, w, deg, iters=200):
    """weighted least squares, iteratively re-weighted towards the minimax of w * (p - y); monomial coefficients, low order first"""
    ww = np.ones_like(tt)
    for _ in range(iters):
        A = np.vander(tt, deg + 1, increasing=True) * (w * ww)[:, None]
        c = np.linalg.lstsq(A, y * w * ww, rcond=None)[0]
        e = np.abs((np.polyval(c[::-1], tt) - y) * w)
        ww = ww * (1 + 2 * e / e.max()) ** 0.5
    return c


def horner32(c_low_first, t):
    c = [np.float32(v) for v in c_low_first]
    r = np.full_like(t, c[-1])
    for v in c[-2::-1]:
        r = (r * t + v).astype(np.float32)
    return r


def fast():
    x = np.linspace(-10, 10, 400001)
    Phi = 0.5 * (1 + erf(x / np.sqrt(2)))
    phi = np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)
    a_true, g_true = x * Phi, Phi + x * phi
    tt = np.linspace(0, 7.5, 15001)
    # forward: exp2(P5(t)) ~ Q(t).  On the negative side gelu = -t Q(t): its RELATIVE error is ln2 * dP; weight fades where t Q(t) < 2e-5
    w = np.maximum(np.minimum(1.0, (tt * Q(tt)) / 2e-5), 1e-3)
    c = minimax(tt, np.log2(Q(tt)), w, 5)
    assert c[-1] < 0, "the leading coefficient must be negative: no clamp on t"
    x32 = x.astype(np.float32); t = np.abs(x32)
    h = np.exp2(horner32(c, t).astype(np.float64)).astype(np.float32)
    a = (np.maximum(x32, np.float32(0)) - t * h).astype(np.float64)
    err = np.abs(a - a_true)
    rel = err / np.maximum(np.abs(a_true), 1e-30)
    print("gelu_tail_bf16dst (Horner order, highest first):", ", ".join("%.9ef" % np.float32(v) for v in c[::-1]))
    print("  max |err| %.2e ; max rel err for |x| <= 4: %.2e (2^-11 = %.2e) ; max |err| for |x| > 4: %.2e" %
          (err.max(), rel[np.abs(x) <= 4].max(), 2.0 ** -11, err[np.abs(x) > 4].max()))
    # backward: gelu'(x) = 1/2 + copysign(1/2 + phi(t) G(t), x), G(t) = t - Q(t) / phi(t); fit under the weight phi(t)
    ph = np.exp(-0.5 * tt * tt) / np.sqrt(2 * np.pi)
    c6 = minimax(tt, tt - Q(tt) / ph, ph, 6)
    e = np.exp2((t * t * np.float32(-0.7213475204) + np.float32(-1.3257480647)).astype(np.float64)).astype(np.float32)
    wv = (e * horner32(c6, t) + np.float32(0.5)).astype(np.float32)
    g = (np.float32(0.5) + np.copysign(wv, x32)).astype(np.float64)
    gerr = np.abs(g - g_true)
    print("gelu_grad_bf16dst (Horner order):", ", ".join("%.9ef" % np.float32(v) for v in c6[::-1]))
    print("  max |err| %.2e ; max rel err where |gelu'| > 0.05: %.2e" % (gerr.max(), (gerr / np.abs(g_true))[np.abs(g_true) > 0.05].max()))


def round3():
    def target(t): return np.log2(Q(t))
    def fit(deg, T, iters=40):
        n = 4000
        t = 0.5 * T * (1 - np.cos(np.pi * (np.arange(n) + 0.5) / n))
        y = target(t)
        base_w = 2.0 ** y * np.maximum(t, 0.5)
        w = base_w.copy()
        V = C.chebvander(2 * t / T - 1, deg)
        for _ in range(iters):
            c = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)[0]
            err = np.abs((V @ c - y) * base_w)
            w = w * (1 + 2.0 * err / err.max()) ** 0.5
        return P.Polynomial(C.cheb2poly(c))(P.Polynomial([-1.0, 2.0 / T])).coef
    x = np.linspace(-9, 9, 2000001)
    phi_true = 0.5 * (1 + erf(x / np.sqrt(2)))
    a_true = x * phi_true
    for deg in (6, 7, 8, 9):
        for T in (5.0, 5.5, 6.0):
            coef = fit(deg, T)
            x32 = x.astype(np.float32); t = np.minimum(np.abs(x32), np.float32(T))
            h = np.exp2(horner32(coef, t)).astype(np.float32)
            a = np.maximum(x32, np.float32(0)) - t * h
            print(f"deg {deg} T {T}: max|gelu err| {np.abs(a - a_true).max():.2e}")
    print("---- coefficients deg 6, T 5.5 (low order first)")
    print(", ".join(f"{np.float32(c):.9e}f" for c in fit(6, 5.5)))


if __name__ == "__main__":
    fast() if "--fast" in sys.argv else round3()
