"""numpy model of the f16x3 product with ONE fp32 accumulator (lo halves unscaled: fp16 subnormals, which gfx950 MFMAs keep --
profiles/r04_mfma_denorm_probe.txt) against today's two accumulators (lo' = lo * 1024, folded as acc + cor / 1024) and float64.
CPU only; prints max abs errors for layer-like operand scales.  DESIGN.md, "What comes next"."""
import numpy as np
rng = np.random.default_rng(0)
def split_scaled(x):
    hi = x.astype(np.float16); lo = ((x - hi.astype(np.float32)) * 1024).astype(np.float16); return hi, lo
def split_plain(x):
    hi = x.astype(np.float16); lo = (x - hi.astype(np.float32)).astype(np.float16); return hi, lo   # subnormals kept
def acc32(terms, K):   # sum over k in fp32, 32 at a time exact (MFMA-like)
    out = np.zeros(terms.shape[:-1], np.float32)
    for k0 in range(0, K, 32):
        out = (out + terms[..., k0:k0+32].astype(np.float64).sum(-1)).astype(np.float32)
    return out
for K, ws, xs in ((1152, 0.02, 1.0), (384, 0.05, 1.0), (6912, 0.01, 0.5), (1152, 0.002, 0.05), (576, 0.05, 30.0)):
    M, N = 64, 64
    x = (rng.standard_normal((M, K)) * xs).astype(np.float32); x[x < 0] *= 0.25
    w = (rng.standard_normal((N, K)) * ws).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    xh, xl = split_scaled(x); wh, wl = split_scaled(w)
    f = lambda a: a.astype(np.float64)
    # current: acc = sum hi*hi ; cor = sum (hi*lo' + lo'*hi); out = acc + cor/1024
    acc = acc32(f(xh)[:, None, :] * f(wh)[None], K)
    cor = acc32(f(xh)[:, None, :] * f(wl)[None] + f(xl)[:, None, :] * f(wh)[None], K)
    cur = (acc.astype(np.float64) + cor.astype(np.float64) / 1024).astype(np.float32)
    xh2, xl2 = split_plain(x); wh2, wl2 = split_plain(w)
    one = acc32(f(xh2)[:, None, :] * f(wh2)[None] + f(xh2)[:, None, :] * f(wl2)[None] + f(xl2)[:, None, :] * f(wh2)[None], K)
    # variant: x lo from the scaled plane times 2^-10 in fp16 (what an in-register v_pk_mul would give)
    xl3 = (xl.astype(np.float32) / 1024).astype(np.float16)
    one3 = acc32(f(xh)[:, None, :] * f(wh2)[None] + f(xh)[:, None, :] * f(wl2)[None] + f(xl3)[:, None, :] * f(wh2)[None], K)
    sc = np.abs(ref).max()
    print(f"K={K:5d} w~{ws} x~{xs}: |out| max {sc:.3g}; max err  two-acc scaled {np.abs(cur-ref).max():.3g}  one-acc plain lo {np.abs(one-ref).max():.3g}  one-acc (plane lo' * 2^-10) {np.abs(one3-ref).max():.3g}  fp32-fma order-of {np.abs((x@w.T)-ref).max():.3g}")
