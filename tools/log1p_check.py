"""sg_log1p_pos / sg_div (alaz_amd/csrc/sg_kernels.h) restated in numpy fp64: the error against long-double log1p and the number of
fp32 results that differ from (float)log1p(x).  The reciprocal seed is rounded to 24 bits: no better than v_rcp_f64."""
import numpy as np
LN2_HI = 6.93147180369123816490e-01; LN2_LO = 1.90821492927058770002e-10
def frcp(d):
    r = (1.0/d).astype(np.float32).astype(np.float64)   # a ~24-bit reciprocal, like v_rcp_f64 at worst
    e = 1.0 - d*r; r = r + r*e
    e = 1.0 - d*r; r = r + r*e
    return r
def fdiv(n, d):
    r = frcp(d); q = n*r
    return q + (n - d*q)*r
def l1p(x):
    y = 1.0 + x
    m, e = np.frexp(y)                      # m in [0.5, 1)
    lo = m < 0.70710678118654752
    m = np.where(lo, m*2.0, m); e = np.where(lo, e-1, e).astype(np.float64)
    s = fdiv(m - 1.0, m + 1.0); s2 = s*s
    p = 1.0/19
    for k in (17, 15, 13, 11, 9, 7, 5, 3): p = p*s2 + 1.0/k
    logm = 2.0*s + 2.0*s*s2*p
    c = (x - (y - 1.0))*frcp(y)
    r = e*LN2_HI + (logm + (e*LN2_LO + c))
    small = x < 1e-4
    return np.where(small, x*(1.0 - x*(0.5 - x*(1.0/3.0 - 0.25*x))), r)
rng = np.random.default_rng(1)
x = np.concatenate([10.0**rng.uniform(-12, 14, 4_000_000), rng.uniform(0, 10, 2_000_000), np.arange(0, 5000, dtype=np.float64), [0.0, 1e-4, 9.99999e-5, 0.41421356, 0.4142136]])
ref = np.log1p(x.astype(np.longdouble)).astype(np.float64)
got = l1p(x)
rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
print("max rel err", rel[ref > 0].max(), "at x =", x[ref > 0][rel[ref > 0].argmax()])
f1 = got.astype(np.float32); f2 = np.log1p(x).astype(np.float32)
print("fp32 mismatches", int((f1 != f2).sum()), "of", len(x))
