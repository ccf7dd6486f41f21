"""The tolerances of the three relaxed forms, as include/pipe_hip.h states them and DESIGN.md section 2 tabulates them --
one copy for the tests that sweep every form (tests/test_gpu_kernel_families.py).  Everything else in the library is
compared with the oracle by numpy.array_equal."""
import numpy as np


def fir_ulps(got, want64, taps, xmax=1.0):
    """overlap-save FIR: |gpu - (float)oracle| in float32 ulps measured at max(|oracle|, 2^-24 ||h||_1 max|x|)."""
    floor = 2.0 ** -24 * float(np.abs(taps).sum()) * xmax
    want32 = want64.astype(np.float32)
    mag = np.maximum(np.abs(want64), floor).astype(np.float32)
    return np.abs(got.astype(np.float64) - want32.astype(np.float64)) / np.spacing(mag).astype(np.float64)


def fir_f64_bound(taps, xmax=1.0):
    """overlap-save FIR on float64 buffers (PIPE_HIP_PARAM_RELAXED_F64): the float32 contract's floor is 2^-24 of the
    filter's full-scale output because float32 RESULTS cannot say more; the transform's own error is c N' 2^-53 of it
    (N' = 1024 points: log2 N' = 10 butterfly layers each way, measured c < 1).  Bound as tested: 64 * 2^-53 ||h||_1 max|x|."""
    return 64.0 * 2.0 ** -53 * float(np.abs(taps).sum()) * xmax


def kappa(q):
    """Largest entry of any power of the cascade's one-frame zero-input transition matrix (what include/pipe_hip.h
    scales the relaxed biquad forms' bound with)."""
    q = np.atleast_2d(q)
    n = 2 * len(q)
    m = np.zeros((n, n))
    for j in range(n):
        st, x = np.zeros(n), 0.0
        st[j] = 1.0
        for s, (b0, b1, b2, a1, a2) in enumerate(q):
            y = b0 * x + st[2 * s]
            st[2 * s] = -a1 * y + (b1 * x + st[2 * s + 1])
            st[2 * s + 1] = -a2 * y + b2 * x
            x = y
        m[:, j] = st
    p, worst = m.copy(), 0.0
    for _ in range(1 << 16):
        mx = np.abs(p).max()
        worst = max(worst, mx)
        if mx < 1e-3 * worst or worst > 1e6:
            break
        p = p @ m
    return worst


def biquad_ulp(q, want):
    """time-segmented biquad: one float32 ulp measured at max(|y|, 2^-19 kappa max|y of the Line|); want: [lines][frames][C]."""
    floor = (2.0 ** -19 * kappa(q) * np.abs(want).max(axis=(1, 2), keepdims=True)).astype(np.float32)
    return np.spacing(np.maximum(np.abs(want).astype(np.float32), floor)).astype(np.float64)   # (a float32 ulp whatever `want`'s type)


def chain_ulps(got, want64):
    """fused chain: float32 ulps measured at max(|oracle|, 2^-24 max|oracle of the Line|); one Line."""
    floor = 2.0 ** -24 * np.abs(want64).max()
    want32 = want64.astype(np.float32)
    mag = np.maximum(np.abs(want64), floor).astype(np.float32)
    return np.abs(got.astype(np.float64) - want32.astype(np.float64)) / np.spacing(mag).astype(np.float64)
