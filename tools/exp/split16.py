import numpy as np
rng = np.random.default_rng(0)
def split_f16(x, S):
    xs = (x.astype(np.float32) * np.float32(S))
    h = xs.astype(np.float16)
    l = (xs - h.astype(np.float32)).astype(np.float16)
    return h.astype(np.float64), l.astype(np.float64)
def split_bf16(x):
    def trunc(v):
        u = v.astype(np.float32).view(np.uint32) & np.uint32(0xffff0000)
        return u.view(np.float32)
    h = trunc(x); r1 = x - h; m = trunc(r1); l = r1 - m
    return h.astype(np.float64), m.astype(np.float64), l.astype(np.float64)
def mfma_acc(terms, kblk=16):
    # terms: list of (A[M,K], B[N,K]) products accumulated per k-block exactly, then added (fp32 rounding) to acc
    M, K = terms[0][0].shape; N = terms[0][1].shape[0]
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, kblk):
        for (a, b) in terms:
            blk = a[:, k0:k0+kblk] @ b[:, k0:k0+kblk].T
            acc = (acc.astype(np.float64) + blk).astype(np.float32)
    return acc.astype(np.float64)
def chain32(A, B):
    M, K = A.shape; N = B.shape[0]
    acc = np.zeros((M, N), np.float32)
    for k in range(K):
        p = A[:, k:k+1].astype(np.float64) * B[:, k][None, :].astype(np.float64)
        acc = (acc.astype(np.float64) + p).astype(np.float32)  # fma: single rounding
    return acc.astype(np.float64)
def pow2_scale(bound):
    return 2.0 ** np.floor(15 - np.log2(bound))
for K in (128, 512, 1024):
  for kind in ("act", "grad"):
    M, N = 96, 96
    if kind == "act":
        A = np.maximum(rng.standard_normal((M, K)) * 1.0 + 0.2, 0).astype(np.float32)
        boundA = np.sqrt(41472.0) * 1.0 + 0.2
    else:
        A = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K)) * 2) * 1e-5).astype(np.float32)
        boundA = np.abs(A).max() * 37.0   # loose bound
    B = (rng.standard_normal((N, K)) * np.sqrt(2.0 / N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    c = chain32(A, B)
    SA, SB = pow2_scale(boundA), pow2_scale(np.abs(B).max())
    ha, la = split_f16(A, SA); hb, lb = split_f16(B, SB)
    s16 = mfma_acc([(ha, lb), (la, hb), (ha, hb)]) / (SA * SB)
    s16 = s16.astype(np.float32).astype(np.float64)
    h, m, l = split_bf16(A); hB, mB, lB = split_bf16(B)
    s6 = mfma_acc([(h, lB), (l, hB), (m, mB), (h, mB), (m, hB), (h, hB)])
    sc = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    def err(x): return np.sqrt(np.mean(((x - ref) / sc) ** 2)), np.max(np.abs(x - ref) / sc)
    print(K, kind, "chain %.3e %.3e | fp16x2 %.3e %.3e | bf16x3(6) %.3e %.3e" % (*err(c), *err(s16), *err(s6)))
print("--- stress: loose bounds / heavy tails")
for loose in (1.0, 2.0**7, 2.0**10, 2.0**13):
  for tail in (2.0, 4.0):
    K, M, N = 512, 96, 96
    A = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K)) * tail) * 1e-5).astype(np.float32)
    B = (rng.standard_normal((N, K)) * np.sqrt(2.0 / N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    c = chain32(A, B)
    SA, SB = pow2_scale(np.abs(A).max() * loose), pow2_scale(np.abs(B).max())
    ha, la = split_f16(A, SA); hb, lb = split_f16(B, SB)
    s16 = (mfma_acc([(ha, lb), (la, hb), (ha, hb)]) / (SA * SB)).astype(np.float32).astype(np.float64)
    sc = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    def err(x): return np.sqrt(np.mean(((x - ref) / sc) ** 2)), np.max(np.abs(x - ref) / sc)
    print("loose 2^%d tail %.0f chain %.3e %.3e | fp16x2 %.3e %.3e" % (np.log2(loose), tail, *err(c), *err(s16)))
