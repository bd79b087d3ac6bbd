"""CPU emulation (numpy) of the two arithmetic forms of the ring GEMM against fp64: three bf16 pieces / 6 products (round 2)
and two fp16 pieces under a power-of-two scale / 3 products (round 4), each k-step of 16 rounded into an fp32 accumulator
per piece product as the MFMA does.  `python tools/gemm16_emulate.py` prints max and rms errors relative to max|result| for
N(0,1), offset, gradient-sized, long-K and ReLU operands; the numbers quoted in DESIGN.md section 4.5e come from it."""
import numpy as np
rng = np.random.default_rng(0)
def split_bf16_3(v):
    v = v.astype(np.float32)
    def trunc(x):
        return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    h = trunc(v); r1 = v - h; m = trunc(r1); r2 = r1 - m; l = trunc(r2)
    return h, m, l
def split_f16_2(v, scale_exp):
    # v * 2^scale_exp -> hi (RN fp16), lo (RN fp16 of the exact fp32 remainder)
    s = np.ldexp(v.astype(np.float32), scale_exp).astype(np.float32)
    hi = s.astype(np.float16)
    r = (s - hi.astype(np.float32)).astype(np.float32)
    lo = r.astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)
def row_exp(v, target=12):
    mx = np.abs(v).max(axis=1, keepdims=True)
    mx = np.where(mx > 0, mx, 1.0)
    e = np.floor(np.log2(mx)).astype(np.int32)
    return (target - e)
def mm32(a, b):
    # fp32 accumulation in k-blocks of 16 (MFMA-like): products exact in fp32? emulate with float64 product per block then round to fp32 accumulate
    M, K = a.shape; N = b.shape[0]
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, 16):
        acc = (acc + (a[:, k0:k0+16].astype(np.float64) @ b[:, k0:k0+16].astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return acc
def gemm_bf16_6(A, W):
    ah, am, al = split_bf16_3(A); wh, wm, wl = split_bf16_3(W)
    acc = np.zeros((A.shape[0], W.shape[0]), np.float32)
    K = A.shape[1]
    for k0 in range(0, K, 16):
        s = slice(k0, k0+16)
        for (x, y) in [(am, wm), (al, wh), (ah, wl), (am, wh), (ah, wm), (ah, wh)]:
            acc = (acc + (x[:, s].astype(np.float64) @ y[:, s].astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return acc
def gemm_f16_3(A, W, ea=None, ew=None, tgt=12):
    if ea is None: ea = row_exp(A, tgt)
    if ew is None: ew = row_exp(W, tgt)
    ah, al = split_f16_2(A, ea); wh, wl = split_f16_2(W, ew)
    acc = np.zeros((A.shape[0], W.shape[0]), np.float32)
    K = A.shape[1]
    for k0 in range(0, K, 16):
        s = slice(k0, k0+16)
        for (x, y) in [(al, wh), (ah, wl), (ah, wh)]:
            acc = (acc + (x[:, s].astype(np.float64) @ y[:, s].astype(np.float64).T).astype(np.float32)).astype(np.float32)
    return np.ldexp(acc, -(ea + ew.T)).astype(np.float32)
def report(name, A, W):
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    scale = np.abs(ref).max()
    lib = (A @ W.T)
    r = {}
    r['np_sgemm'] = np.abs(lib - ref).max() / scale
    r['bf16x6'] = np.abs(gemm_bf16_6(A, W) - ref).max() / scale
    r['f16x3_row'] = np.abs(gemm_f16_3(A, W) - ref).max() / scale
    ea = np.full((A.shape[0],1), row_exp(A.reshape(1,-1))[0,0]); 
    r['f16x3_tensorA'] = np.abs(gemm_f16_3(A, W, ea=ea) - ref).max() / scale
    rms = lambda x: np.sqrt(np.mean((x-ref)**2))/np.sqrt(np.mean(ref**2))
    r['rms_np'] = rms(lib); r['rms_bf16x6'] = rms(gemm_bf16_6(A, W)); r['rms_f16x3'] = rms(gemm_f16_3(A, W))
    print(name, {k: float('%.3g' % v) for k, v in r.items()})
M, K, N = 512, 384, 384
A = rng.standard_normal((M, K)).astype(np.float32)
W = (rng.uniform(-1, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
report('N(0,1)', A, W)
report('offset+3', A + 3, W)
report('offset+100', A + 100, W)
G = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, 1)) * 3) * 1e-6).astype(np.float32)
report('grad-like rows x e^N(0,3) 1e-6', G, W)
A2 = rng.standard_normal((M, 2688)).astype(np.float32); W2 = (rng.uniform(-1, 1, (N, 2688)) / np.sqrt(K)).astype(np.float32)
report('K=2688', A2, W2)
# relu-like (half zeros)
report('relu', np.maximum(A, 0), W)
