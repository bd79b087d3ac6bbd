#!/usr/bin/env python
"""in_proj forward ([7569,384] x [384,1152] + bias) is the one GEMM hipBLASLt runs at 65 TF/s: options."""
import torch
import torch.nn.functional as F

N, d = 7569, 384
dev = torch.device("cuda:0")
x = torch.randn(N, d, device=dev)
w = torch.randn(3 * d, d, device=dev)
b = torch.randn(3 * d, device=dev)


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


ref = F.linear(x, w, b)
print("F.linear (hipBLASLt + bias epilogue)", t(lambda: F.linear(x, w, b)))


def rocblas():
    torch.backends.cuda.preferred_blas_library("cublas")
    y = x.mm(w.t())
    torch.backends.cuda.preferred_blas_library("cublaslt")
    return y.add_(b)


print("rocBLAS mm + bias add", t(rocblas), (rocblas() - ref).abs().max().item())
out = torch.empty(N, 3 * d, device=dev)


def two():
    torch.addmm(b[:2 * d], x, w[:2 * d].t(), out=out[:, :2 * d])
    torch.addmm(b[2 * d:], x, w[2 * d:].t(), out=out[:, 2 * d:])
    return out


print("two addmm into column slices (768 + 384)", t(two), (two() - ref).abs().max().item())


def three():
    for i in range(3):
        torch.addmm(b[i * d:(i + 1) * d], x, w[i * d:(i + 1) * d].t(), out=out[:, i * d:(i + 1) * d])
    return out


print("three addmm into column slices (3 x 384)", t(three), (three() - ref).abs().max().item())
w4 = torch.randn(4 * d, d, device=dev); b4 = torch.randn(4 * d, device=dev)
print("n=1536 F.linear for scale", t(lambda: F.linear(x, w4, b4)))
w7 = torch.randn(7 * d, d, device=dev); b7 = torch.randn(7 * d, device=dev)
print("n=2688 F.linear (ABDE + qkv in one GEMM)", t(lambda: F.linear(x, w7, b7)))
