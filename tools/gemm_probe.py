#!/usr/bin/env python3
"""tools/gemm_probe.py — the dense X·W of a GCN layer (plain library GEMM, fp32) at the Reddit-shaped
sizes, to see what layout rocBLAS/hipBLASLt likes: K as given (602), K padded to a multiple of 8/32/64."""
import sys
import torch

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 232965


def t(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for K, N in ((602, 128), (602, 16), (602, 41), (128, 41), (128, 128), (100, 47)):
    X = torch.randn(M, K, device=dev)
    W = torch.randn(K, N, device=dev)
    base = t(lambda: torch.mm(X, W))
    line = f"M={M} K={K} N={N}: mm {base:.3f} ms ({2.0 * M * K * N / base / 1e9:.1f} TFLOP/s)"
    for Kp in sorted({(K + 7) // 8 * 8, (K + 31) // 32 * 32, (K + 63) // 64 * 64}):
        if Kp == K:
            continue
        Xp = torch.zeros(M, Kp, device=dev); Xp[:, :K] = X
        Wp = torch.zeros(Kp, N, device=dev); Wp[:K] = W
        tp = t(lambda: torch.mm(Xp, Wp))
        err = float((torch.mm(Xp, Wp) - torch.mm(X, W)).abs().max())
        line += f" | K->{Kp}: {tp:.3f} ms (diff {err:.1e})"
    Np = (N + 31) // 32 * 32
    if Np != N:
        Wn = torch.zeros(K, Np, device=dev); Wn[:, :N] = W
        tn = t(lambda: torch.mm(X, Wn))
        line += f" | N->{Np}: {tn:.3f} ms"
    # the transposed product the backward pass needs: dW = X^T · dY
    dY = torch.randn(M, N, device=dev)
    tb = t(lambda: torch.mm(X.t(), dY))
    line += f" | X^T·dY {tb:.3f} ms"
    print(line, flush=True)
    del X, W
