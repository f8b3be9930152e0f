import torch, itertools
dev = torch.device("cuda:0")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, N, K) in ((1024, 320, 3200), (1024, 320, 800)):
    for ta, tb in itertools.product((0, 1), (0, 1)):
        A = torch.randn(36, K, M, device=dev).transpose(1, 2) if ta else torch.randn(36, M, K, device=dev)   # ta: M contiguous
        Bm = torch.randn(36, N, K, device=dev).transpose(1, 2) if tb else torch.randn(36, K, N, device=dev)  # tb: K contiguous
        out = torch.empty(36, M, N, device=dev)
        us = timed(lambda: torch.bmm(A, Bm, out=out))
        print(f"M={M} N={N} K={K} A {'M' if ta else 'K'}-contig, B {'K' if tb else 'N'}-contig: {us:7.1f} us {2*36*M*N*K/us*1e-6:6.1f} TF", flush=True)
    # transposed product: out^T [N][M]
    A = torch.randn(36, M, K, device=dev); Bm = torch.randn(36, N, K, device=dev); out = torch.empty(36, N, M, device=dev)
    us = timed(lambda: torch.bmm(Bm, A.transpose(1, 2), out=out))
    print(f"  swapped (out [N][M]), both K-contig: {us:7.1f} us", flush=True)
