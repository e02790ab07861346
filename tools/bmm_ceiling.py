"""Reference ceiling: what torch.bmm (hipBLASLt/rocBLAS sgemm) reaches on the two GEMM shapes of the path."""
import torch
dev = torch.device('cuda:0')
M, B = 64, 512
for P in (512, 2048, 8192):
    X = torch.randn(M, B, P, device=dev)
    W = torch.randn(M, B, B, device=dev)
    for name, fn in (('dX=W@X  ', lambda: torch.bmm(W, X)), ('K=X@X^T ', lambda: torch.bmm(X, X.transpose(1, 2)))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('%s P=%5d  %8.1f us  %6.1f TF' % (name, P, ms * 1e3, 2.0 * M * B * B * P / ms / 1e9))
