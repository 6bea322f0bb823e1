"""Yardstick only: the same GEMM shapes through torch.mm (rocBLAS / hipBLASLt fp32) -- how far the
hand-written fp32 MFMA kernels are from the vendor library.  Not used by the product."""
import torch, time
n = 232968
dev = "cuda"
shapes = [("fwd 608->128", (n, 608), (608, 128), False, False), ("fwd 128->128", (n, 128), (128, 128), False, False),
          ("fwd 128->41", (n, 128), (128, 41), False, False), ("bwd X^T.G 608x128", (n, 608), (n, 128), True, False),
          ("bwd X^T.G 128x128", (n, 128), (n, 128), True, False), ("bwd X^T.G 128x41", (n, 128), (n, 41), True, False),
          ("bwd G.W^T 128->128", (n, 128), (128, 128), False, True), ("bwd G.W^T 41->128", (n, 41), (128, 41), False, True)]
for name, sa, sb, at, bt in shapes:
    A = torch.randn(sa, device=dev); B = torch.randn(sb, device=dev)
    a = A.t() if at else A; b = B.t() if bt else B
    for _ in range(3): C = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): C = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * a.shape[0] * a.shape[1] * b.shape[1]
    print(f"{name}: {ms*1e3:7.1f} us  {fl/ms/1e9:6.1f} TF", flush=True)
