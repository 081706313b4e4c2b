"""What the vendor library's GEMM reaches on the step's matrix shapes (diagnostic only: the product path never calls it).

Gives a yardstick for the hand-written kernels: the k=3 conv as an im2col GEMM (rows x 3*Cin @ 3*Cin x Cout — the library is
handed the unfolded operand for free, which the conv kernels never materialise), its weight gradient (3*Cin x rows @ rows x Cout),
the two 1x1 projections.  Prints microseconds, TFLOP/s and the kernel the library chose (macro tile in its name).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def kernel_name(fn):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        fn(); torch.cuda.synchronize()
    names = [e.key for e in p.key_averages() if e.device_type == torch.autograd.DeviceType.CUDA]
    return "; ".join(n[:110] for n in names)


def main():
    dev, dt = "cuda:0", torch.bfloat16
    R = 256 * 360
    shapes = [("k=3 conv 320->320 as im2col GEMM", R, 320, 960, "nn"), ("k=3 conv 320->640", R, 640, 960, "nn"),
              ("k=3 weight gradient 320->320", 960, 320, R, "tn"), ("conv_final2 640->1024", R, 1024, 640, "nn"),
              ("conv_final2 weight gradient", 1024, 640, R, "tn"), ("conv_final1 320->640", R, 640, 320, "nn")]
    for name, M, N, K, mode in shapes:
        if mode == "nn":
            A = torch.randn(M, K, device=dev, dtype=dt); B = torch.randn(N, K, device=dev, dtype=dt)
            fn = lambda: torch.matmul(A, B.t())
        else:
            A = torch.randn(K, M, device=dev, dtype=dt); B = torch.randn(K, N, device=dev, dtype=dt)
            fn = lambda: torch.matmul(A.t(), B)
        us = timeit(fn)
        print(f"{name:36s} M={M:6d} N={N:5d} K={K:6d} {us:7.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {kernel_name(fn)}", flush=True)


if __name__ == "__main__":
    main()
