"""What does a CU mask select?  (diagnostic)  Times one non-persistent MFMA-bound kernel (conv_gemm, 1536 workgroups) and one
HBM-bound copy on CU-masked streams: the time ratio against the unmasked stream is the effective share of the chip."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops, lib as L

dev = "cuda:0"
B, T, cin, cout = 256, 360, 320, 320
dtype = torch.bfloat16
x = ops.new_rows(B, T, cin, dtype, dev); x.normal_()
w = torch.randn(cout, cin, 3, device=dev) / math.sqrt(3 * cin)
wp = ops.pack_conv_weight(w, cout, cin, dtype)
y = ops.new_rows(B, T, cout, dtype, dev)
big = torch.empty(256 << 20, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
cus = torch.cuda.get_device_properties(0).multi_processor_count


def mk(bits):
    words = [sum(bits[32 * w_ + j] << j for j in range(32) if 32 * w_ + j < cus) for w_ in range((cus + 31) // 32)]
    return torch.cuda.ExternalStream(ops.stream_create_cumask(words), device=dev)


def timeit(st, fn, n=10):
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n): fn()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


conv = lambda: ops.conv_gemm(x, wp, y, B=B, T=T, KS=3, dil=4)
copy = lambda: big2.copy_(big)
masks = {"unmasked stream": None,
         "all ones": [1] * cus,
         "i % 8 < 4": [1 if i % 8 < 4 else 0 for i in range(cus)],
         "i % 8 < 1": [1 if i % 8 < 1 else 0 for i in range(cus)],
         "i < 128": [1 if i < 128 else 0 for i in range(cus)],
         "i < 32": [1 if i < 32 else 0 for i in range(cus)],
         "(i // 8) % 2 == 0": [1 if (i // 8) % 2 == 0 else 0 for i in range(cus)],
         "(i // 8) < 16": [1 if (i // 8) < 16 else 0 for i in range(cus)]}
base = None
for name, bits in masks.items():
    st = torch.cuda.Stream(device=dev) if bits is None else mk(bits)
    tc, tm = timeit(st, conv), timeit(st, copy)
    base = base or (tc, tm)
    print(f"{name:22s} bits set {sum(bits) if bits else cus:4d}: conv {tc:8.1f} us (x{tc / base[0]:5.2f})   256 MiB copy {tm:8.1f} us (x{tm / base[1]:5.2f})", flush=True)
