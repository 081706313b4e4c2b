"""Throughput of the frozen wav2vec 2.0 embedder at xlsr-53's dimensions (diagnostic; random weights).
Usage: python tools/bench_w2v2.py [seconds_of_audio] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_decoding_amd.wav2vec2 import Wav2Vec2Config, Wav2Vec2Embedder

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    dtype = dict(bf16=torch.bfloat16, fp16=torch.float16, fp32=torch.float32)[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
    cfg = Wav2Vec2Config()
    g = torch.Generator().manual_seed(0)
    sd = {}
    def lin(n, o, i): sd[n + ".weight"] = torch.randn(o, i, generator=g) * (0.7 / i ** 0.5); sd[n + ".bias"] = torch.randn(o, generator=g) * 0.1
    def ln(n, c): sd[n + ".weight"] = 1 + 0.1 * torch.randn(c, generator=g); sd[n + ".bias"] = 0.1 * torch.randn(c, generator=g)
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"feature_extractor.conv_layers.{i}."
        sd[p + "conv.weight"] = torch.randn(c, cin, k, generator=g) * (0.7 / (cin * k) ** 0.5); sd[p + "conv.bias"] = torch.randn(c, generator=g) * 0.1
        ln(p + "layer_norm", c); cin = c
    ln("feature_projection.layer_norm", 512); lin("feature_projection.projection", 1024, 512)
    sd["encoder.pos_conv_embed.conv.weight_g"] = 0.5 + torch.rand(1, 1, 128, generator=g)
    sd["encoder.pos_conv_embed.conv.weight_v"] = torch.randn(1024, 64, 128, generator=g) * 0.05
    sd["encoder.pos_conv_embed.conv.bias"] = torch.randn(1024, generator=g) * 0.1
    ln("encoder.layer_norm", 1024)
    for i in range(24):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"): lin(p + "attention." + n, 1024, 1024)
        ln(p + "layer_norm", 1024); ln(p + "final_layer_norm", 1024)
        lin(p + "feed_forward.intermediate_dense", 4096, 1024); lin(p + "feed_forward.output_dense", 1024, 4096)
    emb = Wav2Vec2Embedder(sd, cfg, dtype=dtype)
    wave = torch.randn(1, int(secs * 16000), generator=g).cuda()
    for _ in range(2): out = emb.embed(wave)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 3
    for _ in range(n): out = emb.embed(wave)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    frames = out.shape[1]
    flop = frames * (2 * (4 * 1024 * 1024 + 2 * 1024 * 4096) * 24 + 2 * 128 * 64 * 1024 + 2 * 512 * 1024 + 2 * 512 * 512 * (3 * 16 + 3 * 8 + 3 * 4 + 3 * 2 + 2 * 1 + 2 * 1) / 1)
    print(f"{secs:.0f} s of audio, {frames} frames, {str(dtype)[6:]}: {dt * 1e3:.1f} ms per call = {secs / dt:.0f} x real time, "
          f"{frames / dt:.0f} frames/s, ~{flop / dt / 1e12:.0f} TFLOP/s (GEMM FLOPs, attention excluded)")

if __name__ == "__main__":
    main()
