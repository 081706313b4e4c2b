"""Per-workgroup timeline of the flat-tile conv kernel (diagnostic flag 32): where each workgroup ran (XCD, CU), when it
started and when each of its tiles finished.  Prints co-residency and the spread of start / end times."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from speech_decoding_amd import ops, lib as L

def main():
    dev = "cuda:0"
    extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    T, cin, cout, dil = 360, 320, int(sys.argv[3]) if len(sys.argv) > 3 else 320, 4
    mode = sys.argv[4] if len(sys.argv) > 4 else "full"        # full | plain (no bias / residual / statistics) | nores
    dtype = torch.bfloat16
    x = ops.new_rows(B, T, cin, dtype, dev); x.normal_()
    w = torch.randn(cout, cin, 3, device=dev) / math.sqrt(3 * cin)
    wp = ops.pack_conv_weight(w, cout, cin, dtype)
    y = ops.new_rows(B, T, cout, dtype, dev)
    bias = torch.zeros(cout, device=dev)
    stats = torch.zeros((ops.conv_stats_rows(B, T, 3, cout, 16384), 2, cout), device=dev)
    dbg = torch.zeros((1024, 32), dtype=torch.int64, device=dev)

    def run(flags, use_dbg):
        a = L.ConvArgs()
        a.x, a.w, a.bias, a.res, a.y, a.y_pre = x.data_ptr(), wp.data_ptr(), bias.data_ptr(), (x.data_ptr() if cin == cout else None), y.data_ptr(), None
        a.widx, a.stats, a.partial = None, stats.data_ptr(), (dbg.data_ptr() if use_dbg else None)
        if mode == "plain":
            a.bias, a.res, a.stats = None, None, None
        if mode == "nores":
            a.res = None
        a.bn_x, a.bn_coef = None, None
        a.B, a.T, a.Cin_p, a.Cout_p, a.KS, a.dil = B, T, cin, cout, 3, dil
        a.x_pitch, a.w_pitch = cin, cin
        a.x_row0, a.x_sample_rows, a.x_rows_limit = L.ROW_PAD, L.rows_tp(T), x.shape[0]
        a.w_rows_limit, a.ksplit = cout, 1
        a.flags, a.dtype = flags, L.BF16
        L.check(L.load().sda_conv_gemm(C.byref(a), torch.cuda.current_stream().cuda_stream), "conv")

    for _ in range(3):
        run(16384 | extra, False)
    torch.cuda.synchronize()
    run(16384 | 32 | extra, True)
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    if os.environ.get("FLAT_TIMELINE_SAVE"):
        np.save(os.environ["FLAT_TIMELINE_SAVE"], d)
    d = d[d[:, 4] != 0]
    t0 = d[:, 4].min()
    hw, xcc = d[:, 0], d[:, 1] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
    place = xcc * 1000 + cu
    uniq, counts = np.unique(place, return_counts=True)
    print(f"{len(d)} workgroups on {len(uniq)} (xcd, cu) places; workgroups per place: {np.bincount(counts)}")
    nst = (d[:, 4:9] != 0).sum(axis=1)
    last = nst - 1
    rows = np.arange(len(d))
    dt_real = (d[rows, 4 + last] - d[:, 4]) / 100.0            # us
    dt_cyc = (d[rows, 10 + last] - d[:, 10]).astype(np.float64)
    ghz = dt_cyc / dt_real / 1e3
    print(f"shader clock over each workgroup's life: median {np.median(ghz):.3f} GHz  min {ghz.min():.3f}  max {ghz.max():.3f}")
    for k in range(int(nst.max())):
        tk = (d[nst > k, 4 + k] - t0) / 100.0
        print(f"stamp {k}: n={len(tk)} min {tk.min():.1f} us  median {np.median(tk):.1f}  max {tk.max():.1f}")
    # spread: when workgroups start / end, how long they live, by XCD
    start = (d[:, 4] - t0) / 100.0
    end = (d[rows, 4 + last] - t0) / 100.0
    life = end - start
    q = lambda v: "  ".join(f"{np.percentile(v, p_):6.1f}" for p_ in (0, 10, 50, 90, 100))
    print(f"start  us (min p10 p50 p90 max): {q(start)}")
    print(f"end    us (min p10 p50 p90 max): {q(end)}")
    print(f"life   us (min p10 p50 p90 max): {q(life)}")
    for xc in np.unique(xcc):
        sel = xcc == xc
        print(f"   xcd {xc}: {sel.sum():3d} workgroups  life p50 {np.median(life[sel]):6.1f}  max {life[sel].max():6.1f}  end max {end[sel].max():6.1f}  clock {np.median(ghz[sel]):.3f}")
    # per-plan durations
    plans = d[:, 2]
    for pl in np.unique(plans):
        sel = d[plans == pl]
        lead, pairs, tail = pl & 0xff, (pl >> 8) & 0xff, (pl >> 16) & 0xff
        dur = np.diff(sel[:, 4:4 + 1 + lead + pairs + tail].astype(np.float64), axis=1) / 100.0
        print(f"plan lead={lead} pairs={pairs} tail={tail}: {len(sel)} workgroups, tile durations (us, median): {np.median(dur, axis=0).round(1)}")
    # co-resident pairs: do they have different plans?
    same = diff = 0
    for u_ in uniq[counts == 2]:
        pp = plans[place == u_]
        if pp[0] == pp[1]: same += 1
        else: diff += 1
    print(f"places with two workgroups: same plan {same}, different plans {diff}")
    if extra & 128:
        seg = d[:, 16:21].astype(np.float64)
        seg = seg[seg[:, 4] > 0]
        per = seg[:, :4] / seg[:, 4:5]
        names = ["barrier exit -> operands in registers", "MFMA + DMA issue", "vmcnt wait", "barrier"]
        print(f"K-loop phases of the stamped build ({int(np.median(seg[:, 4]))} phases per workgroup), cycles per phase (median over workgroups):")
        for i, nme in enumerate(names):
            print(f"   {nme:40s} {np.median(per[:, i]):8.0f}")
        print(f"   {'sum':40s} {np.median(per.sum(axis=1)):8.0f}")

if __name__ == "__main__":
    main()
