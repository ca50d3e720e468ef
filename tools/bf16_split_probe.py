#!/usr/bin/env python3
"""Round-4 probe: do the step's largest fp32 GEMMs run faster as EXACT three-way bf16 splits?

fp32 has a 24-bit significand = three 8-bit bf16 pieces, so a = a_h + a_m + a_l exactly (round-to-nearest pieces, the
residues are exact fp32 subtractions).  All NINE cross products a_i b_j are exact in fp32 (8 x 8 bits) and accumulate in fp32
on `v_mfma_f32_32x32x16_bf16`: the arithmetic is not narrower than the fp32 MFMA's.  Through the library that is ONE bf16
GEMM with fp32 output over K-concatenated planes:

    A' [M, 9K] = [a_l a_l a_m | a_l a_m a_h | a_m a_h a_h]   (smallest products first)
    B' [9K, N] = [b_l b_m b_l | b_h b_m b_l | b_h b_m b_h]^T-blocks in the matching order

Measured per GEMM shape of the 4096-row DPPO step (the three largest + the head):
  * tuned fp32 GEMM (the product's path),
  * the 9-product bf16 GEMM alone (planes already resident = what producer-fused splitting would give),
  * the split passes (torch ops: an upper bound of a fused producer),
  * the 6-product form (drops m*l, l*m, l*l: 2^-23-relative terms -- LABELLED, never `f32`),
  * max / rms error of every variant against float64, relative to sum |a||b|.

Kill criterion (VERDICT): < 1.3x over the tuned fp32 GEMM with the split cost included -> record and stop.
Run on the GPU box:  python tools/bf16_split_probe.py > gpurun_out/bf16_split_probe.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def split3(a):
    h = a.bfloat16()
    r = a - h.float()
    m = r.bfloat16()
    l = (r - m.float()).bfloat16()
    return h, m, l


# (i, j) pairs of A-plane x B-plane, smallest magnitude first; planes indexed 0 = h, 1 = m, 2 = l
ORDER9 = [(2, 2), (2, 1), (1, 2), (2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]
ORDER6 = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]


def concat(planes_a, planes_b, order, batched):
    ka = torch.cat([planes_a[i] for i, _ in order], dim=-1)            # [.., M, len*K]
    kb = torch.cat([planes_b[j] for _, j in order], dim=-2)            # [.., len*K, N]
    return ka.contiguous(), kb.contiguous()


def timed(fn, reps=20, replays=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * replays)


def main():
    from pfpn_amd.tuning import enable_tuned_gemms
    print("tuned selection loaded:", enable_tuned_gemms(tune_missing=True))
    dev = torch.device("cuda")
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    B = 4096
    # name, A shape, B shape (batched when 3-d), transposes as the step issues them
    shapes = [("L1 fwd   2x[4096,1024]@[1024,512]", (2, B, 1024), (2, 1024, 512)),
              ("L1 dW    2x[1024,4096]@[4096,512]", (2, 1024, B), (2, B, 512)),
              ("L1 dX    2x[4096,512]@[512,1024]", (2, B, 512), (2, 512, 1024)),
              ("head fwd [4096,512]@[512,1260]", (B, 512), (512, 1260)),
              ("head dW  [512,4096]@[4096,1260]", (512, B), (B, 1260)),
              ("L0 fwd   [4096,197]@[197,2048]", (B, 197), (197, 2048))]
    tot = {"f32": 0.0, "s9": 0.0, "s9+split": 0.0, "s6": 0.0}
    for name, sa, sb in shapes:
        batched = len(sa) == 3
        # activations in [0, 6] like post-ReLU6 layers / weights ~ N(0, 0.01): realistic magnitudes for the error columns
        a = (torch.rand(sa, device=dev, generator=g) * 6.0) * (torch.rand(sa, device=dev, generator=g) > 0.4)
        b = torch.randn(sb, device=dev, generator=g) * 0.01
        mm = torch.bmm if batched else torch.mm
        out32 = torch.empty(sa[:-1] + sb[-1:], device=dev)
        t32 = timed(lambda: mm(a, b, out=out32))
        pa, pb = split3(a), split3(b)
        assert torch.equal(pa[0].float() + pa[1].float() + pa[2].float(), a), "three-way split is not exact"
        a9, b9 = concat(pa, pb, ORDER9, batched)
        a6, b6 = concat(pa, pb, ORDER6, batched)
        try:
            f9 = lambda: mm(a9, b9, out_dtype=torch.float32)
            f6 = lambda: mm(a6, b6, out_dtype=torch.float32)
            y9, y6 = f9(), f6()
        except Exception as e:
            print(name, "bf16 -> f32 GEMM not available:", repr(e)[:200])
            continue
        t9, t6 = timed(f9), timed(f6)
        tsa = timed(lambda: split3(a))
        tsb = timed(lambda: split3(b))
        ref = mm(a.double(), b.double())
        scale = mm(a.double().abs(), b.double().abs())
        err = lambda y: ((y.double() - ref).abs() / scale.clamp_min(1e-30))
        e32, e9, e6 = err(out32), err(y9), err(y6)
        flop = 2.0 * a.numel() * sb[-1]
        print("%s\n   fp32 tuned %7.1f us (%5.1f TF)   split9 %7.1f us (%5.1f TF-equiv, bf16 rate %6.1f TF)   split6 %7.1f us"
              "   split passes A %6.1f us  B %6.1f us" % (name, t32, flop / t32 / 1e6, t9, flop / t9 / 1e6, 9 * flop / t9 / 1e6,
                                                          t6, tsa, tsb))
        print("   speed-up split9 alone %.2fx, with the activation split %.2fx, with both splits %.2fx; split6 alone %.2fx"
              % (t32 / t9, t32 / (t9 + tsa), t32 / (t9 + tsa + tsb), t32 / t6))
        print("   error / sum|a||b|:  fp32 max %.2e rms %.2e | split9 max %.2e rms %.2e | split6 max %.2e rms %.2e"
              % (e32.max(), e32.pow(2).mean().sqrt(), e9.max(), e9.pow(2).mean().sqrt(), e6.max(), e6.pow(2).mean().sqrt()))
        tot["f32"] += t32
        tot["s9"] += t9
        tot["s9+split"] += t9 + tsa
        tot["s6"] += t6
    print("\nsum over the listed GEMMs: fp32 %.1f us | split9 %.1f us (%.2fx) | split9 + activation split %.1f us (%.2fx) | "
          "split6 %.1f us (%.2fx)" % (tot["f32"], tot["s9"], tot["f32"] / max(tot["s9"], 1e-9), tot["s9+split"],
                                      tot["f32"] / max(tot["s9+split"], 1e-9), tot["s6"], tot["f32"] / max(tot["s6"], 1e-9)))
    print("kill criterion: < 1.30x with the split cost included")


if __name__ == "__main__":
    main()
