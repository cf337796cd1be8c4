"""would a Winograd F(4,3)-along-y form of the split-bf16 conv (half the MFMAs of the direct form, 3/4 of F(2,3)'s) hold
the network's tolerances?  Measured here instead of argued: the conv arithmetic of the HIP kernels is EMULATED inside the oracle graph on the
CPU -- weights transformed in float64 and rounded to fp32 (as engine.pack_conv_weight_wy does), input transform in fp32, every product as
hi*hi + hi*lo + lo*hi with hi = bf16(v), lo = bf16(v - hi) and fp32 accumulation, output transform in fp32 -- and the whole network's output
is compared with the plain fp32 oracle at BASELINE config 2 (7x3x180x320, x4): max-abs, |dPSNR-Y|, |dSSIM-Y| against the same synthetic GT.

    python3 tools/emulate_winograd_f43.py [--h 180 --w 320] [--forms direct,f23,f43] [--where all|batched]

`--where batched` = only the convs of the launches that take the Winograd form with ONE clip in flight (the 6-conv families of the two
propagation directions and the pyramid); `all` = every 3x3 conv with cout % 64 == 0 and cin % 16 == 0 (the throughput mode's choice).
The `f23` row calibrates the emulation: the shipped F(2,3) kernels measure 1.1-1.25e-5 on the GPU at this config.
Test infrastructure (imports oracle/ as the checker, like tools/fuzz_network.py); nothing here is reachable from savsr_amd/."""
import argparse
import math
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import savsr_oracle as O  # noqa: E402
from savsr_amd.metrics import calculate_psnr, calculate_ssim, tensor2img  # noqa: E402
from savsr_amd.utils import synth  # noqa: E402

MATS = {
    2: ([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
        [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]],
        [[1, 1, 1, 0], [0, 1, -1, -1]]),
    4: ([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
        [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
        [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]),
}


def split(v):
    hi = v.to(torch.bfloat16).to(torch.float32)
    lo = (v - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def conv3(xh, xl, wh, wl, **kw):
    """a*b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulation (the order of the three MFMAs of mma3())."""
    return F.conv2d(xl, wh, **kw) + F.conv2d(xh, wl, **kw) + F.conv2d(xh, wh, **kw)


def conv_direct(x, w, padding=1):
    xh, xl = split(x)
    wh, wl = split(w)
    return conv3(xh, xl, wh, wl, padding=padding)


def conv_wino_y(x, w, m):
    """3x3, zero pad 1, as F(m,3) along y on split-bf16 products.  x [B,Cin,H,W] fp32, w [Cout,Cin,3,3] fp32."""
    BT, G, AT = (torch.tensor(a, dtype=torch.float64) for a in MATS[m])
    n = m + 2
    B, cin, H, W = x.shape
    U = torch.einsum("ik,ockx->iocx", G, w.to(torch.float64)).to(torch.float32)          # [n][Cout][Cin][kx]: float64 transform, fp32 rounding
    T = math.ceil(H / m)
    xp = F.pad(x, (1, 1, 1, 1 + T * m - H))
    idx = (torch.arange(T)[:, None] * m + torch.arange(n)[None, :]).reshape(-1)
    d = xp[:, :, idx, :].view(B, cin, T, n, W + 2)
    V = torch.einsum("ij,bctjw->bctiw", BT.to(torch.float32), d)                           # input transform in fp32
    outs = []
    for i in range(n):
        vh, vl = split(V[:, :, :, i, :].contiguous())
        uh, ul = split(U[i].unsqueeze(2).contiguous())                                   # [Cout][Cin][1][3]
        outs.append(conv3(vh, vl, uh, ul))                                               # [B][Cout][T][W]
    M = torch.stack(outs, 3)
    o = torch.einsum("ij,bctjw->bctiw", AT.to(torch.float32), M)                           # output transform in fp32
    return o.reshape(B, -1, T * m, W)[:, :, :H].contiguous()


BATCHED = re.compile(r"^(f2p_win|p2f_win)\.(blocks\.\d+\.(conv0|conv2)\.\d+|blocks\.\d+\.osconv|merge)$|^h_win\.0\.(conv_h\.\d+|blocks\.\d+\.(conv0|conv2)\.\d+)$")


def install(form, where):
    """Replace the oracle's conv calls by the emulated arithmetic; returns a census dict."""
    census = {"emulated": 0, "plain": 0}
    conv0, os0 = O._conv, O.osconv2d

    def pick(pfx, cout, cin, ks):
        if ks != 3 or cout % 64 or cin % 16:
            return "plain"
        if form in ("f23", "f43") and where == "batched" and not BATCHED.match(pfx):
            return "direct"              # (those launches stay in the direct split-bf16 form)
        return form

    def run(kind, x, w, padding):
        if kind == "plain":
            census["plain"] += 1
            return F.conv2d(x, w, None, stride=1, padding=padding)
        census["emulated"] += 1
        if kind == "direct":
            return conv_direct(x, w, padding)
        return conv_wino_y(x, w, 2 if kind == "f23" else 4)

    def _conv(sd, pfx, x, padding):
        w = sd[pfx + ".weight"]
        b = sd.get(pfx + ".bias")
        kind = pick(pfx, w.shape[0], w.shape[1], w.shape[-1])
        if kind == "plain" and w.shape[-1] == 1 and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and form != "fp32":
            kind = "direct"              # the 1x1 convs run the direct split-bf16 kernel too
        y = run(kind, x, w, padding)
        return y if b is None else y + b.view(1, -1, 1, 1)

    def osconv2d(sd, pfx, x, scale):
        b, cin, h, w = x.shape
        weight = sd[pfx + ".weight"]
        cout = weight.size(1)
        s = torch.cat((torch.ones(1, 1) / scale[0], torch.ones(1, 1) / scale[1]), 1).repeat(b, 1)
        pooled = F.adaptive_avg_pool2d(x, 1).view(b, -1)
        v = torch.cat([s, pooled], dim=1)
        v = F.relu(F.linear(v, sd[pfx + ".scale_routing.0.weight"], sd[pfx + ".scale_routing.0.bias"]))
        v = F.relu(F.linear(v, sd[pfx + ".scale_routing.2.weight"], sd[pfx + ".scale_routing.2.bias"]))
        ca, fa, sa, ka = O.scale_attention(sd, pfx + ".attention", v.view(b, cin, 1, 1))
        assert b == 1
        # the engine folds ALL gates into the weight image (savsr_arch.py:148-149's equivalence) and runs a plain conv on x
        agg = torch.sum(sa * ka * weight.unsqueeze(0), dim=1).view(-1, cin, 3, 3) * ca.view(1, cin, 1, 1) * fa.view(cout, 1, 1, 1)
        return run(pick(pfx, cout, cin, 3), x, agg, 1)
    O._conv, O.osconv2d = _conv, osconv2d
    return census, (conv0, os0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=180)
    ap.add_argument("--w", type=int, default=320)
    ap.add_argument("--scale", type=float, nargs=2, default=[4, 4])
    ap.add_argument("--forms", default="direct,f23,f43")
    ap.add_argument("--where", default="all,batched")
    a = ap.parse_args()
    sc = tuple(a.scale)
    sd = synth.synth_state_dict(seed=0)
    lq = synth.synth_clip(7, 3, a.h, a.w, seed=0)
    H, W = O.get_hw(a.h, a.w, sc)
    gt = tensor2img(synth.synth_gt(3, H, W, seed=0))
    t0 = time.time()
    with torch.no_grad():
        ref = O.forward(sd, lq, sc)
    r_img = tensor2img(ref[0])
    p0, s0 = calculate_psnr(r_img, gt, 0, test_y_channel=True), calculate_ssim(r_img, gt, 0, test_y_channel=True)
    print(f"# {a.h}x{a.w} x{sc}: fp32 oracle {time.time() - t0:.1f} s; tolerances: max-abs 5e-5 (tests), |dPSNR-Y| 1e-3 dB, |dSSIM-Y| 1e-4 (north_star)", flush=True)
    for form in a.forms.split(","):
        for where in (a.where.split(",") if form in ("f23", "f43") else ["all"]):
            census, saved = install(form, where)
            t0 = time.time()
            try:
                with torch.no_grad():
                    out = O.forward(sd, lq, sc)
            finally:
                O._conv, O.osconv2d = saved
            img = tensor2img(out[0])
            err = float((out - ref).abs().max())
            dp = abs(calculate_psnr(img, gt, 0, test_y_channel=True) - p0)
            ds = abs(calculate_ssim(img, gt, 0, test_y_channel=True) - s0)
            verdict = "holds" if (err < 5e-5 and dp <= 1e-3 and ds <= 1e-4) else "FAILS"
            print(f"{form:6s} {where:7s}: max-abs {err:.2e}  rms {float((out - ref).pow(2).mean().sqrt()):.2e}  dPSNR-Y {dp:.1e} dB  dSSIM-Y {ds:.1e}  -> {verdict} "
                  f"every tolerance   ({census['emulated']} emulated convs, {census['plain']} plain; {time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
