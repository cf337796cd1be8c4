"""CPU oracle for the SAVSR per-frame inference path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product (savsr_amd/) never does: its forward fails loudly when the HIP library is missing.

What this is: a functional (state_dict driven) restatement of the reference network
`/root/reference/lbasicsr/archs/savsr_arch.py`, written against plain torch CPU ops in the
reference's own (un-fused, un-restructured) formulation, fp32.  Every function cites the
reference lines it follows.  The arithmetic itself is ATen (torch is un-pinned upstream:
requirements.txt:18 `torch>=1.9`); here it is torch 2.10.0 CPU.

Pinning: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, imported in isolation in the build
container: tools/gen_golden.py runs both on identical key-seeded weights/inputs and commits the
reference's outputs as fixtures under tests/golden/; tests/test_oracle_golden.py re-checks the
oracle against those fixtures everywhere, and tests/test_oracle_vs_reference.py re-runs the
live comparison whenever /root/reference is present.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

DEFAULT_CFG = dict(num_in_ch=3, num_feat=64, num_frame=7, slid_win=3, fusion_win=5, interval=0,
                   w1_num_block=4, w2_num_block=2, n_resgroups=4, n_resblocks=8,
                   downsample_scale=2, center_frame_idx=None)


# ----------------------------------------------------------------------------- small helpers
def get_hw(h: int, w: int, scale: Sequence[float]) -> Tuple[int, int]:
    """savsr_arch.py:745-751 -- Python round (half-to-even) of the double product."""
    return round(h * scale[0]), round(w * scale[1])


def _conv(sd: SD, pfx: str, x: Tensor, padding: int) -> Tensor:
    return F.conv2d(x, sd[pfx + ".weight"], sd.get(pfx + ".bias"), stride=1, padding=padding)


def _bn_eval(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """nn.BatchNorm2d in eval mode (running statistics, eps 1e-5)."""
    return F.batch_norm(x, sd[pfx + ".running_mean"], sd[pfx + ".running_var"],
                        sd[pfx + ".weight"], sd[pfx + ".bias"], training=False, eps=1e-5)


def _lrelu(x: Tensor, slope: float = 0.2) -> Tensor:
    return F.leaky_relu(x, slope)


# ----------------------------------------------------------------------------- OSConv (a6, a7)
def scale_attention(sd: SD, pfx: str, v: Tensor, ksize: int = 3):
    """savsr_arch.py:91-96 with getters :69-89.  v: [b, Cin, 1, 1].  Temperature is 1.0 (:22)."""
    b = v.size(0)
    a = F.relu(_bn_eval(sd, pfx + ".bn", F.conv2d(v, sd[pfx + ".fc.weight"])))
    ca = torch.sigmoid(_conv(sd, pfx + ".channel_fc", a, 0).view(b, -1, 1, 1))
    fa = torch.sigmoid(_conv(sd, pfx + ".filter_fc", a, 0).view(b, -1, 1, 1))
    sa = torch.sigmoid(_conv(sd, pfx + ".spatial_fc", a, 0).view(b, 1, 1, 1, ksize, ksize))
    ka = F.softmax(_conv(sd, pfx + ".kernel_fc", a, 0).view(b, -1, 1, 1, 1, 1), dim=1)
    return ca, fa, sa, ka


def osconv2d(sd: SD, pfx: str, x: Tensor, scale: Sequence[float]) -> Tensor:
    """savsr_arch.py:139-172 (_forward_impl_common), kernel 3x3, padding 1, groups 1."""
    b, cin, h, w = x.shape
    weight = sd[pfx + ".weight"]                       # [K, Cout, Cin, 3, 3]
    cout = weight.size(1)
    s = torch.cat((torch.ones(1, 1) / scale[0], torch.ones(1, 1) / scale[1]), 1).repeat(b, 1)   # :143-145
    pooled = F.adaptive_avg_pool2d(x, 1).view(b, -1)                                            # :146
    v = torch.cat([s, pooled], dim=1)
    v = F.relu(F.linear(v, sd[pfx + ".scale_routing.0.weight"], sd[pfx + ".scale_routing.0.bias"]))
    v = F.relu(F.linear(v, sd[pfx + ".scale_routing.2.weight"], sd[pfx + ".scale_routing.2.bias"]))
    ca, fa, sa, ka = scale_attention(sd, pfx + ".attention", v.view(b, cin, 1, 1))
    xg = (x * ca).reshape(1, -1, h, w)                                                          # :156-157
    agg = torch.sum(sa * ka * weight.unsqueeze(0), dim=1).view(-1, cin, 3, 3)                   # :158-163
    out = F.conv2d(xg, agg, None, stride=1, padding=1, groups=b).view(b, cout, h, w)            # :166-170
    return out * fa                                                                            # :171


def osadapt(sd: SD, pfx: str, x: Tensor, scale: Sequence[float]) -> Tensor:
    """savsr_arch.py:186-214.  mask = Sequential indices 0..13 (:189-206)."""
    m = pfx + ".mask"
    t = F.relu(_bn_eval(sd, m + ".1", _conv(sd, m + ".0", x, 1)))
    t = F.avg_pool2d(t, 2)
    t = F.relu(_bn_eval(sd, m + ".5", _conv(sd, m + ".4", t, 1)))
    t = F.relu(_bn_eval(sd, m + ".8", _conv(sd, m + ".7", t, 1)))
    t = F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    mask = torch.sigmoid(_bn_eval(sd, m + ".12", _conv(sd, m + ".11", t, 1)))
    return x + osconv2d(sd, pfx + ".adapt", x, scale) * mask                                    # :214


# ----------------------------------------------------------------------------- propagation (a4, a5, a9)
def residual_block(sd: SD, pfx: str, xs, scale, use_osconv: bool):
    """savsr_arch.py:399-415."""
    n = len(xs)
    x1 = [_lrelu(_conv(sd, f"{pfx}.conv0.{i}", xs[i], 1)) for i in range(n)]
    merge = torch.cat(x1, dim=1)
    if use_osconv:
        base = _lrelu(osconv2d(sd, pfx + ".osconv", merge, scale))
    else:
        base = _lrelu(_conv(sd, pfx + ".conv1", merge, 0))
    x2 = [_lrelu(_conv(sd, f"{pfx}.conv2.{i}", torch.cat([base, x1[i]], 1), 1)) for i in range(n)]
    return [xs[i] + x2[i] for i in range(n)]


def window_unit_l1(sd: SD, pfx: str, it: Tensor, h_past: Tensor, scale, num_block: int) -> Tensor:
    """savsr_arch.py:444-464.  it: [b, 3, c, h, w]; block 0 uses conv1, the rest OSConv (:434-440)."""
    b, t, c, h, w = it.shape
    x_c = it[:, t // 2]
    sup = [i for i in range(t) if i != t // 2]
    x_sup = it[:, sup].reshape(b, (t - 1) * c, h, w)
    h_sup = _lrelu(_conv(sd, pfx + ".conv_sup", x_sup, 1))
    h_c = _lrelu(_conv(sd, pfx + ".conv_c", x_c, 1))
    feats = [h_c, h_sup, h_past]
    for k in range(num_block):
        feats = residual_block(sd, f"{pfx}.blocks.{k}", feats, scale, use_osconv=(k >= 1))
    return _conv(sd, pfx + ".merge", torch.cat(feats, dim=1), 1)


def window_unit_l2(sd: SD, pfx: str, xs, scale, win_size: int, slid_win: int, num_block: int):
    """savsr_arch.py:485-501."""
    hf = [_lrelu(_conv(sd, f"{pfx}.conv_h.{i}", xs[i], 1)) for i in range(win_size)]
    out = list(hf) if len(hf) == 1 else []
    for i in range(win_size - slid_win + 1):
        sw = hf[i:i + slid_win]
        for k in range(num_block):
            sw = residual_block(sd, f"{pfx}.blocks.{k}", sw, scale, use_osconv=True)
        out.append(_conv(sd, pfx + ".merge", torch.cat(sw, dim=1), 1))
    return out


# ----------------------------------------------------------------------------- RCAN (a10)
def rcab(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """savsr_arch.py:504-549 (res_scale 1)."""
    r = F.relu(_conv(sd, pfx + ".rcab.0", x, 1))
    r = _conv(sd, pfx + ".rcab.2", r, 1)
    y = F.adaptive_avg_pool2d(r, 1)
    y = F.relu(_conv(sd, pfx + ".rcab.3.attention.1", y, 0))
    y = torch.sigmoid(_conv(sd, pfx + ".rcab.3.attention.3", y, 0))
    return r * y + x


def residual_group(sd: SD, pfx: str, x: Tensor, n_blocks: int) -> Tensor:
    """savsr_arch.py:552-571."""
    r = x
    for k in range(n_blocks):
        r = rcab(sd, f"{pfx}.residual_group.{k}", r)
    return _conv(sd, pfx + ".conv", r, 1) + x


# ----------------------------------------------------------------------------- SATU (a11-a15)
def sta_conv(feat: Tensor, kernel: Tensor, ksize: int = 5) -> Tensor:
    """savsr_arch.py:297-313: per-pixel, per-channel ksize x ksize dynamic filter, replicate pad."""
    b, c, h, w = feat.shape
    pad = (ksize - 1) // 2
    fp = F.pad(feat, (pad, pad, pad, pad), mode="replicate")
    fp = fp.unfold(2, ksize, 1).unfold(3, ksize, 1)            # [b,c,h,w,ky,kx]
    fp = fp.permute(0, 2, 3, 1, 5, 4).contiguous().reshape(b, h, w, c, -1)
    k = kernel.permute(0, 2, 3, 1).reshape(b, h, w, c, ksize, ksize)
    k = k.permute(0, 1, 2, 3, 5, 4).reshape(b, h, w, c, -1)
    return torch.sum(fp * k, -1).permute(0, 3, 1, 2).contiguous()


def satu_coords(h: int, w: int, scale: Sequence[float]):
    """savsr_arch.py:326-333.  Returns (H, W, coor_h[H], coor_w[W], floor_h[H], floor_w[W]).

    floor_* is the integer LR index grid `floor((Y+.5)/s + 1e-3)` evaluated in fp32 exactly as the
    reference evaluates it -- the bit-exact part of the contract."""
    H, W = get_hw(h, w, scale)
    ys = torch.arange(0, H, 1).float()
    xs = torch.arange(0, W, 1).float()
    fh = torch.floor((ys + 0.5) / scale[0] + 1e-3)
    fw = torch.floor((xs + 0.5) / scale[1] + 1e-3)
    ch = ((ys + 0.5) / scale[0]) - fh - 0.5
    cw = ((xs + 0.5) / scale[1]) - fw - 0.5
    return H, W, ch, cw, fh, fw


def satu_grid_sample(x: Tensor, offset: Tensor, scale: Sequence[float]) -> Tensor:
    """savsr_arch.py:262-295: bilinear, zeros padding, align_corners=True, fp32 grid from a
    float64 numpy meshgrid cast by torch.Tensor()."""
    b, _, h, w = x.shape
    H, W = get_hw(h, w, scale)
    g = np.stack(np.meshgrid(range(W), range(H)), axis=-1).astype(np.float64)
    g = torch.Tensor(g)
    g[:, :, 0] = (g[:, :, 0] + 0.5) / scale[1] - 0.5
    g[:, :, 1] = (g[:, :, 1] + 0.5) / scale[0] - 0.5
    g[:, :, 0] = g[:, :, 0] * 2 / (w - 1) - 1
    g[:, :, 1] = g[:, :, 1] * 2 / (h - 1) - 1
    g = g.permute(2, 0, 1).unsqueeze(0).expand([b, -1, -1, -1])
    o0 = torch.unsqueeze(offset[:, 0] * 2 / (w - 1), dim=1)
    o1 = torch.unsqueeze(offset[:, 1] * 2 / (h - 1), dim=1)
    g = (g + torch.cat((o0, o1), 1)).permute(0, 2, 3, 1)
    return F.grid_sample(x, g, mode="bilinear", padding_mode="zeros", align_corners=True)


def satu_heads(sd: SD, pfx: str, h: int, w: int, scale: Sequence[float]):
    """savsr_arch.py:335-350: coordinate MLP -> (offset, st_offset, routing) on the HR grid."""
    H, W, ch, cw, _, _ = satu_coords(h, w, scale)
    ch = ch.view(H, 1)
    cw = cw.view(1, W)
    inp = torch.cat((
        torch.ones_like(ch).expand([-1, W]).unsqueeze(0) / scale[1],
        torch.ones_like(ch).expand([-1, W]).unsqueeze(0) / scale[0],
        ch.expand([-1, W]).unsqueeze(0),
        cw.expand([H, -1]).unsqueeze(0)), 0).unsqueeze(0)
    e = F.relu(_conv(sd, pfx + ".body.0", inp, 0))
    e = F.relu(_conv(sd, pfx + ".body.2", e, 0))
    off = _conv(sd, pfx + ".offset", e, 0)
    soff = _conv(sd, pfx + ".st_offset", e, 0)
    r = torch.sigmoid(_conv(sd, pfx + ".routing.0", e, 0))
    return off, soff, r


def sta_upsample(sd: SD, pfx: str, x: Tensor, scale: Sequence[float], st_feat: Tensor) -> Tensor:
    """savsr_arch.py:315-376."""
    b, c, h, w = x.shape
    kw = _lrelu(_conv(sd, pfx + ".kernel_conv.0", st_feat, 0), 0.1)                 # :226-228,319
    sta = sta_conv(x, kw)
    H, W = get_hw(h, w, scale)
    off, soff, r = satu_heads(sd, pfx, h, w, scale)
    ne = sd[pfx + ".weight_compress"].size(0)
    rw = r.view(ne, H * W).transpose(0, 1)                                           # :351
    wc = torch.matmul(rw, sd[pfx + ".weight_compress"].view(ne, -1)).view(1, H, W, c // 8, c)
    we = torch.matmul(rw, sd[pfx + ".weight_expand"].view(ne, -1)).view(1, H, W, c, c // 8)
    fea0 = satu_grid_sample(x, off, scale)
    fea = fea0.unsqueeze(-1).permute(0, 2, 3, 1, 4)
    fea = torch.matmul(wc.expand([b, -1, -1, -1, -1]), fea)
    fea = torch.matmul(we.expand([b, -1, -1, -1, -1]), fea).squeeze(-1)
    fea = fea.permute(0, 3, 1, 2) + fea0
    ss = satu_grid_sample(sta, soff, scale)
    return _conv(sd, pfx + ".fusion", torch.cat((ss, fea), dim=1), 0)


# ----------------------------------------------------------------------------- whole network (a2, a3, a16, a17)
def pad_spatial(x: Tensor, multiple: int = 2) -> Tensor:
    """savsr_arch.py:670-690: reflect-pad bottom/right to an even size."""
    n, t, c, h, w = x.shape
    ph = (multiple - h % multiple) % multiple
    pw = (multiple - w % multiple) % multiple
    x = F.pad(x.reshape(-1, c, h, w), [0, pw, 0, ph], mode="reflect")
    return x.view(n, t, c, h + ph, w + pw)


def frame_sample(frames: Tensor, num_frame: int, interval: int):
    """SAVSR.frame_sample, savsr_arch.py:638-659: (past -> future sub-sequence, future -> past sub-sequence) of [b, t, ...]; the
    method's own centre index is num_frame // 2 (:642), both sub-sequences hold that frame."""
    if interval == 0:
        return frames, frames
    c = num_frame // 2
    index = list(range(num_frame))
    if c % 2 == 0:
        fwd = index[1::interval + 1]                                               # :646-648
        fwd.insert(c // 2, c)
        bwd = index[::interval + 1]
    else:
        fwd = index[::interval + 1]                                                # :650-655
        fwd.insert(c // 2 + 1, c)
        bwd = index[1::interval + 1]
        if len(fwd) != len(bwd):
            bwd.append(fwd[-1])
            bwd.insert(0, fwd[0])
    return frames[:, fwd], frames[:, bwd]


def forward(sd: SD, lq: Tensor, scale: Sequence[float], cfg: dict | None = None,
            taps: dict | None = None) -> Tensor:
    """savsr_arch.py:692-742 (interval == 0 is what the test YAMLs use; frame sampling, :638-659, is restated too).

    `taps`, when given, is filled with named intermediate tensors for stage-level parity tests."""
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    nf, t = c["num_feat"], c["num_frame"]
    center = t // 2 if c["center_frame_idx"] is None else c["center_frame_idx"]
    sw, fw = c["slid_win"], c["fusion_win"]
    if c["interval"] == 0:                                                          # :597-604
        iter_win = t
    else:
        iter_win = center + 1 if center % 2 == 0 else center + 2
    b, _, _, h_in, w_in = lq.shape
    H, W = get_hw(h_in, w_in, scale)
    x_center = lq[:, center].contiguous()
    x = pad_spatial(lq)
    x_f, x_b = frame_sample(x, t, c["interval"])                                    # :699
    hp, wp = x.shape[-2:]
    ht_b = torch.zeros(b, nf, hp, wp)
    ht_f = torch.zeros(b, nf, hp, wp)
    lb, lf = [], []
    steps = iter_win - sw + 1
    for idx in range(steps):                                                        # :708-719
        cur = iter_win - 1 - sw // 2 - idx
        ht_b = window_unit_l1(sd, "f2p_win", x_b[:, cur - sw // 2: cur + sw // 2 + 1], ht_b, scale, c["w1_num_block"])
        lb.insert(0, ht_b)
        cur = idx + sw // 2
        ht_f = window_unit_l1(sd, "p2f_win", x_f[:, cur - sw // 2: cur + sw // 2 + 1], ht_f, scale, c["w1_num_block"])
        lf.append(ht_f)
    feats = [torch.cat([lb[i], lf[i]], dim=1) for i in range(steps)]               # :721
    n_l2 = (iter_win - fw + 1) // 2
    for i in range(n_l2):                                                           # :616-618,722
        feats = window_unit_l2(sd, f"h_win.{i}", feats, scale, win_size=steps - 2 * i,
                               slid_win=fw, num_block=c["w2_num_block"])
    hfeat = _lrelu(_conv(sd, "h_win_conv_h", feats[0], 1))                          # :723
    align = hfeat
    share = hfeat
    if taps is not None:
        taps["align_feat"] = align
    for i in range(c["n_resgroups"]):                                               # :728-732
        hfeat = residual_group(sd, f"RG.{i}", hfeat, c["n_resblocks"])
        hfeat = osadapt(sd, f"adapt.{i}", hfeat, scale)
        hfeat = hfeat + sd["gamma"] * share
    hfeat = _conv(sd, "conv_last", hfeat, 1) + share                                # :733-734
    if taps is not None:
        taps["h_feat"] = hfeat
    sr = sta_upsample(sd, "upsample", hfeat[..., :h_in, :w_in], scale, align[..., :h_in, :w_in])
    if taps is not None:
        taps["satu"] = sr
    sr = _conv(sd, "tail", sr, 1)                                                   # :738
    return sr + F.interpolate(x_center, size=(H, W), mode="bilinear", align_corners=False)   # :739
