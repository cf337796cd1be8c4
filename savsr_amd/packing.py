"""Weight images and SATU matrices for the HIP kernels, built once per engine (device-side scatter / split; float64 folds).

Host helpers (integer grids, index maps of the weight-image layouts, split-bf16 packing) and `WeightPacking`, the part of
`HipEngine` that turns a reference state_dict (791 keys, savsr_arch.py:576-636) into what the kernels read: BatchNorm folded into
the convs (:191-204), every conv as a split-bf16 image in MFMA lane order (direct and, for static 3x3 convs, Winograd-y), the OSConv
kernel banks + routing / attention matrices (:139-172), the SATU matrices with the tail conv's channel contraction folded in
(:315-376, :738).  Pure data movement + RNE conversions; include/savsr_hip.h documents every layout.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import SatuWeights

BN_EPS = 1e-5
MAX_SUM_BLOCKS = 256     # workgroups of one savsr_channel_sums launch


# ----------------------------------------------------------------------------- host helpers (integer / grid logic)
def get_hw(h: int, w: int, scale: Sequence[float]) -> Tuple[int, int]:
    """Output size, savsr_arch.py:745-751 (Python round = half-to-even on the double product)."""
    return round(h * scale[0]), round(w * scale[1])


def satu_axis_tables(n_out: int, n_in: int, s: float):
    """Per-axis SATU tables, evaluated in fp32 exactly like the reference's torch CPU ops.

    Returns (coor, floor_idx, grid_norm):
      coor      = (i+.5)/s - floor((i+.5)/s + 1e-3) - .5          savsr_arch.py:331-333
      floor_idx = floor((i+.5)/s + 1e-3)  (the integer LR index grid, bit-exact contract)
      grid_norm = ((i+.5)/s - .5) * 2 / (n_in-1) - 1              savsr_arch.py:270-280
    """
    f32 = np.float32
    i = np.arange(n_out, dtype=np.float32)
    q = (i + f32(0.5)) / f32(s)
    fl = np.floor(q + f32(1e-3))
    coor = (q - fl) - f32(0.5)
    g = (i + f32(0.5)) / f32(s) - f32(0.5)
    g = (g * f32(2)) / f32(n_in - 1) - f32(1)
    return coor.astype(np.float32), fl.astype(np.int32), g.astype(np.float32)


_PACK_IDX_CACHE: Dict[Tuple[int, int, int], Tuple[np.ndarray, int]] = {}

CONV_TH, CONV_TW = 8, 32      # pixel tile of one conv workgroup (mirrors common.hpp)


def conv_pack_geometry(cout: int, cin: int, ks: int):
    kc = 16 if ks == 3 else 32
    cot = 64 if cout > 32 else 32
    if cin % kc:
        raise ValueError(f"conv cin={cin} must be a multiple of {kc} (pad the weight with zero channels)")
    return kc, cot, cin // kc, (cout + cot - 1) // cot


def conv_pack_index(cout: int, cin: int, ks: int):
    """Index map [cout, cin, ks*ks] -> position inside one part of the weight image
    (mirror of savsr_conv_pack_index)."""
    key = (cout, cin, ks)
    if key not in _PACK_IDX_CACHE:
        kc, cot, nchunk, ncob = conv_pack_geometry(cout, cin, ks)
        taps, nt, ksteps = ks * ks, cot // 32, kc // 16
        co = np.arange(cout, dtype=np.int64)[:, None, None]
        ci = np.arange(cin, dtype=np.int64)[None, :, None]
        tap = np.arange(taps, dtype=np.int64)[None, None, :]
        cob, col = co // cot, co % cot
        t, row = col // 32, col % 32
        chunk, cl = ci // kc, ci % kc
        kstep, kh, j = cl // 16, (cl % 16) // 8, cl % 8
        group = (((cob * nchunk + chunk) * taps + tap) * ksteps + kstep) * nt + t
        idx = group * 512 + (kh * 32 + row) * 8 + j
        total = ncob * nchunk * taps * kc * cot
        _PACK_IDX_CACHE[key] = (np.array(np.broadcast_to(idx, (cout, cin, taps))).reshape(-1), total)      # (a writable copy: torch.from_numpy warns on read-only views)
    return _PACK_IDX_CACHE[key]


_IDX_DEV_CACHE: Dict[tuple, torch.Tensor] = {}


def _index_on(kind: str, key: tuple, idx: np.ndarray, device: torch.device) -> torch.Tensor:
    """The (cached) index map of a weight-image layout as a tensor on `device`."""
    k = (kind, key, str(device))
    t = _IDX_DEV_CACHE.get(k)
    if t is None:
        t = torch.from_numpy(idx).to(device)
        _IDX_DEV_CACHE[k] = t
    return t


def _scatter_image(idx: np.ndarray, total: int, values: torch.Tensor, kind: str, key: tuple, device: Optional[torch.device]) -> torch.Tensor:
    """zeros[total] with values scattered to idx: numpy on the host, one index_put on a GPU (round 5: the engine packs its ~190 conv
    images and 12 OSConv banks ON THE DEVICE -- 1.7 s of host scatter / split work per process became a few ms; a rank of an 8-GPU run of
    a YAML spends 2-7 s on the GPU in all).  Pure data movement + RNE conversions: bit-identical either way
    (tests/test_gpu_kernels.py::test_weight_images_packed_on_device_equal_host_packing)."""
    if device is None or device.type == "cpu":
        out = np.zeros(total, dtype=np.float32)
        out[idx] = values.detach().to("cpu", torch.float32).contiguous().numpy().reshape(-1)
        return torch.from_numpy(out)
    out = torch.zeros(total, dtype=torch.float32, device=device)
    out[_index_on(kind, key, idx, device)] = values.detach().to(device, torch.float32).reshape(-1)
    return out


def pack_conv_part(w: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    """[cout, cin, k, k] -> fp32 tensor of one image part (zero padded), lane order; on `device` (default: host)."""
    cout, cin, ks, _ = w.shape
    idx, total = conv_pack_index(cout, cin, ks)
    return _scatter_image(idx, total, w, "direct", (cout, cin, ks), device)


def split_bf16_image(part: torch.Tensor) -> torch.Tensor:
    """fp32 part [n*512] -> int16 image [n][2][512]: hi = bf16(v), lo = bf16(v - hi) (RNE both)."""
    hi = part.to(torch.bfloat16)
    lo = (part - hi.to(torch.float32)).to(torch.bfloat16)
    img = torch.stack([hi.view(-1, 512), lo.view(-1, 512)], dim=1).contiguous()
    return img.view(torch.int16).reshape(-1)


def pack_conv_weight(w: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    """[cout, cin, k, k] -> split-bf16 weight image (int16 tensor) for savsr_conv2d."""
    return split_bf16_image(pack_conv_part(w, device))


_WY_IDX_CACHE: Dict[tuple, tuple] = {}


def conv_wy_pack_index(cout: int, cin: int):
    """Index map [4 pos, cout, cin, 3 kx] -> position inside one part of the Winograd-y weight image (mirror of
    savsr_conv_wy_pack_index): [cob][chunk][hf][s = vr * 3 + kx][t] groups of 512 = (kh * 32 + row) * 8 + j."""
    key = (cout, cin)
    if key not in _WY_IDX_CACHE:
        if cout % 64 or cin % 16:
            raise ValueError("Winograd-y conv image: cout must be a multiple of 64 and cin of 16")
        nchunk = cin // 16
        pos = np.arange(4, dtype=np.int64)[:, None, None, None]
        co = np.arange(cout, dtype=np.int64)[None, :, None, None]
        ci = np.arange(cin, dtype=np.int64)[None, None, :, None]
        kx = np.arange(3, dtype=np.int64)[None, None, None, :]
        cob, col = co // 64, co % 64
        t, row = col // 32, col % 32
        chunk, cl = ci // 16, ci % 16
        kh, j = cl // 8, cl % 8
        hf, vr = pos // 2, pos % 2
        group = (((cob * nchunk + chunk) * 2 + hf) * 6 + (vr * 3 + kx)) * 2 + t
        idx = group * 512 + (kh * 32 + row) * 8 + j
        total = (cout // 64) * nchunk * 12 * 16 * 64
        _WY_IDX_CACHE[key] = (np.array(np.broadcast_to(idx, (4, cout, cin, 3))).reshape(-1), total)
    return _WY_IDX_CACHE[key]


def pack_conv_weight_wy(w: torch.Tensor, device: Optional[torch.device] = None) -> torch.Tensor:
    """[cout, cin, 3, 3] -> split-bf16 Winograd-y weight image (SAVSR_CONV_WINOGRAD_Y): the F(2,3) weight transform over the tap ROWS
    g_ky in float64 -- U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2, per kx -- rounded to fp32, then (hi, lo)."""
    cout, cin, ks, _ = w.shape
    assert ks == 3
    dev = device if device is not None and device.type != "cpu" else torch.device("cpu")
    g = w.detach().to(dev, torch.float64)                                  # [co][ci][ky][kx]
    g0, g1, g2 = g[:, :, 0], g[:, :, 1], g[:, :, 2]
    u = torch.stack([g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2], 0).to(torch.float32)      # [pos][co][ci][kx]
    idx, total = conv_wy_pack_index(cout, cin)
    return split_bf16_image(_scatter_image(idx, total, u, "wy", (cout, cin), device))


def acc_row(r: int, half: int) -> int:
    """Row of register r of a 32x32 MFMA accumulator for lane half `half`."""
    return (r & 3) + 8 * (r >> 2) + 4 * half


class WeightPacking:
    """Mixin of HipEngine: state_dict -> device-resident kernel operands (`pw`, `pw_wy`, `osc`, `se`, `satu_*`, `tail_*`)."""

    def _dev(self, t: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        d = t.to(self.dev, dtype).contiguous()
        self._keep.append(d)
        return d

    def _fold(self, sd, key: str, bn: Optional[str]):
        w = sd[key + ".weight"].to("cpu", torch.float32)
        b = sd.get(key + ".bias")
        b = None if b is None else b.to("cpu", torch.float32)
        if bn is not None:      # eval BatchNorm folded into the conv (savsr_arch.py:191,196,199,204)
            s = sd[bn + ".weight"].cpu() / torch.sqrt(sd[bn + ".running_var"].cpu() + BN_EPS)
            w = w * s.view(-1, 1, 1, 1)
            b0 = b if b is not None else torch.zeros_like(s)
            b = (b0 - sd[bn + ".running_mean"].cpu()) * s + sd[bn + ".bias"].cpu()
        return w, b

    def _register(self, key: str, w: torch.Tensor, b: Optional[torch.Tensor]):
        cout, cin, ks, _ = w.shape
        bias = None if b is None else self._dev(b)
        wd = w.to(self.dev)                                  # (the images are built on the device: _scatter_image)
        self.pw[key] = (self._dev(pack_conv_weight(wd, self.dev), torch.int16), bias, cout, cin, ks)
        if self.conv_wy and ks == 3 and cout % 64 == 0 and cin % 16 == 0:
            # static 3x3 weights also as the Winograd F(2,3)-along-y image (conv_wy.hip: 2/3 of the matrix work); which form a launch takes is
            # decided per launch in conv_launch (the 16-row Winograd tiles need a launch that fills the chip)
            self.pw_wy[key] = self._dev(pack_conv_weight_wy(wd, self.dev), torch.int16)

    def _add_conv(self, sd, key: str, bn: Optional[str] = None):
        w, b = self._fold(sd, key, bn)
        self._register(key, w, b)

    def _add_window_conv(self, sd, d: str):
        """conv_c (3->64) and conv_sup (6->64) of one direction fused into a 16 -> 128 conv over
        the packed window tensor (channels: frame t | t-1 | t+1 | zeros), savsr_arch.py:429-431,456-457."""
        nf = self.nf
        wc, bc = sd[d + ".conv_c.weight"].cpu().float(), sd[d + ".conv_c.bias"].cpu().float()
        ws, bs = sd[d + ".conv_sup.weight"].cpu().float(), sd[d + ".conv_sup.bias"].cpu().float()
        w = torch.zeros(2 * nf, 16, 3, 3)
        w[:nf, 0:3] = wc
        w[nf:, 3:9] = ws
        self._register(d + ".win", w, torch.cat([bc, bs]))

    def _add_osconv(self, sd, key: str):
        bank = sd[key + ".weight"].to(self.dev, torch.float32)    # [K, cout, cin, 3, 3]
        knum, cout, cin = bank.shape[:3]
        packed = torch.stack([pack_conv_part(bank[k], self.dev) for k in range(knum)], 0)
        a = key + ".attention"
        bn_s = sd[a + ".bn.weight"].cpu() / torch.sqrt(sd[a + ".bn.running_var"].cpu() + BN_EPS)
        bn_b = sd[a + ".bn.bias"].cpu() - sd[a + ".bn.running_mean"].cpu() * bn_s
        hidden = sd[a + ".fc.weight"].shape[0]
        g = lambda k: self._dev(sd[k].reshape(sd[k].shape[0], -1) if sd[k].dim() > 1 else sd[k])
        elems = packed.shape[1]
        ent = dict(cin=cin, cout=cout, knum=knum, hidden=hidden, bank=self._dev(packed), nunits=elems // 8,
                   l1_w=g(key + ".scale_routing.0.weight"), l1_b=g(key + ".scale_routing.0.bias"),
                   l2_w=g(key + ".scale_routing.2.weight"), l2_b=g(key + ".scale_routing.2.bias"),
                   fc_w=g(a + ".fc.weight"), bn_scale=self._dev(bn_s), bn_shift=self._dev(bn_b),
                   ch_w=g(a + ".channel_fc.weight"), ch_b=g(a + ".channel_fc.bias"),
                   fl_w=g(a + ".filter_fc.weight"), fl_b=g(a + ".filter_fc.bias"),
                   sp_w=g(a + ".spatial_fc.weight"), sp_b=g(a + ".spatial_fc.bias"),
                   kn_w=g(a + ".kernel_fc.weight"), kn_b=g(a + ".kernel_fc.bias"),
                   **self._osc_scratch(cin, cout, knum, elems))
        self.osc[key] = ent

    def _osc_scratch(self, cin: int, cout: int, knum: int, elems: int) -> dict:
        """Per-engine scratch of one OSConv (routing vectors, gates, the generated weight images), NB_MAX copies: one per clip of a batched
        launch sequence (the tensors handed around are clip 0's; `_bstride` knows the distance to the next)."""
        nb = self.NB_MAX
        al = lambda n, unit: ((n * unit + 255) // 256) * 256 // unit          # copies stay 256-byte aligned
        out = {}
        for name, n, dt in (("v1", 2 * cin, torch.float32), ("v2", cin, torch.float32), ("att", cin + cout + 9 + knum, torch.float32),
                            ("wdyn", 2 * elems, torch.int16), ("wdyn_wy", 2 * (elems * 4 // 3) if cout % 64 == 0 else 0, torch.int16)):      # (12 taps instead of 9)
            unit = 4 if dt == torch.float32 else 2
            pitch = al(n, unit)
            full = torch.empty(nb * pitch, device=self.dev, dtype=dt)
            self._keep.append(full)
            t = full[:n]
            self._bstride[t.data_ptr()] = pitch * unit
            out[name] = t
        return out

    def _pack_satu(self, sd):
        p = "upsample."
        c = self.nf
        f32 = torch.float32
        wk = sd[p + "kernel_conv.0.weight"].to("cpu", f32).reshape(25 * c, c).numpy()     # [n = 25 ch + tap][k]
        bk = sd[p + "kernel_conv.0.bias"].to("cpu", f32).numpy()
        lane = np.arange(64)
        li, lh = lane & 31, lane >> 5
        jj = np.arange(8)
        # kconv part [tap][cg][ks][lane][j] = Wk[25 (32 cg + (lane & 31)) + tap][16 ks + 8 (lane >> 5) + j]
        tap = np.arange(25)[:, None, None, None, None]
        cg = np.arange(2)[None, :, None, None, None]
        ks = np.arange(4)[None, None, :, None, None]
        n_idx = 25 * (32 * cg + li[None, None, None, :, None]) + tap
        k_idx = 16 * ks + 8 * lh[None, None, None, :, None] + jj[None, None, None, None, :]
        n_idx, k_idx = np.broadcast_arrays(n_idx, k_idx)
        kconv = wk[n_idx, k_idx].astype(np.float32)                                  # [25,2,4,64,8]
        kconv_b = bk.reshape(c, 25).T.copy()                                          # [tap][ch]
        fus = sd[p + "fusion.weight"].to("cpu", f32).reshape(c, 2 * c).numpy()
        wa, wb = fus[:, :c], fus[:, c:]                                              # cat((sta, fea)), :374
        comp = sd[p + "weight_compress"].to("cpu", f32).reshape(4, 8, c).numpy()     # C_m[j][c]
        expd = sd[p + "weight_expand"].to("cpu", f32).reshape(4, c, 8).numpy()       # E_n[c][j]
        # projections, one 512-element group per (matrix tile, k step): [lane][j]
        pa = np.zeros((2, 4, 64, 8), dtype=np.float32)
        pb = np.zeros((2, 4, 64, 8), dtype=np.float32)
        pc = np.zeros((4, 64, 8), dtype=np.float32)
        for t in range(2):
            for kidx in range(4):
                cgi, s = kidx // 2, kidx % 2
                # k order of an accumulator used as B operand: row 16 s + 8 (j >> 2) + 4 half + (j & 3)
                ch = 32 * cgi + 16 * s + 8 * (jj[None, :] >> 2) + 4 * lh[:, None] + (jj[None, :] & 3)
                pa[t, kidx] = wa[(32 * t + li)[:, None], ch]
            for ksi in range(4):
                pb[t, ksi] = wb[(32 * t + li)[:, None], 16 * ksi + 8 * lh[:, None] + jj[None, :]]
        cstack = comp.reshape(32, c)                                                  # row 8 m + j (natural order in the record)
        for ksi in range(4):
            pc[ksi] = cstack[li[:, None], 16 * ksi + 8 * lh[:, None] + jj[None, :]]
        proj = np.concatenate([pa.reshape(-1), pb.reshape(-1), pc.reshape(-1)])
        wbe = np.einsum("oc,ncj->noj", wb.astype(np.float64), expd.astype(np.float64)).astype(np.float32)   # (Wb E_n)[co][j]
        wbe_p = np.zeros((2, 2, 64, 8), dtype=np.float32)                            # [t][ks][lane][j], k = 16 ks + 8 kh + j = 8 n + j
        for t in range(2):
            for ksi in range(2):
                wbe_p[t, ksi] = wbe[(2 * ksi + lh)[:, None], (32 * t + li)[:, None], jj[None, :]]
        fb = sd[p + "fusion.bias"].to("cpu", f32).numpy()
        fb_p = np.zeros((2, 32), dtype=np.float32)
        for hh in range(2):
            for t in range(2):
                for r in range(16):
                    fb_p[hh, 16 * t + r] = fb[32 * t + acc_row(r, hh)]
        head_w = torch.cat([sd[p + "routing.0.weight"], sd[p + "offset.weight"], sd[p + "st_offset.weight"]], 0)
        head_b = torch.cat([sd[p + "routing.0.bias"], sd[p + "offset.bias"], sd[p + "st_offset.bias"]], 0)
        t_ = lambda a: self._dev(torch.from_numpy(np.ascontiguousarray(a)))
        img = lambda a: self._dev(split_bf16_image(torch.from_numpy(np.ascontiguousarray(a.reshape(-1)))), torch.int16)
        self.satu_t = dict(
            body0_w=self._dev(sd[p + "body.0.weight"].reshape(64, 4)), body0_b=self._dev(sd[p + "body.0.bias"]),
            body2_w=self._dev(sd[p + "body.2.weight"].reshape(64, 64).t()), body2_b=self._dev(sd[p + "body.2.bias"]),
            head_w=self._dev(head_w.reshape(8, 64)), head_b=self._dev(head_b),
            kconv_w=img(kconv), kconv_b=t_(kconv_b), proj_w=img(proj), wbe_w=img(wbe_p), fusion_b=t_(fb_p))
        sw = SatuWeights()
        for k, v in self.satu_t.items():
            setattr(sw, k, v.data_ptr())
        self.satu_w = sw
        self.tail_w = self._dev(sd["tail.weight"].reshape(3, 64 * 9))
        self.tail_b = self._dev(sd["tail.bias"])
        # ---- tail-projected form (include/savsr_hip.h, savsr_satu_*_tail): the 3x3 tail conv's channel contraction
        # Wt27[p][c] (rows 27..31 zero) multiplied into fusion / expand / the LR projections in float64.  Two row orders:
        # p = 3 (3 ky + kx) + o (savsr_satu_hr_tail + savsr_tail_gather), and the row-summed form's (savsr_satu_hr_tail_q: the three kx
        # of group g = 3 ky + o at MFMA rows acc_row(3 gi + kx, half), groups 0 .. 4 in lane half 0, 5 .. 8 in half 1)
        tw = sd["tail.weight"].to("cpu", torch.float64).numpy()                          # [3 o][64 c][3 ky][3 kx]

        def fold(row_of):
            wt27 = np.zeros((32, c), dtype=np.float64)
            for ky in range(3):
                for kx in range(3):
                    for o in range(3):
                        wt27[row_of(ky, kx, o)] = tw[o, :, ky, kx]
            ta = (wt27 @ wa.astype(np.float64)).astype(np.float32)                           # [32][64] applies to sta
            tb = (wt27 @ wb.astype(np.float64)).astype(np.float32)                           # [32][64] applies to x
            pa1 = np.zeros((1, 4, 64, 8), dtype=np.float32)
            pb1 = np.zeros((1, 4, 64, 8), dtype=np.float32)
            for kidx in range(4):
                cgi, s_ = kidx // 2, kidx % 2
                ch = 32 * cgi + 16 * s_ + 8 * (jj[None, :] >> 2) + 4 * lh[:, None] + (jj[None, :] & 3)
                pa1[0, kidx] = ta[li[:, None], ch]
                pb1[0, kidx] = tb[li[:, None], 16 * kidx + 8 * lh[:, None] + jj[None, :]]
            proj1 = np.concatenate([pa1.reshape(-1), pb1.reshape(-1), pc.reshape(-1)])
            twbe = np.einsum("pc,ncj->npj", wt27 @ wb.astype(np.float64), expd.astype(np.float64)).astype(np.float32)   # (Wt27 Wb E_n)[p][j]
            twbe_p = np.zeros((1, 2, 64, 8), dtype=np.float32)
            for ksi in range(2):
                twbe_p[0, ksi] = twbe[(2 * ksi + lh)[:, None], li[:, None], jj[None, :]]
            tfb = (wt27 @ fb.astype(np.float64)).astype(np.float32)
            tfb_p = np.zeros((2, 16), dtype=np.float32)
            for hh in range(2):
                for r in range(16):
                    tfb_p[hh, r] = tfb[acc_row(r, hh)]
            tens = dict(proj_w=img(proj1), wbe_w=img(twbe_p), fusion_b=t_(tfb_p))
            swt = SatuWeights()
            for k, v in self.satu_t.items():
                setattr(swt, k, v.data_ptr())
            for k, v in tens.items():
                setattr(swt, k, v.data_ptr())
            return tens, swt

        def row_q(ky, kx, o):
            g = 3 * ky + o
            return acc_row(3 * g + kx, 0) if g < 5 else acc_row(3 * (g - 5) + kx, 1)
        self.satu_tail_t, self.satu_w_tail = fold(lambda ky, kx, o: 3 * (3 * ky + kx) + o)
        self.satu_tailq_t, self.satu_w_tailq = fold(row_q)

    def _pack_all(self, sd):
        cfg = self.cfg
        for d in ("f2p_win", "p2f_win"):
            self._add_window_conv(sd, d)
            for k in range(cfg["w1_num_block"]):
                b = f"{d}.blocks.{k}"
                for i in range(3):
                    self._add_conv(sd, f"{b}.conv0.{i}")
                    self._add_conv(sd, f"{b}.conv2.{i}")
                if k >= 1:
                    self._add_osconv(sd, b + ".osconv")
                else:
                    self._add_conv(sd, b + ".conv1")
            self._add_conv(sd, d + ".merge")
        from .archs.savsr_arch import frame_sample_indices, iteration_window
        center = cfg["num_frame"] // 2 if cfg["center_frame_idx"] is None else cfg["center_frame_idx"]
        self.iter_win = iteration_window(cfg["num_frame"], cfg["interval"], center)      # frames per propagation direction (:597-604)
        self.fwd_idx, self.bwd_idx = frame_sample_indices(cfg["num_frame"], cfg["interval"])   # frame_sample (:638-659)
        if cfg["interval"] != 0 and (len(self.fwd_idx) < self.iter_win or len(self.bwd_idx) < self.iter_win):
            raise ValueError("num_frame / interval: the sampled frame lists are shorter than the iteration window")
        steps = self.iter_win - cfg["slid_win"] + 1
        self.n_l2 = (self.iter_win - cfg["fusion_win"] + 1) // 2
        for i in range(self.n_l2):
            u = f"h_win.{i}"
            for j in range(steps - 2 * i):
                self._add_conv(sd, f"{u}.conv_h.{j}")
            for k in range(cfg["w2_num_block"]):
                b = f"{u}.blocks.{k}"
                for j in range(cfg["fusion_win"]):
                    self._add_conv(sd, f"{b}.conv0.{j}")
                    self._add_conv(sd, f"{b}.conv2.{j}")
                self._add_osconv(sd, b + ".osconv")
            self._add_conv(sd, u + ".merge")
        self._add_conv(sd, "h_win_conv_h")
        for g in range(cfg["n_resgroups"]):
            for k in range(cfg["n_resblocks"]):
                r = f"RG.{g}.residual_group.{k}.rcab"
                self._add_conv(sd, r + ".0")
                self._add_conv(sd, r + ".2")
                a = r + ".3.attention"
                cm = sd[a + ".1.weight"].shape[0]
                self.se[r] = (self._dev(sd[a + ".1.weight"].reshape(cm, -1)), self._dev(sd[a + ".1.bias"]),
                              self._dev(sd[a + ".3.weight"].reshape(-1, cm)), self._dev(sd[a + ".3.bias"]), cm)
            self._add_conv(sd, f"RG.{g}.conv")
            m = f"adapt.{g}.mask"
            self._add_conv(sd, m + ".0", bn=m + ".1")
            self._add_conv(sd, m + ".4", bn=m + ".5")
            self._add_conv(sd, m + ".7", bn=m + ".8")
            self._add_conv(sd, m + ".11", bn=m + ".12")
            self._add_osconv(sd, f"adapt.{g}.adapt")
        self._add_conv(sd, "conv_last")
        self.gamma = float(sd["gamma"].reshape(-1)[0])
        self._pack_satu(sd)
        self.se_gate = torch.empty(self.nf, device=self.dev)
