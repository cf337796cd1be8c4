"""CPU-side checks: the C-ABI library loads and exports every symbol of include/savsr_hip.h,
host-side integer/grid logic is bit-exact against the reference goldens, registry / build_network /
state_dict surface match the reference contract, the product refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

import savsr_amd
from savsr_amd import _lib, engine as E
from savsr_amd.utils import synth
from tests.golden_cases import GRID_SIZES, YAML_SCALES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "savsr_hip.h")).read()
    # the section under SAVSR_DIAG is compiled into the instrumented library only (libsavsr_hip_diag.so)
    diag_sec = re.search(r"#ifdef SAVSR_DIAG\n(.*?)#endif /\* SAVSR_DIAG \*/", hdr, flags=re.S)
    assert diag_sec, "diagnostics section not found"
    diag = set(re.findall(r"\b(savsr_[a-z0-9_]+)\s*\(", diag_sec.group(1)))
    product_hdr = hdr.replace(diag_sec.group(0), "")
    product_hdr = re.sub(r"/\*.*?\*/", "", product_hdr, flags=re.S)                      # comments may mention retired names
    declared = set(re.findall(r"\b(savsr_[a-z0-9_]+)\s*\(", product_hdr))
    assert declared, "header parse failed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert diag == set(_lib.DIAG_SIGNATURES), diag ^ set(_lib.DIAG_SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    if not os.environ.get("SAVSR_LIB_PATH"):
        for name in diag:                                   # no diagnostic entry point (and no global switch) in the product library
            assert not hasattr(lib, name), name
    assert lib.savsr_abi_version() == _lib.ABI_VERSION
    assert b"gfx950" in lib.savsr_version()


def test_header_constants_match_the_binding():
    """The enumerations the ctypes binding carries are the header's (`#define SAVSR_...`): conv forms, ABI number."""
    hdr = open(os.path.join(ROOT, "include", "savsr_hip.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define\s+(SAVSR_[A-Z0-9_]+)\s+(-?\d+)\b", hdr)}
    assert defs["SAVSR_ABI_VERSION"] == _lib.ABI_VERSION
    for name in ("CONV_DIRECT", "CONV_DIRECT_THROUGHPUT", "CONV_WINOGRAD_Y", "CONV_WINOGRAD_Y_THROUGHPUT"):
        assert defs["SAVSR_" + name] == getattr(_lib, name), name
    assert set(_lib.CONV_WY_FORMS) == {defs["SAVSR_CONV_WINOGRAD_Y"], defs["SAVSR_CONV_WINOGRAD_Y_THROUGHPUT"]}
    for name, val in defs.items():                      # activations, where the binding names them
        if name.startswith("SAVSR_ACT_") and hasattr(_lib, name[6:]):
            assert getattr(_lib, name[6:]) == val, name


def test_winograd_y_strip_tile_plan():
    """savsr_conv_wy_tile_count: the launcher's tile plan for the image's last rows (conv_mfma.hip `wy_tile_plan`).  Full tiles = 16 rows x 32 px;
    when h % 16 leaves <= 8 rows they can go as strips of 2^l row pairs x (8 >> l) segments.  WINOGRAD_Y takes strips when they save the grid of
    256 persistent workgroups a round of tiles, WINOGRAD_Y_THROUGHPUT also whenever <= 2 row pairs are left."""
    lib = _lib.load()
    WY, TP = _lib.CONV_WINOGRAD_Y, _lib.CONV_WINOGRAD_Y_THROUGHPUT

    def plan(h, w, nconv, algo, cout=64):
        ntx, nty, ncob = -(-w // 32), -(-h // 16), cout // 64
        full = ntx * nty
        left = h % 16
        pairs = (left + 1) // 2
        tiles = full
        if 0 < left <= 8:
            l2 = 0 if pairs <= 1 else (1 if pairs <= 2 else 2)
            strips = ntx * (nty - 1) + -(-ntx // (8 >> l2))
            per = nconv * ncob
            if (pairs <= 2 and algo == TP) or -(-per * strips // 256) < -(-per * full // 256):
                tiles = strips
        return nconv * ncob * tiles

    assert lib.savsr_conv_wy_tile_count(180, 320, 64, 1, WY) == 120            # alone: one round either way -> full tiles
    assert lib.savsr_conv_wy_tile_count(180, 320, 64, 1, TP) == 113            # 11 x 10 full tiles + 3 strips of 2 row pairs x 4 segments
    assert lib.savsr_conv_wy_tile_count(180, 320, 64, 24, WY) == 24 * 113      # 12 -> 11 rounds
    assert lib.savsr_conv_wy_tile_count(180, 320, 64, 6, WY) == 6 * 120        # 3 rounds both
    assert lib.savsr_conv_wy_tile_count(120, 180, 64, 24, TP) == 24 * 48       # 4 row pairs left, 5 rounds both: full tiles
    assert lib.savsr_conv_wy_tile_count(24, 1280, 64, 4, WY) == 4 * 60         # 320 -> 240 tiles: 2 rounds -> 1
    assert lib.savsr_conv_wy_tile_count(4, 70, 64, 1, TP) == 1                 # strips only: 3 segments in one workgroup
    assert lib.savsr_conv_wy_tile_count(16, 32, 128, 2, TP) == 4               # no rows left; two channel blocks
    for bad in ((0, 32, 64, 1, WY), (16, 32, 32, 1, WY), (16, 32, 64, 25, WY), (16, 32, 64, 1, _lib.CONV_DIRECT)):
        assert lib.savsr_conv_wy_tile_count(*bad) == -1
    rng = np.random.RandomState(5)
    for _ in range(2000):
        h, w, n = int(rng.randint(1, 600)), int(rng.randint(1, 1400)), int(rng.randint(1, 25))
        cout = 64 * int(rng.randint(1, 3))
        for algo in (WY, TP):
            got = lib.savsr_conv_wy_tile_count(h, w, cout, n, algo)
            assert got == plan(h, w, n, algo, cout), (h, w, n, cout, algo, got)
            assert got <= n * (cout // 64) * -(-w // 32) * -(-h // 16)           # never more tiles than the full form


def test_every_entry_point_has_a_row_in_integration_md():
    """INTEGRATION.md shows the reference-side binding: every product entry point of the header is named there beside what it replaces."""
    hdr = open(os.path.join(ROOT, "include", "savsr_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = re.sub(r"#ifdef SAVSR_DIAG\n(.*?)#endif /\* SAVSR_DIAG \*/", "", hdr, flags=re.S)
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(savsr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 40
    missing = sorted(n for n in declared if n not in doc)
    assert not missing, missing


def test_pack_index_matches_c_abi():
    lib = _lib.load()
    rng = np.random.RandomState(0)
    for (co, ci, ks) in [(64, 64, 3), (64, 192, 3), (128, 320, 3), (16, 64, 3), (1, 16, 3), (64, 192, 1), (128, 16, 3)]:
        idx, total = E.conv_pack_index(co, ci, ks)
        assert total == lib.savsr_conv_packed_elems(co, ci, ks)
        assert len(np.unique(idx)) == len(idx) and idx.max() < total
        view = idx.reshape(co, ci, ks * ks)
        for _ in range(40):
            a, b, c = rng.randint(co), rng.randint(ci), rng.randint(ks * ks)
            assert view[a, b, c] == lib.savsr_conv_pack_index(co, ci, ks, a, b, c)


def test_winograd_y_pack_index_and_transform():
    """The Winograd-y weight image (SAVSR_CONV_WINOGRAD_Y): the host packer's index map mirrors savsr_conv_wy_pack_index, and the
    packed U satisfy F(2,3)'s identity -- for any four input rows d, (d0 - d2) U0 + (d1 + d2) U1 + (d2 - d1) U2 = d0 g0 + d1 g1 + d2 g2 and
    (d1 + d2) U1 - (d2 - d1) U2 - (d1 - d3) U3 = d1 g0 + d2 g1 + d3 g2 -- to the split-bf16 precision."""
    lib = _lib.load()
    rng = np.random.RandomState(1)
    for (co, ci) in [(64, 64), (64, 192), (128, 16), (128, 320)]:
        idx, total = E.conv_wy_pack_index(co, ci)
        assert total == lib.savsr_conv_wy_packed_elems(co, ci)
        assert len(np.unique(idx)) == len(idx) and idx.max() < total and len(idx) == total        # 12 taps, no padding
        view = idx.reshape(4, co, ci, 3)
        for _ in range(40):
            a, b, c, e = rng.randint(4), rng.randint(co), rng.randint(ci), rng.randint(3)
            assert view[a, b, c, e] == lib.savsr_conv_wy_pack_index(co, ci, b, c, a, e)
    assert lib.savsr_conv_wy_packed_elems(32, 64) == -1 and lib.savsr_conv_wy_packed_elems(64, 24) == -1
    w = torch.from_numpy(rng.standard_normal((64, 32, 3, 3)).astype(np.float32))
    img = E.pack_conv_weight_wy(w).view(torch.bfloat16).view(-1, 2, 512).double()
    idx, total = E.conv_wy_pack_index(64, 32)
    u = (img[:, 0] + img[:, 1]).reshape(-1)[torch.from_numpy(idx)].reshape(4, 64, 32, 3)              # [pos][co][ci][kx]
    g = w.double()
    d = torch.from_numpy(rng.standard_normal(4))
    y0 = (d[0] - d[2]) * u[0] + (d[1] + d[2]) * u[1] + (d[2] - d[1]) * u[2]
    y1 = (d[1] + d[2]) * u[1] - (d[2] - d[1]) * u[2] - (d[1] - d[3]) * u[3]
    r0 = d[0] * g[:, :, 0] + d[1] * g[:, :, 1] + d[2] * g[:, :, 2]
    r1 = d[1] * g[:, :, 0] + d[2] * g[:, :, 1] + d[3] * g[:, :, 2]
    assert float((y0 - r0).abs().max()) < 1e-4 and float((y1 - r1).abs().max()) < 1e-4


def test_invalid_arguments_are_rejected_without_gpu():
    lib = _lib.load()
    assert lib.savsr_conv2d(None, None) == -1
    assert b"null" in lib.savsr_last_error()
    assert lib.savsr_avgpool2(1, 2, 4, 3, 4, None) == -1      # odd height
    assert lib.savsr_conv_packed_elems(64, 64, 5) == -1 and lib.savsr_conv_packed_elems(64, 24, 3) == -1


def test_split_bf16_image_roundtrip():
    """hi + lo reproduces fp32 weights to ~2^-17 relative; image interleaves parts per 512-element group."""
    w = torch.from_numpy(np.random.RandomState(0).standard_normal((64, 32, 3, 3)).astype(np.float32))
    part = E.pack_conv_part(w)
    img = E.pack_conv_weight(w).view(torch.bfloat16).view(-1, 2, 512).float()
    rec = (img[:, 0] + img[:, 1]).reshape(-1)
    assert float(((rec - part).abs() / part.abs().clamp_min(1e-20)).max()) < 2.0 ** -15
    idx, total = E.conv_pack_index(64, 32, 3)
    assert torch.equal(part[torch.from_numpy(idx)], w.reshape(-1)) and total == part.numel()


def test_integer_grids_bit_exact_vs_reference(golden):
    """get_HW (a1) + the floor term of the SATU coordinates (a12), product host code vs reference."""
    for sc in YAML_SCALES:
        for (h, w) in GRID_SIZES:
            key = f"grid/{sc[0]}_{sc[1]}/{h}x{w}"
            H, W = E.get_hw(h, w, sc)
            assert [H, W] == golden[key + "/HW"].tolist()
            assert np.array_equal(E.satu_axis_tables(H, h, sc[0])[1].astype(np.int16), golden[key + "/fh"])
            assert np.array_equal(E.satu_axis_tables(W, w, sc[1])[1].astype(np.int16), golden[key + "/fw"])


def test_axis_tables_match_oracle_bitwise():
    from oracle import savsr_oracle as O
    for sc in [(4, 4), (1.5, 4), (3.9, 3.9), (2.95, 3.75), (1.1, 1.1)]:
        h, w = 45, 64
        H, W, ch, cw, _, _ = O.satu_coords(h, w, sc)
        a = E.satu_axis_tables(H, h, sc[0])
        b = E.satu_axis_tables(W, w, sc[1])
        assert np.array_equal(a[0], ch.numpy()) and np.array_equal(b[0], cw.numpy())
        # normalised base grid of grid_sample (savsr_arch.py:270-280)
        g = torch.Tensor(np.arange(W, dtype=np.float64))
        g = (g + 0.5) / sc[1] - 0.5
        g = g * 2 / (w - 1) - 1
        assert np.array_equal(b[2], g.numpy())


def test_axis_tables_match_oracle_bitwise_on_every_listed_scale():
    """The same bit-for-bit comparison (fp32 coordinate features, integer LR grid, output size) on every scale pair the reference's lists hold,
    at the sizes they are used on: the 42 YAML pairs at LR 180x320 and 144x180 (Vid4's two LR sizes at x4), the 60 training pairs at the LR
    size their as-mod-crop of the 256x448 GT gives (the integer coordinate grid is bit-exact)."""
    from oracle import savsr_oracle as O
    from savsr_amd.utils import workloads as WL
    cases = [(h, w, sc) for sc in WL.YAML_SCALES for (h, w) in ((180, 320), (144, 180))]
    cases += [WL.lr_shape(WL.VIMEO_GT, sc) + (sc,) for sc in WL.TRAIN_SCALES]
    assert len(cases) == 2 * 42 + 60
    for h, w, sc in cases:
        H, W, ch, cw, fh, fw = O.satu_coords(h, w, sc)
        assert (H, W) == E.get_hw(h, w, sc), (h, w, sc)
        a = E.satu_axis_tables(H, h, sc[0])
        b = E.satu_axis_tables(W, w, sc[1])
        assert np.array_equal(a[0], ch.numpy()) and np.array_equal(b[0], cw.numpy()), (h, w, sc)
        assert np.array_equal(a[1], np.asarray(fh).reshape(-1)) and np.array_equal(b[1], np.asarray(fw).reshape(-1)), (h, w, sc)


def test_registry_and_state_dict_surface():
    net = savsr_amd.build_network(dict(type="SAVSR", num_in_ch=3, num_feat=64, num_frame=7, slid_win=3, fusion_win=5,
                                       interval=0, w1_num_block=4, w2_num_block=2, n_resgroups=4, n_resblocks=8,
                                       center_frame_idx=None))
    assert type(net).__name__ == "SAVSR" and "SAVSR" in savsr_amd.ARCH_REGISTRY
    assert synth.manifest_of(net.state_dict()) == synth.load_manifest()        # 791 keys, shapes, order
    assert sum(p.numel() for p in net.parameters()) == 18890044
    net.load_state_dict(synth.synth_state_dict(), strict=True)
    net.set_scale(3)
    assert net.scale == (3, 3)
    with pytest.raises(KeyError):
        savsr_amd.ARCH_REGISTRY.get("NoSuchArch")
    assert "OSConv2d" in str(net)


def test_no_cpu_fallback():
    net = savsr_amd.SAVSR().eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 7, 3, 8, 8))
    with pytest.raises(RuntimeError, match="inference path only"):
        savsr_amd.SAVSR().train()(torch.zeros(1, 7, 3, 8, 8))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "savsr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f


def test_resize_tables_match_torch_cpu():
    """savsr_amd.resize_gpu.aa_tables (ATen's _compute_indices_weights_aa restated in fp32) applied on the CPU with numpy
    reproduces torch's F.interpolate(bicubic, antialias=True) -- the op behind the reference's T.Resize
    (lbasicsr/data/data_util.py:409-410)."""
    import numpy as np
    import torch.nn.functional as F
    from savsr_amd.resize_gpu import aa_tables

    def axis(a, xmin, xsize, wt, ax):
        a = np.moveaxis(a, ax, -1)
        out = np.zeros(a.shape[:-1] + (len(xmin),), np.float32)
        for i in range(len(xmin)):
            acc = np.zeros(a.shape[:-1], np.float32)
            for j in range(xsize[i]):
                acc = acc + a[..., xmin[i] + j] * wt[i, j]
            out[..., i] = acc
        return np.moveaxis(out, -1, ax)

    for h, w, oh, ow in [(64, 80, 16, 20), (37, 53, 10, 35), (30, 30, 40, 45)]:
        x = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(h))
        ref = F.interpolate(x, size=(oh, ow), mode="bicubic", align_corners=False, antialias=True).numpy()
        got = axis(axis(x.numpy(), *aa_tables(w, ow), 3), *aa_tables(h, oh), 2)
        assert float(np.abs(got - ref).max()) < 1e-6


def test_cal_step_and_as_mod_crop():
    """lbasicsr/data/transforms.py:31-69: known answers for the YAML scale list."""
    from savsr_amd.resize_gpu import as_mod_crop_hw, cal_step
    assert [cal_step(s) for s in (4, 3.5, 1.2, 1.1, 2.95, 3.75, 1.62)] == [1, 2, 5, 10, 20, 20, 50]
    assert as_mod_crop_hw(720, 1280, (4, 4)) == (720, 1280)
    assert as_mod_crop_hw(720, 1272, (1.5, 4)) == (720, 1272)              # UDM10, SURVEY 8(d) config 4
    assert as_mod_crop_hw(720, 1272, (3.5, 2)) == (714, 1272)
    assert as_mod_crop_hw(576, 720, (3.7, 3.7)) == (555, 703)              # floor(576/10/3.7)*10*3.7, floor(720/10/3.7)*10*3.7
    import pytest
    with pytest.raises(ValueError):
        cal_step(1.333)
    # golden vectors from the reference function itself (tools/gen_golden_modcrop.py): every YAML scale x 5 GT sizes
    import json
    import os
    rows = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "as_mod_crop.json")))
    assert len(rows) >= 200
    for sc, h, w, oh, ow, st_h, st_w in rows:
        assert as_mod_crop_hw(h, w, tuple(sc)) == (oh, ow), (sc, h, w)
        assert (cal_step(sc[0]), cal_step(sc[1])) == (st_h, st_w)



def test_bench_gpus_flag_fails_loudly_without_the_gpus():
    """`bench.py --gpus N` must never run a smaller job under the N-GPU label: on a node with fewer GPUs (none here) it refuses
    before touching a device, and a launcher environment whose WORLD_SIZE disagrees with --gpus is refused too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    if "node shows" in r.stderr:                  # fewer than 2 GPUs visible (the build container, a 1-GPU box)
        assert r.returncode == 2 and r.stdout.strip() == ""
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


def test_check_readme_logic():
    """The real-data checker (python -m savsr_amd.test --check-readme) on synthetic result tables: README entries are found for
    the YAML's dataset names / scales, the tolerance is 0.01 dB / 1e-4 plus half a unit of the published rounding, a dataset
    without an entry is not a pass."""
    import json
    from savsr_amd import test as T
    table = json.load(open(T.README_TABLE))
    assert len(table["Vid4"]) == 42 and len(table["UDM10"]) == 42
    assert table["Vid4"]["4,4"] == [27.17, 0.8184] and table["UDM10"]["3.5,2"] == [42.23, 0.9798]        # README.md:96,124
    assert T.readme_entry(table, "Vid4_x1.5_x4", (1.5, 4)) == (30.45, 0.9027)
    assert T.readme_entry(table, "Vid4_x4", (4, 4)) == (27.17, 0.8184) and T.readme_entry(table, "Vid4_x4", (4.0, 4.0)) == (27.17, 0.8184)
    assert T.readme_entry(table, "REDS4_x4", (4, 4)) is None

    def res(name, sc, p, s):
        return {"dataset": name, "scale": sc, "metrics": {"psnr_y": p, "ssim_y": s}}
    rows, st = T.check_readme([res("Vid4_x4", (4, 4), 27.1749, 0.81845), res("UDM10_x3.5_x2", (3.5, 2), 42.2251, 0.97972)], table)
    assert st == 0 and all(r["ok"] for r in rows)
    rows, st = T.check_readme([res("Vid4_x4", (4, 4), 27.17 + 0.0151, 0.8184)], table)
    assert st == 1 and not rows[0]["ok"]
    rows, st = T.check_readme([res("Vid4_x4", (4, 4), 27.17, 0.8184 + 0.00016)], table)
    assert st == 1
    rows, st = T.check_readme([res("Vid4_x4", (4, 4), 27.17, 0.8184), res("Vid4_x6", (6, 6), 25.0, 0.7)], table)
    assert st == 2 and rows[0]["ok"] and not rows[1]["ok"]
    assert "NO README ENTRY" in T.format_check(rows) and "ok" in T.format_check(rows)


def test_hr_plans_table_is_wellformed():
    """savsr_amd/hr_plans.json (the SATU HR stage's measured launch plan per scale, tools/tune_hr_plans.py): 16-hex source hash, every scale of
    the shipped YAMLs present, entries [variant, tile rows (multiple of 4, <= 64), tile columns / 32 (1..8), LR h, LR w] -- what the engine's
    loader (HipEngine._load_hr_plans) and savsr_satu_hr_tail_q's argument checks expect."""
    import json
    import re
    from savsr_amd.utils import workloads
    path = os.path.join(ROOT, "savsr_amd", "hr_plans.json")
    t = json.load(open(path))
    assert re.fullmatch(r"[0-9a-f]{16}", t["satu_source_hash"])
    keys = {tuple(float(v) for v in k.split(",")) for k in t["plans"]}
    assert {(float(a), float(b)) for a, b in workloads.YAML_SCALES} <= keys
    assert {(float(a), float(b)) for a, b in workloads.TRAIN_SCALES} <= keys
    for k, p in t["plans"].items():
        assert len(p) == 5 and all(isinstance(v, int) for v in p), (k, p)
        variant, rows, cols32, h, w = p
        assert 0 <= variant < 5 and rows % 4 == 0 and 4 <= rows <= 64 and 1 <= cols32 <= 8 and h >= 2 and w >= 2, (k, p)


def test_engine_config_reads_every_switch_once(monkeypatch):
    """savsr_amd/config.py is the ONE place the engine's SAVSR_* switches are read: defaults = the product configuration (no knobs), the
    environment changes fields, knobs() lists exactly the changed ones, and none of the engine modules looks at the environment itself."""
    import os
    from savsr_amd.config import EngineConfig
    for k in list(os.environ):
        if k.startswith("SAVSR_"):
            monkeypatch.delenv(k)
    c = EngineConfig.from_env()
    assert c == EngineConfig() and c.knobs() == {}
    assert (c.streams, c.streams_large, c.clip_batch, c.graphs, c.conv_wy, c.satu_q, c.cache_gb) == (3, 2, 4, True, True, True, None)
    monkeypatch.setenv("SAVSR_STREAMS", "2")
    monkeypatch.setenv("SAVSR_CLIP_BATCH", "2")
    monkeypatch.setenv("SAVSR_CONV_WY", "0")
    monkeypatch.setenv("SAVSR_CACHE_GB", "6.5")
    monkeypatch.setenv("SAVSR_HR_VARIANT", "1")
    c = EngineConfig.from_env()
    assert c.knobs() == {"streams": 2, "clip_batch": 2, "conv_wy": False, "cache_gb": 6.5, "hr_variant": 1}      # (streams_large follows SAVSR_STREAMS when that is set: 2 = its default)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mod in ("engine", "cache", "launch", "packing"):
        src = open(os.path.join(root, "savsr_amd", mod + ".py")).read()
        assert "os.environ" not in src and "getenv" not in src, mod
