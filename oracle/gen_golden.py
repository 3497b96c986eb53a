#!/usr/bin/env python3
"""Generate ``tests/golden/*`` by running the REFERENCE itself (build container only).

The reference (``/root/reference``) is imported read-only with three in-memory stubs
for packages the image lacks (``timm.models.layers``: init + identity only, no forward
arithmetic; ``cv2``/``imageio``/``flow_vis``: empty modules for ``demo_2x`` /
``benchmark.utils`` imports).  Identical seeded weights
(``atm-vfi_amd/schema.py::synthetic_state_dict``) are loaded into the reference with
``load_state_dict(strict=True)`` -- which also pins the 236-key schema -- and its
outputs on seeded inputs are committed as small ``.npz`` fixtures.

Nothing of the reference travels: only inputs/outputs are stored.  Re-run with
``python oracle/gen_golden.py`` (any working directory).

``--check`` regenerates everything into a temporary directory and asserts that every
array / JSON entry equals the committed one bit for bit (the log of one such run is
committed under ``profiles/``); ``--only a,b`` restricts either mode to the named
cases; ``--imports-only`` stops after the reference has been imported (CPU test).

Import isolation: the repo ships its own ``network/`` and ``benchmark/`` packages (the
drop-in shims of SURVEY.md §8b).  They are regular packages and would shadow the
reference's namespace packages of the same names as soon as the repo root is on
``sys.path``, so the reference is imported FIRST, with the repo root (and the working
directory) off the path; the repo's own modules are imported afterwards under names
that do not collide (``atm-vfi_amd.schema``, ``oracle.atmvfi_oracle``, ``pairs``).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")


def install_stubs():
    layers = types.ModuleType("timm.models.layers")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.to_2tuple = lambda x: (x, x)
    layers.DropPath = torch.nn.Identity
    sys.modules["timm"] = types.ModuleType("timm")
    sys.modules["timm.models"] = types.ModuleType("timm.models")
    sys.modules["timm.models.layers"] = layers
    for name in ("cv2", "imageio", "flow_vis"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cv2"].FONT_HERSHEY_SIMPLEX = 0
    sys.modules["imageio"].imread = sys.modules["imageio"].imwrite = None


def drop_mask_cache(net):
    # the reference caches shift masks keyed on Hp*Wp only (attention.py:279,304-305);
    # drop them between shapes so every case sees a freshly built mask
    for m in net.modules():
        for nm in ("attn_mask", "HW"):
            if nm in m._buffers:
                del m._buffers[nm]


def sub(t: torch.Tensor, step: int) -> np.ndarray:
    return t[..., ::step, ::step].contiguous().numpy()


class Ref:
    """The reference's modules, imported with the repo root off ``sys.path`` (see the module docstring)."""


def import_reference() -> Ref:
    install_stubs()
    saved_path, saved_cwd = list(sys.path), os.getcwd()
    here = {os.path.realpath(ROOT)}

    def is_repo(p):
        return os.path.realpath(p or saved_cwd) in here
    for name in [m for m in sys.modules if m.split(".")[0] in ("network", "benchmark", "network_base", "network_lite", "flow_warp",
                                                                "attention", "demo_2x")]:
        del sys.modules[name]                      # anything a caller imported from the repo's shims
    try:
        os.chdir(REF)                              # demo_2x.py:10 appends './network/' relative to the working directory
        sys.path[:] = [REF, os.path.join(REF, "network")] + [p for p in saved_path if not is_repo(p)]
        r = Ref()
        r.network_base = importlib.import_module("network_base")
        r.network_lite = importlib.import_module("network_lite")
        r.attention = importlib.import_module("attention")
        r.flow_warp = importlib.import_module("flow_warp")
        r.InputPadder = importlib.import_module("benchmark.utils").InputPadder
        torch.Tensor.cuda = lambda self, *a, **k: self       # demo_2x.py:70-71 hard-codes .cuda()
        r.demo_2x = importlib.import_module("demo_2x")
    finally:
        os.chdir(saved_cwd)
        sys.path[:] = saved_path
    for mod in (r.network_base, r.network_lite, r.attention, r.flow_warp, r.demo_2x):
        assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REF) + os.sep), mod.__file__
    assert os.path.realpath(sys.modules["network.attention"].__file__).startswith(os.path.realpath(REF) + os.sep)
    return r


# (name, variant, B, H, W, global, ensemble, input kind, input seed, store step)
E2E_CASES = [
    ("lite_64x64_g", "lite", 1, 64, 64, True, False, "smooth", 11, 1),       # global pad 4x4 -> 12x12
    ("lite_128x192_g_b2", "lite", 2, 128, 192, True, False, "smooth", 12, 1),  # B=2: frame-stack order
    ("lite_256x448_nog", "lite", 1, 256, 448, False, False, "smooth", 13, 2),  # BASELINE config C2
    ("lite_96x160_g_rand", "lite", 1, 96, 160, True, False, "random", 14, 1),  # iid frames; local pad 12x20->16x24
    ("base_64x64_g", "base", 1, 64, 64, True, False, "smooth", 21, 1),
    ("base_128x192_g", "base", 1, 128, 192, True, False, "smooth", 22, 1),
    ("base_160x96_nog_b2", "base", 2, 160, 96, False, False, "random", 23, 1),
    ("lite_384x576_ens", "lite", 1, 384, 576, True, True, "smooth", 31, 4),    # ensemble; Hp*Wp distinct per scale
    ("base_192x320_g", "base", 1, 192, 320, True, False, "smooth", 24, 2),     # global 12x20 -> pad 12x24 (+shift)
    # ensemble on the base variant (network_base.py:564-605); the three global canvases 24x36, 12x24, 12x12 have distinct Hp*Wp, so the
    # reference's shift-mask cache (attention.py:279, keyed on Hp*Wp only) never hands a stale mask to another scale
    ("base_384x576_ens", "base", 1, 384, 576, True, True, "smooth", 32, 4),
    # ensemble with B = 3 whose samples pick DIFFERENT levels -- 0, 1 and 2 (network_base.py:593-603: the per-sample if / elif chain, the
    # x2 and x4 up-sampling branches); the manifest records the picks and tests/test_oracle_golden.py asserts they are [0, 1, 2]
    ("base_384x576_ens_b3_mixed", "base", 3, 384, 576, True, True, "mixed", 40, 4),
    # BASELINE config C5: 2160x4096 through test_xiph.py:115-128's InputPadder(divisor 32) = 2176x4096, untiled, global on
    ("base_2176x4096_g_c5", "base", 1, 2176, 4096, True, False, "smooth", 25, 16),
    # LARGE MOTION (an 11th field: ``motion_gain`` of schema.synthetic_state_dict): the flow rows of every motion head x4, so that the
    # global branch moves content by tens of pixels (network_base.py:391-415, 457-485), the decoder flows reach 50-70 px
    # (:511-525) and taps leave the frame (flow_warp.py:26-60).  The manifest records flow|max| at every level's warp and the number
    # of 32 x 8 output tiles whose taps do not fit the tiled warps' 64 x 24 staged box (golden_util.tiled_warp_fallback_tiles: the
    # gather fallback of pointwise.hip); tests/test_oracle_golden.py asserts both are non-trivial.
    ("base_384x576_g_large", "base", 1, 384, 576, True, False, "smooth", 26, 2, 4.0),
    ("lite_384x576_g_large", "lite", 1, 384, 576, True, False, "smooth", 27, 2, 4.0),
    ("base_1088x1920_g_large", "base", 1, 1088, 1920, True, False, "smooth", 28, 8, 4.0),   # BASELINE config C4's shape, strided store
]
# (name, variant, H, W, global): uint8 frames through demo_2x.inference_2frame (demo_2x.py:54-87)
DEMO_CASES = [
    ("demo_lite_270x480", "lite", 270, 480, True),
    ("demo_lite_256x256", "lite", 256, 256, True),
    ("demo_base_100x180_nog", "base", 100, 180, False),
    ("demo_base_540x960_c3", "base", 540, 960, True),      # BASELINE config C3 through the padder: 540x960 -> 576x960
]
LARGE = {"base_2176x4096_g_c5"}     # minutes of CPU time and ~30 GB of host memory each


def case_fields(c):
    """(name, variant, B, H, W, global, ensemble, kind, seed, step, motion_gain)"""
    return tuple(c) + ((1.0,) if len(c) == 10 else ())


def generate(gold: str, only=None) -> dict:
    """Write the fixtures into ``gold``; returns the manifest.  ``only``: set of case / op names (None = all)."""
    ref = import_reference()
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import pairs
    import golden_util
    schema = importlib.import_module("atm-vfi_amd.schema")
    from oracle import atmvfi_oracle as O

    def want(name):
        return only is None or name in only

    torch.set_grad_enabled(False)
    os.makedirs(gold, exist_ok=True)
    mods = {"base": ref.network_base, "lite": ref.network_lite}
    manifest = {"torch": torch.__version__, "cases": [], "ops": []}

    # ---- 1. schema ----
    if want("schema"):
        sch = {}
        for v, mod in mods.items():
            net = mod.Network()
            sd = net.state_dict()
            sch[v] = {"entries": [[k, list(t.shape)] for k, t in sd.items()],
                      "n_params": sum(p.numel() for p in net.parameters()),
                      "buffers": [k for k, _ in net.named_buffers()]}
        with open(os.path.join(gold, "schema.json"), "w") as f:
            json.dump(sch, f, indent=0)

    # ---- 2. end-to-end cases ----
    nets = {}
    sds = {}
    for v, mod in mods.items():
        sds[v] = schema.synthetic_state_dict(v, seed=1)
        nets[v] = mod.Network().eval()
        nets[v].load_state_dict(sds[v], strict=True)      # pins names+shapes
    for (name, v, b, h, w, g, ens, kind, seed, step, mgain) in map(case_fields, E2E_CASES):
        if not want(name):
            continue
        im0, im1 = pairs.PAIR_KINDS[kind](b, h, w, seed)
        sd_case = sds[v]
        if mgain != 1.0:
            sd_case = schema.synthetic_state_dict(v, seed=1, motion_gain=mgain)
            drop_mask_cache(nets[v])
            nets[v].load_state_dict(sd_case, strict=True)
        net = nets[v]
        drop_mask_cache(net)
        net.global_motion = g
        net.ensemble_global_motion = ens
        warp_log = []          # (channels, H, W, flow|max|) of every flow_warp call of the reference's forward, in call order
        if mgain != 1.0:
            real_warp = mods[v].flow_warp

            def spy(feature, flow, *a, _real=real_warp, **k):
                warp_log.append([int(feature.shape[1]), int(feature.shape[2]), int(feature.shape[3]), float(flow.abs().max())])
                return _real(feature, flow, *a, **k)
            mods[v].flow_warp = spy
        try:
            out = net(im0, im1)
        finally:
            if mgain != 1.0:
                mods[v].flow_warp = real_warp
        arrs = {
            "I_t": sub(out["I_t"], step),
            "im_t0": sub(out["im_t_list"][0], step),
            "opt_flow_0": sub(out["opt_flow_0"], step),
            "opt_flow_1": sub(out["opt_flow_1"], step),
            "occ_mask1": sub(out["occ_mask1"], step),
            "I_t_0": sub(out["I_t_0"], step),
            "I_t_1": sub(out["I_t_1"], step),
            "im_t_coarse": out["im_t_list"][-1].numpy(),
            "im0_warped_coarse": out["im0_warped_list"][-1].numpy(),
            "sums": np.array([out[k].double().sum().item() for k in
                              ("I_t", "opt_flow_0", "opt_flow_1", "occ_mask1", "I_t_0", "I_t_1")]),
            "abs_sums": np.array([out[k].double().abs().sum().item() for k in
                                  ("I_t", "opt_flow_0", "opt_flow_1", "occ_mask1", "I_t_0", "I_t_1")]),
            "in_sums": np.array([im0.double().sum().item(), im1.double().sum().item()]),
        }
        np.savez_compressed(os.path.join(gold, name + ".npz"), **arrs)
        n_lists = len(out["im_t_list"])
        flow_max = out["opt_flow_0"].abs().max().item()
        large = {}
        if mgain != 1.0:
            large = {"motion_gain": mgain, "flow_max": max(flow_max, out["opt_flow_1"].abs().max().item()),
                     "global_flow_max_full_res": max(r[3] for r in warp_log[:12] if r[1] == h),
                     "warp_flow_max": warp_log,
                     "fallback_tiles": [golden_util.tiled_warp_fallback_tiles(out[k].numpy()) for k in ("opt_flow_0", "opt_flow_1")],
                     "tiles": ((h + 7) // 8) * ((w + 31) // 32) * b,
                     "out_of_frame_fraction": float((golden_util.taps_out_of_frame(out["opt_flow_0"].numpy())).mean())}
        keep = {k: out[k] for k in ("I_t",)}
        lists = [t for t in out["im_t_list"]]
        del out
        ora = O.forward(sd_case, im0, im1, global_motion=g, ensemble_global_motion=ens)
        picks = []
        if ens:      # which level each sample's global flow came from (the oracle is bit-identical to the reference on these cases)
            O.ensemble_global_flows(sds[v], im0, im1, schema.VARIANTS[v].global_window, picks)
        d = (keep["I_t"] - ora["I_t"]).abs().max().item()
        dl = max((a - c).abs().max().item() for a, c in zip(lists, ora["im_t_list"]))
        del ora, keep, lists
        manifest["cases"].append({"name": name, "variant": v, "B": b, "H": h, "W": w, "global": g,
                                  "ensemble": ens, "kind": kind, "seed": seed, "step": step,
                                  "n_lists": n_lists, "oracle_vs_ref_I_t": d,
                                  "oracle_vs_ref_lists": dl, **({"ensemble_picks": picks} if ens else {}), **large})
        if mgain != 1.0:
            drop_mask_cache(nets[v])                       # (the lazily registered mask buffers are not in the schema)
            nets[v].load_state_dict(sds[v], strict=True)
        print(f"{name:24s} oracle-vs-reference max|d| I_t {d:.2e} lists {dl:.2e}  flow|max| {flow_max:.2f}", flush=True)

    # ---- 3. demo path: uint8 frames through the reference's inference_2frame ----
    demo_2x = ref.demo_2x
    torch.set_grad_enabled(False)
    for (name, v, h, w, g) in DEMO_CASES:
        if not want(name):
            continue
        f0, f1 = pairs.uint8_pair(h, w, seed=0)
        net = nets[v]
        drop_mask_cache(net)
        net.global_motion = g
        net.ensemble_global_motion = False
        pred = demo_2x.inference_2frame(f0, f1, net, isBGR=True)
        ora = O.inference_2frame(sds[v], f0, f1, isBGR=True, global_motion=g)
        nd = int((pred.astype(np.int32) - ora.astype(np.int32)).__abs__().max())
        np.savez_compressed(os.path.join(gold, name + ".npz"), pred=pred,
                            in_sums=np.array([int(f0.sum()), int(f1.sum())]))
        manifest["cases"].append({"name": name, "variant": v, "H": h, "W": w, "global": g, "kind": "demo_uint8",
                                  "seed": 0, "oracle_vs_ref_uint8": nd})
        print(f"{name:24s} oracle-vs-reference max|d| uint8 {nd}", flush=True)
    # natural image content: a crop of the reference's only real frame pair (asset/example_frame{0,1}.png)
    if want("demo_lite_asset_crop"):
        from PIL import Image
        a0 = np.array(Image.open(os.path.join(REF, "asset/example_frame0.png")).convert("RGB"))
        a1 = np.array(Image.open(os.path.join(REF, "asset/example_frame1.png")).convert("RGB"))
        c0 = a0[200:200 + 150, 120:120 + 200][:, :, ::-1].copy()     # BGR like cv2.imread
        c1 = a1[200:200 + 150, 120:120 + 200][:, :, ::-1].copy()
        net = nets["lite"]
        drop_mask_cache(net)
        net.global_motion = True
        pred = demo_2x.inference_2frame(c0, c1, net, isBGR=True)
        np.savez_compressed(os.path.join(gold, "demo_lite_asset_crop.npz"), f0=c0, f1=c1, pred=pred)
        manifest["cases"].append({"name": "demo_lite_asset_crop", "variant": "lite", "H": 150, "W": 200,
                                  "global": True, "kind": "demo_asset"})
        print("demo_lite_asset_crop     stored")

    # ---- 4. operator fixtures (shapes from the reference's own smoke blocks) ----
    ref_attn, ref_warp = ref.attention, ref.flow_warp
    # attention.py:512-534: C=128, win 7 on 32x32 -> pad 32->35 and shift 3 (B reduced 24 -> 2 pairs)
    gen = torch.Generator().manual_seed(5)
    for shift in (0, 3):
        blk = ref_attn.ATMFormer(dim=128, num_heads=8, window_size=7, shift_size=shift).eval()
        st = blk.state_dict()
        for k in st:
            if "relative_coord" in k:
                continue
            st[k] = torch.randn(st[k].shape, generator=gen) * (0.3 if st[k].dim() > 1 else 0.2) + (1.0 if "norm" in k and "weight" in k else 0.0)
        blk.load_state_dict(st)
        x = torch.randn(4, 32 * 32, 128, generator=gen)
        if not want(f"op_atm_ws7_shift{shift}"):
            continue
        y, mo = blk(x.reshape(4, 32, 32, 128), 32, 32, 2)
        oy, om = O.atm_block({f"b.{k}": t for k, t in st.items()}, "b", x.reshape(4, 32, 32, 128), 7, shift)
        print(f"op atm_ws7_shift{shift}: oracle-vs-reference x {(y - oy).abs().max():.2e} motion {(mo - om).abs().max():.2e}")
        np.savez_compressed(os.path.join(gold, f"op_atm_ws7_shift{shift}.npz"),
                            **{"w." + k: t.numpy() for k, t in st.items() if "relative_coord" not in k},
                            x=x.numpy(), y=y[:, ::4].numpy(), motion=mo.numpy())
        manifest["ops"].append(f"op_atm_ws7_shift{shift}")
    # flow_warp incl. out-of-range taps
    feat = torch.rand(2, 5, 9, 13, generator=gen)
    flow = (torch.rand(2, 2, 9, 13, generator=gen) - 0.5) * 8
    flow[0, :, 0, 0] = torch.tensor([-0.5, 0.0]); flow[0, :, 0, 1] = torch.tensor([-2.5, 0.0])
    if want("op_flow_warp"):
        wv = ref_warp.flow_warp(feat, flow)
        np.savez_compressed(os.path.join(gold, "op_flow_warp.npz"), feat=feat.numpy(), flow=flow.numpy(), out=wv.numpy())
        manifest["ops"].append("op_flow_warp")
        print(f"op flow_warp: oracle-vs-reference {(wv - O.flow_warp(feat, flow)).abs().max():.2e} "
              f"explicit {(wv - O.flow_warp_explicit(feat, flow)).abs().max():.2e}")
    # flow_warp's non-default forms (flow_warp.py:50-60, bilinear_sample :26-47): mask=True and padding_mode 'border' / 'reflection';
    # flows from sub-pixel to several image sizes (reflection folds more than once), exact-edge coordinates included
    if want("op_flow_warp_modes"):
        g2 = torch.Generator().manual_seed(6)
        feat2 = torch.rand(2, 3, 12, 20, generator=g2)
        flow2 = (torch.rand(2, 2, 12, 20, generator=g2) - 0.5) * 10
        flow2[1] *= 9.0                                                   # up to +-45 px on a 12 x 20 image
        flow2[0, :, 0, 0] = torch.tensor([0.0, 0.0]); flow2[0, :, 0, 1] = torch.tensor([-1.0, 0.0])      # coordinate exactly 0
        flow2[0, :, 11, 19] = torch.tensor([0.0, 0.0]); flow2[0, :, 11, 18] = torch.tensor([1.0, 0.0])   # exactly size - 1
        flow2[0, :, 5, 0] = torch.tensor([-1e-9, 0.0])                    # a rounding below zero: 2 p / (w - 1) - 1 == -1 in fp32
        arrs = {"feat": feat2.numpy(), "flow": flow2.numpy()}
        for pm in ("zeros", "border", "reflection"):
            o, m = ref_warp.flow_warp(feat2, flow2, mask=True, padding_mode=pm)
            arrs["out_" + pm] = o.numpy()
            arrs["mask"] = m.numpy()
            o2 = ref_warp.flow_warp(feat2, flow2, padding_mode=pm)
            assert torch.equal(o, o2)
        np.savez_compressed(os.path.join(gold, "op_flow_warp_modes.npz"), **arrs)
        manifest["ops"].append("op_flow_warp_modes")
        print(f"op flow_warp_modes: mask true on {arrs['mask'].mean():.2f} of the pixels")
    # InputPadder (benchmark/utils.py:57-80)
    pads = {}
    for (h, w, dv) in ((270, 480, 64), (1080, 1920, 64), (256, 256, 64), (1080, 2048, 32), (100, 180, 64), (540, 960, 64),
                       (2160, 4096, 32)):
        pads[f"{h}x{w}/{dv}"] = ref.InputPadder((1, 3, h, w), divisor=dv)._pad
    manifest["input_padder"] = pads

    with open(os.path.join(gold, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(gold, n)) for n in os.listdir(gold))
    print(f"wrote {len(os.listdir(gold))} files, {tot / 1e6:.1f} MB")
    return manifest


def _same(a, b) -> bool:
    if isinstance(a, float) and isinstance(b, float):
        return a == b or (a != a and b != b)
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


def check(only=None) -> int:
    """Regenerate into a temporary directory and compare with ``tests/golden`` bit for bit."""
    bad = 0
    with tempfile.TemporaryDirectory(prefix="atmvfi_golden_") as tmp:
        man = generate(tmp, only)
        with open(os.path.join(GOLD, "manifest.json")) as f:
            committed = json.load(f)
        by_name = {c["name"]: c for c in committed["cases"]}
        for c in man["cases"]:
            ok = c["name"] in by_name and _same(c, by_name[c["name"]])
            bad += not ok
            print(f"manifest {c['name']:28s} {'identical' if ok else 'DIFFERS'}")
        if only is None:
            for key in ("ops", "input_padder", "torch"):
                ok = _same(man[key], committed[key])
                bad += not ok
                print(f"manifest[{key}] {'identical' if ok else 'DIFFERS'}")
            if {c["name"] for c in man["cases"]} != set(by_name):
                bad += 1
                print("manifest: case lists differ")
        for fn in sorted(os.listdir(tmp)):
            new_p, old_p = os.path.join(tmp, fn), os.path.join(GOLD, fn)
            if fn == "manifest.json":
                continue
            if not os.path.exists(old_p):
                print(f"{fn:36s} MISSING from tests/golden")
                bad += 1
                continue
            if fn.endswith(".json"):
                with open(new_p) as f1, open(old_p) as f2:
                    ok = _same(json.load(f1), json.load(f2))
            else:
                a, b = np.load(new_p), np.load(old_p)
                ok = sorted(a.files) == sorted(b.files) and all(
                    a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes() for k in a.files)
            bad += not ok
            print(f"{fn:36s} {'bit-identical' if ok else 'DIFFERS'}")
    print("CHECK " + ("PASSED: every regenerated fixture equals the committed one" if bad == 0 else f"FAILED: {bad} differences"))
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--check", action="store_true", help="regenerate into a temp dir and compare with tests/golden bit for bit")
    ap.add_argument("--only", default=None, help="comma-separated case / op names (default: everything)")
    ap.add_argument("--skip-large", action="store_true", help="leave out the 4K case (minutes of CPU, ~30 GB of host memory)")
    ap.add_argument("--imports-only", action="store_true", help="import the reference and exit (checks the import isolation)")
    args = ap.parse_args()
    if args.imports_only:
        r = import_reference()
        print("reference imported:", r.network_base.__file__, r.attention.__file__, r.demo_2x.__file__)
        return 0
    only = set(args.only.split(",")) if args.only else None
    if args.skip_large:
        names = {"schema", "demo_lite_asset_crop", "op_atm_ws7_shift0", "op_atm_ws7_shift3", "op_flow_warp", "op_flow_warp_modes"}
        names |= {c[0] for c in E2E_CASES} | {c[0] for c in DEMO_CASES}
        only = (only if only is not None else names) - LARGE
    if args.check:
        return check(only)
    if only is not None:
        # partial regeneration: keep the other entries of the committed manifest
        with tempfile.TemporaryDirectory(prefix="atmvfi_golden_") as tmp:
            man = generate(tmp, only)
            import shutil
            for fn in os.listdir(tmp):
                if fn != "manifest.json":
                    shutil.copy(os.path.join(tmp, fn), os.path.join(GOLD, fn))
            with open(os.path.join(GOLD, "manifest.json")) as f:
                committed = json.load(f)
            order = [c[0] for c in E2E_CASES] + [c[0] for c in DEMO_CASES] + ["demo_lite_asset_crop"]
            cases = {c["name"]: c for c in committed["cases"]}
            cases.update({c["name"]: c for c in man["cases"]})
            committed["cases"] = [cases[n] for n in order if n in cases]
            committed["ops"] = sorted(set(committed["ops"]) | set(man["ops"]))
            committed["input_padder"] = man["input_padder"]
            with open(os.path.join(GOLD, "manifest.json"), "w") as f:
                json.dump(committed, f, indent=1)
        return 0
    generate(GOLD, None)
    return 0


if __name__ == "__main__":
    sys.exit(main())
