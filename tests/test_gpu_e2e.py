"""GPU: the whole hot path (``Network.forward`` -> libatmvfi_hip.so) against
 (1) the reference's own outputs committed under tests/golden,
 (2) the CPU oracle on the same seeded inputs, up to BASELINE.json's full sizes,
 (3) size-independent properties at full size (run-to-run determinism, per-pair independence).
Tolerance is the north star's: max|d| <= 1e-3 per pixel on fp32 ``I_t``."""
import importlib
import os

import numpy as np
import pytest
import torch

import golden_util as G
import pairs
from oracle import atmvfi_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3            # BASELINE.json north_star: |d| <= 1e-3 per pixel (fp32)
TOL_FLOW = 2e-3       # flows are O(1..10) px; same relative budget

pkg = importlib.import_module("atm-vfi_amd")
host_io = importlib.import_module("atm-vfi_amd.host_io")
hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def nets(dev):
    torch.set_grad_enabled(False)
    out = {}
    for v, cls in (("lite", pkg.NetworkLite), ("base", pkg.NetworkBase)):
        net = cls()
        net.load_state_dict(pkg.synthetic_state_dict(v, seed=1), strict=True)
        out[v] = net.to(dev).eval()
    return out


def run(net, case_global, ens, im0, im1, dev):
    net.global_motion = case_global
    net.ensemble_global_motion = ens
    out = net(im0.to(dev), im1.to(dev))
    torch.cuda.synchronize()
    return out


def large_motion_net(variant, gain, weights, dev):
    """A fresh model on the large-motion weight set (schema.synthetic_state_dict(motion_gain=...), the ``*_large`` fixtures)."""
    net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
    net.load_state_dict(weights(variant, gain), strict=True)
    return net.to(dev).eval()


@pytest.mark.parametrize("case", G.e2e_cases(), ids=lambda c: c["name"])
def test_forward_vs_reference_golden(case, nets, dev, weights):
    gold = G.load_npz(case["name"])
    im0, im1 = G.case_inputs(case)
    G.check_inputs_match(case, gold, im0, im1)
    gain = case.get("motion_gain", 1.0)
    net = nets[case["variant"]] if gain == 1.0 else large_motion_net(case["variant"], gain, weights, dev)
    out = run(net, case["global"], case["ensemble"], im0, im1, dev)
    if gain != 1.0:
        # the fixture is in the large-motion regime ON THIS PATH too: flows of tens of pixels, and the final warp's tiles overflow
        # the staged box (the manifest's figures come from the reference; here the HIP path's own flows)
        fm = max(out["opt_flow_0"].abs().max().item(), out["opt_flow_1"].abs().max().item())
        assert abs(fm - case["flow_max"]) <= 1e-2 and fm >= 32.0
        assert G.tiled_warp_fallback_tiles(out["opt_flow_0"].cpu().numpy()) > 0.25 * case["tiles"]
        net.release_workspace()
    assert len(out["im_t_list"]) == case["n_lists"]
    assert set(out.keys()) == {"I_t", "im_t_list", "im0_warped_list", "im1_warped_list", "opt_flow_0", "opt_flow_1",
                               "I_t_0", "I_t_1", "occ_mask1", "occ_mask2"}
    errs = G.compare_e2e(out, gold, case["step"], TOL, TOL_FLOW)
    print(case["name"], {k: f"{v:.1e}" for k, v in errs.items()})


@pytest.mark.parametrize("case", G.demo_cases(), ids=lambda c: c["name"])
def test_inference_2frame_uint8(case, nets, dev):
    gold = G.load_npz(case["name"])
    f0, f1 = pairs.uint8_pair(case["H"], case["W"], seed=case["seed"])
    net = nets[case["variant"]]
    net.global_motion = case["global"]
    net.ensemble_global_motion = False
    pred = host_io.inference_2frame(f0, f1, net, isBGR=True)
    d = np.abs(pred.astype(np.int32) - gold["pred"].astype(np.int32))
    assert pred.shape == gold["pred"].shape and pred.dtype == np.uint8
    assert d.max() <= 1 and (d > 0).mean() < 5e-3      # |d| <= 1e-3 in fp32 can flip a rounding, never more than one level


def test_asset_crop_natural_image(nets, dev):
    gold = G.load_npz("demo_lite_asset_crop")
    net = nets["lite"]
    net.global_motion = True
    net.ensemble_global_motion = False
    pred = host_io.inference_2frame(gold["f0"], gold["f1"], net, isBGR=True)
    d = np.abs(pred.astype(np.int32) - gold["pred"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3


FULL = [
    # BASELINE.json configs at their full sizes, checked against the CPU oracle
    ("C1 lite 256x256 global on", "lite", 1, 256, 256, True),
    ("C2 lite 256x448 global off", "lite", 1, 256, 448, False),
    ("C3 base 540x960->576x960 global on", "base", 1, 576, 960, True),
    ("C4 base 1080x1920->1088x1920 global on", "base", 1, 1088, 1920, True),
    # the workload bench.py times: i.i.d. random frames at full size -- the worst case for |grad I| x d(flow) through the warps
    # (flow_warp.py:26-60 feeding network_base.py:523-533); bench.py's `parity` block certifies the timed pair itself the same way
    ("C4r base 1088x1920 global on, RANDOM frames (the timed workload)", "base", 1, 1088, 1920, True, "random", 1000),
    # large motion at full size against the oracle, every pixel (the strided fixture base_1088x1920_g_large checks the same forward
    # against the reference's own output)
    ("C4L base 1088x1920 global on, large-motion weights (x4)", "base", 1, 1088, 1920, True, "smooth", 28, 4.0),
]


@pytest.mark.parametrize("cfg", FULL, ids=lambda c: c[0].split()[0] + "_" + c[1])
def test_full_size_vs_oracle(cfg, nets, dev, weights):
    name, v, b, h, w, g = cfg[:6]
    kind, seed = (cfg[6], cfg[7]) if len(cfg) > 6 else ("smooth", 41)
    gain = cfg[8] if len(cfg) > 8 else 1.0
    im0, im1 = pairs.PAIR_KINDS[kind](b, h, w, seed)
    net = nets[v] if gain == 1.0 else large_motion_net(v, gain, weights, dev)
    out = run(net, g, False, im0, im1, dev)
    ref = O.forward(weights(v, gain), im0, im1, global_motion=g)
    errs = {}
    for k in ("I_t", "I_t_0", "I_t_1", "occ_mask1"):
        errs[k] = (out[k].cpu() - ref[k]).abs().max().item()
        assert errs[k] <= TOL, f"{name}: {k} max|d| {errs[k]:.3e}"
    for k in ("opt_flow_0", "opt_flow_1"):
        errs[k] = (out[k].cpu() - ref[k]).abs().max().item()
        assert errs[k] <= TOL_FLOW, f"{name}: {k} max|d| {errs[k]:.3e}"
    for a, r in zip(out["im_t_list"], ref["im_t_list"]):
        assert (a.cpu() - r).abs().max().item() <= TOL
    mse = ((out["I_t"].cpu() - ref["I_t"]) ** 2).mean().item()
    psnr = float("inf") if mse == 0 else -10 * np.log10(mse)
    print(f"{name}: " + " ".join(f"{k}={e:.1e}" for k, e in errs.items()) + f" PSNR-vs-oracle={psnr:.1f} dB")
    assert psnr > 80.0


def test_full_size_properties_1080p(nets, dev):
    """Size-independent properties at the benchmark size: bitwise run-to-run determinism, and
    per-pair independence (a batch of two pairs == the two pairs run separately)."""
    net = nets["base"]
    a0, a1 = pairs.smooth_pair(1, 1088, 1920, seed=51)
    b0, b1 = pairs.random_pair(1, 1088, 1920, seed=52)
    oa = run(net, True, False, a0, a1, dev)["I_t"].clone()
    oa2 = run(net, True, False, a0, a1, dev)["I_t"].clone()
    assert torch.equal(oa, oa2)
    ob = run(net, True, False, b0, b1, dev)["I_t"].clone()
    both = run(net, True, False, torch.cat([a0, b0]), torch.cat([a1, b1]), dev)["I_t"]
    assert (both[0] - oa[0]).abs().max().item() <= 1e-5
    assert (both[1] - ob[0]).abs().max().item() <= 1e-5
    assert both.min().item() >= 0.0 and both.max().item() <= 1.0
    net.release_workspace()


def test_full_size_properties_4k(nets, dev):
    """BASELINE's largest configuration (C5, 2160x4096 padded to 2176x4096, untiled): the forward runs, is bitwise deterministic
    run to run and stays in range; with the global branch off, the warped frame agrees with a direct warp of the input by the
    returned flow (a relation between outputs that holds at any size: I_t_0 = flow_warp(im0, opt_flow_0), flow_warp.py:50-60,
    network_base.py:523-541)."""
    net = nets["base"]
    a0, a1 = pairs.smooth_pair(1, 2176, 4096, seed=71)
    it1 = run(net, True, False, a0, a1, dev)["I_t"].clone()
    o2 = run(net, True, False, a0, a1, dev)
    assert torch.equal(it1, o2["I_t"])
    assert torch.isfinite(it1).all() and it1.min().item() >= 0.0 and it1.max().item() <= 1.0
    assert it1.shape == (1, 3, 2176, 4096)
    # without the global branch the finest-level warp acts on the input frame itself (with it, on the globally pre-warped pyramid)
    o3 = run(net, False, False, a0, a1, dev)
    w0, f0 = o3["I_t_0"].clone(), o3["opt_flow_0"].clone()
    # independent restatement of the backward bilinear warp (zero outside), on the GPU in fp64
    im = a0.to(dev).double()
    H, W = 2176, 4096
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64), indexing="ij")
    px, py = xs + f0[0, 0].double(), ys + f0[0, 1].double()
    x0, y0 = torch.floor(px), torch.floor(py)
    ref = torch.zeros(3, H, W, device=dev, dtype=torch.float64)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wgt = (1 - (px - xi).abs()) * (1 - (py - yi).abs())
            ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()
            ref += torch.where(ok, wgt, torch.zeros_like(wgt)) * im[0].reshape(3, -1)[:, idx.reshape(-1)].reshape(3, H, W)
    assert (w0[0].double() - ref).abs().max().item() <= 2e-4      # fp32 coordinate arithmetic at |coordinate| ~ 4096
    net.release_workspace()


def test_window_size_knob_and_state_dict_roundtrip(nets, dev, weights):
    """__set_{local,global}_window_size__ (network_base.py:262-270) and checkpoint reload."""
    net = pkg.NetworkLite()
    sd = weights("lite")
    net.load_state_dict(sd, strict=True)
    net.to(dev).eval()
    net.__set_global_window_size__(8)
    net.__set_local_window_size__(4)
    im0, im1 = pairs.smooth_pair(1, 128, 192, seed=61)
    out = net(im0.to(dev), im1.to(dev))
    ref = O.forward(sd, im0, im1, global_motion=True, local_window=4, global_window=8)
    assert (out["I_t"].cpu() - ref["I_t"]).abs().max().item() <= TOL
    assert net.state_dict()["global_motion_atmformer.0.attn.relative_coord"].shape[-1] == 64
    # a saved-and-stripped checkpoint (lazy attn_mask/HW keys) loads back bit-identically
    polluted = dict(net.state_dict())
    polluted["local_motion_atmformer.1.attn_mask"] = torch.zeros(3)
    polluted["local_motion_atmformer.1.HW"] = torch.zeros(1)
    net2 = pkg.NetworkLite()
    net2.__set_global_window_size__(8)
    net2.__set_local_window_size__(4)
    net2.load_state_dict(host_io.strip_lazy_buffers(polluted), strict=True)
    net2.to(dev).eval()
    out2 = net2(im0.to(dev), im1.to(dev))
    assert torch.equal(out2["I_t"], out["I_t"])


def test_rejects_cpu_inputs_and_bad_shapes(nets, dev):
    net = nets["lite"]
    with pytest.raises((RuntimeError, TypeError)):
        net(torch.rand(1, 3, 64, 64), torch.rand(1, 3, 64, 64))
    net.global_motion = True
    with pytest.raises(ValueError):
        net(torch.rand(1, 3, 72, 64, device=dev), torch.rand(1, 3, 72, 64, device=dev))


def test_c5_4k_untiled_properties(nets, dev):
    """BASELINE config C5 (network_base 2160x4096 -> padded 2176x4096, global on) on one GPU, untiled (the
    reference has no tiling).  Too large for the CPU oracle inside a test budget, so the
    size-independent properties are checked: finite, range, bitwise determinism and the structural identities of
    the returned dict."""
    net = nets["base"]
    pad = host_io.InputPadder((1, 3, 2160, 4096), divisor=64)
    a0, a1 = pairs.smooth_pair(1, 2160, 4096, seed=71)
    a0, a1 = pad.pad(a0, a1)
    assert tuple(a0.shape[-2:]) == (2176, 4096)
    o1 = run(net, True, False, a0, a1, dev)
    it = o1["I_t"].clone()
    assert torch.isfinite(it).all() and it.min().item() >= 0.0 and it.max().item() <= 1.0
    assert len(o1["im_t_list"]) == 5 and tuple(o1["opt_flow_0"].shape) == (1, 2, 2176, 4096)
    o2 = run(net, True, False, a0, a1, dev)
    assert torch.equal(o2["I_t"], it)
    # structure of the outputs (network_base.py:525-533): masks are complementary, the returned frame is the clamped
    # refined frame, and the refinement residual 2*sigmoid(r)-1 stays inside (-1, 1) around the blend
    m1 = o2["occ_mask1"]
    assert torch.equal(o2["occ_mask2"], 1 - m1)
    assert torch.equal(o2["I_t"], o2["im_t_list"][0].clamp(0, 1))
    blend = m1 * o2["I_t_0"] + (1 - m1) * o2["I_t_1"]
    assert (o2["im_t_list"][0] - blend).abs().max().item() < 1.0 + 1e-5
    net.release_workspace()
    torch.cuda.empty_cache()


def test_exact_fp32_engine_end_to_end(dev, weights):
    """Network.set_precision("f32"): every contraction on the exact-fp32 MFMA engine (gemm_mfma_f32)."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    net.set_precision("f32")
    case = [c for c in G.e2e_cases() if c["name"] == "lite_128x192_g_b2"][0]
    im0, im1 = G.case_inputs(case)
    out = net(im0.to(dev), im1.to(dev))
    assert net._ops_obj.precision == "f32"
    errs = G.compare_e2e(out, G.load_npz(case["name"]), case["step"], TOL, TOL_FLOW)
    net.set_precision("f16x3")
    out2 = net(im0.to(dev), im1.to(dev))
    errs2 = G.compare_e2e(out2, G.load_npz(case["name"]), case["step"], TOL, TOL_FLOW)
    print("f32", {k: f"{v:.1e}" for k, v in errs.items()}, "f16x3", {k: f"{v:.1e}" for k, v in errs2.items()})
    # the two engines agree far inside the parity budget
    assert (out["I_t"] - out2["I_t"]).abs().max().item() <= 2e-4


@pytest.mark.parametrize("variant,name", [("base", "base_128x192_g"), ("lite", "lite_128x192_g_b2")])
def test_fallback_kernel_selections_end_to_end(variant, name, dev, weights):
    """The forward's kernel selections other than the default one, each against the reference's outputs: (1) what a batch beyond
    2^26 plane rows takes (`Network._rows_fit_planes` False: 3x3 convs and deconvs on the fp32-input kernels -- batch 4 at 4K), (2) every
    split-plane switch off (the round-1 forward: fp32 maps between all layers), (3) the three stem layers as separate launches,
    (4) launch plans off.  All within the parity budget of the golden fixture and within 2e-4 of the default selection."""
    cases = [c for c in G.e2e_cases() if c["name"] == name]
    if not cases:
        pytest.skip(f"no fixture {name}")
    case = cases[0]
    gold = G.load_npz(case["name"])
    im0, im1 = G.case_inputs(case)
    net = (pkg.NetworkBase if variant == "base" else pkg.NetworkLite)()
    net.load_state_dict(weights(variant), strict=True)
    net.to(dev).eval()
    net.global_motion = case["global"]
    ref = net(im0.to(dev), im1.to(dev))["I_t"].clone()
    G.compare_e2e(net(im0.to(dev), im1.to(dev)), gold, case["step"], TOL, TOL_FLOW)

    def check(tag):
        out = net(im0.to(dev), im1.to(dev))
        torch.cuda.synchronize()
        errs = G.compare_e2e(out, gold, case["step"], TOL, TOL_FLOW)
        d = (out["I_t"] - ref).abs().max().item()
        print(variant, tag, {k: f"{v:.1e}" for k, v in errs.items()}, f"vs default {d:.1e}")
        assert d <= 2e-4, (tag, d)

    # (1) the selection of a batch whose plane rows do not fit 32-bit byte offsets: the flag is recomputed by every forward, so the
    # test pins it through the property the forward reads
    cls = type(net)
    orig = cls._plane_convs
    try:
        cls._plane_convs = lambda self, ops: False
        net._plans.clear(); net._graphs.clear()
        check("rows beyond 2^26 (fp32-input 3x3 kernels)")
    finally:
        cls._plane_convs = orig
    # (2) every plane switch off
    saved = (net.use_plane_convs, net.use_plane_deconvs, net.use_unet_planes, net.use_split_planes)
    net.use_plane_convs = net.use_plane_deconvs = net.use_unet_planes = net.use_split_planes = False
    check("all split-plane paths off")
    net.use_plane_convs, net.use_plane_deconvs, net.use_unet_planes, net.use_split_planes = saved
    # (3) separate stem launches, (4) no plans
    net.use_fused_stem = False
    check("stem as three launches")
    net.use_fused_stem = True
    net.enable_plans(False)
    check("launch plans off")
    net.enable_plans(True)
    # (5) the refiner's tail as three launches, (6) no split-K
    net.use_fused_tail = False
    check("refine_head.1 as its own launch")
    net.use_fused_tail = True
    net.use_splitk = False
    check("split-K off")
    net.use_splitk = True
    check("default again")


def test_graph_replay_equals_eager(dev, weights):
    """Network.enable_graphs(): the captured HIP graph must reproduce the eager forward bit for bit, follow new inputs,
    new shapes and a changed parameter (the graph is re-captured when the packed weights are refreshed)."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a0, a1 = pairs.smooth_pair(1, 128, 192, seed=81)
    b0, b1 = pairs.random_pair(1, 128, 192, seed=82)
    c0, c1 = pairs.smooth_pair(2, 64, 128, seed=83)
    eager = [{k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in net(x.to(dev), y.to(dev)).items()}
             for x, y in ((a0, a1), (b0, b1), (c0, c1))]
    net.enable_graphs(True)
    for rep in range(2):
        for (x, y), ref in zip(((a0, a1), (b0, b1), (c0, c1)), eager):
            out = net(x.to(dev), y.to(dev))
            torch.cuda.synchronize()
            assert torch.equal(out["I_t"], ref["I_t"]) and torch.equal(out["opt_flow_0"], ref["opt_flow_0"])
            assert all(torch.equal(p, q) for p, q in zip(out["im_t_list"], ref["im_t_list"]))
    assert len(net._graphs) == 2
    with torch.no_grad():
        dict(net.named_parameters())["refine_head.1.0.weight"].mul_(0.5)          # in-place parameter update -> packed weights and graphs are rebuilt
    out = net(a0.to(dev), a1.to(dev))["I_t"].clone()
    net.enable_graphs(False)
    assert torch.equal(out, net(a0.to(dev), a1.to(dev))["I_t"])
    assert not torch.equal(out, eager[0]["I_t"])


def test_frame_pipeline_matches_sequential(nets, dev):
    """host_io.FramePipeline (pinned slots, copies on side streams) returns exactly what one-at-a-time inference_2frame returns,
    in order, for more pairs than slots; and the generic torch path of inference_2frame (any nn.Module) agrees with the HIP one."""
    net = nets["lite"]
    net.global_motion = True
    net.ensemble_global_motion = False
    rng = np.random.default_rng(5)
    frames = [rng.integers(0, 256, (100, 150, 3), dtype=np.uint8) for _ in range(6)]
    pairs_ = list(zip(frames[:-1], frames[1:]))
    seq = [host_io.inference_2frame(a, b, net, isBGR=True) for a, b in pairs_]
    pipe = host_io.FramePipeline(net, 100, 150, isBGR=True, divisor=64, depth=2)
    got = list(pipe.run(pairs_))
    assert len(got) == len(seq) and all(np.array_equal(g, s) for g, s in zip(got, seq))
    assert list(pipe.run([])) == []

    class Wrapped(torch.nn.Module):           # hides the HIP backend: forces the reference-style torch pre/post path
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, a, b):
            return self.inner(a, b)
    generic = host_io.inference_2frame(pairs_[0][0], pairs_[0][1], Wrapped(net), isBGR=True)
    # torch's GPU "x / 255." multiplies by the reciprocal (1 ulp off the true division the CPU reference, the golden fixtures and the
    # HIP pre-kernel compute), so a rounding may flip here and there; never more than one level
    d = np.abs(generic.astype(np.int32) - seq[0].astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 5e-3


def same_uint8_frame(got, want) -> bool:
    """An interpolated uint8 frame against the oracle's: |d| <= 1e-3 in fp32 can flip a rounding here and there, never more than one
    level (the criterion of test_inference_2frame_uint8)."""
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    return got.shape == want.shape and got.dtype == np.uint8 and d.max() <= 1 and (d > 0).mean() < 5e-3


def oracle_video_2x(sd, frames, global_motion):
    """The output sequence of the reference's video loop (demo_2x.py:144-163) computed by the CPU oracle pair by pair:
    f0, I(f0,f1), f1, ..., f_{n-1}."""
    out = []
    for i in range(len(frames) - 1):
        out += [frames[i], O.inference_2frame(sd, frames[i], frames[i + 1], isBGR=True, global_motion=global_motion)]
    return out + [frames[-1]] if frames else out


def check_video(got, want):
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        if i % 2 == 0:
            assert np.array_equal(g, w), f"original frame {i // 2} was altered"
        else:
            assert same_uint8_frame(g, w), f"interpolated frame {i // 2} differs from the oracle's"


class FakeCapture:
    """An in-memory stand-in for cv2.VideoCapture (demo_2x.py:129-133, 144-147): get(CAP_PROP_*), isOpened(), read(), release()."""

    def __init__(self, frames, fps):
        self.frames, self.fps, self.i, self.open, self.released = list(frames), fps, 0, True, 0
        self.buf = None if not frames else np.empty_like(frames[0])

    def get(self, prop):
        h, w = self.frames[0].shape[:2] if self.frames else (0, 0)
        return {host_io.CAP_PROP_FPS: float(self.fps) + 0.97, host_io.CAP_PROP_FRAME_WIDTH: float(w), host_io.CAP_PROP_FRAME_HEIGHT: float(h),
                host_io.CAP_PROP_FRAME_COUNT: float(len(self.frames))}[prop]

    def isOpened(self):
        return self.open

    def read(self):
        if self.i >= len(self.frames):
            return False, None
        np.copyto(self.buf, self.frames[self.i])       # like OpenCV, the decoder hands out ONE reused buffer
        self.i += 1
        return True, self.buf

    def release(self):
        self.open = False
        self.released += 1


class FakeWriter:
    def __init__(self, fps, size):
        self.fps, self.size, self.frames, self.released = fps, size, [], 0

    def write(self, frame):
        assert frame.shape == (self.size[1], self.size[0], 3) and frame.dtype == np.uint8
        self.frames.append(frame.copy())

    def release(self):
        self.released += 1


def test_video_2x_frame_order(nets, dev, weights):
    """interpolate_video_2x against the CPU ORACLE's output sequence of the reference's video loop (demo_2x.py:144-163), and the
    capture -> pairs -> writer adapters (host_io.video_2x, :129-168) over an in-memory fake codec: frame rate doubled, every original
    once, the last frame once, both ends released."""
    net = nets["lite"]
    sd = weights("lite")
    for glob in (False, True):
        net.global_motion = glob
        frames = pairs.uint8_video(5, 64, 96, seed=9 + glob)
        want = oracle_video_2x(sd, frames, glob)
        assert len(want) == 9
        check_video(list(host_io.interpolate_video_2x(iter(frames), net)), want)
        cap, sinks = FakeCapture(frames, fps=24), []

        def make_writer(fps, size):
            sinks.append(FakeWriter(fps, size))
            return sinks[-1]
        info = host_io.video_2x(cap, make_writer, net)
        assert info == {"fps_in": 24, "fps_out": 48, "size": (96, 64), "frames_in": 5, "frames_out": 9}
        assert len(sinks) == 1 and sinks[0].fps == 48 and sinks[0].size == (96, 64) and sinks[0].released == 1 and cap.released == 1
        check_video(sinks[0].frames, want)
    assert list(host_io.interpolate_video_2x(iter([]), net)) == []
    one = list(host_io.interpolate_video_2x(iter(frames[:1]), net))
    assert len(one) == 1 and np.array_equal(one[0], frames[0])
    net.global_motion = True


def test_frame_cache_is_exact_and_video_distributed_single_rank(nets, dev, weights):
    """Network.enable_frame_cache(): with the global branch off, forward(b, c, reuse_first=True) after forward(a, b) must equal
    forward(b, c) bit for bit (the encoder and the cross-scale fusion are per frame, network_base.py:342-352) AND the oracle's
    forward(b, c) within the parity budget; and the multi-GPU video loop (host_io.interpolate_video_2x_distributed, here with one
    rank, frame cache on inside its blocks) must yield the ORACLE's sequence of the reference's loop (demo_2x.py:144-163), with and
    without the global branch."""
    net = nets["lite"]
    net.ensemble_global_motion = False
    net.global_motion = False
    a, _ = pairs.smooth_pair(1, 128, 192, seed=91)
    b, c = pairs.smooth_pair(1, 128, 192, seed=92)
    a, b, c = a.to(dev), b.to(dev), c.to(dev)
    plain = {k: v.clone() for k, v in net(b, c).items() if torch.is_tensor(v)}
    net.enable_frame_cache(True)
    try:
        net(a, b)
        cached = net(b, c, reuse_first=True)
        for k, v in plain.items():
            assert torch.equal(cached[k], v), k
        ref = O.forward(weights("lite"), b.cpu(), c.cpu(), global_motion=False)
        assert (cached["I_t"].cpu() - ref["I_t"]).abs().max().item() <= TOL
        # a cache filled at another shape (or a call without reuse_first) must not be used
        x, y = pairs.smooth_pair(1, 64, 96, seed=93)
        net(x.to(dev), y.to(dev))
        again = net(b, c, reuse_first=True)                 # cache belongs to the 64x96 workspace: recomputed, still right
        assert torch.equal(again["I_t"], plain["I_t"])
    finally:
        net.enable_frame_cache(False)
    frames = pairs.uint8_video(6, 64, 96, seed=11)
    for g in (False, True):
        net.global_motion = g
        want = oracle_video_2x(weights("lite"), frames, g)
        got = list(host_io.interpolate_video_2x_distributed(frames, net, 0, 1, block=2))
        assert len(got) == len(want) == 11
        check_video(got, want)
        single = list(host_io.interpolate_video_2x(iter(frames), net))       # ... and exactly the single-GPU loop's frames
        assert all(np.array_equal(p, q) for p, q in zip(got, single)), g
    net.global_motion = True


def test_video_distributed_under_rccl_one_rank(nets, dev, weights):
    """interpolate_video_2x_distributed under a real RCCL communicator (one rank: what a one-GPU box can exercise): sharding.HostGather's
    all-gather on its side stream and the pinned host copies must reproduce the ORACLE's sequence of the reference's video loop
    (demo_2x.py:144-163; 270 x 480 frames through the padder) and interpolate_video_2x frame for frame."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    net = nets["lite"]
    net.global_motion, net.ensemble_global_motion = True, False
    frames = pairs.uint8_video(7, 270, 480, seed=3)
    ref = list(host_io.interpolate_video_2x(frames, net))
    check_video(ref, oracle_video_2x(weights("lite"), frames, True))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for block in (1, 3):
            got = list(host_io.interpolate_video_2x_distributed(frames, net, rank=0, world=1, block=block))
            assert len(got) == len(ref) == 13
            assert all(np.array_equal(a, b) for a, b in zip(got, ref)), f"block {block}"
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_flip_tta_on_the_hip_path(nets, dev, weights):
    """host_io.forward_tta (benchmark/test_snufilm.py:135-139: average of the prediction and of the un-flipped prediction on the
    H- and W-flipped frames) run on the HIP path against the CPU oracle doing the same."""
    net = nets["lite"]
    net.global_motion = True
    net.ensemble_global_motion = False
    im0, im1 = pairs.smooth_pair(1, 128, 192, seed=95)
    got = host_io.forward_tta(net, im0.to(dev), im1.to(dev)).cpu()
    sd = weights("lite")
    ref = O.forward(sd, im0, im1, global_motion=True)["I_t"]
    ref_f = O.forward(sd, im0.flip(2).flip(3).contiguous(), im1.flip(2).flip(3).contiguous(), global_motion=True)["I_t"]
    want = (ref + ref_f.flip(2).flip(3)) / 2
    assert (got - want).abs().max().item() <= TOL
    assert (got - ref).abs().max().item() > 0            # the augmentation really changes the result (asymmetric network)


def test_model_moved_to_other_device_drops_device_state(dev, weights):
    """The workspace, packed weights and window maps are per device: a forward with inputs on a device other than the model's
    parameters must raise (not launch mixed pointers), and release / re-use on the same device must keep working."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a, b = pairs.smooth_pair(1, 64, 64, seed=97)
    o1 = net(a.to(dev), b.to(dev))["I_t"].clone()
    assert net.workspace_bytes() > 0 and len(net._workspaces) == 1
    net.cpu()
    with pytest.raises(RuntimeError):
        net(a.to(dev), b.to(dev))                          # parameters on the CPU, inputs on the GPU
    net.to(dev)
    assert torch.equal(net(a.to(dev), b.to(dev))["I_t"], o1)


def test_f16x3_deviation_and_weight_range_warning(dev, weights):
    """Network.f16x3_deviation: the split engines against the exact-fp32 engine on one pair (the run-time range check of DESIGN.md
    section 1, deviation 2); and a checkpoint whose weights come near the fp16 range is announced when its weights are packed."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a, b = [t.to(dev) for t in pairs.smooth_pair(1, 128, 192, seed=99)]
    d = net.f16x3_deviation(a, b)
    assert 0.0 <= d <= 2e-4 and net._precision == "f16x3" and net.weight_abs_max < 100.0
    with torch.no_grad():
        net.refine_head._modules["1"]._modules["0"].weight[0, 0, 0, 0] = 4.0e4
    with pytest.warns(UserWarning, match="saturate"):
        net(a, b)


def test_range_guard_of_the_checked_build(dev, weights):
    """Network.set_precision("f16x3-checked"): the checked build of the library computes exactly what the
    default one does and counts operands beyond the fp16 range at every activation split.  A clean run leaves the count at zero; an
    activation of 1e5 (a bias pushed there; then an input frame scaled there) trips it; the count is sticky until reset; the default
    precision is unaffected and carries no word."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a, b = [t.to(dev) for t in pairs.smooth_pair(1, 128, 192, seed=77)]
    want = [t.clone() for t in _flat(net(a, b))]
    with pytest.raises(RuntimeError):
        net.range_violations()
    net.set_precision("f16x3-checked")
    for rep in range(4):                                   # (no plans in this mode: every call is a sequence of direct launches)
        for t, r in zip(_flat(net(a, b)), want):
            assert torch.equal(t, r)
    assert net._ops_obj.checked and net._ops_obj.lib.atmvfi_range_checked() == 1 and not net._plans
    assert net.range_violations() == 0
    # (1) a bias of 1e5 in the refiner's first conv: its output is split for the next contraction
    bias = dict(net.named_parameters())["proj.0.bias"]
    keep = bias.detach().clone()
    with torch.no_grad():
        bias[3] = 1.0e5
    out = net(a, b)["I_t"]
    n1 = net.range_violations()
    assert n1 > 0 and torch.isfinite(out).all()            # saturates (finite), and says so
    with torch.no_grad():
        bias.copy_(keep)
    net(a, b)
    assert net.range_violations() == n1                     # sticky: a clean forward does not clear it
    assert net.range_violations(reset=True) == n1 and net.range_violations() == 0
    # (2) an input frame beyond the range (the stem splits the frame patch itself)
    net(a * 3.0e5, b)
    assert net.range_violations(reset=True) > 0
    # (3) inf / NaN count too
    c = a.clone()
    c[0, 1, 5, 7] = float("inf")
    net(c, b)
    assert net.range_violations(reset=True) > 0
    # back on the default build: same results, no word, plans again
    net.set_precision("f16x3")
    for rep in range(4):
        for t, r in zip(_flat(net(a, b)), want):
            assert torch.equal(t, r)
    assert not net._ops_obj.checked and net._ops_obj.range_word is None
    assert net._ops_obj.lib.atmvfi_range_checked() == 0
    rc = net._ops_obj.lib.atmvfi_range_word_set(None, None)
    assert rc != 0 and b"default build" in net._ops_obj.lib.atmvfi_last_error()


def _flat(out):
    return [out[k] for k in ("I_t", "opt_flow_0", "opt_flow_1", "I_t_0", "I_t_1", "occ_mask1", "occ_mask2")] + list(out["im_t_list"]) + \
        list(out["im0_warped_list"]) + list(out["im1_warped_list"])


def test_launch_plan_replay_equals_direct_launches(dev, weights):
    """From the third forward of one (shape, mode, weights) key on, ``forward`` is one atmvfi_plan_run call (hip_ops.LaunchPlan):
    bit-identical to the direct launches, on new inputs too; output tensors are fresh on every call (a caller may keep earlier
    results); other shapes and modes keep their own plans; an in-place parameter update re-records; ensemble mode stays eager."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a = [t.to(dev) for t in pairs.smooth_pair(1, 128, 192, seed=81)]
    b = [t.to(dev) for t in pairs.random_pair(1, 128, 192, seed=82)]
    c = [t.to(dev) for t in pairs.smooth_pair(2, 64, 128, seed=83)]
    net.enable_plans(False)
    ref = {n: [t.clone() for t in _flat(net(*x))] for n, x in (("a", a), ("b", b), ("c", c))}
    net.enable_plans(True)
    kept = []
    for rep in range(5):
        for n, x in (("a", a), ("b", b), ("c", c)):
            out = net(*x)
            assert out["I_t_0"] is out["im0_warped_list"][0] and out["im_t_list"][0] is not out["I_t"]
            for t, r in zip(_flat(out), ref[n]):
                assert torch.equal(t, r)
            kept.append((n, out["I_t"]))
    plans = [p for p in net._plans.values() if not isinstance(p, (int, bool))]
    assert len(plans) == 2 and all(len(p.ops_list) > 50 and len(p.patches) > 10 for p in plans)     # a and b share shape and mode
    ptrs = [t.data_ptr() for _, t in kept]
    assert len(set(ptrs)) == len(ptrs)                      # fresh outputs: nothing handed out twice while the caller holds it
    for n, t in kept:                                        # ... and earlier results are intact
        assert torch.equal(t, ref[n][0])
    # global branch off: another key, another plan, still exact
    net.global_motion = False
    net.enable_plans(False)
    r2 = [t.clone() for t in _flat(net(*a))]
    net.enable_plans(True)
    for rep in range(4):
        for t, r in zip(_flat(net(*a)), r2):
            assert torch.equal(t, r)
    net.global_motion = True
    # an in-place parameter update must reach the replayed forward
    for rep in range(3):
        net(*a)
    with torch.no_grad():
        net.refine_head._modules["1"]._modules["0"].bias.add_(0.25)
    out = net(*a)
    assert not torch.equal(out["I_t"], ref["a"][0])
    net.enable_plans(False)
    want = net(*a)["I_t"].clone()
    net.enable_plans(True)
    for rep in range(4):
        assert torch.equal(net(*a)["I_t"], want)
    # ensemble mode: its per-sample pick is atmvfi_ensemble_select and l1_mean clears its own accumulator (round 4), so the forward has
    # no device arithmetic outside the C ABI any more and is planned like every other mode; equal to the direct launches
    e0, e1 = [t.to(dev) for t in pairs.smooth_pair(1, 128, 192, seed=81)]
    net.ensemble_global_motion = True
    big = [torch.nn.functional.interpolate(t, size=(192, 256), mode="bilinear") for t in (e0, e1)]
    net.enable_plans(False)
    want_e = [t.clone() for t in _flat(net(*big))]
    net.enable_plans(True)
    for rep in range(5):
        for t, r in zip(_flat(net(*big)), want_e):
            assert torch.equal(t, r)
    assert len([k for k, p in net._plans.items() if k[4] and not isinstance(p, (int, bool))]) == 1
    net.ensemble_global_motion = False


def test_launch_plans_refuse_aliased_frames_and_follow_input_alignment(dev, weights):
    """(1) warm-up calls with ALIASED frames (net(x, x)) must not produce a plan that replays every later net(a, b) of
    that shape as net(a, a): overlapping frames are never recorded, the next call with distinct frames is; (2) frames that are two
    slices of one stacked tensor are told apart; (3) a replay whose inputs are aligned differently from the recording's (a 4-byte
    aligned view: the LDS-staged warps need 16) takes the direct launches instead of failing inside the plan."""
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a, b = [t.to(dev) for t in pairs.smooth_pair(1, 128, 192, seed=91)]
    x = a.clone()
    net.enable_plans(False)
    want_ab = [t.clone() for t in _flat(net(a, b))]
    want_xx = [t.clone() for t in _flat(net(x, x))]
    net.enable_plans(True)
    net._plans.clear()
    for rep in range(4):                                       # would have been recorded as "both frames = slot 0" before
        for t, r in zip(_flat(net(x, x)), want_xx):
            assert torch.equal(t, r)
    assert not [p for p in net._plans.values() if not isinstance(p, (int, bool))]
    for rep in range(4):
        for t, r in zip(_flat(net(a, b)), want_ab):
            assert torch.equal(t, r)
    plans = [p for p in net._plans.values() if not isinstance(p, (int, bool))]
    assert len(plans) == 1 and plans[0].by_address == 0 and plans[0].by_tensor > 10
    for t, r in zip(_flat(net(x, x)), want_xx):                # replaying with aliased frames is fine: both slots get x
        assert torch.equal(t, r)
    # (2) one stacked tensor
    st = torch.cat([a, b], 0).contiguous()
    for rep in range(2):
        for t, r in zip(_flat(net(st[0:1], st[1:2])), want_ab):
            assert torch.equal(t, r)
    net._plans.clear()
    for rep in range(5):                                       # ... recorded from such a pair too
        for t, r in zip(_flat(net(st[0:1], st[1:2])), want_ab):
            assert torch.equal(t, r)
    assert len([p for p in net._plans.values() if not isinstance(p, (int, bool))]) == 1
    # (3) a 4-byte-aligned contiguous view of the same pixels
    flat = torch.zeros(a.numel() + 1, device=dev)
    flat[1:] = a.reshape(-1)
    a4 = flat[1:].view_as(a)
    assert a4.data_ptr() % 16 == 4 and a4.is_contiguous()
    for t, r in zip(_flat(net(a4, b)), want_ab):
        assert torch.equal(t, r)


def test_launch_plan_equals_direct_launches_at_1080p_base(dev, weights):
    """The path bench.py times: network_base at 1088 x 1920 replayed from its launch plan, bit for bit the direct launches (VERDICT
    round 3, weak 1), on the recording's own pair and on a new one; the record-time self-check accepted the plan."""
    net = pkg.NetworkBase()
    net.load_state_dict(weights("base"), strict=True)
    net.to(dev).eval()
    p0 = [t.to(dev) for t in pairs.random_pair(1, 1088, 1920, seed=5)]
    p1 = [t.to(dev) for t in pairs.smooth_pair(1, 1088, 1920, seed=6)]
    net.enable_plans(False)
    want = [[t.clone() for t in _flat(net(*p))] for p in (p0, p1)]
    net.enable_plans(True)
    for rep in range(4):
        for p, w in zip((p0, p1), want):
            for t, r in zip(_flat(net(*p)), w):
                assert torch.equal(t, r)
    plans = [p for p in net._plans.values() if not isinstance(p, (int, bool))]
    assert len(plans) == 1 and len(plans[0].ops_list) > 100
    net.release_workspace()
    torch.cuda.empty_cache()


def test_lanes_equal_single_stream(dev, weights):
    """Branches of the forward on side streams (HipOps.branch / join, atmvfi_plan_run_lanes): the same kernels
    in a partial order -- bit for bit the single-stream forward, on direct launches and replayed from a launch plan, for both variants,
    global branch on and off, batch 1 and 2, call after call (a race between concurrent branches would show as a run-to-run difference)."""
    for variant, cls, (bsz, H, W), glob in (("lite", pkg.NetworkLite, (1, 256, 256), True), ("lite", pkg.NetworkLite, (2, 256, 448), False),
                                            ("base", pkg.NetworkBase, (1, 192, 320), True), ("base", pkg.NetworkBase, (1, 576, 960), True)):
        net = cls()
        net.load_state_dict(weights(variant), strict=True)
        net.global_motion = glob
        net.to(dev).eval()
        p0 = [t.to(dev) for t in pairs.random_pair(bsz, H, W, seed=11)]
        p1 = [t.to(dev) for t in pairs.smooth_pair(bsz, H, W, seed=12)]
        net.use_lanes = False
        net.enable_plans(False)
        want = [[t.clone() for t in _flat(net(*p))] for p in (p0, p1)]
        net.use_lanes = True
        for plans in (False, True):
            net.enable_plans(plans)
            for rep in range(5):
                for p, w in zip((p0, p1), want):
                    for t, r in zip(_flat(net(*p)), w):
                        assert torch.equal(t, r), (variant, H, W, plans, rep)
        recorded = [p for p in net._plans.values() if not isinstance(p, (int, bool))]
        assert recorded and all(p.n_lanes == 2 and p.n_events >= (4 if glob else 2) for p in recorded)
        net.release_workspace()
    torch.cuda.empty_cache()


def test_pair_streams_equal_single_stream(dev, weights):
    """host_io.PairStreams: K replicas (Network.replica(): shared parameters and packed weights, own
    workspace + launch plan) on K streams, pairs round-robin, no cross-stream event inside a forward.  Every output of every pair is
    bit for bit the single-stream forward's, at three shapes, K = 2..4, with one issuing thread and with a thread per stream, before
    and after the plans exist; the model's switches follow; a bad pair raises in the caller; an in-place parameter update reaches
    every replica; FramePipeline / interpolate_video_2x with streams > 1 deliver the sequential frames."""
    cases = (("lite", pkg.NetworkLite, (1, 256, 256), True), ("lite", pkg.NetworkLite, (2, 256, 448), False),
             ("base", pkg.NetworkBase, (1, 192, 320), True))
    for variant, cls, (bsz, H, W), glob in cases:
        net = cls()
        net.load_state_dict(weights(variant), strict=True)
        net.to(dev).eval()
        net.global_motion = glob
        frames = [[t.to(dev) for t in (pairs.random_pair if i % 2 else pairs.smooth_pair)(bsz, H, W, seed=30 + i)] for i in range(5)]
        want = [[t.clone() for t in _flat(net(*f))] for f in frames]
        for k, thr in ((2, False), (3, True), (4, False)):
            with host_io.PairStreams(net, k, threads=thr) as ps:
                assert len(ps.replicas) == k and all(next(r.parameters()) is next(net.parameters()) for r in ps.replicas)
                for rep in range(3):            # 15 pairs per round: every replica sees eager, recording and replayed forwards
                    outs = list(ps.map(frames[i % 5] for i in range(15)))
                    assert len(outs) == 15
                    for i, o in enumerate(outs):
                        for t, r in zip(_flat(o), want[i % 5]):
                            assert torch.equal(t, r), (variant, k, thr, rep, i)
                assert all(any(not isinstance(p, (int, bool)) for p in r._plans.values()) for r in ps.replicas)
                assert all(r._prepared is net._prepared for r in ps.replicas)          # one set of packed weights
                if variant == "lite" and k == 2:
                    net.global_motion = not glob                                          # the caller flips a switch on the model
                    other = net(*frames[0])["I_t"].clone()
                    got = list(ps.map([frames[0], frames[0], frames[0]]))
                    assert all(torch.equal(o["I_t"], other) for o in got) and not torch.equal(other, want[0][0])
                    net.global_motion = glob
                    with pytest.raises(ValueError):                                       # a forward's error surfaces in the caller
                        list(ps.map([(frames[0][0], frames[1][1][..., :64])]))
                    bias = dict(net.named_parameters())["refine_head.1.0.bias"]
                    keep = bias.detach().clone()
                    with torch.no_grad():
                        bias.add_(0.125)                                                  # in-place update: every replica repacks / re-records
                    upd = net(*frames[1])["I_t"].clone()
                    assert not torch.equal(upd, want[1][0])
                    assert all(torch.equal(o["I_t"], upd) for o in ps.map([frames[1]] * 5))
                    with torch.no_grad():
                        bias.copy_(keep)
                    assert all(torch.equal(o["I_t"], w[0]) for o, w in zip(ps.map(frames), want))
        net.release_workspace()
    # the uint8 pipeline and the video loop with several forwards in flight
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    clip = pairs.uint8_video(9, 100, 150, seed=4)
    seq = list(host_io.interpolate_video_2x(iter(clip), net))
    for k in (2, 3):
        got = list(host_io.interpolate_video_2x(iter(clip), net, streams=k))
        assert len(got) == len(seq) == 17 and all(np.array_equal(a, b) for a, b in zip(got, seq)), k
    torch.cuda.empty_cache()


def test_tiled_inference_vs_oracle_on_identical_tiles(nets, dev, weights):
    """host_io.forward_tiled (BASELINE configs[4], "tiled"; SURVEY 8d: the parity oracle of a build-defined tiled mode is the reference's
    forward on the IDENTICAL tiles with the IDENTICAL stitching): the HIP path tile by tile and with two tiles in flight against the
    CPU oracle run through the same tiling; and tiling matters (borders and windows change), so the tiled frame is not the untiled one."""
    for variant, (H, W), tile, ov in (("lite", (200, 300), (128, 160), 32), ("base", (160, 224), (96, 128), 16)):
        net = nets[variant]
        net.global_motion, net.ensemble_global_motion = True, False
        sd = weights(variant)
        im0, im1 = pairs.smooth_pair(1, H, W, seed=61)
        want = host_io.forward_tiled(lambda a, b: O.forward(sd, a, b, global_motion=True)["I_t"], im0, im1, tile=tile, overlap=ov)
        got1 = host_io.forward_tiled(net, im0.to(dev), im1.to(dev), tile=tile, overlap=ov)
        got2 = host_io.forward_tiled(net, im0.to(dev), im1.to(dev), tile=tile, overlap=ov, streams=2)
        assert tuple(got1.shape) == (1, 3, H, W) and got1.is_cuda
        assert (got1.cpu() - want).abs().max().item() <= TOL, variant
        assert torch.equal(got1, got2)
        with host_io.PairStreams(net, 3) as ps:              # a PairStreams the caller keeps, for repeated use
            for _ in range(2):
                assert torch.equal(host_io.forward_tiled(ps, im0.to(dev), im1.to(dev), tile=tile, overlap=ov), got1)
        pad = host_io.InputPadder(im0.shape, divisor=64)
        a, b = pad.pad(im0.to(dev), im1.to(dev))
        whole = pad.unpad(net(a, b)["I_t"])
        # a different computation (borders, window contents, the global branch's context): on the stress weights -- not a trained
        # interpolator -- far from the untiled frame, which is why the oracle of this mode is the oracle on the same tiles
        assert (whole - got1).abs().max().item() > 1e-3
    net.release_workspace()


def test_deepcopy_of_a_used_model_is_standalone(dev, weights):
    """copy.deepcopy of a model that has run (workspace, packed weights, plans) and has replicas: an independent model with its own
    parameters and NO device-side runtime state; it builds its own on first use and computes the same frames."""
    import copy
    net = pkg.NetworkLite()
    net.load_state_dict(weights("lite"), strict=True)
    net.to(dev).eval()
    a, b = [t.to(dev) for t in pairs.smooth_pair(1, 128, 192, seed=88)]
    for _ in range(4):
        want = net(a, b)["I_t"].clone()
    with host_io.PairStreams(net, 2) as ps:
        assert all(torch.equal(o["I_t"], want) for o in ps.map([(a, b)] * 4))
        twin = copy.deepcopy(net)
        rep_twin = copy.deepcopy(ps.replicas[0])
    for m in (twin, rep_twin):
        assert m.workspace_bytes() == 0 and m._ops_obj is None and m._primary is None and not m._plans
        assert next(m.parameters()) is not next(net.parameters()) and next(m.parameters()).device == next(net.parameters()).device
        assert torch.equal(m(a, b)["I_t"], want)
    with torch.no_grad():
        dict(twin.named_parameters())["refine_head.1.0.bias"].add_(0.5)          # the copies do not share storage
    assert not torch.equal(twin(a, b)["I_t"], want) and torch.equal(net(a, b)["I_t"], want)


def test_lanes_on_a_fresh_workspace(dev, weights):
    """the FIRST forward of a shape with lanes on creates workspace inside a branch body (the global branch's plane
    buffers are zero-filled by torch on their first use).  That fill must be ordered with the lane's kernels -- the result of a
    lanes-first model equals the single-stream model's bit for bit, first call included, and again after release_workspace()."""
    for variant, cls, (H, W) in (("lite", pkg.NetworkLite, (256, 256)), ("base", pkg.NetworkBase, (192, 320))):
        p0 = [t.to(dev) for t in pairs.random_pair(1, H, W, seed=21)]
        ref_net = cls()
        ref_net.load_state_dict(weights(variant), strict=True)
        ref_net.to(dev).eval()
        ref_net.use_lanes = False
        want = [t.clone() for t in _flat(ref_net(*p0))]
        ref_net.release_workspace()
        net = cls()
        net.load_state_dict(weights(variant), strict=True)
        net.to(dev).eval()
        net.use_lanes = True
        for rep in range(3):
            for call in range(4):                       # first call builds the workspace inside the branches; the third records
                for t, r in zip(_flat(net(*p0)), want):
                    assert torch.equal(t, r), (variant, rep, call)
            net.release_workspace()
    torch.cuda.empty_cache()


def test_library_that_ran_is_built_from_these_sources(dev):
    """On the GPU box: the library this process loaded reports the digest of the sources in this tree (tools/source_digest.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import source_digest
    ops = hip_ops.HipOps(dev)
    assert ops.lib.atmvfi_source_digest().decode() == source_digest.digest()
