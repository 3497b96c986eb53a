"""GPU: every HIP kernel through the C ABI against a plain PyTorch fp32 restatement of the
same op contract (tests/cpu_ops.py) on seeded inputs, including ragged channel counts,
channel-slice views, row groups, scatter maps, padded/shifted windows and out-of-range flows.
Tolerances are absolute on O(1) data: 2e-5 for bandwidth ops, 1e-4 for contractions
(fp32 MFMA = k-ordered fmaf chain vs oneDNN's blocked summation)."""
import importlib
import os

import numpy as np
import pytest
import torch

import golden_util as G
from cpu_ops import CpuOps

pytestmark = pytest.mark.gpu

hip_ops = importlib.import_module("atm-vfi_amd.hip_ops")
windows = importlib.import_module("atm-vfi_amd.windows")
GEMM_CONV, GEMM_LINEAR, GEMM_DECONV = 0, 1, 2


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def hip(dev):
    return hip_ops.HipOps(dev)


@pytest.fixture(scope="module")
def cpu():
    return CpuOps()


def rnd(gen, *shape, scale=1.0):
    return (torch.rand(*shape, generator=gen) * 2 - 1) * scale


def maxdiff(a, b):
    return (a.detach().float().cpu() - b.detach().float().cpu()).abs().max().item()


# ------------------------------------------------------------------ contractions
CONV_CASES = [
    # cin, cout, k, stride, pad, dil, H, W, in_off, in_ld, out_off, out_ld, N
    (3, 24, 3, 1, 1, 1, 20, 28, 0, 4, 0, 24, 2),          # first layer: NHWC4 input, Cin=3 (direct vector-ALU kernel)
    (3, 16, 3, 1, 1, 1, 21, 19, 0, 4, 8, 32, 1),          # lite's first layer, written into a channel slice, ragged tiles
    (3, 32, 3, 1, 1, 1, 16, 16, 0, 4, 0, 32, 1),
    (24, 48, 3, 2, 1, 1, 20, 28, 0, 24, 0, 48, 2),
    (48, 48, 3, 4, 1, 1, 24, 40, 0, 48, 96, 384, 2),       # fusion conv, stride 4, writes a slice
    (48, 48, 3, 4, 2, 2, 24, 40, 0, 48, 144, 384, 2),      # stride 4, dilation 2
    (13, 21, 3, 1, 1, 1, 9, 11, 4, 24, 8, 40, 1),          # ragged Cin/Cout, offsets
    (101, 101, 3, 1, 1, 1, 16, 24, 0, 104, 0, 116, 1),     # decoder tail shape (7 n-tiles)
    (197, 197, 3, 1, 1, 1, 8, 12, 0, 200, 128, 328, 1),    # 13 n-tiles -> two n-blocks
    (389, 389, 3, 1, 1, 1, 6, 6, 0, 392, 256, 648, 1),     # 25 n-tiles -> five n-blocks
    (64, 5, 1, 1, 0, 1, 10, 14, 0, 64, 768, 776, 2),       # 1x1 motion head into the decoder input
    (116, 64, 3, 1, 1, 1, 12, 20, 0, 116, 64, 128, 1),
    (33, 20, 3, 1, 1, 1, 10, 12, 0, 36, 0, 20, 1),         # one-channel tail: tap-packed tail block of the 3x3 kernel
    (104, 48, 3, 1, 1, 1, 19, 17, 0, 104, 0, 48, 2),       # eight-channel tail, ragged tile borders
    (40, 24, 3, 2, 1, 1, 10, 12, 0, 40, 0, 24, 1),         # same tail through the strided (generic GEMM) path
    (256, 128, 3, 2, 1, 1, 12, 20, 64, 328, 0, 128, 1),    # down2: reads [feat1 | dec1[:192]] slice
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"cin{c[0]}_cout{c[1]}_k{c[2]}_s{c[3]}_d{c[5]}")
def test_conv2d(case, hip, cpu, dev):
    cin, cout, k, stride, pad, dil, H, W, ioff, ild, ooff, old, N = case
    g = torch.Generator().manual_seed(cin * 131 + cout)
    buf = rnd(g, N, H, W, ild)
    w = rnd(g, cout, cin, k, k, scale=1.0 / np.sqrt(cin * k * k))
    bias = rnd(g, cout, scale=0.2)
    slope = torch.rand(cout, generator=g) * 0.4
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    for use_prelu, precision in ((True, "f32"), (False, "f32"), (True, "f16x3"), (False, "f16x3")):
        obuf_c = torch.full((N, Ho, Wo, old), 7.0)
        cpu.conv(buf[..., ioff:ioff + cin], cpu.pack_weight(GEMM_CONV, w), obuf_c[..., ooff:ooff + cout], stride, pad, dil,
                 bias, slope if use_prelu else None)
        dbuf = buf.to(dev)
        obuf_g = torch.full((N, Ho, Wo, old), 7.0, device=dev)
        pw = hip.pack_weight(GEMM_CONV, w.to(dev))
        hip.precision = precision
        try:
            hip.conv(dbuf[..., ioff:ioff + cin], pw, obuf_g[..., ooff:ooff + cout], stride, pad, dil, bias.to(dev),
                     slope.to(dev) if use_prelu else None)
        finally:
            hip.precision = "f16x3"
        torch.cuda.synchronize()
        # the whole buffer is compared: channels outside the view must stay untouched (7.0).
        # f16x3 (3x3/s1 only) keeps ~22 significand bits: same budget as the fp32-MFMA path.
        assert maxdiff(obuf_g, obuf_c) <= 1e-4


def test_conv3x3_f16x3_large_and_ragged(hip, cpu, dev):
    """Split-precision halo kernel: partial 16x16 tiles, image borders, several images, large
    magnitudes, tiny magnitudes (lo' is scaled so it never goes subnormal) and operands beyond the
    fp16 range, which SATURATE at +-65504*(1+2^-10) instead of turning into inf/NaN (documented limit)."""
    g = torch.Generator().manual_seed(77)
    for (cin, cout, H, W, N, scale, overflow) in ((48, 37, 23, 41, 2, 1.0, False), (773 // 4, 64, 17, 16, 1, 1.0, False),
                                                  (32, 16, 16, 16, 1, 3.0e4, False), (32, 16, 16, 16, 1, 3.0e4, True),
                                                  (64, 128, 33, 18, 1, 1e-3, False)):
        r4 = lambda c: (c + 3) // 4 * 4
        x = rnd(g, N, H, W, r4(cin), scale=scale)[..., :cin]
        if overflow:
            x[0, 3, 3, :4] = torch.tensor([1.0e5, -1.2e5, 65504.0, 7.0e4])
        w = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin))
        bias = rnd(g, cout, scale=0.2 * scale)
        oc = torch.empty(N, H, W, r4(cout))[..., :cout]
        cpu.conv(x, cpu.pack_weight(GEMM_CONV, w), oc, 1, 1, 1, bias, None)
        og = torch.empty(N, H, W, r4(cout), device=dev)[..., :cout]
        xg = torch.zeros(N, H, W, r4(cin), device=dev)
        xg[..., :cin] = x.to(dev)
        hip.conv(xg[..., :cin], hip.pack_weight(GEMM_CONV, w.to(dev)), og, 1, 1, 1, bias.to(dev), None)
        torch.cuda.synchronize()
        assert hip.precision == "f16x3"
        assert not torch.isnan(og).any() and not torch.isinf(og).any()
        xr = x.clamp(-65504.0 * (1 + 2.0 ** -10), 65504.0 * (1 + 2.0 ** -10)) if overflow else x
        ref64 = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2).double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        err_hip = (og.cpu().double() - ref64).abs().max().item()
        err_cpu32 = (oc.double() - ref64).abs().max().item()
        if overflow:
            assert err_hip <= 0.5, (err_hip, err_cpu32)       # equals the convolution of the saturated input
        else:
            # within a small factor of what plain fp32 arithmetic itself loses against fp64
            assert err_hip <= max(8 * err_cpu32, 2e-6 * scale), (cin, cout, err_hip, err_cpu32)


DECONV_CASES = [(773, 389, 5, 7, 776, 392, 1), (13, 21, 6, 5, 16, 24, 2), (256, 128, 6, 10, 256, 128, 1), (128, 64, 4, 4, 328, 128, 2)]


@pytest.mark.parametrize("case", DECONV_CASES, ids=lambda c: f"cin{c[0]}_cout{c[1]}")
def test_deconv2x2(case, hip, cpu, dev):
    cin, cout, H, W, ild, old, N = case
    g = torch.Generator().manual_seed(cin + 7 * cout)
    buf = rnd(g, N, H, W, ild)
    w = rnd(g, cin, cout, 2, 2, scale=1.0 / np.sqrt(cin))
    bias = rnd(g, cout, scale=0.2)
    slope = torch.rand(cout, generator=g) * 0.4
    inslope = torch.rand(cin, generator=g) * 0.4
    for use_in in (False, True):
        oc = torch.full((N, 2 * H, 2 * W, old), 7.0)
        cpu.deconv(buf[..., :cin], cpu.pack_weight(GEMM_DECONV, w), oc[..., :cout], bias, slope,
                   cpu.pad_channels(inslope) if use_in else None)
        og = torch.full((N, 2 * H, 2 * W, old), 7.0, device=dev)
        hip.deconv(buf.to(dev)[..., :cin], hip.pack_weight(GEMM_DECONV, w.to(dev)), og[..., :cout], bias.to(dev), slope.to(dev),
                   hip.pad_channels(inslope.to(dev)) if use_in else None)
        torch.cuda.synchronize()
        assert maxdiff(og, oc) <= 1e-4


def test_linear_groups_scatter_residual(hip, cpu, dev):
    g = torch.Generator().manual_seed(3)
    frames, h, w, ws, shift, C = 4, 6, 10, 4, 2, 48
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    mw = geo.row_map.numel()
    x = rnd(g, mw, C)
    res = rnd(g, mw, C)
    wt = rnd(g, 72, C, scale=0.2)
    bias = rnd(g, 72, scale=0.2)
    # (a) scatter through the window map into a frame-stacked (grouped) destination slice
    B = frames // 2
    dst_c = torch.full((B, h, w, 8 + 2 * 72), 7.0)
    dst_g = dst_c.to(dev)

    def stacked(buf):
        b, hh, ww, ld = buf.shape
        return buf.reshape(b * hh * ww, ld)[:, 8:8 + 144].unflatten(1, (2, 72)).permute(1, 0, 2)
    cpu.linear(x, cpu.pack_weight(GEMM_LINEAR, wt), stacked(dst_c), bias, None, geo.row_map)
    hip.linear(x.to(dev), hip.pack_weight(GEMM_LINEAR, wt.to(dev)), stacked(dst_g), bias.to(dev), None, geo.row_map.to(dev))
    torch.cuda.synchronize()
    assert not torch.isnan(dst_c).any()
    assert maxdiff(dst_g, dst_c) <= 1e-4
    # (b) grouped input + residual, plain output
    wt2 = rnd(g, C, 72, scale=0.2)
    res2 = rnd(g, 2 * B * h * w, C)
    out_c = torch.empty(2 * B * h * w, C)
    out_g = torch.empty(2 * B * h * w, C, device=dev)
    cpu.linear(stacked(dst_c), cpu.pack_weight(GEMM_LINEAR, wt2), out_c, None, res2)
    hip.linear(stacked(dst_g), hip.pack_weight(GEMM_LINEAR, wt2.to(dev)), out_g, None, res2.to(dev))
    torch.cuda.synchronize()
    assert maxdiff(out_g, out_c) <= 1e-4


def test_gemm_rejects_bad_views(hip, dev):
    w = hip.pack_weight(GEMM_CONV, torch.rand(8, 6, 3, 3, device=dev))
    x = torch.rand(1, 8, 8, 7, device=dev)          # ld = 7: not a multiple of 4
    with pytest.raises(RuntimeError):
        hip.conv(x[..., :6], w, torch.empty(1, 8, 8, 8, device=dev), 1, 1, 1)
    with pytest.raises(TypeError):
        hip.conv(torch.rand(1, 8, 8, 8)[..., :6], w, torch.empty(1, 8, 8, 8, device=dev), 1, 1, 1)


# ------------------------------------------------------------------ transformer pieces
def test_layernorm_gather_groups(hip, cpu, dev):
    g = torch.Generator().manual_seed(5)
    frames, h, w, ws, shift, C = 2, 5, 6, 4, 2, 224
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    assert (geo.row_map < 0).any()
    src = rnd(g, 1, h, w, 8 + 2 * C, scale=3.0)
    gamma, beta = 1 + rnd(g, C, scale=0.2), rnd(g, C, scale=0.2)

    def stacked(buf):
        return buf.reshape(h * w, 8 + 2 * C)[:, 8:].unflatten(1, (2, C)).permute(1, 0, 2)
    out_c = torch.empty(geo.row_map.numel(), C)
    out_g = torch.empty(geo.row_map.numel(), C, device=dev)
    cpu.layernorm(stacked(src), out_c, gamma, beta, geo.row_map)
    hip.layernorm(stacked(src.to(dev)), out_g, gamma.to(dev), beta.to(dev), geo.row_map.to(dev))
    torch.cuda.synchronize()
    assert maxdiff(out_g, out_c) <= 2e-5
    plain = rnd(g, 37, 672, scale=2.0)
    g2, b2 = 1 + rnd(g, 672, scale=0.2), rnd(g, 672, scale=0.2)
    oc, og = torch.empty(37, 672), torch.empty(37, 672, device=dev)
    cpu.layernorm(plain, oc, g2, b2)
    hip.layernorm(plain.to(dev), og, g2.to(dev), b2.to(dev))
    assert maxdiff(og, oc) <= 2e-5


@pytest.mark.parametrize("shape", [(2, 7, 9, 448), (1, 19, 37, 704), (2, 5, 6, 100), (1, 16, 16, 64)],
                         ids=lambda s: "x".join(map(str, s)))
def test_dwconv_gelu(shape, hip, cpu, dev):
    """C % 64 == 0 -> sliding-window kernel (partial strips / x-blocks / image borders); otherwise per-pixel kernel."""
    n, h, w_, c = shape
    g = torch.Generator().manual_seed(6 + c)
    x = rnd(g, n, h, w_, c, scale=2.0)
    w = rnd(g, c, 1, 3, 3, scale=0.5)
    b = rnd(g, c, scale=0.3)
    oc, og = torch.empty(n, h, w_, c), torch.full((n, h, w_, c), 9.0, device=dev)
    cpu.dwconv_gelu(x, oc, w, b)
    hip.dwconv_gelu(x.to(dev), og, hip.pack_dw_weight(w.to(dev)), b.to(dev))
    assert maxdiff(og, oc) <= 2e-5


@pytest.mark.parametrize("shape", [(2, 34, 24, 192), (1, 17, 37, 448), (2, 16, 9, 64), (1, 32, 40, 128), (1, 24, 5, 64), (2, 19, 21, 320),
                                   (1, 11, 8, 64), (1, 8, 64, 64), (1, 136, 16, 1536)], ids=lambda s: "x".join(map(str, s)))
def test_dwconv_gelu_dma_planes(shape, hip, dev):
    """Plane sink only -> the LDS-DMA kernel (a private row ring per wave, counted vmcnt): strips of 17 / 16 / 8 rows that divide H, the
    overlapping last strip when none does, x-groups cut by the image edge, W < 8.  Its
    planes are the exact split of the sliding-window kernel's fp32 rows (same arithmetic in the same order), Inf in an edge column
    included (zero padding adds nothing: ADVICE round 4)."""
    n, h, w_, c = shape
    g = torch.Generator().manual_seed(60 + h + c)
    x = rnd(g, n, h, w_, c, scale=2.0).to(dev)
    if h == 19:      # a channel slice of a wider map (row pitch > C): the DMA sources step by the pitch
        wide = torch.full((n, h, w_, c + 64), 7.0, device=dev)
        wide[..., 64:] = x
        x = wide[..., 64:]
    x[0, h // 2, 0, 3] = float("inf")
    x[-1, 0, w_ - 1, c - 1] = float("-inf")
    wt = hip.pack_dw_weight(rnd(g, c, 1, 3, 3, scale=0.5).to(dev))
    b = rnd(g, c, scale=0.3).to(dev)
    of = torch.empty(n, h, w_, c, device=dev)
    hip.dwconv_gelu(x, of, wt, b)                      # fp32 rows: the sliding-window (or per-pixel) kernel
    p = hip_ops.Planes.alloc(n * h * w_, c, dev)
    hip.dwconv_gelu(x, None, wt, b, planes=p)
    torch.cuda.synchronize()
    rows = of.reshape(-1, c)
    hi = torch.where(torch.isinf(rows), rows, rows.clamp(-65504, 65504)).half()      # (finite values saturate, an infinity stays one)
    lo = ((rows - hi.float()) * 1024).clamp(-65504, 65504).half()
    got = p.to_rows()
    same = lambda a, b_: torch.equal(torch.nan_to_num(a.float(), nan=-7.0), torch.nan_to_num(b_.float(), nan=-7.0))
    assert same(got[0, :, :c], hi) and same(got[1, :, :c], lo)
    assert torch.isinf(rows).sum() > 0


ATTN_CASES = [
    # ws, hd, frames, h, w, shift, cross
    (8, 48, 2, 16, 24, 0, True), (8, 48, 2, 16, 24, 4, True), (8, 28, 2, 12, 20, 4, True),     # 12x20 -> pad 16x24 + shift
    (12, 84, 2, 12, 20, 6, True), (12, 44, 4, 4, 4, 0, True), (12, 44, 4, 4, 4, 6, True),      # 4x4 -> 12x12
    (8, 48, 4, 16, 16, 4, False), (7, 16, 2, 32, 32, 3, True), (16, 24, 2, 16, 32, 8, False),
    (4, 8, 2, 8, 8, 2, True),
]


ATTN_CASES_X3 = [(8, 128, 2, 8, 16, 4, True), (16, 64, 2, 16, 16, 8, True), (10, 36, 2, 10, 20, 5, False), (14, 20, 2, 14, 14, 7, True)]


@pytest.mark.parametrize("engine", ["f16x3", "f32"])
@pytest.mark.parametrize("case", ATTN_CASES + ATTN_CASES_X3,
                         ids=lambda c: f"ws{c[0]}_hd{c[1]}_{c[3]}x{c[4]}_s{c[5]}_{'x' if c[6] else 'self'}")
def test_window_attention(case, engine, hip, cpu, dev):
    """Both kernels behind the entry points: atmvfi_window_attention_f16x3 (what the f16x3 forward calls) and the exact-fp32
    atmvfi_window_attention."""
    ws, hd, frames, h, w, shift, cross = case
    hip = hip_ops.HipOps(dev)
    hip.attention_f16x3 = engine == "f16x3"
    heads = 8
    C = heads * hd
    g = torch.Generator().manual_seed(ws * 100 + hd + shift)
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    bw = frames * geo.n_windows
    n = ws * ws
    qkv = rnd(g, bw * n, 3 * C, scale=1.5)
    oc, og = torch.empty(bw * n, C), torch.full((bw * n, C), 9.0, device=dev)
    mc = torch.empty(bw * n, heads, 2) if cross else None
    mg = torch.full((bw * n, heads, 2), 9.0, device=dev) if cross else None
    kv_shift = bw // 2 if cross else 0
    cpu.window_attention(qkv, oc, mc, geo.labels, bw, geo.n_windows, ws, heads, hd, kv_shift)
    hip.window_attention(qkv.to(dev), og, mg, None if geo.labels is None else geo.labels.to(dev), bw, geo.n_windows, ws,
                         heads, hd, kv_shift)
    torch.cuda.synchronize()
    assert maxdiff(og, oc) <= 1e-4
    if cross:
        assert maxdiff(mg, mc) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["f16x3", "f32"])
@pytest.mark.parametrize("case", [(8, 32, 4, 13, 3), (8, 48, 6, 21, 0), (12, 20, 2, 5, 4), (5, 64, 3, 37, 11)],
                         ids=lambda c: f"ws{c[0]}_hd{c[1]}_heads{c[2]}_bw{c[3]}_kv{c[4]}")
def test_window_attention_other_head_counts(case, engine, cpu, dev):
    """Head counts other than the network's 8, window counts that are not a multiple of 8 (the persistent walk's padded tail), an
    arbitrary K/V shift, no label mask -- through the raw op against the CPU double."""
    ws, hd, heads, bw, kv_shift = case
    hip = hip_ops.HipOps(dev)
    hip.attention_f16x3 = engine == "f16x3"
    C, n = heads * hd, ws * ws
    g = torch.Generator().manual_seed(ws + hd + heads + bw)
    qkv = rnd(g, bw * n, 3 * C, scale=1.5)
    oc, og = torch.empty(bw * n, C), torch.full((bw * n, C), 9.0, device=dev)
    mc, mg = torch.empty(bw * n, heads, 2), torch.full((bw * n, heads, 2), 9.0, device=dev)
    cpu.window_attention(qkv, oc, mc, None, bw, 1, ws, heads, hd, kv_shift)
    hip.window_attention(qkv.to(dev), og, mg, None, bw, 1, ws, heads, hd, kv_shift)
    torch.cuda.synchronize()
    assert maxdiff(og, oc) <= 1e-4
    assert maxdiff(mg, mc) <= 1e-4


def test_motion_head(hip, cpu, dev):
    g = torch.Generator().manual_seed(8)
    frames, h, w, ws, shift = 4, 6, 10, 4, 2
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    rows = geo.row_map.numel()
    mo = rnd(g, rows, 8, 2, scale=3.0)
    w0, b0, w1, b1 = rnd(g, 4, 8), rnd(g, 4), rnd(g, 1, 4), rnd(g, 1)
    B = frames // 2
    dc = torch.full((B * h * w, 24), 7.0)
    dg = dc.to(dev)

    def view(buf):
        return buf[:, 4:8].unflatten(1, (2, 2)).permute(1, 0, 2)
    cpu.motion_head(mo, geo.row_map, w0, b0, w1, b1, view(dc))
    hip.motion_head(mo.to(dev), geo.row_map.to(dev), w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev), view(dg))
    torch.cuda.synchronize()
    assert maxdiff(dg, dc) <= 2e-5
    # the plane sink (the motion channels of the motion MLP's plane input without a split pass): exactly the split of the fp32 values,
    # at channels c0 + 2 * frame + {0, 1}; everything else in the planes untouched
    sink = hip_ops.Planes.alloc(B * h * w, 40, dev)
    dg2 = torch.full((B * h * w, 24), 7.0, device=dev)
    hip.motion_head(mo.to(dev), geo.row_map.to(dev), w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev), view(dg2), planes=sink, planes_c0=4, planes_gc=2)
    want = hip_ops.Planes.alloc(B * h * w, 40, dev)
    hip.split_planes(dg2[:, 4:8].contiguous(), want, c0=8)          # (split_planes_at needs an offset that is a multiple of 8)
    torch.cuda.synchronize()
    assert torch.equal(dg2, dg)
    assert torch.equal(sink.to_rows()[:, :, 4:8], want.to_rows()[:, :, 8:12])
    rest = sink.to_rows().clone()
    rest[:, :, 4:8] = 0
    assert (rest == 0).all() and (sink.t[:, :, sink.rows:] == 0).all()


# ------------------------------------------------------------------ warps / resampling
def test_flow_warp_golden_and_random(hip, cpu, dev):
    gold = G.load_npz("op_flow_warp")
    feat, flow = torch.from_numpy(gold["feat"]), torch.from_numpy(gold["flow"])
    out = torch.empty_like(feat, device=dev)
    hip.flow_warp(feat.to(dev), flow.to(dev), out)
    assert np.abs(out.cpu().numpy() - gold["out"]).max() <= 2e-6          # the reference's own output
    g = torch.Generator().manual_seed(9)
    src = rnd(g, 2, 3, 33, 47)
    fl = rnd(g, 2, 2, 33, 47, scale=6.0)
    fl[0, :, :3] = 500.0            # far outside: must be exactly zero
    fl[1, 0, 5:8] = -1e9
    fl[1, 1, 9, 9] = float("inf")
    oc, og = torch.empty_like(src), torch.empty_like(src, device=dev)
    cpu.flow_warp(src, torch.nan_to_num(fl, posinf=1e9), oc)
    hip.flow_warp(src.to(dev), fl.to(dev), og)
    assert not torch.isnan(og).any()
    assert maxdiff(og, oc) <= 2e-5
    assert og[0, :, :3].abs().max().item() == 0.0


def test_flow_warp_mask_and_padding_modes_vs_reference(hip, dev):
    """flow_warp(feature, flow, mask=True, padding_mode=...) of the drop-in shim (network/flow_warp.py -> atmvfi_flow_warp_ex)
    against the reference's own outputs (tests/golden/op_flow_warp_modes.npz, flow_warp.py:26-60): the three padding modes of
    grid_sample, the in-range mask bit for bit (incl. exact-edge coordinates and one a rounding below zero), flows of several image
    sizes (reflection folds repeatedly); then a larger random case against torch's grid_sample on the GPU."""
    import importlib
    shim = importlib.import_module("network.flow_warp")
    gold = G.load_npz("op_flow_warp_modes")
    feat, flow = torch.from_numpy(gold["feat"]).to(dev), torch.from_numpy(gold["flow"]).to(dev)
    for pm in ("zeros", "border", "reflection"):
        out, m = shim.flow_warp(feat, flow, mask=True, padding_mode=pm)
        assert m.dtype == torch.bool and tuple(m.shape) == (2, 12, 20)
        assert np.array_equal(m.cpu().numpy(), gold["mask"]), pm
        assert np.abs(out.cpu().numpy() - gold["out_" + pm]).max() <= 2e-6, pm
        assert torch.equal(shim.flow_warp(feat, flow, padding_mode=pm), out)          # without the mask: the same values
    assert torch.equal(shim.flow_warp(feat, flow), shim.flow_warp(feat, flow, mask=True)[0])     # hot-path kernel == the general one
    g = torch.Generator().manual_seed(19)
    src = rnd(g, 2, 5, 37, 53).to(dev)
    fl = rnd(g, 2, 2, 37, 53, scale=60.0).to(dev)
    ys, xs = torch.meshgrid(torch.arange(37, dtype=torch.float32, device=dev), torch.arange(53, dtype=torch.float32, device=dev), indexing="ij")
    gx, gy = 2 * (xs[None] + fl[:, 0]) / 52 - 1, 2 * (ys[None] + fl[:, 1]) / 36 - 1
    for pm in ("zeros", "border", "reflection"):
        want = torch.nn.functional.grid_sample(src, torch.stack([gx, gy], -1), mode="bilinear", padding_mode=pm, align_corners=True)
        got, m = shim.flow_warp(src, fl, mask=True, padding_mode=pm)
        # (torch's own GPU kernel, not the reference: coordinates of a few hundred px carry ~2e-5 px of fp32 rounding each way)
        assert maxdiff(got, want) <= 1e-4, pm
        assert torch.equal(m, (gx >= -1) & (gy >= -1) & (gx <= 1) & (gy <= 1))
    with pytest.raises(ValueError):
        shim.flow_warp(src, fl, padding_mode="wrap")


def test_flow_warp_nhwc_views(hip, cpu, dev):
    g = torch.Generator().manual_seed(10)
    B, H, W, C = 2, 12, 20, 224
    src = rnd(g, 2 * B, H, W, C)
    motion = rnd(g, B, H, W, 2 * C + 8, scale=4.0)     # flows live in channels 2C..2C+4 of the destination buffer
    dst_c = motion.clone()
    dst_g = motion.to(dev)
    for which, lo in ((0, 0), (1, C)):
        f_c = dst_c[..., 2 * C + 2 * which:2 * C + 2 * which + 2].permute(0, 3, 1, 2)
        f_g = dst_g[..., 2 * C + 2 * which:2 * C + 2 * which + 2].permute(0, 3, 1, 2)
        cpu.flow_warp_nhwc(src[which * B:(which + 1) * B], f_c, dst_c[..., lo:lo + C])
        hip.flow_warp_nhwc(src.to(dev)[which * B:(which + 1) * B], f_g, dst_g[..., lo:lo + C])
    torch.cuda.synchronize()
    assert maxdiff(dst_g, dst_c) <= 2e-5


def test_warp_blend_all_outputs(hip, cpu, dev):
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 24, 40
    im0, im1, o0, o1 = (torch.rand(B, 3, H, W, generator=g) for _ in range(4))
    buf = rnd(g, B, H, W, 116, scale=3.0)
    outs_c = [torch.empty(B, 3, H, W) for _ in range(3)] + [torch.empty(B, 2, H, W) for _ in range(2)] + [torch.empty(B, 1, H, W) for _ in range(2)]
    outs_g = [torch.empty_like(t, device=dev) for t in outs_c]
    bc, bg = buf.clone(), buf.to(dev)
    cpu.warp_blend(im0, im1, bc[..., 96:101], *outs_c, o0, o1, bc[..., 101:116])
    hip.warp_blend(im0.to(dev), im1.to(dev), bg[..., 96:101], *outs_g, o0.to(dev), o1.to(dev), bg[..., 101:116])
    torch.cuda.synchronize()
    for a, b in zip(outs_g, outs_c):
        assert maxdiff(a, b) <= 2e-5
    assert maxdiff(bg, bc) <= 2e-5
    # minimal form (coarse levels)
    outs_c = [torch.empty(B, 3, H, W) for _ in range(3)]
    outs_g = [torch.empty_like(t, device=dev) for t in outs_c]
    cpu.warp_blend(im0, im1, bc[..., 96:101], *outs_c)
    hip.warp_blend(im0.to(dev), im1.to(dev), bg[..., 96:101], *outs_g)
    for a, b in zip(outs_g, outs_c):
        assert maxdiff(a, b) <= 2e-5


def test_resize_align_corners(hip, cpu, dev):
    g = torch.Generator().manual_seed(12)
    x = rnd(g, 2, 3, 34, 50)
    for (oh, ow, sc) in ((17, 25, 1.0), (68, 100, 2.0), (34, 50, 1.0)):
        oc, og = torch.empty(2, 3, oh, ow), torch.empty(2, 3, oh, ow, device=dev)
        cpu.resize(x, oc, sc)
        hip.resize(x.to(dev), og, sc)
        assert maxdiff(og, oc) <= 2e-5
    nh = rnd(g, 2, 9, 11, 8, scale=4.0)           # flow pair inside an NHWC motion map
    v_c = nh[..., 2:4].permute(0, 3, 1, 2)
    v_g = nh.to(dev)[..., 2:4].permute(0, 3, 1, 2)
    oc, og = torch.empty(2, 2, 18, 22), torch.empty(2, 2, 18, 22, device=dev)
    cpu.resize(v_c, oc, 2.0)
    hip.resize(v_g, og, 2.0)
    assert maxdiff(og, oc) <= 2e-5


def test_pack_final_l1(hip, cpu, dev):
    g = torch.Generator().manual_seed(13)
    B, H, W = 2, 10, 14
    im0, im1 = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    pc, pg = torch.empty(2 * B, H, W, 4), torch.empty(2 * B, H, W, 4, device=dev)
    cpu.pack_frames(im0, im1, pc)
    hip.pack_frames(im0.to(dev), im1.to(dev), pg)
    assert maxdiff(pg, pc) == 0.0
    r = rnd(g, B, H, W, 4, scale=3.0)
    sc, cc = torch.empty(B, 3, H, W), torch.empty(B, 3, H, W)
    sg, cg = torch.empty(B, 3, H, W, device=dev), torch.empty(B, 3, H, W, device=dev)
    cpu.final_residual(im0, r[..., :3], sc, cc)
    hip.final_residual(im0.to(dev), r.to(dev)[..., :3], sg, cg)
    assert maxdiff(sg, sc) <= 2e-6 and maxdiff(cg, cc) <= 2e-6
    lc, lg = torch.empty(B), torch.empty(B, device=dev)
    cpu.l1_mean(im0, im1, lc)
    hip.l1_mean(im0.to(dev), im1.to(dev), lg)
    assert maxdiff(lg, lc) <= 1e-6
    hip.l1_mean(im0.to(dev), im1.to(dev), lg)                 # a second call does not add up
    assert maxdiff(lg, lc) <= 1e-6
    # ensemble pick (network_base.py:591-603): per sample the candidate pair of the smallest loss, the FIRST on ties
    g2 = torch.Generator().manual_seed(77)
    B = 5
    cands = [(torch.rand(B, 2, 6, 9, generator=g2).to(dev), torch.rand(B, 2, 6, 9, generator=g2).to(dev)) for _ in range(3)]
    losses = [torch.tensor(v, device=dev) for v in ([0.3, 0.2, 0.5, 0.1, 0.4], [0.2, 0.2, 0.4, 0.1, 0.3], [0.9, 0.1, 0.4, 0.1, 0.3])]
    picks = [1, 2, 1, 0, 1]
    o0, o1 = torch.full((B, 2, 6, 9), 7.0, device=dev), torch.full((B, 2, 6, 9), 7.0, device=dev)
    hip.ensemble_select(losses, cands, o0, o1)
    torch.cuda.synchronize()
    for i, pk in enumerate(picks):
        assert torch.equal(o0[i], cands[pk][0][i]) and torch.equal(o1[i], cands[pk][1][i]), (i, pk)
    # NaN losses: Python's min(l0, l1, l2) keeps its first argument unless a later one compares LESS, and nothing equals a NaN -- so a
    # NaN l0 makes the reference's chain fall through to level 2, a NaN l1 / l2 is never the minimum (network_base.py:591-603)
    nan = float("nan")
    losses = [torch.tensor(v, device=dev) for v in ([nan, 0.2, 0.5, nan, 0.4], [0.2, nan, 0.4, nan, 0.3], [0.9, 0.1, nan, nan, nan])]
    def ref_pick(a, b, c):
        m = min(a, b, c)
        return 0 if a == m else (1 if b == m else 2)
    picks = [ref_pick(*(float(l[i]) for l in losses)) for i in range(B)]
    assert picks == [2, 2, 1, 2, 1]
    hip.ensemble_select(losses, cands, o0, o1)
    torch.cuda.synchronize()
    for i, pk in enumerate(picks):
        assert torch.equal(o0[i], cands[pk][0][i]) and torch.equal(o1[i], cands[pk][1][i]), (i, pk)
    # l1_mean sums in a fixed order: bit-identical from call to call, at a size of many blocks per sample
    big0, big1 = torch.rand(3, 3, 544, 960, generator=g2).to(dev), torch.rand(3, 3, 544, 960, generator=g2).to(dev)
    outs = []
    for _ in range(4):
        o = torch.empty(3, device=dev)
        hip.l1_mean(big0, big1, o)
        outs.append(o.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert maxdiff(outs[0].cpu(), (big0 - big1).abs().double().mean(dim=[1, 2, 3]).float().cpu()) <= 1e-6


# ------------------------------------------------------------------ a whole ATMFormer block (reference fixture)
@pytest.mark.parametrize("shift", [0, 3])
def test_atm_block_reference_fixture(shift, hip, dev):
    """The reference's own smoke shape (attention.py:512-534): C=128, window 7 (N=49, padded to 64
    inside the kernel), 32x32 -> canvas 35x35, shift 3; expected values from the reference itself."""
    pkg = importlib.import_module("atm-vfi_amd")
    gold = G.load_npz(f"op_atm_ws7_shift{shift}")
    net = pkg.NetworkLite()
    net.set_ops(hip)
    P = {}
    for k in gold.files:
        if k.startswith("w."):
            P["b." + k[2:]] = torch.from_numpy(gold[k]).to(dev)
    qkv = torch.cat([P["b.attn.q.weight"], P["b.attn.kv.weight"]], 0)
    P["pk:b.attn.qkv.weight"] = hip.pack_weight(GEMM_LINEAR, qkv)
    for nm in ("attn.proj", "mlp.fc1", "mlp.fc2"):
        P[f"pk:b.{nm}.weight"] = hip.pack_weight(GEMM_LINEAR, P[f"b.{nm}.weight"])
    P["pk:b.mlp.dwconv.dwconv.weight"] = hip.pack_dw_weight(P["b.mlp.dwconv.dwconv.weight"])
    x = torch.from_numpy(gold["x"]).to(dev)               # [4, 1024, 128]
    out = torch.empty(4 * 1024, 128, device=dev)
    mdst = torch.empty(2, 2 * 1024, 2, device=dev)      # [frame, B*hw, 2]
    net._block(hip, P, "b", x.reshape(4 * 1024, 128), 4, 32, 32, 7, shift, True, out, mdst, "t")
    torch.cuda.synchronize()
    y = out.reshape(4, 1024, 128)[:, ::4].cpu().numpy()
    assert np.abs(y - gold["y"]).max() <= 2e-4
    mo = mdst.reshape(4, 1024, 2).cpu().numpy()
    assert np.abs(mo - gold["motion"]).max() <= 2e-4


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_gemm_modes_both_precisions(precision, hip, cpu, dev):
    """LINEAR (groups + scatter + residual), DECONV (in_prelu) and strided/dilated CONV on both engines."""
    hip.precision = precision
    try:
        test_linear_groups_scatter_residual(hip, cpu, dev)
        for case in DECONV_CASES:
            test_deconv2x2(case, hip, cpu, dev)
        g = torch.Generator().manual_seed(99)
        for (cin, cout, k, stride, pad, dil, H, W) in ((96, 192, 3, 2, 1, 1, 18, 26), (48, 48, 3, 4, 2, 2, 24, 40),
                                                      (768, 5, 1, 1, 0, 1, 9, 7), (37 * 4, 21, 3, 2, 1, 1, 11, 13)):
            x = rnd(g, 2, H, W, cin)
            w = rnd(g, cout, cin, k, k, scale=1.0 / np.sqrt(cin * k * k))
            b = rnd(g, cout, scale=0.2)
            Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
            Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
            r4 = (cout + 3) // 4 * 4
            oc = torch.full((2, Ho, Wo, r4), 7.0)
            og = torch.full((2, Ho, Wo, r4), 7.0, device=dev)
            cpu.conv(x, cpu.pack_weight(GEMM_CONV, w), oc[..., :cout], stride, pad, dil, b, None)
            hip.conv(x.to(dev), hip.pack_weight(GEMM_CONV, w.to(dev)), og[..., :cout], stride, pad, dil, b.to(dev), None)
            torch.cuda.synchronize()
            assert maxdiff(og, oc) <= 1e-4
    finally:
        hip.precision = "f16x3"


# ------------------------------------------------------------------ split-plane activations (LDS-DMA GEMM)
def planes_to_f32(p):
    """hi + lo'/1024: the value the split GEMM sees (fp64 so that the reconstruction itself adds nothing)."""
    t = p.to_rows().double().cpu()
    return (t[0] + t[1] / 1024.0)[:, :p.c]


def test_split_planes_format(hip, dev):
    g = torch.Generator().manual_seed(21)
    x = torch.cat([rnd(g, 50, 70, scale=3.0), rnd(g, 50, 70, scale=1e-3), rnd(g, 50, 70, scale=7e4)])      # incl. > fp16 max
    x[0, :4] = torch.tensor([0.0, -0.0, 65504.0, -1e30])
    buf = torch.zeros(150, 72)
    buf[:, :70] = x
    p = hip_ops.Planes.alloc(150, 70, dev)
    assert p.chunks == 3 and p.rows == 150 and tuple(p.t.shape) == (2, 3, 151, 32)     # chunk major: [plane, chunk, row (+1 spare zero row), 32]
    hip.split_planes(buf.to(dev)[:, :70], p)
    torch.cuda.synchronize()
    hi = x.clamp(-65504, 65504).half()
    lo = ((x.clamp(-3e38, 3e38) - hi.float()) * 1024).clamp(-65504, 65504).half()
    pr = p.to_rows()
    assert torch.equal(pr[0, :, :70].cpu(), hi) and torch.equal(pr[1, :, :70].cpu(), lo)
    assert torch.equal(p.t[0, 2, :150, 5].cpu(), hi[:, 69])               # channel 69 = chunk 2, element 5
    assert (p.t[:, :, 150] == 0).all()                                    # the spare row stays zero
    assert (pr[:, :, 70:] == 0).all()                                     # pad channels are written as zero
    err = (planes_to_f32(p) - x.double()).abs()
    normal = (x.abs() <= 65504) & (x.abs() >= 2.0 ** -14)
    assert (err / x.double().abs().clamp_min(1e-30))[normal].max().item() <= 2.0 ** -21    # ~22 significant bits in fp16's normal range
    assert err[x.abs() < 2.0 ** -14].max().item() <= 2.0 ** -34                            # below it: fp16-subnormal lo' / 1024


LIN_SPLIT_CASES = [
    # M, N, K, bias, residual, scatter
    (1000, 200, 100, True, True, False),        # every tail: rows, columns, K (100 -> 128)
    (256, 128, 32, False, False, False),        # exactly one tile, one k-step
    (257, 129, 64, True, False, False),         # one row / one column past a tile, two k-steps (prologue tails)
    (777, 672, 672, True, True, False),         # 21 k-steps
    (3000, 384, 1536, True, True, False),       # fc2 shape (long K)
    (1920, 96, 96, True, True, True),           # scatter through a window map
]


@pytest.mark.parametrize("case", LIN_SPLIT_CASES, ids=lambda c: f"M{c[0]}_N{c[1]}_K{c[2]}" + ("_scatter" if c[5] else ""))
def test_linear_from_split_planes(case, hip, dev):
    """The LDS-DMA GEMM must give bit-identical results to the fp32-input f16x3 GEMM (same split, same
    accumulation order), and agree with fp64 to fp32 accuracy."""
    m, n, k, use_b, use_r, scatter = case
    g = torch.Generator().manual_seed(m + n + k)
    x = rnd(g, m, k, scale=2.0).to(dev)
    w = rnd(g, n, k, scale=1.0 / k ** 0.5).to(dev)
    b = rnd(g, n, scale=0.5).to(dev) if use_b else None
    r = rnd(g, m, n, scale=0.5).to(dev) if use_r else None
    rows_out = m
    rmap = None
    if scatter:
        geo = windows.build_window_geometry(2, 30, 30, 8, 4)
        assert geo.row_map.numel() >= m
        rmap = geo.row_map[:m].contiguous().to(dev)
        rows_out = 2 * 30 * 30
    pw = hip.pack_weight(GEMM_LINEAR, w)
    n4 = (n + 3) // 4 * 4                                                   # views need ld % 4 == 0
    y0 = torch.full((rows_out, n4), 5.0, device=dev)[:, :n]
    y1 = torch.full((rows_out, n4), 5.0, device=dev)[:, :n]
    if r is not None:
        r = torch.cat([r, torch.zeros(m, n4 - n, device=dev)], 1)[:, :n]
    p = hip_ops.Planes.alloc(m, k, dev)
    hip.split_planes(x, p)
    hip.linear(x, pw, y0, b, r, rmap)
    hip.linear(p, pw, y1, b, r, rmap)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    if not scatter:
        ref = x.double() @ w.double().t() + (b.double() if use_b else 0) + (r.double() if use_r else 0)
        assert (y1.double() - ref).abs().max().item() <= 2e-5


def test_split_plane_sinks(hip, cpu, dev):
    """LayerNorm / dwconv+GELU / window attention writing their result as split planes (with and without the fp32
    copy): the planes must be the exact split of the fp32 result."""
    g = torch.Generator().manual_seed(31)

    def split_of(t):
        hi = t.clamp(-65504, 65504).half()
        return hi, ((t - hi.float()) * 1024).clamp(-65504, 65504).half()
    # LayerNorm with a gather map (zero-padded tokens -> beta)
    frames, h, w, ws, shift, C = 2, 5, 6, 4, 2, 224
    geo = windows.build_window_geometry(frames, h, w, ws, shift)
    src = rnd(g, frames * h * w, C, scale=3.0).to(dev)
    gamma, beta = (1 + rnd(g, C, scale=0.2)).to(dev), rnd(g, C, scale=0.2).to(dev)
    rows = geo.row_map.numel()
    of = torch.empty(rows, C, device=dev)
    hip.layernorm(src, of, gamma, beta, geo.row_map.to(dev))
    for keep in (True, False):
        o2 = torch.empty(rows, C, device=dev) if keep else None
        p = hip_ops.Planes.alloc(rows, C, dev)
        hip.layernorm(src, o2, gamma, beta, geo.row_map.to(dev), planes=p)
        torch.cuda.synchronize()
        hi, lo = split_of(of)
        assert torch.equal(p.to_rows()[0, :, :C], hi) and torch.equal(p.to_rows()[1, :, :C], lo)
        if keep:
            assert torch.equal(o2, of)
    # dwconv + GELU (both kernels: C % 64 == 0 and not)
    for n, hh, ww, c in ((2, 7, 9, 448), (1, 5, 6, 100)):
        x = rnd(g, n, hh, ww, c, scale=2.0).to(dev)
        wt = hip.pack_dw_weight(rnd(g, c, 1, 3, 3, scale=0.5).to(dev))
        b = rnd(g, c, scale=0.3).to(dev)
        of = torch.empty(n, hh, ww, c, device=dev)
        hip.dwconv_gelu(x, of, wt, b)
        p = hip_ops.Planes.alloc(n * hh * ww, c, dev)
        hip.dwconv_gelu(x, None, wt, b, planes=p)
        torch.cuda.synchronize()
        hi, lo = split_of(of.reshape(-1, c))
        assert torch.equal(p.to_rows()[0, :, :c], hi) and torch.equal(p.to_rows()[1, :, :c], lo)
    # window attention
    ws, hd, heads = 8, 28, 8
    geo = windows.build_window_geometry(2, 12, 20, ws, 4)
    bw, n, C = 2 * geo.n_windows, ws * ws, heads * hd
    qkv = rnd(g, bw * n, 3 * C, scale=1.5).to(dev)
    of = torch.empty(bw * n, C, device=dev)
    lab = geo.labels.to(dev)
    hip.window_attention(qkv, of, None, lab, bw, geo.n_windows, ws, heads, hd, bw // 2)
    p = hip_ops.Planes.alloc(bw * n, C, dev)
    hip.window_attention(qkv, None, None, lab, bw, geo.n_windows, ws, heads, hd, bw // 2, planes=p)
    torch.cuda.synchronize()
    hi, lo = split_of(of)
    assert torch.equal(p.to_rows()[0, :, :C], hi) and torch.equal(p.to_rows()[1, :, :C], lo)
    with pytest.raises(ValueError):
        hip.layernorm(src, None, gamma, beta)                              # no output at all


# ------------------------------------------------------------------ host-boundary frame formats
@pytest.mark.parametrize("geom", [(37, 53, 64), (64, 128, 64), (270, 480, 64), (45, 33, 32)], ids=lambda g: f"{g[0]}x{g[1]}_div{g[2]}")
@pytest.mark.parametrize("bgr", [True, False], ids=["bgr", "rgb"])
def test_frame_u8_f32_roundtrip_bit_exact(geom, bgr, hip, dev):
    """uint8 HWC -> fp32 planar (+ flip, /255, replicate pad) and back (unpad, np.round(x*255)): bit-exact vs the numpy/torch path of
    inference_2frame (demo_2x.py:64-85) and InputPadder (benchmark/utils.py:57-80)."""
    host_io = importlib.import_module("atm-vfi_amd.host_io")
    h, w, div = geom
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    pad = host_io.InputPadder((1, 3, h, w), divisor=div)
    l, r, t, b = pad._pad
    hp, wp = h + t + b, w + l + r
    ref_in = img[:, :, ::-1].copy() if bgr else img
    ref = pad.pad((torch.tensor(ref_in.transpose(2, 0, 1)) / 255.).unsqueeze(0))[0]          # the reference's arithmetic, on CPU
    out = torch.empty(3, hp, wp, device=dev)
    hip.frame_u8_to_f32(torch.from_numpy(img).to(dev), out, t, l, bgr)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), ref)
    # back: arbitrary fp32 values in [0,1] incl. exact .5 ties after scaling
    f = torch.rand(3, hp, wp, generator=torch.Generator().manual_seed(h))
    f[:, :4, :4] = torch.tensor([0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255]).repeat(3, 4, 1)
    f[0, 5, 5], f[1, 5, 5], f[2, 5, 5] = 0.0, 1.0, 0.999999
    want = np.round(pad.unpad(f.unsqueeze(0))[0].numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
    if bgr:
        want = want[:, :, ::-1].copy()
    got = torch.empty(h, w, 3, dtype=torch.uint8, device=dev)
    hip.frame_f32_to_u8(f.to(dev), got, t, l, bgr)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        hip.frame_f32_to_u8(f.to(dev), got, t + 100, l, bgr)


def test_deconv_from_split_planes(hip, cpu, dev):
    """split_planes(+PReLU) -> deconv on the LDS-DMA GEMM == the fp32-input deconv with in_prelu, bit for bit."""
    g = torch.Generator().manual_seed(77)
    n, h, w, cin, cout = 1, 9, 14, 197, 101
    xb = rnd(g, n, h, w, 200, scale=2.0).to(dev)
    x = xb[..., :cin]
    wt = rnd(g, cin, cout, 2, 2, scale=0.1).to(dev)
    bias, slope, inp = rnd(g, cout, scale=0.3).to(dev), (0.25 + rnd(g, cout, scale=0.1)).to(dev), (0.25 + rnd(g, cin, scale=0.1)).to(dev)
    pw = hip.pack_weight(GEMM_DECONV, wt)
    y0 = torch.full((n, 2 * h, 2 * w, 104), 3.0, device=dev)
    y1 = torch.full((n, 2 * h, 2 * w, 104), 3.0, device=dev)
    hip.deconv(x, pw, y0[..., :cout], bias, slope, in_prelu=hip.pad_channels(inp))
    p = hip_ops.Planes.alloc(n * h * w, cin, dev)
    hip.split_planes(x.flatten(0, 2), p, prelu=inp)
    hip.deconv(x, pw, y1[..., :cout], bias, slope, planes=p)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    xa = torch.where(x > 0, x, x * inp)
    ref = torch.nn.functional.conv_transpose2d(xa.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), stride=2)
    ref = torch.where(ref > 0, ref, ref * slope.double()[None, :, None, None]).permute(0, 2, 3, 1)
    assert (y1[..., :cout].double() - ref).abs().max().item() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [0, 1], ids=["row", "half"])
def test_conv3x3_every_kernel_instance(schedule, hip, cpu, dev):
    """All 16 instances of the 3x3 kernel (1..8 n-tiles x row / half schedule; from 3 n-tiles on the fragment reads run under
    hand-counted LDS waits, 7-8 tiles of the row schedule with a single activation register set): ragged image, two images,
    a channel count with a tap-packed tail and one without, against the CPU restatement."""
    g = torch.Generator().manual_seed(4242 + schedule)
    r4 = lambda c: (c + 3) // 4 * 4
    try:
        for wn in range(1, 9):
            hip.conv3_instance = (schedule, wn)
            for cin in (37, 64):
                cout = 2 * 16 * wn - (5 if wn % 2 else 0)            # two column blocks, the second one partial for odd wn
                N, H, W = 2, 21, 35
                x = rnd(g, N, H, W, r4(cin))[..., :cin]
                w = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin))
                bias = rnd(g, cout, scale=0.2)
                slope = torch.rand(cout, generator=g) * 0.4
                oc = torch.empty(N, H, W, r4(cout))[..., :cout]
                cpu.conv(x, cpu.pack_weight(GEMM_CONV, w), oc, 1, 1, 1, bias, slope)
                xg = torch.zeros(N, H, W, r4(cin), device=dev)
                xg[..., :cin] = x.to(dev)
                og = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
                hip.conv(xg[..., :cin], hip.pack_weight(GEMM_CONV, w.to(dev)), og[..., :cout], 1, 1, 1, bias.to(dev), slope.to(dev))
                torch.cuda.synchronize()
                assert maxdiff(og[..., :cout], oc) <= 1e-4, (schedule, wn, cin)
                assert (og[..., cout:] == 7.0).all()
    finally:
        hip.conv3_instance = None


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", [0, 1], ids=["row", "half"])
def test_conv3x3_plane_sink(schedule, hip, dev):
    """The 3x3 kernel's second output: the planes must be the exact split of prelu(result) -- identical, bit for bit, to
    split_planes(+PReLU) applied to the fp32 output -- with the fp32 output unchanged, for a 32k+5-wide layer (partial last
    group of 4, partial last chunk), ragged tiles, two images; and a deconv fed from them must equal the one fed by the split pass."""
    g = torch.Generator().manual_seed(909 + schedule)
    n, h, w, cin, cout = 2, 19, 21, 40, 101
    x = rnd(g, n, h, w, cin, scale=2.0).to(dev)
    wt = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)).to(dev)
    bias = rnd(g, cout, scale=0.3).to(dev)
    inp = (0.25 + rnd(g, cout, scale=0.1)).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    hip.conv3_instance = (schedule, 0)
    try:
        for slopes in (inp, None):
            y0 = torch.full((n, h, w, 104), 3.0, device=dev)
            y1 = torch.full((n, h, w, 104), 3.0, device=dev)
            hip.conv(x, pw, y0[..., :cout], 1, 1, 1, bias, None)
            p = hip_ops.Planes.alloc(n * h * w + 7, cout, dev)          # more plane rows than pixels: the row pitch is plane_rows
            hip.conv(x, pw, y1[..., :cout], 1, 1, 1, bias, None, planes=p, planes_prelu=hip.pad_channels(slopes) if slopes is not None else None)
            q = hip_ops.Planes.alloc(n * h * w, cout, dev)
            hip.split_planes(y0[..., :cout].flatten(0, 2), q, prelu=slopes)
            torch.cuda.synchronize()
            assert torch.equal(y0, y1)
            assert torch.equal(p.t[:, :, :n * h * w], q.t[:, :, :n * h * w])
            assert (p.t[:, :, n * h * w:] == 0).all()
            # and through the consumer: deconv from the sink's planes == deconv from the split pass's
            if slopes is not None:
                wd = rnd(g, cout, 24, 2, 2, scale=0.1).to(dev)
                pwd = hip.pack_weight(GEMM_DECONV, wd)
                z0 = torch.empty(n, 2 * h, 2 * w, 24, device=dev)
                z1 = torch.empty(n, 2 * h, 2 * w, 24, device=dev)
                hip.deconv(y0[..., :cout], pwd, z0, None, None, planes=q)
                hip.deconv(y0[..., :cout], pwd, z1, None, None, planes=p)
                torch.cuda.synchronize()
                assert torch.equal(z0, z1)
    finally:
        hip.conv3_instance = None


@pytest.mark.gpu
@pytest.mark.parametrize("wn", list(range(1, 9)))
def test_conv3x3_planes_every_width(wn, hip, cpu, dev):
    """The 3x3 kernel on split-plane input (LDS-DMA halo, ping-pong wave groups), every tile width: ragged image, two images,
    channel counts with a 5- and an 8-channel tap-packed tail (one and three full chunks in front) and without a tail.  Must equal the fp32-input
    kernel BIT FOR BIT (same split, same k order), and the CPU restatement within the contraction tolerance; the plane sink at a
    channel offset must be the exact split of the fp32 result."""
    g = torch.Generator().manual_seed(7000 + wn)
    r4 = lambda c: (c + 3) // 4 * 4
    N, H, W = 2, 21, 35
    for cin in (37, 64, 104, 40):
        cout = 2 * 16 * wn - (5 if wn % 2 else 0)            # two column blocks, the second one partial for odd wn
        x = rnd(g, N, H, W, r4(cin))[..., :cin]
        w = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin))
        bias = rnd(g, cout, scale=0.2)
        slope = torch.rand(cout, generator=g) * 0.4
        oc = torch.empty(N, H, W, r4(cout))[..., :cout]
        cpu.conv(x, cpu.pack_weight(GEMM_CONV, w), oc, 1, 1, 1, bias, slope)
        xg = torch.zeros(N, H, W, r4(cin), device=dev)
        xg[..., :cin] = x.to(dev)
        pw = hip.pack_weight(GEMM_CONV, w.to(dev))
        y0 = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
        hip.conv(xg[..., :cin], pw, y0[..., :cout], 1, 1, 1, bias.to(dev), slope.to(dev))
        xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
        hip.split_planes(xg[..., :cin].flatten(0, 2), xp)
        y1 = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
        c0 = 8
        sink = hip_ops.Planes.alloc(N * H * W, c0 + cout, dev)
        hip.conv3x3_planes(xp, N, H, W, pw, out=y1[..., :cout], bias=bias.to(dev), prelu=slope.to(dev), planes=sink, planes_c0=c0, wn=wn)
        only = hip_ops.Planes.alloc(N * H * W, cout, dev)
        hip.conv3x3_planes(xp, N, H, W, pw, out=None, bias=bias.to(dev), prelu=slope.to(dev), planes=only, wn=wn)
        q = hip_ops.Planes.alloc(N * H * W, cout, dev)
        hip.split_planes(y0[..., :cout].flatten(0, 2), q)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1), (wn, cin, maxdiff(y0, y1))
        assert maxdiff(y1[..., :cout], oc) <= 1e-4, (wn, cin)
        assert torch.equal(only.t, q.t), (wn, cin)
        got = sink.to_rows()[:, :, c0:c0 + cout]
        assert torch.equal(got, q.to_rows()[:, :, :cout]), (wn, cin)
        assert (sink.to_rows()[:, :, :c0] == 0).all() and (sink.t[:, :, N * H * W:] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("wn", [2, 4, 7, 8])
def test_conv3x3_planes_prelu_forms(wn, hip, dev):
    """The plane kernel applies PReLU as max(v, s v) when every slope of a tile's columns lies in [0, 1] and as the select otherwise,
    decided per tile from the tile's constants in LDS (the 8-n-tile instance always selects).  Slopes in range, slopes with negative /
    > 1 entries in ONE of two column blocks only (the two blocks of a pixel tile then take different forms), the same for the plane
    sink's own slopes, and signed zeros / huge values in the data: always bit-identical to the fp32-input kernel (select form) and to
    split_planes(prelu) of its result."""
    g = torch.Generator().manual_seed(7300 + wn)
    r4 = lambda c: (c + 3) // 4 * 4
    N, H, W, cin = 1, 40, 52, 40
    cout = 2 * 16 * wn - 3
    x = rnd(g, N, H, W, r4(cin), scale=1.5).to(dev)
    x[0, 3, 5, :cin] = 0.0
    x[0, 7, 9, :cin] = 3.0e4
    wt = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)).to(dev)
    bias = rnd(g, cout, scale=0.2).to(dev)
    bias[::5] = 0.0                                            # exact zeros (and -0 * slope) at the all-zero pixel
    pw = hip.pack_weight(GEMM_CONV, wt)
    xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
    hip.split_planes(x[..., :cin].flatten(0, 2), xp)
    base = torch.rand(cout, generator=g)
    for name, edit in (("in range", lambda t: t), ("edges 0 and 1", lambda t: torch.where(torch.arange(cout) % 2 == 0, torch.zeros(()), torch.ones(()))),
                       ("second block out of range", lambda t: torch.cat([t[:16 * wn], t[16 * wn:] * 3.0 - 1.0])),
                       ("first block out of range", lambda t: torch.cat([-t[:16 * wn], t[16 * wn:]]))):
        slope = edit(base.clone()).to(dev).contiguous()
        pslope = torch.zeros((cout + 31) // 32 * 32, device=dev)
        pslope[:cout] = edit(torch.rand(cout, generator=g)).to(dev)
        y0 = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
        hip.conv(x[..., :cin], pw, y0[..., :cout], 1, 1, 1, bias, slope)
        y1 = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
        s1 = hip_ops.Planes.alloc(N * H * W, cout, dev)
        hip.conv3x3_planes(xp, N, H, W, pw, out=y1[..., :cout], bias=bias, prelu=slope, planes=s1, planes_prelu=pslope, wn=wn)
        qp = hip_ops.Planes.alloc(N * H * W, cout, dev)
        hip.split_planes(y0[..., :cout].flatten(0, 2), qp, prelu=pslope[:cout].contiguous())
        torch.cuda.synchronize()
        assert torch.equal(y0.view(torch.int32), y1.view(torch.int32)), (wn, name, maxdiff(y0, y1))
        assert torch.equal(s1.t.view(torch.int16), qp.t.view(torch.int16)), (wn, name)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(1, 16, 16, 712, 352, 0), (1, 32, 32, 456, 224, 0), (1, 36, 60, 1352, 768, 0), (2, 21, 35, 520, 61, 0),
                                  (1, 32, 32, 456, 224, 4), (1, 20, 28, 392, 101, 0)],
                         ids=lambda c: f"n{c[0]}_{c[1]}x{c[2]}_cin{c[3]}_cout{c[4]}_wn{c[5]}")
def test_conv3x3_planes_split_k(case, hip, dev):
    """Split-K of under-filled long-K launches (atmvfi_conv3x3_planes3 with a workspace; the motion-MLP shapes of network_lite at
    256 x 256 / 256 x 448 and of network_base at 576 x 960, ragged images, tap-packed tails, ragged Cout): K ranges by whole chunks,
    raw fp32 partial sums, a fixed-order reduce with the whole epilogue (bias, PReLU, fp32 rows from out_cmin on, both plane sinks, the
    first through its own PReLU).  Against the unsplit launch: equal within fp32 summation-order error (1e-5 of the value scale);
    two split launches are bit-identical; pad channels of the planes stay zero; without a workspace nothing changes."""
    N, H, W, cin, cout, wn = case
    g = torch.Generator().manual_seed(7100 + cin + cout + H)
    need = hip.conv3x3_workspace_floats(N, H, W, cin, cout)
    assert need > 0, "the launcher does not split this shape: the case tests nothing"
    xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
    x = rnd(g, N * H * W, (cin + 3) // 4 * 4, scale=1.5).to(dev)[:, :cin]
    hip.split_planes(x, xp)
    wt = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)).to(dev)
    bias, slope = rnd(g, cout, scale=0.2).to(dev), (torch.rand(cout, generator=g) * 0.4).to(dev)
    pslope = torch.zeros((cout + 31) // 32 * 32, device=dev)
    pslope[:cout] = torch.rand(cout, generator=g).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    ws = torch.empty(need, device=dev)
    cmin = (cout - 5) // 4 * 4
    r4 = lambda c: (c + 3) // 4 * 4

    def launch(workspace):
        y = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
        ycmp = torch.full((N, H, W, 8), 7.0, device=dev)
        s1 = hip_ops.Planes.alloc(N * H * W, 8 + cout, dev)
        s2 = hip_ops.Planes.alloc(N * H * W, cout, dev)
        hip.conv3x3_planes(xp, N, H, W, pw, out=y[..., :cout], bias=bias, prelu=slope, wn=wn, workspace=workspace)
        hip.conv3x3_planes(xp, N, H, W, pw, out=ycmp[..., :cout - cmin], bias=bias, prelu=None, planes=s1, planes_c0=8, planes_prelu=pslope,
                           planes2=s2, out_cmin=cmin, wn=wn, workspace=workspace)
        torch.cuda.synchronize()
        return y, ycmp, s1, s2
    y0, c0_, a0, b0 = launch(None)
    y1, c1_, a1, b1 = launch(ws)
    y2, c2_, a2, b2 = launch(ws)
    scale = max(1.0, float(y0[..., :cout].abs().max()))
    assert not torch.equal(y0, y1), "the workspace launch produced the unsplit result bit for bit: no split happened"
    assert maxdiff(y0[..., :cout], y1[..., :cout]) <= 1e-5 * scale
    assert (y1[..., cout:] == 7.0).all()
    assert maxdiff(c0_[..., :cout - cmin], c1_[..., :cout - cmin]) <= 1e-5 * scale and (c1_[..., cout - cmin:] == 7.0).all()
    for p0, p1 in ((a0, a1), (b0, b1)):
        assert maxdiff(p0.to_float(), p1.to_float()) <= 1e-5 * scale
        assert (p1.t[:, :, N * H * W:] == 0).all()
    assert (a1.to_rows()[:, :, :8] == 0).all() and (a1.to_rows()[:, :, 8 + cout:] == 0).all() and (b1.to_rows()[:, :, cout:] == 0).all()
    # the second sink is the raw map: exactly the split of the fp32 result without the first sink's PReLU
    q = hip_ops.Planes.alloc(N * H * W, cout, dev)
    yraw = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
    hip.conv3x3_planes(xp, N, H, W, pw, out=yraw[..., :cout], bias=bias, prelu=None, wn=wn, workspace=ws)
    hip.split_planes(yraw[..., :cout].flatten(0, 2), q)
    torch.cuda.synchronize()
    assert torch.equal(b1.to_rows()[:, :, :cout], q.to_rows()[:, :, :cout])
    assert torch.equal(c1_[..., :cout - cmin], yraw[..., cmin:cout])
    # run-to-run determinism of the split launch
    assert torch.equal(y1, y2) and torch.equal(c1_, c2_) and torch.equal(a1.t, a2.t) and torch.equal(b1.t, b2.t)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(64, 64, 0), (101, 101, 0), (200, 130, 0), (72, 40, 3), (37, 200, 8), (136, 389, 0)],
                         ids=lambda c: f"cin{c[0]}_cout{c[1]}_wn{c[2]}")
def test_conv3x3_planes_persistent_grid(case, hip, dev):
    """More tiles than CUs: the plane kernel's workgroups walk several tiles each, with the next tile's first halo and weights put in
    flight by the current tile's last k-steps (regular chunks, and the three-piece halo issue of the tap-packed tail steps), ragged
    right / bottom tiles, one or several column blocks, both sinks.  Must equal the fp32-input kernel bit for bit."""
    cin, cout, wn = case
    g = torch.Generator().manual_seed(7200 + cin + cout)
    r4 = lambda c: (c + 3) // 4 * 4
    N, H, W = 2, 150, 250                                    # 2 x 10 x 16 = 320 spatial tiles (x column blocks)
    x = rnd(g, N, H, W, r4(cin), scale=1.5).to(dev)
    wt = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)).to(dev)
    bias, slope = rnd(g, cout, scale=0.2).to(dev), (torch.rand(cout, generator=g) * 0.4).to(dev)
    pslope = torch.zeros((cout + 31) // 32 * 32, device=dev)
    pslope[:cout] = torch.rand(cout, generator=g).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    y0 = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
    hip.conv(x[..., :cin], pw, y0[..., :cout], 1, 1, 1, bias, slope)
    xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
    hip.split_planes(x[..., :cin].flatten(0, 2), xp)
    y1 = torch.full((N, H, W, r4(cout)), 7.0, device=dev)
    s1 = hip_ops.Planes.alloc(N * H * W, cout, dev)
    s2 = hip_ops.Planes.alloc(N * H * W, 8 + cout, dev)
    hip.conv3x3_planes(xp, N, H, W, pw, out=y1[..., :cout], bias=bias, prelu=slope, planes=s1, planes_prelu=pslope, planes2=s2, planes2_c0=8, wn=wn)
    q = hip_ops.Planes.alloc(N * H * W, cout, dev)
    hip.split_planes(y0[..., :cout].flatten(0, 2), q)
    qp = hip_ops.Planes.alloc(N * H * W, cout, dev)
    hip.split_planes(y0[..., :cout].flatten(0, 2), qp, prelu=pslope[:cout].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), maxdiff(y0, y1)
    assert torch.equal(s1.t, qp.t)
    assert torch.equal(s2.to_rows()[:, :, 8:8 + cout], q.to_rows()[:, :, :cout])


@pytest.mark.gpu
def test_conv3x3_planes_compact_fp32_tail(hip, dev):
    """out_cmin: only the channels >= out_cmin reach the fp32 output -- inside the full-width map (columns below untouched), or in a
    compact buffer of just those channels (the decoder's five flow / mask channels for warp_blend); the planes carry everything."""
    g = torch.Generator().manual_seed(7300)
    N, H, W, cin, cout = 1, 40, 72, 101, 101
    cmin = (cout - 5) // 4 * 4
    x = rnd(g, N, H, W, 104, scale=1.5).to(dev)
    wt = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)).to(dev)
    bias = rnd(g, cout, scale=0.2).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    xp = hip_ops.Planes.alloc(N * H * W, cin, dev)
    hip.split_planes(x[..., :cin].flatten(0, 2), xp)
    full = torch.empty(N, H, W, 104, device=dev)
    hip.conv3x3_planes(xp, N, H, W, pw, out=full[..., :cout], bias=bias)
    wide = torch.full((N, H, W, 104), 7.0, device=dev)
    s1 = hip_ops.Planes.alloc(N * H * W, cout, dev)
    hip.conv3x3_planes(xp, N, H, W, pw, out=wide[..., :cout], bias=bias, planes=s1, out_cmin=cmin)
    compact = torch.full((N, H, W, 8), 7.0, device=dev)
    s2 = hip_ops.Planes.alloc(N * H * W, cout, dev)
    hip.conv3x3_planes(xp, N, H, W, pw, out=compact[..., :cout - cmin], bias=bias, planes=s2, out_cmin=cmin)
    torch.cuda.synchronize()
    assert torch.equal(wide[..., cmin:cout], full[..., cmin:cout]) and (wide[..., :cmin] == 7.0).all()
    assert torch.equal(compact[..., :cout - cmin], full[..., cmin:cout]) and (compact[..., cout - cmin:] == 7.0).all()
    assert torch.equal(s1.t, s2.t)


@pytest.mark.gpu
def test_conv3x3_planes_chunk_offset_and_auto_width(hip, dev):
    """Input view starting at a 32-channel chunk of wider planes, automatic tile width, a large-ish ragged map: equals the fp32 kernel."""
    g = torch.Generator().manual_seed(7100)
    N, H, W, cin, cout = 1, 50, 70, 69, 101
    xb = rnd(g, N, H, W, 32 + 72, scale=1.5).to(dev)
    x = xb[..., 32:32 + cin]
    wt = rnd(g, cout, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)).to(dev)
    bias = rnd(g, cout, scale=0.2).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    y0 = torch.empty(N, H, W, 104, device=dev)
    hip.conv(x, pw, y0[..., :cout], 1, 1, 1, bias, None)
    xp = hip_ops.Planes.alloc(N * H * W, 32 + cin, dev)
    hip.split_planes(xb[..., :32 + cin].flatten(0, 2), xp)
    y1 = torch.empty(N, H, W, 104, device=dev)
    hip.conv3x3_planes(xp, N, H, W, pw, out=y1[..., :cout], bias=bias, in_chunk0=1, cin=cin)
    torch.cuda.synchronize()
    assert torch.equal(y0[..., :cout], y1[..., :cout])


PLANE_CONV_CASES = [
    # cin, cout, k, stride, pad, dil, H, W, N
    (24, 48, 3, 2, 1, 1, 20, 28, 2),           # encoder stride 2
    (48, 48, 3, 4, 1, 1, 24, 40, 2),           # fusion conv, stride 4
    (48, 48, 3, 4, 2, 2, 24, 40, 2),           # stride 4, dilation 2
    (64, 5, 1, 1, 0, 1, 10, 14, 2),            # 1x1 motion head
    (40, 24, 3, 2, 1, 1, 11, 13, 1),           # ragged channels and sizes
    (192, 288, 3, 2, 1, 1, 9, 15, 2),          # last_feat_extract.0
    (96, 37, 3, 1, 1, 1, 7, 9, 1),             # stride 1 through the GEMM path as well
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", PLANE_CONV_CASES, ids=lambda c: f"cin{c[0]}_cout{c[1]}_k{c[2]}_s{c[3]}_d{c[5]}")
def test_conv_from_split_planes(case, hip, dev):
    """CONV mode of the LDS-DMA GEMM (per-tap source rows, zero row for the padding): bit-identical to the fp32-input f16x3
    engine on the same values, fp32 output and plane sink (at a channel offset) alike; input view at a chunk offset."""
    cin, cout, k, stride, pad, dil, H, W, N = case
    g = torch.Generator().manual_seed(sum(case))
    r4 = lambda c: (c + 3) // 4 * 4
    xb = rnd(g, N, H, W, 32 + r4(cin), scale=1.5).to(dev)
    x = xb[..., 32:32 + cin]
    wt = rnd(g, cout, cin, k, k, scale=1.0 / np.sqrt(k * k * cin)).to(dev)
    bias, slope = rnd(g, cout, scale=0.2).to(dev), (torch.rand(cout, generator=g) * 0.4).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    oh = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ow = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y0 = torch.full((N, oh, ow, r4(cout)), 7.0, device=dev)
    forced = hip.conv3_instance
    hip.conv(x, pw, y0[..., :cout], stride, pad, dil, bias, slope)
    xp = hip_ops.Planes.alloc(N * H * W, 32 + cin, dev)
    hip.split_planes(xb[..., :32 + cin].flatten(0, 2), xp)
    y1 = torch.full((N, oh, ow, r4(cout)), 7.0, device=dev)
    c0 = 8
    sink = hip_ops.Planes.alloc(N * oh * ow, c0 + cout, dev)
    hip.conv_planes(xp, N, H, W, pw, out=y1[..., :cout], stride=stride, pad=pad, dil=dil, bias=bias, prelu=slope, sink=sink, sink_c0=c0,
                    in_chunk0=1)
    q = hip_ops.Planes.alloc(N * oh * ow, cout, dev)
    hip.split_planes(y0[..., :cout].flatten(0, 2), q)
    torch.cuda.synchronize()
    if not (k == 3 and stride == 1):           # (the 3x3 / stride-1 fp32 path is the halo kernel: other k order, compare to tolerance)
        assert torch.equal(y0, y1), maxdiff(y0, y1)
    assert maxdiff(y0[..., :cout], y1[..., :cout]) <= 2e-5
    if not (k == 3 and stride == 1):
        assert torch.equal(sink.to_rows()[:, :, c0:c0 + cout], q.to_rows()[:, :, :cout])
    assert hip.conv3_instance is forced


@pytest.mark.gpu
def test_conv_from_two_plane_sources(hip, dev):
    """down2.0 of the refiner reads cat(feat1, dec1[:, :192]) (network_base.py:421-422): as two plane buffers, no concat."""
    g = torch.Generator().manual_seed(8123)
    N, H, W = 1, 14, 18
    a = rnd(g, N, H, W, 64, scale=1.2).to(dev)
    bfull = rnd(g, N, H, W, 200, scale=1.2).to(dev)
    wt = rnd(g, 128, 256, 3, 3, scale=1.0 / np.sqrt(9 * 256)).to(dev)
    bias, slope = rnd(g, 128, scale=0.2).to(dev), (torch.rand(128, generator=g) * 0.4).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    cat = torch.cat([a, bfull[..., :192]], -1).contiguous()
    y0 = torch.empty(N, H // 2, W // 2, 128, device=dev)
    hip.conv(cat, pw, y0, 2, 1, 1, bias, slope)
    pa = hip_ops.Planes.alloc(N * H * W, 128, dev)                   # feat1 lives in channels 64..127 of its buffer
    hip.split_planes(torch.cat([torch.zeros_like(a), a], -1).flatten(0, 2), pa)
    pb = hip_ops.Planes.alloc(N * H * W, 197, dev)
    hip.split_planes(bfull[..., :197].flatten(0, 2), pb)
    y1 = torch.empty_like(y0)
    hip.conv_planes(pa, N, H, W, pw, out=y1, stride=2, pad=1, dil=1, bias=bias, prelu=slope, in_chunk0=2, x2=pb, split_chunks=2)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), maxdiff(y0, y1)


@pytest.mark.gpu
def test_gemm_split_k(dev):
    """Split-K of under-filled long-K plane-input GEMM launches (atmvfi_gemm_params.workspace; gemm_duo.hip + gemm_splitk_reduce_kernel):
    a linear with bias, residual and the window-reverse row map into a grouped view + plane sink, a stride-2 convolution from two
    plane sources into a plane sink (the refiner's down3.0 at 256 x 256), a deconv into a plane sink.  Against the unsplit launch:
    equal within fp32 summation-order error; bit-identical between two split launches; nothing changes without a workspace."""
    g = torch.Generator().manual_seed(8400)
    hip = hip_ops.HipOps(dev)
    used = []

    def scratch(n):
        used.append(n)
        return torch.empty(n, device=dev)

    def both(fn):
        hip.gemm_workspace = None
        a = fn()
        hip.gemm_workspace = scratch
        n0 = len(used)
        b = fn()
        c = fn()
        hip.gemm_workspace = None
        torch.cuda.synchronize()
        assert len(used) == n0 + 2, "the launcher did not ask for split-K scratch: the case tests nothing"
        return a, b, c

    # ---- linear: M 512, N 256, K 2048, residual + scatter map (a permutation with dropped rows) ----
    m, n, k = 512, 256, 2048
    x = rnd(g, m, k, scale=1.5).to(dev)
    xp = hip_ops.Planes.alloc(m, k, dev)
    hip.split_planes(x, xp)
    w = hip.pack_weight(GEMM_LINEAR, rnd(g, n, k, scale=1.0 / np.sqrt(k)).to(dev))
    bias, res = rnd(g, n, scale=0.2).to(dev), rnd(g, m, n).to(dev)
    perm = torch.randperm(m, generator=g).to(torch.int32)
    perm[::37] = -1
    rmap = perm.to(dev)

    def lin():
        out = torch.full((2, m // 2, n), 7.0, device=dev)
        sink = hip_ops.Planes.alloc(m // 2, 8 + 2 * n, dev)
        hip.linear(xp, w, out, bias=bias, residual=res, out_row_map=rmap, sink=sink, sink_c0=8, sink_gc=n)
        return out, sink
    (o0, s0), (o1, s1), (o2, s2) = both(lin)
    scale = max(1.0, float(o0[o0 != 7.0].abs().max()))
    assert not torch.equal(o0, o1) and maxdiff(o0, o1) <= 1e-5 * scale and torch.equal(o1, o2) and torch.equal(s1.t, s2.t)
    assert maxdiff(s0.to_float(), s1.to_float()) <= 1e-5 * scale
    assert (o1 == 7.0).sum() == (o0 == 7.0).sum()           # dropped rows stay untouched

    # ---- stride-2 3x3 convolution from two plane sources, plane sink (down3.0 of network_lite at 256 x 256: M 1024, N 128, K 2592) ----
    N, H, W = 1, 64, 64
    pa, pb = hip_ops.Planes.alloc(N * H * W, 64, dev), hip_ops.Planes.alloc(N * H * W, 229, dev)
    hip.split_planes(rnd(g, N * H * W, 64, scale=1.2).to(dev), pa)
    hip.split_planes(rnd(g, N * H * W, 232, scale=1.2).to(dev)[:, :229], pb)
    wc = hip.pack_weight(GEMM_CONV, rnd(g, 128, 64 + 224, 3, 3, scale=1.0 / np.sqrt(9 * 288)).to(dev))
    cb, cs = rnd(g, 128, scale=0.2).to(dev), (torch.rand(128, generator=g) * 0.4).to(dev)

    def conv():
        out = torch.full((N, H // 2, W // 2, 128), 7.0, device=dev)
        sink = hip_ops.Planes.alloc(N * (H // 2) * (W // 2), 128, dev)
        hip.conv_planes(pa, N, H, W, wc, out=out, stride=2, pad=1, dil=1, bias=cb, prelu=cs, sink=sink, x2=pb, split_chunks=2)
        return out, sink
    (o0, s0), (o1, s1), (o2, s2) = both(conv)
    scale = max(1.0, float(o0.abs().max()))
    assert not torch.equal(o0, o1) and maxdiff(o0, o1) <= 1e-5 * scale and torch.equal(o1, o2) and torch.equal(s1.t, s2.t)
    assert maxdiff(s0.to_float(), s1.to_float()) <= 1e-5 * scale

    # ---- deconv 2x2 / stride 2 into a plane sink: M 1024, N 4 x 61 -> 256, K 1024 ----
    n_, h_, w_, cin, cout = 1, 32, 32, 1024, 61
    xd = hip_ops.Planes.alloc(n_ * h_ * w_, cin, dev)
    hip.split_planes(rnd(g, n_ * h_ * w_, cin, scale=1.2).to(dev), xd)
    wd = hip.pack_weight(GEMM_DECONV, rnd(g, cin, cout, 2, 2, scale=1.0 / np.sqrt(cin)).to(dev))
    db, dsl = rnd(g, cout, scale=0.2).to(dev), (torch.rand(cout, generator=g) * 0.4).to(dev)

    def dec():
        sink = hip_ops.Planes.alloc(n_ * 2 * h_ * 2 * w_, cout, dev)
        hip.deconv(None, wd, None, bias=db, prelu=dsl, planes=xd, sink=sink, in_shape=(n_, h_, w_, cin))
        return sink
    s0, s1, s2 = both(dec)
    assert not torch.equal(s0.t, s1.t) and maxdiff(s0.to_float(), s1.to_float()) <= 1e-5 * max(1.0, float(s0.to_float().abs().max()))
    assert torch.equal(s1.t, s2.t) and (s1.t[:, :, s1.rows:] == 0).all() and (s1.to_rows()[:, :, cout:] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 3, 17, 29), (2, 3, 17, 28), (1, 3, 68, 120), (1, 5, 34, 60)], ids=lambda s: "x".join(map(str, s)))
def test_flow_warp_up2_equals_two_launches(shape, hip, dev):
    """flow_warp + x2 flow up-sampling in one launch == the two kernels, bit for bit (incl. flows that leave the image); with W a
    multiple of 4 the launch is the LDS-staged form (atmvfi_flow_warp_up2_tiled), which must equal the direct one as well."""
    g = torch.Generator().manual_seed(99)
    b, c, h, w = shape
    src = torch.rand(b, c, h, w, generator=g).to(dev)
    flow = ((torch.rand(b, 2, h, w, generator=g) - 0.5) * 12).to(dev)
    flow[0, :, 0, 0] = 1e6
    flow[0, :, h // 2:, : w // 2] *= 8.0            # a region whose flows leave the staged box
    hip.warp_tiles = False
    d0, u0 = torch.empty_like(src), torch.empty(b, 2, 2 * h, 2 * w, device=dev)
    hip.flow_warp(src, flow, d0); hip.resize(flow, u0, 2.0)
    for tiles in (False, True):
        hip.warp_tiles = tiles
        d1, u1 = torch.full_like(src, 3.0), torch.full((b, 2, 2 * h, 2 * w), 3.0, device=dev)
        hip.flow_warp_up2(src, flow, d1, u1)
        torch.cuda.synchronize()
        assert torch.equal(d0, d1) and torch.equal(u0, u1), f"tiles={tiles}"
    hip.warp_tiles = True


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 96), (1, 37, 44), (3, 8, 4), (1, 136, 256), (1, 1088, 1920)], ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}")
@pytest.mark.parametrize("amp", [1.5, 9.0, 60.0], ids=["flows~1px", "flows~9px", "flows~60px"])
def test_tiled_warps_equal_direct_warps(shape, amp, hip, dev):
    """flow_warp / warp_blend with LDS-staged source tiles (atmvfi_flow_warp_tiled, atmvfi_warp_blend_tiled) == the direct-gather kernels,
    bit for bit: flows inside the staged box, flows beyond it (per-tile fallback to global gathers), taps that leave the image, inf / NaN
    flows, ragged tiles (H not a multiple of 8, W not a multiple of 32), several channel groups."""
    b, h, w = shape
    g = torch.Generator().manual_seed(int(amp * 10) + h * w)
    low = torch.randn(b, 5, max(2, h // 8), max(2, w // 8), generator=g)
    mot = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=True) * amp
    mot[0, 0, 0, 0] = float("inf"); mot[0, 1, h // 2, w // 2] = float("nan"); mot[-1, 2, -1, -1] = -1e9; mot[0, 3, 1, :] = 500.0
    mot[:, :, :2, : w // 2] *= 30.0            # one corner with far-reaching flows beside quiet tiles
    motion = mot.permute(0, 2, 3, 1).contiguous().to(dev)
    im0, im1 = torch.rand(b, 3, h, w, generator=g).to(dev), torch.rand(b, 3, h, w, generator=g).to(dev)
    src = torch.rand(b, 7, h, w, generator=g).to(dev)
    res = {}
    for tiles in (False, True):
        hip.warp_tiles = tiles
        outs = [torch.full((b, 3, h, w), 7.0, device=dev) for _ in range(3)]
        f0, f1 = (torch.full((b, 2, h, w), 7.0, device=dev) for _ in range(2))
        m1, m2 = (torch.full((b, 1, h, w), 7.0, device=dev) for _ in range(2))
        pl = hip_ops.Planes.alloc(b * h * w, 32, dev)
        pl.t.zero_()
        hip.warp_blend(im0, im1, motion, *outs, f0, f1, m1, m2, im0, im1, None, pack_planes=pl, pack_c0=8)
        pack = torch.full((b, h, w, 20), 7.0, device=dev)
        outs2 = [torch.full((b, 3, h, w), 7.0, device=dev) for _ in range(3)]
        hip.warp_blend(im0, im1, motion, *outs2, None, None, None, None, im0, im1, pack[..., 2:17])
        # flow_warp: 7 channels = three staging groups, planar flow and the NHWC motion view
        d1, d2 = torch.full((b, 7, h, w), 7.0, device=dev), torch.full((b, 7, h, w), 7.0, device=dev)
        hip.flow_warp(src, mot[:, 2:4].contiguous().to(dev), d1)
        hip.flow_warp(src, motion[..., 0:2].permute(0, 3, 1, 2), d2)
        torch.cuda.synchronize()
        res[tiles] = [*outs, f0, f1, m1, m2, pl.t.clone(), pack, *outs2, d1, d2]
    hip.warp_tiles = True
    for i, (x, y) in enumerate(zip(res[False], res[True])):
        assert torch.equal(torch.nan_to_num(x.float(), nan=-3.0), torch.nan_to_num(y.float(), nan=-3.0)), f"output {i} differs"
    assert not torch.isnan(res[True][0]).any()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 64, 96), (2, 136, 248), (1, 8, 8)], ids=lambda s: f"{s[0]}x{s[1]}x{s[2]}")
def test_image_pyramid_equals_sequential_resizes(shape, hip, dev):
    """One launch for the x0.5 pyramid levels 1..3 of both frames == F.interpolate-style resizes applied level by level, bit for bit."""
    b, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    im0, im1 = torch.rand(b, 3, h, w, generator=g).to(dev), torch.rand(b, 3, h, w, generator=g).to(dev)
    lv = [torch.full((2 * b, 3, h >> l, w >> l), 5.0, device=dev) for l in (1, 2, 3)]
    hip.image_pyramid(im0, im1, *lv)
    ref = [torch.empty_like(t) for t in lv]
    hip.resize(im0, ref[0][:b]); hip.resize(im1, ref[0][b:])
    hip.resize(ref[0], ref[1]); hip.resize(ref[1], ref[2])
    torch.cuda.synchronize()
    for a, r in zip(lv, ref):
        assert torch.equal(a, r), maxdiff(a, r)
    want = torch.nn.functional.interpolate(torch.cat([im0, im1], 0), scale_factor=0.5, mode="bilinear", align_corners=True)
    assert maxdiff(lv[0], want) <= 1e-6
    # with `pack`: pack_frames' NHWC4 stack of the two frames in the same launch, the levels unchanged
    lv2 = [torch.full_like(t, 5.0) for t in lv]
    x0, x1 = torch.full((2 * b, h, w, 4), 5.0, device=dev), torch.full((2 * b, h, w, 4), 6.0, device=dev)
    hip.image_pyramid(im0, im1, *lv2, pack=x0)
    hip.pack_frames(im0, im1, x1)
    torch.cuda.synchronize()
    assert torch.equal(x0, x1) and all(torch.equal(a, r) for a, r in zip(lv2, lv))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 68, 120, 576, 5), (2, 17, 23, 768, 5), (1, 9, 11, 100, 8), (1, 5, 7, 40, 1)], ids=lambda s: f"{s[3]}to{s[4]}")
def test_head1x1_planes(shape, hip, dev):
    """The 5-channel read-out of a motion MLP (network_base.py:158,195) on split-plane input: fp32 FMAs per pixel row against fp64, and
    against the GEMM path it replaces (which multiplies the same hi / lo' operands on the f16x3 MFMA)."""
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = rnd(g, n * h * w, cin, scale=2.0).to(dev)
    wt = rnd(g, cout, cin, 1, 1, scale=1.0 / cin ** 0.5).to(dev)
    b = rnd(g, cout, scale=0.5).to(dev)
    pw = hip.pack_weight(GEMM_CONV, wt)
    pl = hip_ops.Planes.alloc(n * h * w, cin, dev)
    hip.split_planes(x, pl)
    buf = torch.full((n, h, w, 8), 3.0, device=dev)
    hip.head1x1_planes(pl, n, h, w, pw, buf[..., :cout], bias=b)
    ref_buf = torch.full((n, h, w, 8), 3.0, device=dev)
    hip.conv_planes(pl, n, h, w, pw, out=ref_buf[..., :cout], stride=1, pad=0, dil=1, bias=b)
    torch.cuda.synchronize()
    ref = (x.double() @ wt.reshape(cout, cin).double().t() + b.double()).reshape(n, h, w, cout)
    assert (buf[..., :cout].double() - ref).abs().max().item() <= 2e-5
    assert (buf[..., :cout] - ref_buf[..., :cout]).abs().max().item() <= 2e-5
    assert torch.all(buf[..., cout:] == 3.0)


PP_CASES = [
    # kind, rows / geometry, Cin, Cout, extras -- every one with several tiles per workgroup of the persistent grid (> 256 tiles)
    ("linear", 70001, 96, 384, dict(bias=True, res=True)),            # 3 k-steps, ragged last row tile, 3 column blocks
    ("linear", 66000, 64, 200, dict(bias=True, res=False)),           # 2 k-steps (the shortest persistent K), ragged columns
    ("linear", 40000, 160, 1152, dict(bias=False, res=False)),        # 5 k-steps, 9 column blocks
    ("linear_map", 2 * 150 * 150, 96, 384, dict(bias=True, res=True)),   # window-reverse scatter with dropped (padded) rows
    ("linear_groups", 2 * 20000, 128, 128, dict()),                   # grouped [2, R, C] output view + plane sink at a group offset
    ("deconv", (1, 150, 220), 101, 101, dict()),                      # plane sink, ragged position blocks (104-wide), 4 k-steps
    ("deconv", (2, 96, 130), 64, 32, dict()),                         # 2 k-steps
    ("conv", (2, 300, 400), 48, 96, dict(stride=2, k=3, dil=1)),      # strided 3x3, 18 k-steps, taps outside the image
    ("conv", (1, 270, 480), 64, 160, dict(stride=1, k=1, dil=1)),     # 1x1 (2 k-steps), fp32 out + sink
    ("conv", (2, 260, 300), 32, 96, dict(stride=4, k=3, dil=2)),      # stride 4, dilation 2
]


@pytest.mark.gpu
@pytest.mark.parametrize("engine", [-3, -2, -4, 0], ids=["pp", "duo", "duo64", "auto"])
@pytest.mark.parametrize("case", PP_CASES, ids=lambda c: f"{c[0]}_{c[2]}to{c[3]}")
def test_gemm_pp_matches_reference_schedule(case, engine, dev):
    """gemm_pp.hip (ping-pong wave groups, persistent grid with the DMA ring flowing across tiles, LDS-transposed epilogue) and
    gemm_duo.hip (128 x 128 tiles, two workgroups per CU; atmvfi_gemm_params.tile_wn = -2) against the reference schedule
    gemm_split.hip (tile_wn = -1): same arithmetic, so bit-identical outputs, on shapes with several tiles per workgroup (the op tests
    above fit one round of 256 workgroups)."""
    kind, geom, cin, cout, ex = case
    # the reference schedule lives in the diagnostic library (round 6: the product library does not carry engines its forward never
    # launches); __graft_entry__.build() / `make` produce it next to the product
    ref_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "lib", "libatmvfi_hip_ref.so")
    assert os.path.exists(ref_lib), f"{ref_lib} missing: python -c 'import __graft_entry__ as g; g.build()'"
    hp, hr = hip_ops.HipOps(dev), hip_ops.HipOps(dev, lib_path=ref_lib)
    assert hr.lib.atmvfi_source_digest() == hp.lib.atmvfi_source_digest()          # the same sources
    hp.gemm_tile_wn = engine
    hr.gemm_tile_wn = -1
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    outs = []
    if kind.startswith("linear"):
        m = geom
        x = rnd(g, m, cin, scale=2.0).to(dev)
        w = rnd(g, cout, cin, scale=1.0 / cin ** 0.5).to(dev)
        b = rnd(g, cout, scale=0.5).to(dev) if ex.get("bias") else None
        r = rnd(g, m, cout, scale=0.5).to(dev) if ex.get("res") else None
        pw = hp.pack_weight(GEMM_LINEAR, w)
        pl = hip_ops.Planes.alloc(m, cin, dev)
        hp.split_planes(x, pl)
        for h in (hp, hr):
            if kind == "linear_map":
                geo = windows.build_window_geometry(2, 150, 150, 8, 4)
                mw = geo.row_map.numel()
                assert mw >= m
                rmap = geo.row_map.contiguous().to(dev)
                plw = hip_ops.Planes.alloc(mw, cin, dev)
                h.split_planes(torch.cat([x, x[:mw - m]], 0), plw)
                rr = torch.cat([r, r[:mw - m]], 0)
                y = torch.full((m, cout), 7.0, device=dev)
                h.linear(plw, pw, y, b, rr, rmap)
                outs.append([y])
            elif kind == "linear_groups":
                half = m // 2
                buf = torch.full((half, 8 + 2 * cout), 3.0, device=dev)
                y = buf[:, 8:8 + 2 * cout].unflatten(1, (2, cout)).permute(1, 0, 2)          # [2, half, cout] view, group stride cout
                sk = hip_ops.Planes.alloc(half, 8 + 2 * cout, dev)
                h.linear(pl, pw, y, sink=sk, sink_c0=8, sink_gc=cout)
                outs.append([buf, sk.t])
            else:
                n4 = (cout + 3) // 4 * 4
                y = torch.full((m, n4), 5.0, device=dev)[:, :cout]
                h.linear(pl, pw, y, b, r)
                outs.append([y])
    elif kind == "deconv":
        n, hh, ww = geom
        x = rnd(g, n * hh * ww, cin, scale=2.0).to(dev)
        wt = rnd(g, cin, cout, 2, 2, scale=1.0 / cin ** 0.5).to(dev)
        b, pr = rnd(g, cout, scale=0.5).to(dev), (torch.rand(cout, generator=g) * 0.4).to(dev)
        pw = hp.pack_weight(GEMM_DECONV, wt)
        pl = hip_ops.Planes.alloc(n * hh * ww, cin, dev)
        hp.split_planes(x, pl)
        for h in (hp, hr):
            sk = hip_ops.Planes.alloc(n * 4 * hh * ww, cout, dev)
            h.deconv(None, pw, None, bias=b, prelu=pr, planes=pl, sink=sk, in_shape=(n, hh, ww, cin))
            outs.append([sk.t])
    else:
        n, hh, ww = geom
        k, st, dil = ex["k"], ex["stride"], ex["dil"]
        pad = dil * (k // 2)
        x = rnd(g, n * hh * ww, cin, scale=2.0).to(dev)
        wt = rnd(g, cout, cin, k, k, scale=1.0 / (k * k * cin) ** 0.5).to(dev)
        b, pr = rnd(g, cout, scale=0.5).to(dev), (torch.rand(cout, generator=g) * 0.4).to(dev)
        pw = hp.pack_weight(GEMM_CONV, wt)
        pl = hip_ops.Planes.alloc(n * hh * ww, cin, dev)
        hp.split_planes(x, pl)
        oh, ow = (hh + 2 * pad - dil * (k - 1) - 1) // st + 1, (ww + 2 * pad - dil * (k - 1) - 1) // st + 1
        for h in (hp, hr):
            sk = hip_ops.Planes.alloc(n * oh * ow, cout, dev)
            y = torch.full((n, oh, ow, cout), 9.0, device=dev) if k == 1 else None
            h.conv_planes(pl, n, hh, ww, pw, out=y, stride=st, pad=pad, dil=dil, bias=b, prelu=pr, sink=sk)
            outs.append([sk.t] + ([y] if y is not None else []))
    torch.cuda.synchronize()
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_), maxdiff(a.float(), b_.float())


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [0, 3])
def test_atmformer_module_reference_fixture(shift, dev):
    """``from network.attention import ATMFormer`` (network_base.py:8): the stand-alone module, same constructor / state dict /
    forward signature as attention.py:216-334, on the reference's own smoke shape against the reference's outputs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from network.attention import ATMFormer, RefineBottleneck
    gold = G.load_npz(f"op_atm_ws7_shift{shift}")
    blk = ATMFormer(dim=128, num_heads=8, window_size=7, shift_size=shift)
    sd = {k[2:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("w.")}
    sd["attn.relative_coord"] = blk.state_dict()["attn.relative_coord"]
    assert set(sd) == set(blk.state_dict())
    blk.load_state_dict(sd, strict=True)
    blk.to(dev).eval()
    x = torch.from_numpy(gold["x"]).to(dev).reshape(4, 32, 32, 128)
    y, mo = blk(x, 32, 32, 2)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (4, 1024, 128) and tuple(mo.shape) == (4, 1024, 2)
    assert np.abs(y[:, ::4].cpu().numpy() - gold["y"]).max() <= 2e-4
    assert np.abs(mo.cpu().numpy() - gold["motion"]).max() <= 2e-4
    # RefineBottleneck: shape contract + agreement with Network's own enhancement block on the same weights
    rb = RefineBottleneck(dim=64, window_size=8, shift_size=4, mlp_ratio=2.0).to(dev).eval()
    out = rb(torch.randn(2, 16, 24, 64, device=dev))
    assert tuple(out.shape) == (2, 16 * 24, 64) and torch.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(128, 64, 1, 40, 72), (64, 32, 2, 21, 35), (128, 64, 1, 300, 520), (64, 32, 1, 16, 16), (128, 64, 2, 17, 50)],
                         ids=lambda c: f"cin{c[0]}_c{c[1]}_n{c[2]}_{c[3]}x{c[4]}")
def test_tail_fused(case, dev):
    """atmvfi_conv3x3_planes_readout + atmvfi_refine_tail (refine_head.0 -> refine_head.1 -> 2 sigmoid - 1 -> += I_t -> clamp,
    network_base.py:257-260, 429, 532-533, with the hidden map r1 kept in registers) against the chain in torch float64: both refiner
    widths (64: network_base, 32: network_lite), ragged tiles, several images, a map of one tile, and 627 tiles on 256 persistent
    workgroups; image borders are where the zero padding of the SECOND convolution has to be right (a tap contribution from outside
    the image is nothing, not conv(0) + bias)."""
    cin, c, n, h, w = case
    g = torch.Generator().manual_seed(9100 + cin + h + w)
    hip = hip_ops.HipOps(dev)
    xp = hip_ops.Planes.alloc(n * h * w, cin, dev)
    hip.split_planes(rnd(g, n * h * w, cin, scale=1.5).to(dev), xp)
    x = xp.to_float().cpu().double().reshape(n, h, w, cin).permute(0, 3, 1, 2)       # exactly what the kernel reads
    wa, ba, pa = rnd(g, c, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin)), rnd(g, c, scale=0.2), 0.25 + rnd(g, c, scale=0.2)
    wb, bb, pb = rnd(g, 3, c, 3, 3, scale=2.0 / np.sqrt(9 * c)), rnd(g, 3, scale=0.2), 0.25 + rnd(g, 3, scale=0.2)
    it = torch.rand(n, 3, h, w, generator=g)
    F = torch.nn.functional
    r1 = F.prelu(F.conv2d(x, wa.double(), ba.double(), padding=1), pa.double())
    r = F.prelu(F.conv2d(r1, wb.double(), bb.double(), padding=1), pb.double())
    want = it.double() + (2.0 * torch.sigmoid(r) - 1.0)
    pw = hip.pack_weight(GEMM_CONV, wa.to(dev))
    w2 = hip.pack_readout(wb.to(dev))
    contrib = torch.full((27, n * h * w), float("nan"), device=dev)
    hip.conv3x3_planes_readout(xp, n, h, w, pw, ba.to(dev), pa.to(dev), w2, contrib)
    s, cl = torch.full((n, 3, h, w), 7.0, device=dev), torch.full((n, 3, h, w), 7.0, device=dev)
    hip.refine_tail(contrib, bb.to(dev), pb.to(dev), it.to(dev), s, cl)
    torch.cuda.synchronize()
    assert not torch.isnan(contrib).any(), "a tap contribution was never written"
    err = (s.cpu().double() - want).abs().max().item()
    assert err <= 3e-5, f"max|d| {err:.3e}"
    assert torch.equal(cl, s.clamp(0, 1))
    # against the three unfused launches (same arithmetic class, another summation order)
    r1g = torch.empty(n, h, w, c, device=dev)
    hip.conv3x3_planes(xp, n, h, w, pw, out=r1g, bias=ba.to(dev), prelu=pa.to(dev))
    rg = torch.empty(n, h, w, 4, device=dev)
    hip.conv(r1g, hip.pack_weight(GEMM_CONV, wb.to(dev)), rg[..., :3], 1, 1, 1, bb.to(dev), pb.to(dev))
    s2, cl2 = torch.empty_like(s), torch.empty_like(s)
    hip.final_residual(it.to(dev), rg[..., :3], s2, cl2)
    torch.cuda.synchronize()
    assert maxdiff(s, s2) <= 2e-5
    # run-to-run
    contrib2 = torch.empty_like(contrib)
    hip.conv3x3_planes_readout(xp, n, h, w, pw, ba.to(dev), pa.to(dev), w2, contrib2)
    torch.cuda.synchronize()
    assert torch.equal(contrib, contrib2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(24, 48, 2, 64, 96), (24, 48, 1, 36, 52), (16, 32, 2, 40, 72), (24, 48, 1, 16, 32), (16, 32, 1, 2, 2),
                                  (24, 48, 3, 70, 34), (24, 48, 2, 544, 960), (16, 32, 1, 528, 992)],
                         ids=lambda c: f"c{c[0]}_{c[1]}_f{c[2]}_{c[3]}x{c[4]}")
def test_stem_fused(case, dev):
    """atmvfi_stem_fused (feat_extracts.0.0 -> 0.1 -> 1.0, network_base.py:99-110, in one launch, the two full-resolution maps in LDS)
    against the three layers in torch fp32: whole tiles, ragged tiles (sizes that are no multiples of the 16 x 32 full-resolution
    tile), a map smaller than one tile, both variants' channel counts, several frames.  Image borders are where the zero padding of
    EVERY layer has to be right (a layer-1 value outside the image is 0, not conv(0) + bias).  The two large cases have 2 040 and
    1 023 tiles on 256 persistent workgroups: every workgroup walks 4-8 tiles, which is what exercises the cross-tile hazards of the
    kernel (the frame patch overlaying the layer-3 weight region, the per-tile LDS-DMA re-staging of those weights, the next patch
    requested under phase E) at the unit tolerance; a second launch must reproduce the planes bit for bit."""
    c0, c1, f, h, w = case
    g = torch.Generator().manual_seed(100 * c0 + h + w)
    x = torch.zeros(f, h, w, 4)
    x[..., :3] = torch.rand(f, h, w, 3, generator=g)
    ws = [rnd(g, c0, 3, 3, 3, scale=0.6), rnd(g, c0, c0, 3, 3, scale=0.25), rnd(g, c1, c0, 3, 3, scale=0.25)]
    bs = [rnd(g, c, scale=0.3) for c in (c0, c0, c1)]
    ps = [0.25 + rnd(g, c, scale=0.2) for c in (c0, c0, c1)]
    t = x[..., :3].permute(0, 3, 1, 2).double()
    for wt, b, p, st in zip(ws, bs, ps, (1, 1, 2)):
        t = torch.nn.functional.prelu(torch.nn.functional.conv2d(t, wt.double(), b.double(), stride=st, padding=1), p.double())
    want = t.permute(0, 2, 3, 1).reshape(-1, c1).float()
    hip = hip_ops.HipOps(dev)
    args = []
    for wt, b, p in zip(ws, bs, ps):
        args += [wt.to(dev), b.to(dev), p.to(dev)]
    pk = hip.pack_stem(*args)
    out = hip_ops.Planes.alloc(f * (h // 2) * (w // 2), c1, dev)
    hip.stem_fused(x.to(dev), pk, out)
    torch.cuda.synchronize()
    got = out.to_float().cpu()
    err = (got - want).abs().max().item()
    assert err <= 3e-5 * max(1.0, want.abs().max().item()), f"max|d| {err:.3e} (|ref| max {want.abs().max().item():.2f})"
    # the planes' pad channels and spare row stay zero
    assert out.t[:, :, out.rows:].abs().max().item() == 0
    if c1 % 32:
        assert out.t[:, -1, :, c1 % 32:].abs().max().item() == 0
    out2 = hip_ops.Planes.alloc(f * (h // 2) * (w // 2), c1, dev)
    hip.stem_fused(x.to(dev), pk, out2)
    torch.cuda.synchronize()
    assert torch.equal(out.t, out2.t), "two launches of the fused stem differ"
