"""``Network``: the drop-in replacement for the reference's model class.

Same constructor, attributes, methods, ``state_dict`` and ``forward(im0, im1) -> dict`` as
``network/network_base.py:88-546`` / ``network/network_lite.py``, but
``forward`` is a sequence of hand-written HIP kernel launches (``hip_ops.HipOps`` ->
``libatmvfi_hip.so``) over an NHWC workspace:

* feature maps live channel-last; every ``torch.cat`` / ``einops.rearrange`` of the
  reference is realised by letting the producer write into a channel slice ("view") of
  the consumer's buffer, so none of them moves data;
* ``pad_if_needed`` / ``torch.roll`` / ``window_partition`` and their inverses are int32
  row maps (``windows.py``) consumed by the LayerNorm gather and the GEMM scatter;
* warps generate their coordinates in-kernel; flows and masks are read straight from the
  last five channels of the decoder maps.

The module holds parameters exactly like the reference (236 entries, same names), so
``load_state_dict(strict=True)`` of published checkpoints works; GEMM-layout copies of
the weights are derived lazily and refreshed when a parameter changes.

There is no CPU path: ``forward`` requires CUDA(HIP) tensors and the built library.
"""
from __future__ import annotations

import math
import operator
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import schema as S
from .hip_ops import GEMM_CONV, GEMM_DECONV, GEMM_LINEAR, HipOps, PackedWeight, Planes, PlanUnsupported
from .windows import WindowGeometry, build_window_geometry


_VERSION_OF = operator.attrgetter("_version")
_DATA_PTR_OF = torch.Tensor.data_ptr


_PARAM_EPOCH = [0]       # bumped whenever a Parameter OBJECT is (re)assigned, registered or deleted on any _Node / Network


class _ParamWatch:
    """Mixin: replacing a parameter object (``module.weight = nn.Parameter(...)``, ``register_parameter``, pruning or
    re-parametrisation hooks, ``del module.weight``) OR a whole sub-module (``net.x = other_node``, ``add_module`` /
    ``register_module``, ``del net.x``: parametrize / prune wrappers swap modules) bumps ``_PARAM_EPOCH``, which ``Network._param_sig``
    compares on every forward: its cached parameter list would otherwise keep the OLD objects and the packed weights, plans and graphs
    would stay stale."""

    def __setattr__(self, name, value):
        d = self.__dict__
        if isinstance(value, (nn.Parameter, nn.Module)) or name in d.get("_parameters", ()) or name in d.get("_modules", ()):
            _PARAM_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        d = self.__dict__
        if name in d.get("_parameters", ()) or name in d.get("_modules", ()):
            _PARAM_EPOCH[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        _PARAM_EPOCH[0] += 1
        super().register_parameter(name, param)

    def add_module(self, name, module):
        _PARAM_EPOCH[0] += 1
        super().add_module(name, module)

    def register_module(self, name, module):
        _PARAM_EPOCH[0] += 1
        super().register_module(name, module)


class _Node(_ParamWatch, nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _r4(c: int) -> int:
    return (c + 3) // 4 * 4


class _PMap:
    """A feature map that lives in split planes only: channels [32 * chunk0, 32 * chunk0 + c) of ``p``, rows = pixels of [n, h, w]."""
    __slots__ = ("p", "n", "h", "w", "chunk0", "c")

    def __init__(self, p, n, h, w, chunk0, c):
        self.p, self.n, self.h, self.w, self.chunk0, self.c = p, n, h, w, chunk0, c


class BlockRunner:
    """Workspace + one-transformer-block machinery shared by ``Network`` and the stand-alone ``ATMFormer`` / ``RefineBottleneck``
    modules (``atm-vfi_amd/blocks.py``): named device buffers, split-plane buffers, cached window maps, and ``_block``."""

    def _init_runner(self):
        self._ops_obj = None
        self.use_split_planes = True       # kernel selection (Network's `selections` argument): split planes between contraction layers
        self._bufs: Dict[Tuple, object] = {}
        self._geo: Dict[Tuple, Tuple[WindowGeometry, torch.Tensor, Optional[torch.Tensor]]] = {}

    def buf(self, name: str, *shape) -> torch.Tensor:
        key = (name,) + tuple(shape)
        t = self._bufs.get(key)
        if t is None:
            if getattr(self._ops_obj, "recording", None) is not None:
                raise PlanUnsupported(f"workspace buffer {name} created while recording")
            t = self._ops_obj.empty(*shape)
            self._bufs[key] = t
        return t

    def planes(self, name: str, rows: int, c: int) -> Planes:
        """Workspace rows in the split-plane format (zero-initialised once: the pad channels must stay finite)."""
        key = ("planes", name, rows, c)
        p = self._bufs.get(key)
        if p is None:
            if getattr(self._ops_obj, "recording", None) is not None:
                raise PlanUnsupported(f"workspace planes {name} created while recording")
            p = Planes.alloc(rows, c, self._ops_obj.device)
            self._bufs[key] = p
        return p

    def geometry(self, frames, h, w, ws, shift):
        key = (frames, h, w, ws, shift)
        g = self._geo.get(key)
        if g is None:
            if getattr(self._ops_obj, "recording", None) is not None:
                raise PlanUnsupported("window maps created while recording")
            geo = build_window_geometry(frames, h, w, ws, shift)
            ops = self._ops_obj
            g = (geo, ops.to_device_int(geo.row_map), None if geo.labels is None else ops.to_device_int(geo.labels))
            self._geo[key] = g
        return g


    def _block(self, ops, P, p, x, frames, h, w, ws, shift, cross, out, motion_dst, tag, out_sink=None, motion_sink=None):
        """One shifted-window transformer block (ATMFormer attention.py:265-334 when ``cross``,
        RefineBottleneck :433-495 otherwise).  x/out: token-matrix views in image order."""
        c = x.shape[-1]
        heads = S.NUM_HEADS
        hd = c // heads
        geo, row_map, labels = self.geometry(frames, h, w, ws, shift)
        mw = frames * geo.n_windows * geo.tokens
        bw = frames * geo.n_windows
        # With the f16x3 engine every nn.Linear input is written by its producer as split planes (fp16 hi / lo', same
        # bytes as fp32) and read by the GEMM through LDS-DMA; fp32 copies are kept only where something else reads them
        # (xn is the residual of proj: the reference adds onto the post-norm tensor).
        pl = getattr(ops, "precision", None) == "f16x3" and getattr(ops, "split_planes_ok", False) and self.use_split_planes
        xn = self.buf(f"{tag}xn", mw, c)
        xn_p = self.planes(f"{tag}xn_p", mw, c) if pl else None
        ops.layernorm(x, xn, P[f"{p}.norm1.weight"], P[f"{p}.norm1.bias"], src_row_map=row_map, **({"planes": xn_p} if pl else {}))
        qkv = self.buf(f"{tag}qkv", mw, 3 * c)
        ops.linear(xn_p if pl else xn, P[f"pk:{p}.attn.qkv.weight"], qkv)
        ao = None if pl else self.buf(f"{tag}ao", mw, c)
        ao_p = self.planes(f"{tag}ao_p", mw, c) if pl else None
        mo = self.buf(f"{tag}mo", mw, heads, 2) if cross else None
        ops.window_attention(qkv, ao, mo, labels, bw, geo.n_windows, ws, heads, hd, bw // 2 if cross else 0,
                             **({"planes": ao_p} if pl else {}))
        xb = self.buf(f"{tag}xb", frames * h * w, c)
        ops.linear(ao_p if pl else ao, P[f"pk:{p}.attn.proj.weight"], xb, bias=P[f"{p}.attn.proj.bias"], residual=xn, out_row_map=row_map)
        if cross:
            mk = {} if motion_sink is None else {"planes": motion_sink[0], "planes_c0": motion_sink[1], "planes_gc": motion_sink[2]}
            ops.motion_head(mo, row_map, P[f"{p}.attn.mlp.0.weight"], P[f"{p}.attn.mlp.0.bias"],
                            P[f"{p}.attn.mlp.2.weight"], P[f"{p}.attn.mlp.2.bias"], motion_dst, **mk)
        hid = P[f"{p}.mlp.fc1.bias"].shape[0]
        f1 = self.buf(f"{tag}fc1", frames, h, w, hid)
        if pl:
            y_p = self.planes(f"{tag}ln2_p", frames * h * w, c)
            ops.layernorm(xb, None, P[f"{p}.norm2.weight"], P[f"{p}.norm2.bias"], planes=y_p)
            ops.linear(y_p, P[f"pk:{p}.mlp.fc1.weight"], f1.reshape(frames * h * w, hid), bias=P[f"{p}.mlp.fc1.bias"])
            f2_p = self.planes(f"{tag}dw_p", frames * h * w, hid)
            ops.dwconv_gelu(f1, None, P[f"pk:{p}.mlp.dwconv.dwconv.weight"], P[f"{p}.mlp.dwconv.dwconv.bias"], planes=f2_p)
            sk = {} if out_sink is None else {"sink": out_sink[0], "sink_c0": out_sink[1], "sink_gc": out_sink[2]}
            ops.linear(f2_p, P[f"pk:{p}.mlp.fc2.weight"], out, bias=P[f"{p}.mlp.fc2.bias"], residual=xb, **sk)
        else:
            y = self.buf(f"{tag}ln2", frames * h * w, c)
            ops.layernorm(xb, y, P[f"{p}.norm2.weight"], P[f"{p}.norm2.bias"])
            ops.linear(y, P[f"pk:{p}.mlp.fc1.weight"], f1.reshape(frames * h * w, hid), bias=P[f"{p}.mlp.fc1.bias"])
            f2 = self.buf(f"{tag}dw", frames, h, w, hid)
            ops.dwconv_gelu(f1, f2, P[f"pk:{p}.mlp.dwconv.dwconv.weight"], P[f"{p}.mlp.dwconv.dwconv.bias"])
            ops.linear(f2.reshape(frames * h * w, hid), P[f"pk:{p}.mlp.fc2.weight"], out, bias=P[f"{p}.mlp.fc2.bias"], residual=xb)


class Network(_ParamWatch, BlockRunner, nn.Module):
    VARIANT = "base"

    SELECTIONS = ("use_split_planes", "use_plane_deconvs", "use_plane_convs", "use_unet_planes", "use_fused_stem", "use_fused_tail",
                  "use_splitk", "use_lanes", "use_plans")

    def __init__(self, global_motion: bool = True, ensemble_global_motion: bool = False, variant: Optional[str] = None,
                 selections: Optional[Dict[str, bool]] = None):
        """``global_motion`` / ``ensemble_global_motion``: the reference's constructor (network_base.py:89).  ``selections`` (not in
        the reference): overrides of the kernel-selection attributes in ``SELECTIONS`` (all default to the measured-best choice)."""
        super().__init__()
        self.variant_name = variant or self.VARIANT
        v = S.VARIANTS[self.variant_name]
        self._v = v
        # ---- attributes the reference exposes (network_base.py:91-201) ----
        self.pyramid_level = S.PYRAMID_LEVELS
        self.hidden_dims = list(v.hidden_dims)
        self.global_motion = global_motion
        self.ensemble_global_motion = ensemble_global_motion
        self.local_motion_args = {"window_size": v.local_window, "num_heads": S.NUM_HEADS, "patch_size": 1,
                                  "dim": v.local_dim, "enhance_window": 8}
        self.global_motion_args = {"window_size": v.global_window, "num_heads": S.NUM_HEADS, "patch_size": 1,
                                   "dim": v.global_dim}
        if self.variant_name == "lite":
            self.local_motion_args["mlp_ratio"] = v.mlp_ratio
            self.global_motion_args["mlp_ratio"] = v.mlp_ratio
        self.fused_dim = v.fused_dim
        self.motion_out_dim = S.MOTION_OUT
        self.fused_dim1, self.fused_dim2, self.fused_dim3 = v.decoder_widths
        self.fused_dims = [self.fused_dim1, self.fused_dim2, self.fused_dim3, 2 * self.fused_dim1]
        # ---- parameters / buffers under the reference's names ----
        gen = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        for sp in S.param_schema(v):
            t = S.init_tensor(sp, gen)
            parts = sp.key.split(".")
            node: nn.Module = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            if sp.is_buffer:
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t))
        # ---- runtime state (not part of the state dict) ----
        self._init_runner()
        self._precision = "f16x3"
        self._checked = False                        # "f16x3-checked": the library build that counts out-of-range operands
        # Kernel selections: every one of these has been "on" since round 2 and is bit-compatible with its alternative inside the
        # parity budget (tests/test_gpu_e2e.py::test_fallback_kernel_selections_end_to_end runs each alternative).  They are plain
        # attributes / constructor arguments (`Network(selections={"use_fused_stem": False})`), not environment variables: the
        # product reads no environment variable.
        self.use_plane_deconvs = True     # decoder deconvs from split planes
        self.use_plane_convs = True       # 3x3 convs on split-plane input
        self.use_unet_planes = True       # the refiner's strided convs on split planes
        self.use_fused_stem = True        # the encoder's first three layers in one launch
        self.use_fused_tail = True        # refine_head.1 folded into refine_head.0's epilogue
        self.use_splitk = True            # split-K of under-filled long-K 3x3 launches
        # independent branches of a forward on side streams (HipOps.branch).  OFF by default: bit-identical and planned
        # like everything else, but not faster -- 256x256 860.7 -> 853.3 frames/s, 256x448 1062.6 -> 1063.0, 576x960 169.9 -> 170.7,
        # 1088x1920 51.9 -> 52.0 (profiles/r05_lanes_ab.txt): what the overlapped launches save (~100 us of 1.2 ms at 256x256) the four
        # cross-queue event waits of a forward cost again.  Independent FORWARDS on streams of their own do pay: host_io.PairStreams.
        self.use_lanes = False
        self._prepared: Dict[str, object] = {}
        self._prepared_sig = None
        # Workspaces: one dict of named buffers per (device, input shape, mode) key, least recently used first.  The reference's
        # evaluation loops (benchmark/test_snufilm.py, test_xiph.py) feed one model frames of many sizes and never call
        # release_workspace(), so the cache evicts by itself: at most `max_workspaces` shapes and `workspace_cap_bytes` in total
        # (checked when a forward starts; the workspace in use always stays).  ~13 GB per 1080p shape of the GPU's 288 GB.
        self.max_workspaces = 2
        self.workspace_cap_bytes = 96 << 30
        self._workspaces: Dict[Tuple, Dict[Tuple, object]] = {}
        self._ws_key: Optional[Tuple] = None
        self._bufs: Dict[Tuple, object] = {}
        self._geo: Dict[Tuple, Tuple[WindowGeometry, torch.Tensor, Optional[torch.Tensor]]] = {}
        self.use_graphs = False
        self._frame_cache_on = False
        self._rows_fit_planes = True
        self._frame_cache = None          # (workspace key, tokens of the last call's second frame)
        self._reuse_first = False
        self._graphs: Dict[Tuple, Tuple] = {}
        self._graph_sig = None
        # launch plans (hip_ops.LaunchPlan): from the third forward with one (shape, mode, weights) key on, a forward is ONE
        # atmvfi_plan_run call with fresh output tensors; enable_plans(False) turns it off
        self.use_plans = True
        self._plans: Dict[Tuple, object] = {}       # key -> LaunchPlan | int (eager forwards seen so far) | False (cannot be planned)
        self._plan_sig = None
        self._plist = None                          # cached parameter list of the per-forward currency check (_param_sig)
        self._pepoch = -1
        self._primary: Optional["Network"] = None   # set on replicas (replica()): the model whose packed weights this one reads
        self._prepare_lock = None                   # created on the primary with its first replica
        for name, flag in (selections or {}).items():
            if name not in self.SELECTIONS:
                raise ValueError(f"unknown kernel selection {name!r}; one of {self.SELECTIONS}")
            setattr(self, name, bool(flag))

    # ------------------------------------------------------------------ API parity
    def __set_local_window_size__(self, window_size):       # network_base.py:262-265
        self.local_motion_args["window_size"] = window_size
        self._reset_relative_coord("local_motion_atmformer", window_size)

    def __set_global_window_size__(self, window_size):      # network_base.py:267-270
        self.global_motion_args["window_size"] = window_size
        self._reset_relative_coord("global_motion_atmformer", window_size)

    def _reset_relative_coord(self, branch: str, ws: int):
        # attention.py:167-170 re-registers the table for the new window
        for b in range(2):
            attn = self._modules[branch]._modules[str(b)]._modules["attn"]
            dev = attn._buffers["relative_coord"].device
            attn._buffers["relative_coord"] = S.relative_coord_table(ws).to(dev)

    _GLOBAL_PARTS = ("last_feat_extract", "global_feature_fusion", "global_motion_atmformer", "global_motion_mlp")
    _REFINE_PARTS = ("proj", "down1", "down2", "down3", "up1", "up2", "up3", "refine_head")
    _LOCAL_PARTS = ("feat_extracts", "cross_scale_feature_fusion", "local_motion_atmformer", "local_motion_mlp",
                    "feat_enhance_transformer", "upsample_pyramid") + _REFINE_PARTS

    def _grad(self, parts, flag):
        for p in parts:
            self._modules[p].requires_grad_(flag)

    def __freeze_global_motion__(self):       # network_base.py:272-276
        self._grad(self._GLOBAL_PARTS, False)

    def __finetune_global_motion__(self):     # :278-282
        self._grad(self._GLOBAL_PARTS, True)

    def __freeze_local_motion__(self):        # :284-298
        self._grad(self._LOCAL_PARTS, False)

    def __finetune_local_motion__(self):      # :300-314
        self._grad(self._LOCAL_PARTS, True)

    def __finetune_refinenet_only__(self):    # :316-334 (base only in the reference)
        self._grad(self._GLOBAL_PARTS, False)
        self._grad([p for p in self._LOCAL_PARTS if p not in self._REFINE_PARTS], False)
        self._grad(self._REFINE_PARTS, True)

    # ------------------------------------------------------------------ plumbing
    def set_ops(self, ops):
        """Inject the op backend (tests inject a CPU double to check host logic; the
        default is the HIP library and nothing else)."""
        self._ops_obj = ops
        self._drop_device_state()

    def set_precision(self, precision: str):
        """"f16x3" (default): contractions as fp16 hi/lo split, three 16-bit MFMAs per product, fp32 accumulate
        (~22 significand bits, operands must stay within the fp16 range).  "f32": every contraction on the exact-fp32
        MFMA (about 2.5x slower).  "f16x3-checked": the f16x3 arithmetic, bit for bit, on the CHECKED build of the library
        (libatmvfi_hip_checked.so), which counts every activation pair whose split hit the fp16 limit (|x| >= 65488, inf, NaN) --
        ``range_violations()`` reads the count; launch plans and graphs are off in this mode.  Not part of the reference's API."""
        if precision not in ("f16x3", "f32", "f16x3-checked"):
            raise ValueError("precision must be 'f16x3', 'f32' or 'f16x3-checked'")
        checked = precision == "f16x3-checked"
        if checked != self._checked:
            # another library: packed weights, workspaces, plans and the op backend belong to the one that made them
            self._ops_obj = None
            self._drop_device_state()
        self._checked = checked
        self._precision = "f16x3" if checked else precision
        if self._ops_obj is not None and hasattr(self._ops_obj, "precision"):
            self._ops_obj.precision = self._precision

    def range_violations(self, reset: bool = False) -> int:
        """Activation pairs split beyond the fp16 operand range since the model entered "f16x3-checked" (or since the last
        ``reset``): 0 means every contraction of every forward so far saw operands the f16x3 engines represent (DESIGN.md section 1,
        deviation 2); anything else means results may differ from the fp32 reference -- use ``set_precision("f32")``.  Synchronises."""
        if not self._checked:
            raise RuntimeError("range_violations(): the model is not in 'f16x3-checked' precision")
        ops = self._ops_obj
        if ops is None or getattr(ops, "range_word", None) is None:
            return 0
        n = int(ops.range_word.item()) & 0xffffffff
        if reset:
            ops.range_word.zero_()
        return n

    def _drop_device_state(self):
        """Everything that lives on one device or was derived there: packed weights, workspaces, window maps, captured graphs."""
        self._prepared = {}
        self._prepared_sig = None
        self._plist = None
        self._graphs.clear()
        self._plans.clear()
        self._workspaces.clear()
        self._bufs = {}
        self._ws_key = None
        self._geo.clear()
        self._frame_cache = None

    def _ops(self, device: torch.device):
        cur = getattr(self._ops_obj, "device", None)
        if isinstance(self._ops_obj, HipOps) and cur is not None and torch.device(cur) != torch.device(device):
            # the model (or its inputs) moved to another GPU: buffers, packed weights and maps of the old device must not meet
            # tensors of the new one in a launch
            self._ops_obj = None
            self._drop_device_state()
        if self._ops_obj is None:
            if device.type != "cuda":
                raise RuntimeError("atm-vfi_amd.Network.forward runs on MI355X only: move the model and its inputs to "
                                   "'cuda' (HIP). There is no CPU implementation in the product; the CPU oracle lives in oracle/.")
            self._ops_obj = HipOps(device, checked=self._checked)
            self._ops_obj.precision = self._precision
            self._ops_obj.gemm_workspace = self._gemm_scratch
        return self._ops_obj

    def _gemm_scratch(self, floats: int) -> Optional[torch.Tensor]:
        """Split-K scratch of the plane-input GEMM (hip_ops.HipOps.gemm_workspace): workspace memory like every other buffer."""
        if not self.use_splitk:
            return None
        lane = getattr(self._ops_obj, "lane", 0)          # (a branch on a side stream gets scratch of its own: HipOps.branch)
        return self.buf("gemm_splitk_ws" if not lane else f"gemm_splitk_ws_l{lane}", floats)

    def release_workspace(self):
        self._plans.clear()           # recorded plans and
        self._graphs.clear()          # captured graphs launch into the workspace
        self._workspaces.clear()
        self._bufs = {}
        self._ws_key = None
        self._frame_cache = None

    @staticmethod
    def _nbytes(ws: Dict) -> int:
        n = 0
        for t in ws.values():
            t = t.t if isinstance(t, Planes) else t
            n += t.numel() * t.element_size()
        return n

    def workspace_bytes(self) -> int:
        return sum(self._nbytes(ws) for ws in self._workspaces.values())

    def _select_workspace(self, key: Tuple):
        """Make the workspace of ``key`` current (creating it empty) and evict the least recently used others beyond the limits."""
        ws = self._workspaces.pop(key, None)
        self._workspaces[key] = {} if ws is None else ws          # most recently used last
        self._bufs = self._workspaces[key]
        self._ws_key = key
        while len(self._workspaces) > 1 and (len(self._workspaces) > self.max_workspaces or self.workspace_bytes() > self.workspace_cap_bytes):
            old = next(iter(self._workspaces))
            del self._workspaces[old]
            for gk in [g for g in self._graphs if g[-1] == old]:      # graphs captured into that workspace
                del self._graphs[gk]
            for gk in [g for g in self._plans if g[-1] == old]:       # plans recorded into it
                del self._plans[gk]
        if len(self._geo) > 64:                                        # window maps are small; bound them all the same
            for gk in list(self._geo)[:len(self._geo) - 64]:
                del self._geo[gk]
            self._plans.clear()                                        # a plan may hold a map that just went
            self._graphs.clear()

    # ------------------------------------------------------------------ weights
    def _apply(self, fn, *a, **k):          # .to() / .cuda() / .float(): parameters get new storage
        self._plist = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plist = None
        return super().load_state_dict(*a, **k)

    def invalidate_weights(self):
        """Forget the GEMM-layout copies of the weights.  Needed only after replacing a parameter's storage behind the module's back
        (``p.data = ...``): in-place updates, ``load_state_dict`` and ``.to()`` are noticed by themselves."""
        self._plist = None
        self._prepared_sig = None

    def _param_sig(self):
        """Per-forward check that the packed weights are current, ~40 us for the 236 entries: (1) the parameter OBJECTS are the cached
        ones (``_PARAM_EPOCH``: any assignment / registration / deletion of a Parameter on a sub-module re-walks the tree); (2) their
        storage pointers (``p.data = ...`` swaps); (3) their version counters (every in-place update through the parameter bumps
        one).  What nothing sees: writes through ``p.data`` (``p.data.add_()`` has a version counter of its own) --
        ``invalidate_weights()`` is for those."""
        pl = self._plist
        if pl is None or self._pepoch != _PARAM_EPOCH[0]:
            self._pepoch = _PARAM_EPOCH[0]
            pl = self._plist = [p for _, p in self.named_parameters()]
        return (tuple(map(_DATA_PTR_OF, pl)), tuple(map(_VERSION_OF, pl)), tuple(map(id, pl)))

    _RUNTIME_RESET = {"_ops_obj": None, "_prepared_sig": None, "_ws_key": None, "_graph_sig": None, "_plan_sig": None, "_plist": None,
                      "_pepoch": -1, "_frame_cache": None, "_primary": None, "_prepare_lock": None, "_reuse_first": False}
    _RUNTIME_DICTS = ("_bufs", "_geo", "_prepared", "_workspaces", "_graphs", "_plans")

    def __getstate__(self):
        """copy.deepcopy / pickle / torch.save(module): parameters, buffers and settings travel; device-side runtime state
        (workspaces, packed weights, plans, graphs, the op backend, a replica's link to its primary and its lock) does not -- the copy
        is a stand-alone model that builds its own on first use."""
        d = dict(self.__dict__)
        d.update(self._RUNTIME_RESET)
        for k in self._RUNTIME_DICTS:
            d[k] = {}
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        _PARAM_EPOCH[0] += 1

    def replica(self) -> "Network":
        """A second front end on the SAME parameters: shares this model's parameter objects and its GEMM-layout weight copies, owns
        its workspaces, window maps, launch plans and op backend.  What ``host_io.PairStreams`` builds K of to keep K independent
        forwards in flight on K streams (the path's parallel axis is the pair axis: demo_2x.py:129-168,
        benchmark/test_vimeo90k.py:80-110): no buffer is shared between two replicas, so no cross-stream event is needed inside a
        forward.  Results are bit-identical to this model's.  Not part of the reference's API."""
        import copy
        rep = copy.copy(self)                     # nn.Module: _parameters / _buffers / _modules are shared by reference
        rep._ops_obj = None
        # (not through nn.Module.__setattr__: that would register the primary as a sub-module of its own replica)
        object.__setattr__(rep, "_primary", self if self._primary is None else self._primary)
        if rep._primary.__dict__.get("_prepare_lock") is None:
            import threading
            object.__setattr__(rep._primary, "_prepare_lock", threading.Lock())
        rep.local_motion_args = dict(self.local_motion_args)
        rep.global_motion_args = dict(self.global_motion_args)
        rep._prepared, rep._prepared_sig = {}, None
        rep._workspaces, rep._ws_key, rep._bufs, rep._geo = {}, None, {}, {}
        rep._graphs, rep._graph_sig, rep._plans, rep._plan_sig = {}, None, {}, None
        rep._frame_cache_on, rep._frame_cache, rep._reuse_first = False, None, False
        rep._plist, rep._pepoch = None, -1
        return rep

    def _prepare(self, ops):
        if self._primary is not None:
            # a replica reads the primary's packed weights; if they have to be (re)built now, that happens on THIS call's stream and
            # the other replicas' streams must not read them early: wait for the packing kernels once (rare: a parameter changed)
            pm = self._primary
            with pm._prepare_lock:                 # replicas may run on worker threads (host_io.PairStreams)
                before = pm._prepared_sig
                P = pm._prepare_unlocked(pm._ops(ops.device) if isinstance(ops, HipOps) else ops)
                if pm._prepared_sig is not before and isinstance(ops, HipOps):
                    torch.cuda.current_stream(ops.device).synchronize()
                self._prepared, self._prepared_sig, self._plist = P, pm._prepared_sig, pm._plist
            return P
        lock = self._prepare_lock
        if lock is None:
            return self._prepare_unlocked(ops)
        with lock:                                 # a model with replicas: they may be preparing on other threads right now
            return self._prepare_unlocked(ops)

    def _prepare_unlocked(self, ops):
        sig = self._param_sig()
        if sig == self._prepared_sig:
            return self._prepared
        sd = {k: v for k, v in self.named_parameters()}
        P: Dict[str, object] = {}
        for k, p in sd.items():
            P[k] = p.detach()
        for sp in S.param_schema(self._v):
            k, shp = sp.key, sp.shape
            if sp.is_buffer or not k.endswith(".weight"):
                continue
            w = sd[k].detach()
            if len(shp) == 4 and "dwconv" in k:
                P["pk:" + k] = ops.pack_dw_weight(w)
            elif len(shp) == 4 and shp[2] == 2:                      # ConvTranspose2d (Cin,Cout,2,2)
                P["pk:" + k] = ops.pack_weight(GEMM_DECONV, w)
            elif len(shp) == 4 and shp[2] == 1 and ".proj." in k:    # fusion 1x1 conv used as a row GEMM
                P["pk:" + k] = ops.pack_weight(GEMM_LINEAR, w.reshape(shp[0], shp[1]))
            elif len(shp) == 4:
                P["pk:" + k] = ops.pack_weight(GEMM_CONV, w)
            elif len(shp) == 2 and ".attn.mlp." not in k:
                if k.endswith("attn.kv.weight"):
                    continue
                if k.endswith("attn.q.weight"):                      # [Wq; Wkv] -> one [3C,C] projection
                    kv = sd[k.replace(".q.weight", ".kv.weight")].detach()
                    P["pk:" + k.replace(".q.weight", ".qkv.weight")] = ops.pack_weight(GEMM_LINEAR, torch.cat([w, kv], 0))
                else:
                    P["pk:" + k] = ops.pack_weight(GEMM_LINEAR, w)
        # refiner input as split planes: [dec2 (w3+5) | zeros to the next multiple of 8 | 15 image channels]: the 16-channel pack of
        # warp_blend's plane sink then starts on an 8-channel boundary (proj.0.weight gets zero input channels at the gap)
        w3 = self._v.decoder_widths[2] + S.MOTION_OUT
        gap = (-w3) % 8
        if getattr(ops, "split_planes_ok", False):
            wp = sd["proj.0.weight"].detach()
            wpp = torch.cat([wp[:, :w3], torch.zeros(wp.shape[0], gap, 3, 3, dtype=wp.dtype, device=wp.device), wp[:, w3:]], 1)
            P["pk:proj.0.weight:planes"] = ops.pack_weight(GEMM_CONV, wpp.contiguous())
        if hasattr(ops, "pack_readout") and getattr(ops, "split_planes_ok", False) and self._v.refine_hidden in (32, 64):
            P["readout"] = ops.pack_readout(sd["refine_head.1.0.weight"].detach())
        if hasattr(ops, "pack_stem") and getattr(ops, "split_planes_ok", False):
            P["stem"] = ops.pack_stem(*(sd[f"feat_extracts.{a}.{b}"].detach() for a in ("0.0", "0.1", "1.0") for b in ("0.weight", "0.bias", "1.weight")))
        # f16x3 contraction operands saturate at +-65504 (deviation 2).  Weights are known here: say so once per
        # parameter version if a checkpoint comes near; for activations there is f16x3_deviation() below.
        if isinstance(ops, HipOps):
            wmax = float(torch.stack([sd[sp.key].detach().abs().max() for sp in S.param_schema(self._v)
                                      if not sp.is_buffer and sp.key.endswith(".weight") and len(sp.shape) >= 2]).max())
            self.weight_abs_max = wmax
            if wmax > 16384.0:
                import warnings
                warnings.warn(f"atm-vfi_amd: largest |weight| of this checkpoint is {wmax:.0f}; the f16x3 engines saturate operands at "
                              "65504 -- check Network.f16x3_deviation(im0, im1) or use set_precision('f32')")
        for st in (1, 2):     # leading PReLU of decoder stages 1-2, applied on the deconv's input load
            P[f"inprelu:{st}"] = ops.pad_channels(sd[f"upsample_pyramid.{st}.0.weight"])
        self._prepared = P
        self._prepared_sig = sig
        return P

    # ------------------------------------------------------------------ layers
    def _conv_act(self, ops, P, p, x, out, stride=1):
        ops.conv(x, P[f"pk:{p}.0.weight"], out, stride=stride, pad=1, dil=1, bias=P[f"{p}.0.bias"], prelu=P[f"{p}.1.weight"])

    def _conv_plain(self, ops, P, p, x, out, stride=1, pad=1, dil=1, planes=None, planes_prelu=None):
        if planes is not None:       # 3x3 epilogue also writes the split planes the next deconv reads (HIP backend only)
            ops.conv(x, P[f"pk:{p}.weight"], out, stride=stride, pad=pad, dil=dil, bias=P[f"{p}.bias"], planes=planes,
                     planes_prelu=planes_prelu)
            return
        ops.conv(x, P[f"pk:{p}.weight"], out, stride=stride, pad=pad, dil=dil, bias=P[f"{p}.bias"])

    def _plane_convs(self, ops) -> bool:
        """3x3 / stride-1 convs on the split-plane kernel (LDS-DMA halo, ping-pong wave groups): their inputs are written as
        split planes by the producing layer's sink, fp32 copies only where something other than a contraction reads them."""
        return self._plane_deconvs(ops) and self.use_plane_convs and self._rows_fit_planes

    def _c3p(self, ops, P, p, xp, n, h, w, out=None, act=True, sink=None, sink_c0=0, sink_prelu=None, wkey=None, out_cmin=0, sink2=None):
        """conv()/Conv2d 3x3 s1 p1 of the reference on split-plane input ``xp`` ([n*h*w rows]); ``p`` = parameter prefix
        (``p.0.weight``/``p.0.bias``/``p.1.weight`` with ``act``, ``p.weight``/``p.bias`` without)."""
        if act:
            wk, bias, prelu = wkey or f"pk:{p}.0.weight", P[f"{p}.0.bias"], P[f"{p}.1.weight"]
        else:
            wk, bias, prelu = wkey or f"pk:{p}.weight", P[f"{p}.bias"], None
        extra = {} if sink2 is None else {"planes2": sink2}
        if self.use_splitk and hasattr(ops, "conv3x3_workspace_floats"):
            # under-filled grids with long K (the motion MLPs of small frames): split K over the idle CUs; the scratch for the partial
            # sums is workspace memory like every other buffer
            need = ops.conv3x3_workspace_floats(n, h, w, P[wk].cin, P[wk].cout)
            if need:
                extra["workspace"] = self.buf("splitk_ws" if not getattr(ops, "lane", 0) else f"splitk_ws_l{ops.lane}", need)
        ops.conv3x3_planes(xp, n, h, w, P[wk], out=out, bias=bias, prelu=prelu, planes=sink, planes_c0=sink_c0, planes_prelu=sink_prelu,
                           out_cmin=out_cmin, **extra)

    def _conv_s2_sink(self, ops, P, p, x, sink, shape, out=None, sink_c0=0):
        """conv() 3x3 stride 2 (+PReLU) on the fp32-input GEMM engine, result to a plane sink (and ``out`` if given)."""
        ops.conv(x, P[f"pk:{p}.0.weight"], out, stride=2, pad=1, dil=1, bias=P[f"{p}.0.bias"], prelu=P[f"{p}.1.weight"],
                 planes=sink, planes_c0=sink_c0, out_shape=shape)

    def _conv_p(self, ops, P, p, src: "_PMap", stride=1, pad=1, dil=1, act=True, out=None, sink=None, sink_c0=0, src2: "Optional[_PMap]" = None):
        """conv() / nn.Conv2d of the reference (any stride / dilation, k 1 or 3) on a split-plane map through the LDS-DMA GEMM."""
        if act:
            wk, bias, prelu = f"pk:{p}.0.weight", P[f"{p}.0.bias"], P[f"{p}.1.weight"]
        else:
            wk, bias, prelu = f"pk:{p}.weight", P[f"{p}.bias"], None
        two = {} if src2 is None else {"x2": src2.p, "x2_chunk0": src2.chunk0, "split_chunks": src.c // 32}
        ops.conv_planes(src.p, src.n, src.h, src.w, P[wk], out=out, stride=stride, pad=pad, dil=dil, bias=bias, prelu=prelu, sink=sink,
                        sink_c0=sink_c0, in_chunk0=src.chunk0, **two)

    def _plane_deconvs(self, ops) -> bool:
        """Decoder deconvs on the LDS-DMA GEMM (operands as split planes) instead of the fp32-input engine."""
        return (getattr(ops, "precision", None) == "f16x3" and getattr(ops, "split_planes_ok", False) and self.use_split_planes
                and self.use_plane_deconvs)

    def _deconv_act(self, ops, P, p, x, out, in_prelu=None, split: Optional[str] = None, planes=None):
        """``planes``: the input rows already in split-plane form, through ``in_prelu`` (written by the producing 3x3 conv).
        ``split``: workspace name -- split the input rows into fp16 planes first (one pass, through ``in_prelu``).  Either way the
        deconv runs on the LDS-DMA GEMM: the fp32-input engine redoes the split once per column block (7x for 389 -> 4*197)."""
        w = P[f"pk:{p}.0.weight"]
        if planes is not None:
            ops.deconv(x, w, out, bias=P[f"{p}.0.bias"], prelu=P[f"{p}.1.weight"], planes=planes)
            return
        if split is not None and self._plane_deconvs(ops) and w.hi is not None:
            b, h, wd, cin = x.shape
            xp = self.planes(split, b * h * wd, cin)
            ops.split_planes(x.flatten(0, 2), xp, prelu=in_prelu)
            ops.deconv(x, w, out, bias=P[f"{p}.0.bias"], prelu=P[f"{p}.1.weight"], planes=xp)
            return
        ops.deconv(x, w, out, bias=P[f"{p}.0.bias"], prelu=P[f"{p}.1.weight"], in_prelu=in_prelu)

    def _encoder(self, ops, P, x0, tag: str):
        """shared_feat_extraction + cross-scale fusion buffers.  x0: [F,H,W,4].  Returns (e1, e2, fuse_l)
        with s3 already written into fuse_l[..., -d3:] (network_base.py:342-352)."""
        d = self._v.hidden_dims
        f, h, w, _ = x0.shape
        planes_path = self._plane_convs(ops) and min(d[1:]) >= 32 and (self._v.local_dim - d[3]) % 32 == 0
        fused = planes_path and self.use_fused_stem and "stem" in P and x0.is_contiguous()
        if not fused:
            a = self.buf(f"{tag}e0a", f, h, w, d[0]); self._conv_act(ops, P, "feat_extracts.0.0", x0[..., :3], a)
            e0 = self.buf(f"{tag}e0", f, h, w, d[0])
        if planes_path:
            # Stages 1-3 entirely in split planes: stride-2 conv (LDS-DMA GEMM, CONV mode) -> planes -> 3x3 conv (plane kernel) ->
            # planes; the last one writes s3 straight into the fusion buffer's planes.  No fp32 copy of e1, e2, s3 exists: their
            # only readers are contractions.
            # (The first stride-2 conv, 24 -> 48 channels at full resolution, stays on the fp32-input engine: the LDS-DMA GEMM's
            # 128-column tile is 62 % padding at N = 48 -- 0.40 ms against 0.33 -- and e0 exists in fp32 anyway.)
            # The full-resolution stem (3 -> d0 -> d0 -> d1 stride 2) is ONE launch whose two d0-channel maps stay in LDS
            # (atmvfi_stem_fused); A/B: use_fused_stem = False runs the three layers one by one through fp32 maps in HBM.
            if not fused:
                self._conv_act(ops, P, "feat_extracts.0.1", a, e0)
            ld = self._v.local_dim
            fuse_p = self.planes(f"{tag}fuse_l_p", f * (h // 8) * (w // 8), ld)
            src, outs = None, []
            for st in (1, 2, 3):
                hs, ws = h >> st, w >> st
                ap = self.planes(f"{tag}e{st}a_p", f * hs * ws, d[st])
                if st == 1 and fused:
                    ops.stem_fused(x0, P["stem"], ap)
                elif st == 1:
                    self._conv_s2_sink(ops, P, "feat_extracts.1.0", e0, ap, (f, hs, ws, d[1]))
                else:
                    self._conv_p(ops, P, f"feat_extracts.{st}.0", src, stride=2, sink=ap)
                if st == 3:
                    self._c3p(ops, P, "feat_extracts.3.1", ap, f, hs, ws, sink=fuse_p, sink_c0=ld - d[3])
                    outs.append(_PMap(fuse_p, f, hs, ws, 0, ld))
                else:
                    ep = self.planes(f"{tag}e{st}_p", f * hs * ws, d[st])
                    self._c3p(ops, P, f"feat_extracts.{st}.1", ap, f, hs, ws, sink=ep)
                    outs.append(_PMap(ep, f, hs, ws, 0, d[st]))
                    src = outs[-1]
            return outs[0], outs[1], outs[2]
        self._conv_act(ops, P, "feat_extracts.0.1", a, e0)
        a = self.buf(f"{tag}e1a", f, h // 2, w // 2, d[1]); self._conv_act(ops, P, "feat_extracts.1.0", e0, a, 2)
        e1 = self.buf(f"{tag}e1", f, h // 2, w // 2, d[1]); self._conv_act(ops, P, "feat_extracts.1.1", a, e1)
        a = self.buf(f"{tag}e2a", f, h // 4, w // 4, d[2]); self._conv_act(ops, P, "feat_extracts.2.0", e1, a, 2)
        e2 = self.buf(f"{tag}e2", f, h // 4, w // 4, d[2]); self._conv_act(ops, P, "feat_extracts.2.1", a, e2)
        a = self.buf(f"{tag}e3a", f, h // 8, w // 8, d[3]); self._conv_act(ops, P, "feat_extracts.3.0", e2, a, 2)
        fuse = self.buf(f"{tag}fuse_l", f, h // 8, w // 8, self._v.local_dim)
        self._conv_act(ops, P, "feat_extracts.3.1", a, fuse[..., self._v.local_dim - d[3]:])
        return e1, e2, fuse

    def _fusion(self, ops, P, p, fine, mid, fuse, c_mid, c_fine, tag):
        """CrossScaleFeatureFusion (network_base.py:73-85); the coarsest scale is already in
        fuse[..., c_mid+2*c_fine:].  Returns LayerNorm'ed tokens [F*h*w, C]."""
        if isinstance(fuse, _PMap):
            # the three strided convs sink into the fusion buffer's planes next to the coarsest scale; the 1x1 projection is the
            # LDS-DMA GEMM on those planes (no fp32 fusion buffer, no fp32 -> fp16-pair conversion per column block)
            f, h, w, c = fuse.n, fuse.h, fuse.w, fuse.c
            self._conv_p(ops, P, f"{p}.layers.0", mid, stride=2, pad=1, dil=1, act=False, sink=fuse.p, sink_c0=0)
            self._conv_p(ops, P, f"{p}.layers.1", fine, stride=4, pad=1, dil=1, act=False, sink=fuse.p, sink_c0=c_mid)
            self._conv_p(ops, P, f"{p}.layers.2", fine, stride=4, pad=2, dil=2, act=False, sink=fuse.p, sink_c0=c_mid + c_fine)
            t = self.buf(f"{tag}fproj", f * h * w, c)
            ops.linear(fuse.p, P[f"pk:{p}.proj.weight"], t, bias=P[f"{p}.proj.bias"])
            out = self.buf(f"{tag}fnorm", f * h * w, c)
            ops.layernorm(t, out, P[f"{p}.norm.weight"], P[f"{p}.norm.bias"])
            return out
        f, h, w, c = fuse.shape
        self._conv_plain(ops, P, f"{p}.layers.0", mid, fuse[..., 0:c_mid], stride=2, pad=1, dil=1)
        self._conv_plain(ops, P, f"{p}.layers.1", fine, fuse[..., c_mid:c_mid + c_fine], stride=4, pad=1, dil=1)
        self._conv_plain(ops, P, f"{p}.layers.2", fine, fuse[..., c_mid + c_fine:c_mid + 2 * c_fine], stride=4, pad=2, dil=2)
        t = self.buf(f"{tag}fproj", f * h * w, c)
        ops.linear(fuse.reshape(f * h * w, c), P[f"pk:{p}.proj.weight"], t, bias=P[f"{p}.proj.bias"])
        out = self.buf(f"{tag}fnorm", f * h * w, c)
        ops.layernorm(t, out, P[f"{p}.norm.weight"], P[f"{p}.norm.bias"])
        return out

    @staticmethod
    def _stacked(buf, off, c):
        """View the channel range [off, off+2c) of an [B,h,w,LD] buffer as the frame-stacked token
        matrix [2, B*h*w, c]: '(N B) (H W) C -> B (N C) H W' of the reference without moving data."""
        b, h, w, ld = buf.shape
        flat = buf.reshape(b * h * w, ld)[:, off:off + 2 * c]
        return flat.unflatten(1, (2, c)).permute(1, 0, 2)

    def _motion_branch(self, ops, P, branch, mlp, x_tokens, b, h, w, ws, tag, mlp_lane=None):
        """Two ATMFormer blocks + motion MLP (estimate_{local,global}_motion, network_base.py:367-415).
        Returns (mlp_in buffer, last hidden map) -- the caller runs the 1x1 head into its own slice."""
        c = x_tokens.shape[-1]
        cin = 8 + 2 * c
        mlp_in = self.buf(f"{tag}mlp_in", b, h, w, cin)
        flat = mlp_in.reshape(b * h * w, cin)
        pc = self._plane_convs(ops)
        mlp_in_p = self.planes(f"{tag}mlp_in_p", b * h * w, cin) if pc else None
        x = x_tokens
        for blk in range(2):
            mdst = flat[:, 4 * blk:4 * blk + 4].unflatten(1, (2, 2)).permute(1, 0, 2)       # '(N B) L K -> B L (N K)'
            out = self._stacked(mlp_in, 8, c) if blk == 1 else self.buf(f"{tag}blk0", 2 * b * h * w, c)
            # the block's four motion channels (frame 0 dx dy, frame 1 dx dy at 4 blk ..) go into the plane input of the motion MLP
            # straight from the motion head (no separate split pass)
            msink = (mlp_in_p, 4 * blk, 2) if (pc and getattr(ops, "motion_head_sink", False)) else None
            self._block(ops, P, f"{branch}.{blk}", x, 2 * b, h, w, ws, 0 if blk == 0 else ws // 2, True, out, mdst, tag,
                        out_sink=(mlp_in_p, 8, c) if (pc and blk == 1) else None, motion_sink=msink)
            x = out
        hid = P[f"{mlp}.0.0.bias"].shape[0]
        if pc:
            # [motion 8 | frame0 C | frame1 C] as split planes: the features come from fc2's plane sink, the eight motion
            # channels (written by the two motion heads) from one small split pass
            if not getattr(ops, "motion_head_sink", False):
                ops.split_planes(flat[:, 0:8], mlp_in_p, c0=0)
            t1p = self.planes(f"{tag}mm1_p", b * h * w, hid)
            t2p = self.planes(f"{tag}mm2_p", b * h * w, hid)

            def convs():
                self._c3p(ops, P, f"{mlp}.0", mlp_in_p, b, h, w, sink=t1p)
                self._c3p(ops, P, f"{mlp}.1", t1p, b, h, w, sink=t2p)
            if mlp_lane is not None:
                with ops.branch(mlp_lane):
                    convs()
            else:
                convs()
            return mlp_in, t2p
        t1 = self.buf(f"{tag}mm1", b, h, w, hid); self._conv_act(ops, P, f"{mlp}.0", mlp_in, t1)
        t2 = self.buf(f"{tag}mm2", b, h, w, hid); self._conv_act(ops, P, f"{mlp}.1", t1, t2)
        return mlp_in, t2

    def _global_tokens(self, ops, P, e2, fuse_l, tag):
        """The per-frame half of estimate_global_motion (network_base.py:391-400): last_feat_extract + the global cross-scale fusion.
        Returns the LayerNorm'ed tokens [F*h_*w_, global_dim] of the F frames in ``e2`` / ``fuse_l``."""
        v = self._v
        d = v.hidden_dims
        if isinstance(fuse_l, _PMap):
            f, h8, w8 = fuse_l.n, fuse_l.h, fuse_l.w
            h_, w_ = h8 // 2, w8 // 2
            s3 = _PMap(fuse_l.p, f, h8, w8, (v.local_dim - d[3]) // 32, d[3])
            fuse_g = _PMap(self.planes(f"{tag}fuse_g_p", f * h_ * w_, v.global_dim), f, h_, w_, 0, v.global_dim)
            ap = self.planes(f"{tag}ga_p", f * h_ * w_, v.last_feat_dim)
            self._conv_p(ops, P, "last_feat_extract.0", s3, stride=2, sink=ap)
            self._c3p(ops, P, "last_feat_extract.1", ap, f, h_, w_, sink=fuse_g.p, sink_c0=d[3] + 2 * d[2])
            return self._fusion(ops, P, "global_feature_fusion", e2, s3, fuse_g, d[3], d[2], tag + "g")
        f, h8, w8, _ = fuse_l.shape
        h_, w_ = h8 // 2, w8 // 2
        s3 = fuse_l[..., v.local_dim - d[3]:]
        fuse_g = self.buf(f"{tag}fuse_g", f, h_, w_, v.global_dim)
        if self._plane_convs(ops):
            ap = self.planes(f"{tag}ga_p", f * h_ * w_, v.last_feat_dim)
            self._conv_s2_sink(ops, P, "last_feat_extract.0", s3, ap, (f, h_, w_, v.last_feat_dim))
            self._c3p(ops, P, "last_feat_extract.1", ap, f, h_, w_, out=fuse_g[..., d[3] + 2 * d[2]:])
        else:
            a = self.buf(f"{tag}ga", f, h_, w_, v.last_feat_dim)
            self._conv_act(ops, P, "last_feat_extract.0", s3, a, 2)
            self._conv_act(ops, P, "last_feat_extract.1", a, fuse_g[..., d[3] + 2 * d[2]:])
        return self._fusion(ops, P, "global_feature_fusion", e2, s3, fuse_g, d[3], d[2], tag + "g")

    def _global_from_tokens(self, ops, P, tokens, b, h_, w_, tag):
        """The pair half of estimate_global_motion (network_base.py:401-415): two ATMFormer blocks + the motion MLP on the frame-stacked
        tokens [2B*h_*w_, global_dim].  Returns the raw 5-channel map [B,h_,w_,5] (in an 8-float-per-pixel buffer)."""
        mlp_in, t2 = self._motion_branch(ops, P, "global_motion_atmformer", "global_motion_mlp", tokens, b, h_, w_,
                                         self.global_motion_args["window_size"], tag + "g")
        gout = self.buf(f"{tag}gout", b, h_, w_, 8)
        self._head1x1(ops, P, "global_motion_mlp.2", t2, b, h_, w_, gout[..., :5])
        return gout

    def _global_motion(self, ops, P, e2, fuse_l, b, tag):
        """estimate_global_motion (network_base.py:391-415): returns the raw 5-channel map [B,h_,w_,5]."""
        if isinstance(fuse_l, _PMap):
            h_, w_ = fuse_l.h // 2, fuse_l.w // 2
        else:
            h_, w_ = fuse_l.shape[1] // 2, fuse_l.shape[2] // 2
        tokens = self._global_tokens(ops, P, e2, fuse_l, tag)
        return self._global_from_tokens(ops, P, tokens, b, h_, w_, tag)

    def _head1x1(self, ops, P, p, t2, b, h, w, out):
        """The 1x1 head of a motion MLP (network_base.py:158,195): on the hidden map's planes when the branch left it there."""
        if isinstance(t2, Planes) and P[f"pk:{p}.weight"].cout <= 8 and hasattr(ops, "head1x1_planes"):
            ops.head1x1_planes(t2, b, h, w, P[f"pk:{p}.weight"], out, bias=P[f"{p}.bias"])       # 5 channels: a lane per pixel, no GEMM tile
        elif isinstance(t2, Planes):
            ops.conv_planes(t2, b, h, w, P[f"pk:{p}.weight"], out=out, stride=1, pad=0, dil=1, bias=P[f"{p}.bias"])
        else:
            self._conv_plain(ops, P, p, t2, out, pad=0)

    def _ensemble_flows(self, ops, P, im0, im1):
        """multiscale_global_motion_ensemble (network_base.py:564-605): global flows from the x1, x1/2
        and x1/4 inputs; per sample keep the one whose warped full-resolution frames agree best."""
        b, _, h, w = im0.shape
        cands, losses = [], []
        c0, c1 = im0, im1
        for lvl in range(3):
            if lvl:
                n0 = self.buf(f"ens{lvl}i0", b, 3, h >> lvl, w >> lvl); ops.resize(c0, n0)
                n1 = self.buf(f"ens{lvl}i1", b, 3, h >> lvl, w >> lvl); ops.resize(c1, n1)
                c0, c1 = n0, n1
            hh, ww = h >> lvl, w >> lvl
            if hh % 16 or ww % 16:
                raise ValueError(f"ensemble_global_motion needs H, W divisible by 64 (level {lvl} is {hh}x{ww})")
            x0 = self.buf(f"ens{lvl}x0", 2 * b, hh, ww, 4); ops.pack_frames(c0, c1, x0)
            _, e2, fuse = self._encoder(ops, P, x0, f"ens{lvl}")
            gout = self._global_motion(ops, P, e2, fuse, b, f"ens{lvl}")
            f0 = gout[..., 0:2].permute(0, 3, 1, 2)
            f1 = gout[..., 2:4].permute(0, 3, 1, 2)
            factor = 16 << lvl
            u0 = self.buf(f"ens{lvl}u0", b, 2, h, w); ops.resize(f0, u0, float(factor))      # global_alignmentness (:548-562)
            u1 = self.buf(f"ens{lvl}u1", b, 2, h, w); ops.resize(f1, u1, float(factor))
            wa = self.buf("ens_wa", b, 3, h, w); ops.flow_warp(im0, u0, wa)
            wb = self.buf("ens_wb", b, 3, h, w); ops.flow_warp(im1, u1, wb)
            loss = self.buf(f"ens{lvl}loss", b)
            ops.l1_mean(wa, wb, loss, workspace=self.buf("ens_l1_ws", ops.l1_mean_workspace_floats(b, 3 * h * w)) if hasattr(ops, "l1_mean_workspace_floats") else None)
            # candidate at the level-0 flow resolution (H/16): x1, x2, x4 up-sampling of the coarser flows
            if lvl == 0:
                k0 = self.buf("ens_c0_0", b, 2, h // 16, w // 16); ops.resize(f0, k0, 1.0)
                k1 = self.buf("ens_c1_0", b, 2, h // 16, w // 16); ops.resize(f1, k1, 1.0)
            else:
                k0 = self.buf(f"ens_c0_{lvl}", b, 2, h // 16, w // 16); ops.resize(f0, k0, float(1 << lvl))
                k1 = self.buf(f"ens_c1_{lvl}", b, 2, h // 16, w // 16); ops.resize(f1, k1, float(1 << lvl))
            cands.append((k0, k1)); losses.append(loss)
        if hasattr(ops, "ensemble_select"):                # the pick inside the C ABI: no device arithmetic outside it, plannable
            s0, s1 = self.buf("ens_sel0", b, 2, h // 16, w // 16), self.buf("ens_sel1", b, 2, h // 16, w // 16)
            ops.ensemble_select(losses, cands, s0, s1)
            return s0, s1
        ls = torch.stack(losses, 0)                        # [3,B]; first minimum wins like the reference's if/elif chain
        pick = ls.argmin(dim=0)
        sel0 = torch.stack([c[0] for c in cands], 0)       # [3,B,2,h_,w_]
        sel1 = torch.stack([c[1] for c in cands], 0)
        idx = torch.arange(b, device=pick.device)
        return sel0[pick, idx].contiguous(), sel1[pick, idx].contiguous()

    # ------------------------------------------------------------------ forward
    def enable_graphs(self, flag: bool = True):
        """Replay ``forward`` from a captured HIP graph (one per input shape / mode): the ~140 kernel launches of a pass are
        submitted as one graph launch, which removes the launch gaps (a few % at 1080p, most of the time on small frames).
        The returned tensors are the graph's static outputs: they are overwritten by the next call with the same shape.
        Not part of the reference's API; off by default."""
        self.use_graphs = bool(flag)
        if not flag:
            self._graphs.clear()

    def enable_frame_cache(self, flag: bool = True):
        """Video mode (demo_2x.py:129-168: pair i+1's first frame is pair i's second): keep the second frame's encoder + fusion
        tokens of every call, so that ``forward(im0, im1, reuse_first=True)`` runs ``shared_feat_extraction`` and the cross-scale
        fusion (network_base.py:342-352, 451-455) on the new frame only -- and, with ``global_motion`` on, the per-frame half of
        ``estimate_global_motion`` too (last_feat_extract + the global fusion, :391-400).  Exact: all of that is per frame, the pair
        meets in the ATMFormers.  Not with the ensemble (three input scales).  Not part of the reference's API; off by default."""
        self._frame_cache_on = bool(flag)
        self._frame_cache = None

    def f16x3_deviation(self, im0: torch.Tensor, im1: torch.Tensor) -> float:
        """max |I_t(f16x3) - I_t(exact fp32 engine)| on this frame pair: the run-time check that a checkpoint's activations stay inside
        the fp16 range of the split engines (an operand beyond 65504 saturates silently there; on the stress weights the largest
        |activation| is 24).  Costs one forward on each engine; expect ~1e-4, investigate above 1e-3."""
        keep = "f16x3-checked" if self._checked else self._precision
        try:
            self.set_precision("f32")
            ref = self.forward(im0, im1)["I_t"].clone()
            self.set_precision("f16x3")
            got = self.forward(im0, im1)["I_t"]
            return float((got - ref).abs().max())
        finally:
            self.set_precision(keep)

    def enable_plans(self, flag: bool = True):
        """Launch plans (default on): once a (shape, mode, weights) combination has run twice, its forward is recorded
        (``hip_ops.LaunchPlan``) and every later one is a single ``atmvfi_plan_run`` call -- same launches, same arguments, fresh
        output tensors -- instead of ~120 Python-side op calls.  Not used with the frame cache, graphs or per-launch profiling (those
        forwards contain copies outside the C ABI or need the individual launches).  Ensemble mode is planned too since round 4 (its
        per-sample pick is `atmvfi_ensemble_select`)."""
        self.use_plans = bool(flag)
        if not flag:
            self._plans.clear()

    def _mode_key(self, ops, im0, im1) -> Tuple:
        return (tuple(im0.shape), tuple(im1.shape), str(im0.device), self.global_motion, self.ensemble_global_motion,
                self._precision, self._checked, self.use_split_planes, self.use_plane_convs, self.use_unet_planes, self.use_plane_deconvs,
                self.use_fused_stem, self.use_splitk, self.use_fused_tail, getattr(ops, "attention_f16x3", None), self.local_motion_args["window_size"],
                self.global_motion_args["window_size"], getattr(ops, "warp_tiles", None), getattr(ops, "conv3_instance", None),
                getattr(ops, "gemm_tile_wn", None), self.use_lanes, self._workspace_key(im0))

    def forward(self, im0: torch.Tensor, im1: torch.Tensor, reuse_first: bool = False):
        self._reuse_first = bool(reuse_first)
        if not im0.is_cuda or self._frame_cache_on or self._checked:
            return self._forward_eager(im0, im1)
        if self.use_graphs:
            return self._forward_graph(im0, im1)
        ops = self._ops_obj
        pl = self._plist
        if (not self.use_plans or not isinstance(ops, HipOps) or ops.profile is not None
                or pl is None or pl[0].device != im0.device or torch.cuda.is_current_stream_capturing()):
            return self._forward_eager(im0, im1)         # (also every case that must raise: it validates devices and shapes)
        self._prepare(ops)
        if self._plan_sig is not self._prepared_sig:         # parameters changed: recorded launches hold stale weight pointers
            self._plans.clear()
            self._plan_sig = self._prepared_sig
        key = self._mode_key(ops, im0, im1)
        ent = self._plans.get(key, 0)
        if ent is False:
            return self._forward_eager(im0, im1)
        if isinstance(ent, int):
            if ent < 2:                                      # the first two forwards build the workspace, maps, kernel attributes
                self._plans[key] = ent + 1
                return self._forward_eager(im0, im1)
            return self._record_plan(ops, key, im0, im1)
        if im0.shape != im1.shape or im0.device != im1.device or im0.device != ops.device:
            return self._forward_eager(im0, im1)             # raises the proper error
        with torch.cuda.device(im0.device):
            a = im0.detach().contiguous().float()
            b = im1.detach().contiguous().float()
            if (a.data_ptr() & 15, b.data_ptr() & 15) != ent.align:
                # the recording chose kernels for its inputs' alignment (the LDS-staged warps load rows 16 bytes at a time): a
                # differently aligned view takes the direct launches, which choose again
                return self._forward_eager(im0, im1)
            self._select_workspace(key[-1])                  # replay counts as a use for the workspace LRU
            if ent.n_lanes > 1:
                ls, le = ops.lane_handles()
                return ent.run((a, b), ops.device, ops._stream(), lane_streams=ls, lane_events=le)
            return ent.run((a, b), ops.device, ops._stream())

    @staticmethod
    def _same_results(x, y) -> bool:
        if isinstance(x, torch.Tensor):
            return isinstance(y, torch.Tensor) and x.shape == y.shape and bool(torch.equal(x, y))
        if isinstance(x, dict):
            return isinstance(y, dict) and x.keys() == y.keys() and all(Network._same_results(v, y[k]) for k, v in x.items())
        if isinstance(x, (list, tuple)):
            return isinstance(y, (list, tuple)) and len(x) == len(y) and all(Network._same_results(u, v) for u, v in zip(x, y))
        return x == y

    def _record_plan(self, ops, key, im0, im1):
        a = im0.detach().contiguous().float()
        b = im1.detach().contiguous().float()
        na, nb = a.numel() * 4, b.numel() * 4
        if a.data_ptr() < b.data_ptr() + nb and b.data_ptr() < a.data_ptr() + na:
            # aliased / overlapping frames (net(x, x)): a recording could not tell a pointer into one from a pointer into the other
            # and every later net(a, b) of this shape would replay as net(a, a).  Stay eager; the next call with distinct frames records.
            return self._forward_eager(im0, im1)
        try:
            ops.begin_plan((a, b))
            out = self._forward_eager(a, b)
            plan = ops.end_plan(out)
        except PlanUnsupported:
            ops.abort_plan()
            self._plans[key] = False
            return self._forward_eager(im0, im1)
        except Exception:
            ops.abort_plan()
            raise
        # Record-time self-check: replay the fresh plan once into NaN-filled outputs and require the recording forward's results bit
        # for bit (the forward is run-to-run deterministic).  A pointer patched into the wrong slot, a missed patch or a launch that
        # was not recorded shows up here, before the plan ever serves a caller; such a shape stays on direct launches.
        with torch.cuda.device(a.device):
            ls, le = ops.lane_handles() if plan.n_lanes > 1 else (None, None)
            again = plan.run((a, b), ops.device, ops._stream(), poison=True, lane_streams=ls, lane_events=le)
        if self._same_results(out, again):
            self._plans[key] = plan
        else:
            import warnings
            warnings.warn("atm-vfi_amd: a recorded launch plan did not reproduce its own forward; this input shape stays on direct launches")
            self._plans[key] = False
        return out

    def _forward_graph(self, im0: torch.Tensor, im1: torch.Tensor):
        ops = self._ops(im0.device)
        self._prepare(ops)
        if self._graph_sig is not self._prepared_sig:        # parameters changed: the captured launches hold stale weights
            self._graphs.clear()
            self._graph_sig = self._prepared_sig
        key = self._mode_key(ops, im0, im1)
        ent = self._graphs.get(key)
        if ent is None:
            for _ in range(2):                               # validation, workspace, window maps, packed weights, kernel attributes
                self._forward_eager(im0, im1)
            torch.cuda.synchronize(im0.device)
            s0 = im0.detach().float().contiguous().clone()
            s1 = im1.detach().float().contiguous().clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._forward_eager(s0, s1)
            ent = (graph, s0, s1, out)
            self._graphs[key] = ent
        graph, s0, s1, out = ent
        self._select_workspace(key[-1])                      # replay counts as a use: the hot shape must not be the LRU victim
        s0.copy_(im0)
        s1.copy_(im1)
        graph.replay()
        return out

    def _workspace_key(self, im0: torch.Tensor) -> Tuple:
        return (str(im0.device), tuple(im0.shape), bool(self.global_motion), bool(self.ensemble_global_motion))

    def _forward_eager(self, im0: torch.Tensor, im1: torch.Tensor):
        if im0.shape != im1.shape or im0.dim() != 4 or im0.shape[1] != 3:
            raise ValueError(f"expected two [B,3,H,W] frames, got {tuple(im0.shape)} and {tuple(im1.shape)}")
        if im0.device != im1.device:
            raise ValueError(f"the two frames live on different devices ({im0.device}, {im1.device})")
        ops = self._ops(im0.device)
        if im0.is_cuda:
            for prm in self.parameters():
                if prm.device != im0.device:
                    raise RuntimeError(f"model parameters are on {prm.device}, inputs on {im0.device}: move the model with .to(device)")
                break
            with torch.cuda.device(im0.device):       # the kernels launch on the current device's stream
                return self._forward_on_device(ops, im0, im1)
        return self._forward_on_device(ops, im0, im1)

    def _lanes_on(self, ops, b: int, H: int, W: int) -> bool:
        """Independent branches of the forward on side streams (HipOps.branch)?  ``use_lanes``, except on the CPU test double and under
        per-launch profiling (which times launches on one stream)."""
        if not hasattr(ops, "branch") or getattr(ops, "profile", None) is not None:
            return False
        return bool(self.use_lanes)

    def _forward_on_device(self, ops, im0: torch.Tensor, im1: torch.Tensor):
        self._select_workspace(self._workspace_key(im0))
        if hasattr(ops, "begin_forward"):
            ops.begin_forward()
        b, _, H, W = im0.shape
        lanes = self._lanes_on(ops, b, H, W)
        # the plane kernels address a chunk's pixel rows with 32-bit byte offsets (64 B per row): beyond 2^26 rows of the largest
        # map (2 B frames at full resolution; e.g. batch 8 at 4K) the forward takes the fp32-input kernels instead
        self._rows_fit_planes = 2 * b * H * W < (1 << 26)
        need = 16 if self.global_motion else 8
        if H % need or W % need:
            raise ValueError(f"H and W must be multiples of {need} (got {H}x{W}); pad with InputPadder as the reference's callers do")
        v = self._v
        d = v.hidden_dims
        C = v.local_dim
        with torch.no_grad():
            im0 = im0.detach().contiguous().float()
            im1 = im1.detach().contiguous().float()
            P = self._prepare(ops)
            h, w = H // 8, W // 8
            # image pyramids (network_base.py:444-448)
            # Levels >= 1 of both frames live stacked along the batch axis ([:b] = frame 0, [b:] = frame 1): one launch per level
            # instead of two (the same kernels on 2b images); level 0 are the caller's two tensors.
            pyr0, pyr1 = [im0], [im1]
            pyr_st = [None]
            for l in range(1, 4):
                t = self.buf(f"pyr_{l}", 2 * b, 3, H >> l, W >> l)
                pyr_st.append(t)
                pyr0.append(t[:b]); pyr1.append(t[b:])
            # encoder + local fusion (:451-455)
            x0 = self.buf("x0", 2 * b, H, W, 4)
            if getattr(ops, "pyramid_packs", False):
                ops.image_pyramid(im0, im1, pyr_st[1], pyr_st[2], pyr_st[3], pack=x0)   # one launch: three levels of both frames + torch.cat as NHWC4
            else:
                ops.image_pyramid(im0, im1, pyr_st[1], pyr_st[2], pyr_st[3])
                ops.pack_frames(im0, im1, x0)
            cache_ok = self._frame_cache_on and not (self.global_motion and self.ensemble_global_motion)
            glob = self.global_motion and not self.ensemble_global_motion
            cg = v.global_dim
            ck = (self._ws_key, id(self._prepared_sig))     # same device, shape, mode and weights as the call that filled the cache
            hit = cache_ok and self._reuse_first and self._frame_cache is not None and self._frame_cache[0] == ck
            gtok = None
            if hit:
                # frame 0 of this pair was frame 1 of the previous call: encoder + fusions (everything that is per frame) on the
                # new frame only; the previous call's tokens of the shared frame are copied in front of the new ones
                e1, e2, fuse_l = self._encoder(ops, P, x0[b:], "fc")
                one = self._fusion(ops, P, "cross_scale_feature_fusion", e1, e2, fuse_l, d[2], d[1], "fcl")   # [B*h*w, C]
                feat = self.buf("lfnorm", 2 * b * h * w, C)
                feat[:b * h * w].copy_(self._frame_cache[1])
                feat[b * h * w:].copy_(one)
                if glob:
                    g_one = self._global_tokens(ops, P, e2, fuse_l, "fc")                                     # [B*h_*w_, cg]
                    n_g = b * (H // 16) * (W // 16)
                    gtok = self.buf("gfnorm", 2 * n_g, cg)
                    gtok[:n_g].copy_(self._frame_cache[2])
                    gtok[n_g:].copy_(g_one)
            else:
                e1, e2, fuse_l = self._encoder(ops, P, x0, "")
                if glob and lanes and isinstance(fuse_l, _PMap):
                    # the global branch's per-frame half (last_feat_extract + global fusion: reads e2 and the s3 chunks of the fusion
                    # planes, writes its own buffers) beside the local fusion (writes the OTHER chunks of those planes)
                    with ops.branch(1):
                        gtok = self._global_tokens(ops, P, e2, fuse_l, "")                                    # [2B*h_*w_, cg]
                    feat = self._fusion(ops, P, "cross_scale_feature_fusion", e1, e2, fuse_l, d[2], d[1], "l")
                    ops.join(1)
                else:
                    feat = self._fusion(ops, P, "cross_scale_feature_fusion", e1, e2, fuse_l, d[2], d[1], "l")   # [2B*h*w, C]
                    if glob:
                        gtok = self._global_tokens(ops, P, e2, fuse_l, "")                                    # [2B*h_*w_, cg]
            if cache_ok:
                keep = self.buf("frame_cache_tokens", b * h * w, C)
                keep.copy_(feat[b * h * w:])
                keep_g = None
                if glob:
                    n_g = b * (H // 16) * (W // 16)
                    keep_g = self.buf("frame_cache_gtokens", n_g, cg)
                    keep_g.copy_(gtok[n_g:])
                self._frame_cache = (ck, keep, keep_g)
            self._reuse_first = False
            it_list: List[torch.Tensor] = []
            w0_list: List[torch.Tensor] = []
            w1_list: List[torch.Tensor] = []
            x_tokens = feat
            if self.global_motion:                                                        # (:457-485)
                h_, w_ = H // 16, W // 16
                if self.ensemble_global_motion:
                    g0, g1 = self._ensemble_flows(ops, P, im0, im1)
                else:
                    gout = self._global_from_tokens(ops, P, gtok, b, h_, w_, "")
                    i_16 = self.buf("im_16", 2 * b, 3, h_, w_); ops.resize(pyr_st[3], i_16)
                    a, c, t = (ops.empty(b, 3, h_, w_) for _ in range(3))
                    ops.warp_blend(i_16[:b], i_16[b:], gout[..., :5], a, c, t)
                    w0_list.insert(0, a); w1_list.insert(0, c); it_list.insert(0, t)
                    g0 = gout[..., 0:2].permute(0, 3, 1, 2)
                    g1 = gout[..., 2:4].permute(0, 3, 1, 2)
                # the two global flows stacked like the frames: every warp / x2 up-sampling below is one launch for both
                gf = self.buf("gf_3", 2 * b, 2, h, w)
                if b == 1 and not self.ensemble_global_motion:
                    # one pair: [flow0 | flow1] of the motion map's first four channels IS the stacked layout [2, 2, h, w]: one launch
                    ops.resize(gout[..., 0:4].permute(0, 3, 1, 2), gf.view(1, 4, h, w), 2.0)
                else:
                    ops.resize(g0, gf[:b], 2.0); ops.resize(g1, gf[b:], 2.0)
                featw = self.buf("featw", 2 * b, h, w, C)
                ops.flow_warp_nhwc(feat.reshape(2 * b, h, w, C), gf, featw)
                x_tokens = featw.reshape(2 * b * h * w, C)
                for i in (3, 2, 1, 0):
                    nw = self.buf(f"pw_{i}", 2 * b, 3, H >> i, W >> i)
                    if i:       # the level's warp and the flow's x2 up-sampling to the next level: one launch
                        u = self.buf(f"gf_{i - 1}", 2 * b, 2, H >> (i - 1), W >> (i - 1))
                        ops.flow_warp_up2(pyr_st[i], gf, nw, u)
                        gf = u
                    else:
                        ops.flow_warp(im0, gf[:b], nw[:b]); ops.flow_warp(im1, gf[b:], nw[b:])
                    pyr0[i], pyr1[i] = nw[:b], nw[b:]
            # local motion (:490) -> raw motion map goes straight into the decoder input
            cdec = 2 * C + S.MOTION_OUT
            dec_in = self.buf("dec_in", b, h, w, _r4(cdec))
            motion8 = dec_in[..., 2 * C:2 * C + 5]
            a, c, t = (ops.empty(b, 3, h, w) for _ in range(3))
            # With lanes: the motion MLP (two 3x3 convs on the blocks' plane output), its 1x1 head and the H/8 synthesis run on lane 1
            # beside the two feature-enhancement blocks (which read the blocks' fp32 output and write their own buffers); both meet
            # at the warps of the enhanced features below.
            mlp_in, t2 = self._motion_branch(ops, P, "local_motion_atmformer", "local_motion_mlp", x_tokens, b, h, w,
                                             self.local_motion_args["window_size"], "l", mlp_lane=1 if lanes else None)

            def motion_tail():
                self._head1x1(ops, P, "local_motion_mlp.2", t2, b, h, w, motion8)
                ops.warp_blend(pyr0[3], pyr1[3], motion8, a, c, t)      # synthesis at H/8 (:496-506)
            if lanes:
                with ops.branch(1):
                    motion_tail()
            else:
                motion_tail()
            # feature enhancement (:493-494)
            x = self._stacked(mlp_in, 8, C)
            e_mid = self.buf("enh0", 2 * b * h * w, C)
            self._block(ops, P, "feat_enhance_transformer.0", x, 2 * b, h, w, 8, 0, False, e_mid, None, "e")
            enh = self.buf("enh1", 2 * b, h, w, C)
            self._block(ops, P, "feat_enhance_transformer.1", e_mid, 2 * b, h, w, 8, 4, False, enh.reshape(2 * b * h * w, C), None, "e")
            if lanes:
                ops.join(1)
            # warped features (:496-506)
            fl0 = motion8[..., 0:2].permute(0, 3, 1, 2)
            fl1 = motion8[..., 2:4].permute(0, 3, 1, 2)
            w0_list.insert(0, a); w1_list.insert(0, c); it_list.insert(0, t)
            ops.flow_warp_nhwc(enh[:b], fl0, dec_in[..., 0:C])
            ops.flow_warp_nhwc(enh[b:], fl1, dec_in[..., C:2 * C])
            # decoder (:511-528) writing into the U-Net's skip buffers
            rh = v.refine_hidden
            w1d, w2d, w3d = v.decoder_widths
            bufC = self.buf("bufC", b, H // 4, W // 4, 4 * rh + _r4(w1d + 5))      # [feat2_ | feat2 | dec0]
            bufB = self.buf("bufB", b, H // 2, W // 2, 2 * rh + _r4(w2d + 5))      # [feat1_ | feat1 | dec1]
            bufA = self.buf("bufA", b, H, W, 2 * rh)                               # [feat0_ | feat0]
            rin = self.buf("refine_in", b, H, W, v.refine_in)                      # [dec2 | im0 I0 im1 I1 It]
            dsts = (bufC[..., 4 * rh:4 * rh + w1d + 5], bufB[..., 2 * rh:2 * rh + w2d + 5], rin[..., 0:w3d + 5])
            x = dec_in[..., 0:cdec]
            flow0 = flow1 = m1 = m2 = None
            pd = self._plane_deconvs(ops)
            pc = self._plane_convs(ops) and rh >= 32
            # the refiner's strided convs on split planes too (ping-pong GEMM, CONV mode, two plane sources for the concats): the
            # first source (feat) must end on a 32-channel chunk boundary; the second (dec[:, :w]) may end inside a chunk -- what
            # follows there in the raw decoder planes (its flow / mask channels, then zeros) meets zero-padded weight channels
            unet_p = pc and self.use_unet_planes and rh % 32 == 0
            xp_next = None
            pack_c0 = (w3d + 5 + 7) // 8 * 8                                       # the 15 image channels inside the refiner's input planes
            rin_p = self.planes("refine_in_p", b * H * W, pack_c0 + 15) if pc else None
            for st, scale in enumerate((2, 1, 0)):
                pfx = f"upsample_pyramid.{st}"
                o = 1 if st else 0
                cout = dsts[st].shape[-1]
                hs, wsz = H >> scale, W >> scale
                mot_c = None
                if pc:
                    # planes end to end: deconv (LDS-DMA GEMM) -> planes -> conv+PReLU -> planes -> conv -> fp32 map (flows, masks
                    # and the refiner's strided convs read it) + planes for the next stage (through its leading PReLU) or for
                    # the refiner's first conv
                    if st == 0:
                        xp_next = self.planes("dec_xp_0", b * (hs // 2) * (wsz // 2), cdec)
                        ops.split_planes(x.flatten(0, 2), xp_next)
                    t1p = self.planes(f"dec_t1_{st}_p", b * hs * wsz, cout)
                    ops.deconv(None, P[f"pk:{pfx}.{o}.0.weight"], None, bias=P[f"{pfx}.{o}.0.bias"], prelu=P[f"{pfx}.{o}.1.weight"],
                               planes=xp_next, sink=t1p, in_shape=(b, hs // 2, wsz // 2, xp_next.c))
                    t2p = self.planes(f"dec_t2_{st}_p", b * hs * wsz, cout)
                    self._c3p(ops, P, f"{pfx}.{o + 1}", t1p, b, hs, wsz, sink=t2p)
                    if st < 2:
                        xp_next = self.planes(f"dec_xp_{st + 1}", b * hs * wsz, cout)
                        if unet_p:
                            # the map goes on twice as planes -- through the next stage's leading PReLU (its deconv) and raw (the
                            # U-Net's strided conv reads cat(feat, dec[:, :w]) from two plane buffers); fp32 only for the five
                            # flow / mask channels that warp_blend reads
                            # (in a compact 8-float-per-pixel buffer: as the tail of the skip buffer's 400-600-byte rows each
                            # pixel's 20 bytes cost warp_blend a cache line of their own)
                            raw = self.planes(f"dec_raw_{st}", b * hs * wsz, cout)
                            cmin = (cout - 5) // 4 * 4
                            mot_c = self.buf(f"dec_mot_{st}", b, hs, wsz, 8)[..., :cout - cmin]
                            self._c3p(ops, P, f"{pfx}.{o + 2}", t2p, b, hs, wsz, out=mot_c, act=False, sink=xp_next,
                                      sink_prelu=P[f"inprelu:{st + 1}"], sink2=raw, out_cmin=cmin)
                        else:
                            self._c3p(ops, P, f"{pfx}.{o + 2}", t2p, b, hs, wsz, out=dsts[st], act=False, sink=xp_next,
                                      sink_prelu=P[f"inprelu:{st + 1}"])
                    else:
                        # finest level: only the five flow / mask channels are read in fp32 (by warp_blend); the features go on
                        # to the refiner as planes
                        cmin = (cout - 5) // 4 * 4
                        mot_c = self.buf(f"dec_mot_{st}", b, hs, wsz, 8)[..., :cout - cmin]
                        self._c3p(ops, P, f"{pfx}.{o + 2}", t2p, b, hs, wsz, out=mot_c, act=False, sink=rin_p, out_cmin=cmin)
                else:
                    t1 = self.buf(f"dec_t1_{st}", b, hs, wsz, _r4(cout))[..., :cout]
                    # The deconvs run on the LDS-DMA GEMM from split planes (1.81 against 2.46 ms on the fp32-input engine for the six
                    # deconvs of the network, tools/bench_deconv_planes.py).  Stage 0's input has several producers (two warps, a 1x1
                    # conv): one split pass (0.03 ms); stages 1-2 get their planes from the epilogue of the 3x3 conv that produces
                    # their input, already through the stage's leading PReLU.
                    if st == 0 or not pd:
                        self._deconv_act(ops, P, f"{pfx}.{o}", x, t1, in_prelu=P[f"inprelu:{st}"] if st else None,
                                         split=f"dec_xp_{st}" if pd else None)
                    else:
                        self._deconv_act(ops, P, f"{pfx}.{o}", x, t1, planes=xp_next)
                    t2b = self.buf(f"dec_t2_{st}", b, hs, wsz, _r4(cout))[..., :cout]
                    self._conv_act(ops, P, f"{pfx}.{o + 1}", t1, t2b)
                    if pd and st < 2:
                        xp_next = self.planes(f"dec_xp_{st + 1}", b * hs * wsz, cout)
                        self._conv_plain(ops, P, f"{pfx}.{o + 2}", t2b, dsts[st], planes=xp_next, planes_prelu=P[f"inprelu:{st + 1}"])
                    else:
                        self._conv_plain(ops, P, f"{pfx}.{o + 2}", t2b, dsts[st])
                x = dsts[st]
                mot = x[..., cout - 5:cout] if mot_c is None else mot_c[..., mot_c.shape[-1] - 5:]
                a, c, t = (ops.empty(b, 3, hs, wsz) for _ in range(3))
                if scale == 0:
                    flow0, flow1 = ops.empty(b, 2, H, W), ops.empty(b, 2, H, W)
                    m1, m2 = ops.empty(b, 1, H, W), ops.empty(b, 1, H, W)
                    if pc:
                        ops.warp_blend(pyr0[0], pyr1[0], mot, a, c, t, flow0, flow1, m1, m2, im0, im1, None, pack_planes=rin_p,
                                       pack_c0=pack_c0)
                    else:
                        ops.warp_blend(pyr0[0], pyr1[0], mot, a, c, t, flow0, flow1, m1, m2, im0, im1, rin[..., w3d + 5:w3d + 20])
                else:
                    ops.warp_blend(pyr0[scale], pyr1[scale], mot, a, c, t)
                w0_list.insert(0, a); w1_list.insert(0, c); it_list.insert(0, t)
            # residual refinement U-Net (:417-431)
            r1 = None
            if pc:
                h2, w2, h4, w4 = H // 2, W // 2, H // 4, W // 4
                bufA_p = self.planes("bufA_p", b * H * W, 2 * rh)                  # [up3 out | feat0]
                bufB_p = self.planes("bufB_p", b * h2 * w2, 2 * rh)                # [up2 out | feat1]
                bufC_p = self.planes("bufC_p", b * h4 * w4, 4 * rh)                # [up1 out | feat2]
                d2a_p = self.planes("d2a_p", b * h4 * w4, 2 * rh)
                d3a_p = self.planes("d3a_p", b * h * w, 4 * rh)
                if unet_p:
                    # feat0 / feat1 / feat2 exist as planes only (chunks of bufA_p / bufB_p / bufC_p): their readers are the strided
                    # convs (CONV mode of the ping-pong GEMM) and the up-path's 3x3 convs
                    self._c3p(ops, P, "proj", rin_p, b, H, W, sink=bufA_p, sink_c0=rh, wkey="pk:proj.0.weight:planes")
                    self._conv_p(ops, P, "down1.0", _PMap(bufA_p, b, H, W, rh // 32, rh), stride=2, sink=bufB_p, sink_c0=rh)
                    self._conv_p(ops, P, "down2.0", _PMap(bufB_p, b, h2, w2, rh // 32, rh), stride=2, sink=d2a_p,
                                 src2=_PMap(self.planes("dec_raw_1", b * h2 * w2, w2d + 5), b, h2, w2, 0, w2d))
                    self._c3p(ops, P, "down2.1", d2a_p, b, h4, w4, sink=bufC_p, sink_c0=2 * rh)
                    self._conv_p(ops, P, "down3.0", _PMap(bufC_p, b, h4, w4, 2 * rh // 32, 2 * rh), stride=2, sink=d3a_p,
                                 src2=_PMap(self.planes("dec_raw_0", b * h4 * w4, w1d + 5), b, h4, w4, 0, w1d))
                else:
                    feat0 = bufA[..., rh:2 * rh]
                    self._c3p(ops, P, "proj", rin_p, b, H, W, out=feat0, sink=bufA_p, sink_c0=rh, wkey="pk:proj.0.weight:planes")
                    feat1 = bufB[..., rh:2 * rh]
                    # (64 -> 64 at full resolution: too narrow for the LDS-DMA GEMM's 128-column tile, see _encoder)
                    self._conv_s2_sink(ops, P, "down1.0", feat0, bufB_p, (b, h2, w2, rh), out=feat1, sink_c0=rh)
                    self._conv_s2_sink(ops, P, "down2.0", bufB[..., rh:2 * rh + w2d], d2a_p, (b, h4, w4, 2 * rh))
                    feat2 = bufC[..., 2 * rh:4 * rh]
                    self._c3p(ops, P, "down2.1", d2a_p, b, h4, w4, out=feat2, sink=bufC_p, sink_c0=2 * rh)
                    self._conv_s2_sink(ops, P, "down3.0", bufC[..., 2 * rh:4 * rh + w1d], d3a_p, (b, h, w, 4 * rh))
                d3b_p = self.planes("d3b_p", b * h * w, 4 * rh)
                self._c3p(ops, P, "down3.1", d3a_p, b, h, w, sink=d3b_p)
                d3c_p = self.planes("d3c_p", b * h * w, 4 * rh)
                self._c3p(ops, P, "down3.2", d3b_p, b, h, w, sink=d3c_p)
                u1a_p = self.planes("u1a_p", b * h4 * w4, 2 * rh)
                ops.deconv(None, P["pk:up1.0.0.weight"], None, bias=P["up1.0.0.bias"], prelu=P["up1.0.1.weight"], planes=d3c_p, sink=u1a_p,
                           in_shape=(b, h, w, 4 * rh))
                self._c3p(ops, P, "up1.1", u1a_p, b, h4, w4, sink=bufC_p, sink_c0=0)
                u2a_p = self.planes("u2a_p", b * h2 * w2, 2 * rh)
                ops.deconv(None, P["pk:up2.0.0.weight"], None, bias=P["up2.0.0.bias"], prelu=P["up2.0.1.weight"], planes=bufC_p, sink=u2a_p,
                           in_shape=(b, h4, w4, 4 * rh))
                self._c3p(ops, P, "up2.1", u2a_p, b, h2, w2, sink=bufB_p, sink_c0=0)
                ops.deconv(None, P["pk:up3.0.0.weight"], None, bias=P["up3.0.0.bias"], prelu=P["up3.0.1.weight"], planes=bufB_p, sink=bufA_p,
                           in_shape=(b, h2, w2, 2 * rh))
                fused_tail = self.use_fused_tail and "readout" in P and hasattr(ops, "conv3x3_planes_readout")
                if fused_tail:
                    # refine_head.0 -> refine_head.1 -> 2 sigmoid - 1 -> += -> clamp in two launches; the 64-channel full-resolution
                    # map r1 never reaches HBM: the first launch leaves 27 "tap contributions" per pixel, the second adds the nine
                    # shifted ones of every output pixel (include/atmvfi.h, atmvfi_conv3x3_planes_readout)
                    contrib = self.buf("tail_contrib", 27, b * H * W)
                    ops.conv3x3_planes_readout(bufA_p, b, H, W, P["pk:refine_head.0.0.weight"], P["refine_head.0.0.bias"],
                                               P["refine_head.0.1.weight"], P["readout"], contrib)
                else:
                    r1 = self.buf("r1", b, H, W, rh)
                    self._c3p(ops, P, "refine_head.0", bufA_p, b, H, W, out=r1)
            else:
                fused_tail = False
                r1 = self.buf("r1", b, H, W, rh)
                feat0 = bufA[..., rh:2 * rh]; self._conv_act(ops, P, "proj", rin, feat0)
                feat1 = bufB[..., rh:2 * rh]; self._conv_act(ops, P, "down1.0", feat0, feat1, 2)
                d2a = self.buf("d2a", b, H // 4, W // 4, 2 * rh); self._conv_act(ops, P, "down2.0", bufB[..., rh:2 * rh + w2d], d2a, 2)
                feat2 = bufC[..., 2 * rh:4 * rh]; self._conv_act(ops, P, "down2.1", d2a, feat2)
                d3a = self.buf("d3a", b, h, w, 4 * rh); self._conv_act(ops, P, "down3.0", bufC[..., 2 * rh:4 * rh + w1d], d3a, 2)
                d3b = self.buf("d3b", b, h, w, 4 * rh); self._conv_act(ops, P, "down3.1", d3a, d3b)
                feat3 = self.buf("d3c", b, h, w, 4 * rh); self._conv_act(ops, P, "down3.2", d3b, feat3)
                u1a = self.buf("u1a", b, H // 4, W // 4, 2 * rh); self._deconv_act(ops, P, "up1.0", feat3, u1a)
                self._conv_act(ops, P, "up1.1", u1a, bufC[..., 0:2 * rh])
                u2a = self.buf("u2a", b, H // 2, W // 2, 2 * rh); self._deconv_act(ops, P, "up2.0", bufC[..., 0:4 * rh], u2a)
                self._conv_act(ops, P, "up2.1", u2a, bufB[..., 0:rh])
                self._deconv_act(ops, P, "up3.0", bufB[..., 0:2 * rh], bufA[..., 0:rh])
                self._conv_act(ops, P, "refine_head.0", bufA, r1)
            it_sum, it_final = ops.empty(b, 3, H, W), ops.empty(b, 3, H, W)
            if fused_tail:
                ops.refine_tail(contrib, P["refine_head.1.0.bias"], P["refine_head.1.1.weight"], it_list[0], it_sum, it_final)
            else:
                r = self.buf("r", b, H, W, 4); self._conv_act(ops, P, "refine_head.1", r1, r[..., :3])
                ops.final_residual(it_list[0], r[..., :3], it_sum, it_final)
            i_t_0, i_t_1 = w0_list[0], w1_list[0]
            it_list[0] = it_sum          # the reference adds the residual in place (network_base.py:532)
        return {"I_t": it_final, "im_t_list": it_list, "im0_warped_list": w0_list, "im1_warped_list": w1_list,
                "opt_flow_0": flow0, "opt_flow_1": flow1, "I_t_0": i_t_0, "I_t_1": i_t_1,
                "occ_mask1": m1, "occ_mask2": m2}


class NetworkBase(Network):
    VARIANT = "base"


class NetworkLite(Network):
    VARIANT = "lite"
