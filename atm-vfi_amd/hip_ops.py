"""ctypes binding of ``libatmvfi_hip.so`` (C ABI in ``include/atmvfi.h``).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic
operation of the hot path is a hand-written HIP kernel reached through this module.
There is NO fallback: if the shared library is missing, or a tensor is not a CUDA
(= HIP) fp32 tensor, the call raises.

Tensors are passed as strided *views*: a channel slice ``buf[..., a:b]`` of an NHWC
buffer is a view with ``ld = buf.stride(-2)``; a frame-stacked token matrix is a 3-D
view ``[groups, rows, C]``.  The helpers below turn such views into the raw
(pointer, ld, group stride) triples of the C ABI and validate them on the host.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libatmvfi_hip.so")
CHECKED_LIB_PATH = os.path.join(_HERE, "libatmvfi_hip_checked.so")     # the same sources with -DATMVFI_RANGE_CHECK (csrc/Makefile)

GEMM_CONV, GEMM_LINEAR, GEMM_DECONV = 0, 1, 2

c_f = ctypes.c_void_p      # device pointers travel as void*
c_i = ctypes.c_int
c_l = ctypes.c_int64


class HipLibraryMissing(RuntimeError):
    pass


class GemmParams(ctypes.Structure):
    _fields_ = [
        ("mode", ctypes.c_int32),
        ("in_", c_f), ("in_ld", ctypes.c_int32), ("N", ctypes.c_int32), ("H", ctypes.c_int32),
        ("W", ctypes.c_int32), ("Cin", ctypes.c_int32),
        ("in_gstride", ctypes.c_int64), ("in_rpg", ctypes.c_int32),
        ("weight", c_f),
        ("Cout", ctypes.c_int32), ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride", ctypes.c_int32),
        ("pad", ctypes.c_int32), ("dil", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
        ("M", ctypes.c_int64),
        ("out", c_f), ("out_ld", ctypes.c_int32),
        ("out_gstride", ctypes.c_int64), ("out_rpg", ctypes.c_int32),
        ("out_row_map", c_f),
        ("bias", c_f), ("prelu", c_f), ("in_prelu", c_f), ("residual", c_f),
        ("res_ld", ctypes.c_int32),
        ("precision", ctypes.c_int32), ("weight_hi", c_f), ("weight_lo", c_f),
        ("in_hi", c_f), ("in_lo", c_f),
        ("out_hi", c_f), ("out_lo", c_f), ("out_plane_rows", ctypes.c_int64), ("out_plane_c0", ctypes.c_int32),
        ("out_plane_gc", ctypes.c_int32), ("tile_wn", ctypes.c_int32),
        ("in_hi2", c_f), ("in_lo2", c_f), ("in_ld2", ctypes.c_int32), ("in_split_chunks", ctypes.c_int32),
        ("workspace", c_f), ("workspace_floats", ctypes.c_int64),
    ]


class PlanArg(ctypes.Union):
    _fields_ = [("u", ctypes.c_uint64), ("i", ctypes.c_int64), ("f", ctypes.c_double)]


PLAN_MAX_ARGS = 28


class PlanOp(ctypes.Structure):
    _fields_ = [("fn", ctypes.c_int32), ("nargs", ctypes.c_int32), ("a", PlanArg * PLAN_MAX_ARGS)]


class PlanPatch(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("arg", ctypes.c_int32), ("slot", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("offset", ctypes.c_int64)]


# name -> (restype, argtypes); also the list of symbols the header declares (checked by the CPU tests)
SIGNATURES = {
    "atmvfi_version": (c_i, []),
    "atmvfi_last_error": (ctypes.c_char_p, []),
    "atmvfi_source_digest": (ctypes.c_char_p, []),
    "atmvfi_range_word_set": (c_i, [c_f, c_f]),
    "atmvfi_range_checked": (c_i, []),
    "atmvfi_gemm": (c_i, [ctypes.POINTER(GemmParams), c_f]),
    "atmvfi_gemm_workspace_floats": (c_l, [c_l, c_i, c_i]),
    "atmvfi_split_planes": (c_i, [c_f, c_i, c_l, c_i, c_f, c_f, c_f, c_i, c_f]),
    "atmvfi_split_planes_at": (c_i, [c_f, c_i, c_l, c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_conv2d": (c_i, [ctypes.POINTER(GemmParams), c_f]),
    "atmvfi_linear": (c_i, [ctypes.POINTER(GemmParams), c_f]),
    "atmvfi_deconv2x2": (c_i, [ctypes.POINTER(GemmParams), c_f]),
    "atmvfi_packed_weight_floats": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "atmvfi_pack_weight": (c_i, [c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_split_weight_halves": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "atmvfi_pack_weight_split": (c_i, [c_i, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_conv3x3_weight_halves": (c_l, [c_i, c_i]),
    "atmvfi_pack_weight_conv3x3": (c_i, [c_f, c_f, c_f, c_i, c_i, c_f]),
    "atmvfi_conv3x3_f16x3": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_l, c_f, c_i, c_i, c_f]),
    "atmvfi_conv3x3_planes": (c_i, [c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_l, c_i, c_f, c_i, c_i, c_f]),
    "atmvfi_flow_warp_up2": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_flow_warp_up2_tiled": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_image_pyramid": (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_image_pyramid_pack": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_head1x1_planes": (c_i, [c_f, c_f, c_l, c_l, c_i, c_f, c_f, c_i, c_f, c_i, c_f]),
    "atmvfi_conv3x3_planes2": (c_i, [c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_l, c_i, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f]),
    "atmvfi_conv3x3_planes3": (c_i, [c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_i, c_f, c_f, c_f, c_f, c_l, c_i, c_f, c_f, c_f, c_l, c_i, c_i, c_i, c_f, c_l, c_f]),
    "atmvfi_conv3x3_planes_workspace_floats": (c_l, [c_i, c_i, c_i, c_i, c_i]),
    "atmvfi_conv3x3_planes_readout": (c_i, [c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_l, c_f]),
    "atmvfi_refine_tail": (c_i, [c_f, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_layernorm": (c_i, [c_f, c_i, c_l, c_i, c_f, c_f, c_i, c_f, c_f, c_l, c_i, c_f, c_f, c_i, c_f]),
    "atmvfi_dwconv3x3_gelu": (c_i, [c_f, c_i, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f]),
    "atmvfi_pack_dw_weight": (c_i, [c_f, c_f, c_i, c_f]),
    "atmvfi_window_attention": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f]),
    "atmvfi_window_attention_f16x3": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_i, c_f]),
    "atmvfi_window_attn_cross_motion": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_window_attn_self": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_motion_head": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_l, c_i, c_l, c_i, c_f]),
    "atmvfi_motion_head_planes": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_l, c_i, c_l, c_i, c_f, c_f, c_l, c_i, c_i, c_f]),
    "atmvfi_flow_warp": (c_i, [c_f, c_f, c_l, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_flow_warp_tiled": (c_i, [c_f, c_f, c_l, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_flow_warp_ex": (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_flow_warp_nhwc": (c_i, [c_f, c_i, c_l, c_f, c_l, c_i, c_i, c_f, c_i, c_l, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_warp_blend": (c_i, [c_f, c_f, c_f, c_i, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i,
                                c_i, c_i, c_i, c_f]),
    "atmvfi_warp_blend_planes": (c_i, [c_f, c_f, c_f, c_i, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i,
                                       c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_warp_blend_tiled": (c_i, [c_f, c_f, c_f, c_i, c_l, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i,
                                      c_f, c_f, c_l, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_resize_bilinear_ac": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_i, c_i, c_i, c_i, c_i, c_i, ctypes.c_float, c_f]),
    "atmvfi_frame_u8_to_f32": (c_i, [c_f, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_i, c_f]),
    "atmvfi_frame_f32_to_u8": (c_i, [c_f, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_pack_frames": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_final_residual": (c_i, [c_f, c_f, c_i, c_f, c_f, c_i, c_i, c_i, c_f]),
    "atmvfi_l1_mean": (c_i, [c_f, c_f, c_f, c_i, c_l, c_f, c_l, c_f]),
    "atmvfi_l1_mean_workspace_floats": (c_l, [c_i, c_l]),
    "atmvfi_ensemble_select": (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_l, c_f]),
    "atmvfi_stem_fused": (c_i, [c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_l, c_f]),
    "atmvfi_plan_fn_id": (c_i, [ctypes.c_char_p]),
    "atmvfi_plan_run": (c_i, [ctypes.POINTER(PlanOp), c_i, ctypes.POINTER(PlanPatch), c_i, ctypes.POINTER(ctypes.c_uint64), c_i,
                              ctypes.POINTER(c_i), c_f]),
    "atmvfi_plan_run_lanes": (c_i, [ctypes.POINTER(PlanOp), c_i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(PlanPatch), c_i,
                                    ctypes.POINTER(ctypes.c_uint64), c_i, ctypes.POINTER(c_i), ctypes.POINTER(ctypes.c_void_p), c_i,
                                    ctypes.POINTER(ctypes.c_void_p), c_i]),
}
PLAN_RECORD, PLAN_WAIT = -2, -3        # include/atmvfi.h: the two synchronisation ops of atmvfi_plan_run_lanes


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


@dataclass
class PackedWeight:
    mode: int
    cout: int
    cin: int
    kh: int
    kw: int
    orig: torch.Tensor                 # the parameter (OIHW / [out,in] / IOHW)
    packed: Optional[torch.Tensor]     # GEMM layout on the device (None only for test doubles)
    hi: Optional[torch.Tensor] = None  # split-precision planes (fp16), k-step major, of the two GEMM engines (gemm_f16x3 / gemm_split)
    lo: Optional[torch.Tensor] = None
    hi3: Optional[torch.Tensor] = None  # 3x3 weights again in the conv3x3_f16x3 layout (k-step major, tap-packed channel tail)
    lo3: Optional[torch.Tensor] = None


@dataclass
class StemWeights:
    """Operands of ``atmvfi_stem_fused`` (HipOps.pack_stem)."""
    c0: int
    c1: int
    w1h: torch.Tensor
    w1l: torch.Tensor
    b1: torch.Tensor
    p1: torch.Tensor
    w2h: torch.Tensor
    w2l: torch.Tensor
    b2: torch.Tensor
    p2: torch.Tensor
    w3h: torch.Tensor
    w3l: torch.Tensor
    b3: torch.Tensor
    p3: torch.Tensor


class Planes:
    """Split-plane activation rows: one fp16 tensor [2, chunks, rows, 32] -- plane 0 = hi, plane 1 = lo' = (x - hi) * 1024, CHUNK
    MAJOR (channel c of row r at [c // 32, r, c % 32]) -- holding ``c`` real channels per row; the pad channels of the last chunk
    are finite (zero).  Producers write it in their epilogue; the split GEMM reads it by LDS-DMA, 16 rows x 64 bytes = one
    contiguous KiB per instruction."""

    def __init__(self, t: torch.Tensor, c: int, rows: Optional[int] = None):
        if t.dtype != torch.float16 or t.dim() != 4 or t.shape[0] != 2 or t.shape[3] != 32 or not t.is_contiguous() or c > t.shape[1] * 32 \
                or c <= (t.shape[1] - 1) * 32:
            raise ValueError(f"Planes: expected contiguous fp16 [2, ceil(C/32), rows, 32], got {tuple(t.shape)} {t.dtype} C={c}")
        if not t.is_cuda:
            raise TypeError("Planes: tensor must live on the GPU")
        if rows is not None and not 0 < rows <= t.shape[2]:
            raise ValueError(f"Planes: {rows} valid rows do not fit the {t.shape[2]} allocated ones")
        self.t, self.c = t, c
        self._rows = t.shape[2] if rows is None else rows

    @property
    def rows(self):
        """Rows that hold data.  ``ld_rows`` (>= rows) is the allocated row count = the chunk stride every kernel is given; rows past
        ``rows`` are never written and stay zero (the 3x3 kernel on plane input reads row ``rows`` for pixels outside the image)."""
        return self._rows

    @property
    def ld_rows(self):
        return self.t.shape[2]

    @property
    def chunks(self):
        return self.t.shape[1]

    def to_rows(self) -> torch.Tensor:
        """[2, rows, chunks*32] row-major copy (tests / debugging)."""
        return self.t[:, :, :self.rows].permute(0, 2, 1, 3).reshape(2, self.rows, self.chunks * 32)

    def to_float(self) -> torch.Tensor:
        """[rows, C] fp32 value hi + lo' / 1024 (tests / debugging)."""
        r = self.to_rows().float()
        return (r[0] + r[1] / 1024.0)[:, :self.c]

    @staticmethod
    def alloc(rows: int, c: int, device) -> "Planes":
        """Zero-initialised planes with one spare (zero) row behind the data."""
        return Planes(torch.zeros(2, (c + 31) // 32, rows + 1, 32, dtype=torch.float16, device=device), c, rows)


class TPtr(ctypes.c_void_p):
    """A device pointer that remembers the tensor it was derived from (``src``).  Launch plans classify pointer arguments by that tensor
    -- per-call memory or workspace -- never by what allocation the raw address happens to fall into: a biased or one-past-the-end
    pointer (the compact fp32 view of the 3x3 plane kernel passes ``buffer - 4 * out_cmin``) lies inside a NEIGHBOURING allocation."""
    __slots__ = ("src",)


def _ptr(t: Optional[torch.Tensor], byte_offset: int = 0):
    """-> ``TPtr`` of ``t.data_ptr() + byte_offset`` (None for a missing optional tensor)."""
    if t is None:
        return None
    p = TPtr(t.data_ptr() + byte_offset)
    p.src = t
    return p


def _chk(t: torch.Tensor, what: str):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.float32:
        raise TypeError(f"{what}: expected a CUDA/HIP float32 tensor, got "
                        f"{getattr(t, 'device', None)} {getattr(t, 'dtype', None)}")


def nhwc_view(t: torch.Tensor, what: str):
    """-> (ld, N, H, W, C) of an NHWC view (channel slice of a contiguous NHWC buffer)."""
    _chk(t, what)
    if t.dim() != 4:
        raise ValueError(f"{what}: expected [N,H,W,C], got {tuple(t.shape)}")
    n, h, w, c = t.shape
    ld = t.stride(2)
    if (c > 1 and t.stride(3) != 1) or (h > 1 and t.stride(1) != w * ld) or (n > 1 and t.stride(0) != h * w * ld):
        raise ValueError(f"{what}: not an NHWC pixel-contiguous view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld, n, h, w, c


def rows_view(t: torch.Tensor, what: str):
    """-> (ld, M, C, gstride, rpg) of a token matrix [M,C] or a grouped one [G,R,C]."""
    _chk(t, what)
    if t.dim() == 2:
        if t.stride(1) != 1:
            raise ValueError(f"{what}: rows must be channel-contiguous")
        return t.stride(0), t.shape[0], t.shape[1], 0, 0
    if t.dim() == 3:
        if t.stride(2) != 1:
            raise ValueError(f"{what}: rows must be channel-contiguous")
        g, r, c = t.shape
        return t.stride(1), g * r, c, t.stride(0), r
    raise ValueError(f"{what}: expected [M,C] or [G,R,C], got {tuple(t.shape)}")


def flow_view(flow: torch.Tensor, h: int, w: int, what: str):
    """[B,2,H,W] view (planar, or a permuted channel pair of an NHWC map) -> (bstride, pstride, cstride)."""
    _chk(flow, what)
    if flow.dim() != 4 or flow.shape[1] != 2 or flow.shape[2] != h or flow.shape[3] != w:
        raise ValueError(f"{what}: expected [B,2,{h},{w}], got {tuple(flow.shape)}")
    ps = flow.stride(3)
    if flow.stride(2) != w * ps:
        raise ValueError(f"{what}: rows of the flow view must be pixel-contiguous")
    return flow.stride(0), ps, flow.stride(1)


def _planar(t: torch.Tensor, c: int, what: str):
    _chk(t, what)
    if t.dim() != 4 or t.shape[1] != c or not t.is_contiguous():
        raise ValueError(f"{what}: expected contiguous [B,{c},H,W], got {tuple(t.shape)} strides {t.stride()}")


class PlanUnsupported(Exception):
    """Raised while recording when a forward cannot be expressed as a launch plan (the caller falls back to direct launches)."""


class LaunchPlan:
    """One forward of the hot path recorded as an array of ``atmvfi_plan_op`` (include/atmvfi.h, "Launch plans") and replayed by ONE
    ``atmvfi_plan_run`` call: per-call memory -- the caller's two frames and the output tensors, fresh on every call -- enters through
    a slot table that the library patches into the recorded arguments.

    Recording (``HipOps.begin_plan`` ... ``end_plan``): every launch of ``HipOps._run`` is executed as usual AND appended; every
    ``HipOps.empty`` becomes an output slot; pointer arguments (by the entry point's declared types) that fall inside a slot's
    memory become patches.  A pointer field of a GEMM parameter block inside per-call memory cannot be patched: PlanUnsupported."""

    debug = False      # diagnostic: replay one op per call, synchronised, announced on stderr first (finds the op that faults)

    def __init__(self, lib, inputs):
        self.lib = lib
        self.ops_list = []                 # (fn id, [values], [is_float])
        self.keep = []                     # parameter blocks referenced by address
        self.slots = []                    # (base, bytes): inputs first, then outputs
        self._slot_store = []              # per slot: address of the storage the slot's tensor lives in
        for t in inputs:
            self._add_slot(t)
        for i in range(len(self.slots)):
            for j in range(i):
                (bi, ni), (bj, nj) = self.slots[i], self.slots[j]
                if bi < bj + nj and bj < bi + ni:
                    raise PlanUnsupported("the input frames overlap in memory: a pointer into one could not be told from a pointer into the other")
        self.n_inputs = len(inputs)
        self.align = tuple(t.data_ptr() & 15 for t in inputs)      # kernel choices (16-byte row loads) were made for this alignment
        self.out_meta = []                 # (shape, dtype) per output slot, allocation order
        self.out_tensors = []              # the recording call's own outputs (kept alive until end_plan)
        self.patches = []
        self.template = None               # result structure with ("slot", k) leaves
        self.c_ops = self.c_patches = self.c_slots = None
        self._fn_ids = {}
        self.by_tensor = self.by_address = 0      # pointer arguments classified per-call by their tensor / refused by raw address
        self.lanes = []                    # per op: the lane (stream index) it is issued on; lane 0 = the caller's stream
        self.n_events = 0                  # synchronisation ops (HipOps.branch / join) number their events per plan

    # ---- recording ----
    def _add_slot(self, t: torch.Tensor):
        self.slots.append((t.data_ptr(), t.numel() * t.element_size()))
        self._slot_store.append(t.untyped_storage().data_ptr())

    def _slot_of(self, ptr: int):
        """The slot whose byte range holds the raw address ``ptr`` (assertions and untyped parameter-block fields only)."""
        for k, (base, nb) in enumerate(self.slots):
            if base <= ptr < base + nb:
                return k, ptr - base
        return None

    def _slot_of_tensor(self, t: torch.Tensor):
        """The slot a tensor belongs to: same storage AND the tensor starts inside the slot's bytes (two inputs may be slices of one
        stacked tensor).  None: the tensor is workspace / parameter memory, whatever its neighbours in the address space are."""
        st = t.untyped_storage().data_ptr()
        p = t.data_ptr()
        for k, (base, nb) in enumerate(self.slots):
            if self._slot_store[k] == st and base <= p < base + max(nb, 1):
                return k
        return None

    def add_output(self, t: torch.Tensor):
        self._add_slot(t)
        self.out_meta.append((tuple(t.shape), t.dtype))
        self.out_tensors.append(t)

    def add_sync(self, kind: int, event: int, lane: int):
        """A PLAN_RECORD / PLAN_WAIT op of event ``event`` on ``lane`` (HipOps.branch / join while recording)."""
        self.ops_list.append((kind, [("i", event)]))
        self.lanes.append(lane)
        self.n_events = max(self.n_events, event + 1)

    def add_op(self, fn, args, lane: int = 0):
        """Pointer arguments are ``TPtr`` (hip_ops._ptr): the tensor they were derived from decides whether they are per-call
        pointers (patched on replay, offset relative to the slot's base -- it may be negative or past the end) or fixed ones.  A
        pointer WITHOUT a tensor (a raw integer) must not lie in per-call memory: the plan refuses it rather than guess."""
        name = fn.__name__
        fid = self._fn_ids.get(name)
        if fid is None:
            fid = self.lib.atmvfi_plan_fn_id(name.encode())
            if fid < 0:
                raise PlanUnsupported(f"{name} is not a launch entry point")
            self._fn_ids[name] = fid
        types = SIGNATURES[name][1]
        args = args[:-1]                   # the stream is the plan's
        if len(args) > PLAN_MAX_ARGS:
            raise PlanUnsupported(f"{name}: {len(args)} arguments")
        vals = []
        k = len(self.ops_list)
        for j, (v, ty) in enumerate(zip(args, types)):
            if isinstance(v, float):
                vals.append(("f", v))
                continue
            if hasattr(v, "_obj"):         # ctypes.byref(parameter block): referenced by address, kept alive with the plan
                blk = v._obj
                for t in getattr(blk, "_srcs", ()):                 # the tensors its pointer fields were taken from (by tensor first)
                    if isinstance(t, torch.Tensor) and self._slot_of_tensor(t) is not None:
                        raise PlanUnsupported(f"{name}: a parameter-block operand lives in per-call memory")
                for fname, ftype in blk._fields_:
                    if ftype is c_f and getattr(blk, fname) and self._slot_of(getattr(blk, fname)) is not None:
                        raise PlanUnsupported(f"{name}: parameter block field {fname} points into per-call memory")
                self.keep.append(blk)
                vals.append(("u", ctypes.addressof(blk)))
                continue
            src = getattr(v, "src", None) if isinstance(v, TPtr) else None
            if isinstance(v, ctypes.c_void_p):
                v = v.value
            v = 0 if v is None else int(v)
            if ty is c_f and v:
                if src is not None:
                    slot = self._slot_of_tensor(src)
                    if slot is not None:
                        self.patches.append((k, j, slot, v - self.slots[slot][0]))
                        self.by_tensor += 1
                elif self._slot_of(v) is not None:
                    self.by_address += 1
                    raise PlanUnsupported(f"{name}: argument {j} is a raw address inside per-call memory with no tensor to attribute it to")
            vals.append(("u" if v >= 0 else "i", v))
        self.ops_list.append((fid, vals))
        self.lanes.append(lane)

    def finish(self, result):
        """Freeze the plan; ``result`` is what the recorded forward returned (tensors must be whole output slots)."""
        by_ptr = {base: k for k, (base, _) in enumerate(self.slots) if k >= self.n_inputs}

        def walk(x):
            if isinstance(x, torch.Tensor):
                k = by_ptr.get(x.data_ptr())
                if k is None or tuple(x.shape) != self.out_meta[k - self.n_inputs][0]:
                    raise PlanUnsupported("a returned tensor is not a whole per-call output")
                return ("slot", k)
            if isinstance(x, dict):
                return {kk: walk(v) for kk, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [walk(v) for v in x]
            return x
        self.template = walk(result)
        n = len(self.ops_list)
        self.c_ops = (PlanOp * n)()
        for i, (fid, vals) in enumerate(self.ops_list):
            op = self.c_ops[i]
            op.fn, op.nargs = fid, len(vals)
            for j, (kind, v) in enumerate(vals):
                setattr(op.a[j], kind, v)
        self.c_patches = (PlanPatch * max(1, len(self.patches)))()
        for i, (k, j, slot, off) in enumerate(self.patches):
            pp = self.c_patches[i]
            pp.op, pp.arg, pp.slot, pp.offset = k, j, slot, off
        self.c_slots = (ctypes.c_uint64 * len(self.slots))()
        self.failed = c_i(-1)
        self.out_tensors = []
        self.n_lanes = max(self.lanes) + 1 if self.lanes else 1
        self.c_lanes = (ctypes.c_int32 * n)(*self.lanes) if self.n_lanes > 1 else None
        return self

    # ---- replay ----
    def run(self, inputs, device, stream, poison: bool = False, lane_streams=None, lane_events=None):
        """``poison``: fill the fresh outputs with NaN first (the record-time self-check: an element the replay does not write, or
        writes somewhere else, then differs from the recording forward's result).  A plan recorded with branches (``n_lanes`` > 1)
        needs ``lane_streams`` (stream handles of lanes 1..; lane 0 is ``stream``) and ``lane_events`` (``n_events`` event handles)."""
        outs = [torch.empty(shape, dtype=dt, device=device) for shape, dt in self.out_meta]
        if poison:
            for t in outs:
                t.fill_(float("nan"))
        sl = self.c_slots
        for k, t in enumerate(inputs):
            sl[k] = t.data_ptr()
        base = self.n_inputs
        for k, t in enumerate(outs):
            sl[base + k] = t.data_ptr()
        if LaunchPlan.debug:          # diagnostic (set LaunchPlan.debug = True): one op per call, synchronised, announced on stderr first
            import sys
            names = {v: k for k, v in self._fn_ids.items()}
            for k, j, slot, off in self.patches:
                self.c_ops[k].a[j].u = sl[slot] + off
            rc = 0
            for i in range(len(self.ops_list)):
                if self.c_ops[i].fn in (PLAN_RECORD, PLAN_WAIT):        # (one stream, synchronised after every op: the lane ordering is moot)
                    continue
                print(f"plan op {i} {names.get(self.c_ops[i].fn)} args " + " ".join(hex(self.c_ops[i].a[j].u) for j in range(self.c_ops[i].nargs)),
                      file=sys.stderr, flush=True)
                one = ctypes.cast(ctypes.byref(self.c_ops[i]), ctypes.POINTER(PlanOp))
                rc = self.lib.atmvfi_plan_run(one, 1, None, 0, None, 0, ctypes.byref(self.failed), stream)
                torch.cuda.synchronize()
                if rc:
                    break
        elif self.n_lanes > 1:
            if lane_streams is None or len(lane_streams) < self.n_lanes - 1 or lane_events is None or len(lane_events) < self.n_events:
                raise RuntimeError(f"the plan has {self.n_lanes} lanes and {self.n_events} events: the caller must supply their streams and events")
            st = (ctypes.c_void_p * self.n_lanes)(stream, *lane_streams[:self.n_lanes - 1])
            ev = (ctypes.c_void_p * max(1, self.n_events))(*lane_events[:self.n_events])
            rc = self.lib.atmvfi_plan_run_lanes(self.c_ops, len(self.ops_list), self.c_lanes, self.c_patches, len(self.patches), sl, len(self.slots),
                                                ctypes.byref(self.failed), st, self.n_lanes, ev, self.n_events)
        else:
            rc = self.lib.atmvfi_plan_run(self.c_ops, len(self.ops_list), self.c_patches, len(self.patches), sl, len(self.slots),
                                          ctypes.byref(self.failed), stream)
        if rc != 0:
            raise RuntimeError(f"plan_run failed ({rc}) at op {self.failed.value}: {self.lib.atmvfi_last_error().decode()}")

        def build(x):
            if isinstance(x, tuple) and len(x) == 2 and x[0] == "slot":
                return outs[x[1] - base]
            if isinstance(x, dict):
                return {kk: build(v) for kk, v in x.items()}
            if isinstance(x, list):
                return [build(v) for v in x]
            return x
        return build(self.template)


class HipOps:
    """The op vocabulary of the hot path, each a single HIP kernel launch."""

    def __init__(self, device: torch.device, checked: bool = False, lib_path: Optional[str] = None):
        # checked: the build that counts operands beyond the f16x3 engines' fp16 range (libatmvfi_hip_checked.so, include/atmvfi.h
        # atmvfi_range_word_set).  One device word per HipOps; attached at the top of every forward (begin_forward): the library has
        # one attachment per device, so one checked model runs at a time on a device.
        self.checked = bool(checked)
        # lib_path: another build of the same ABI (tests and tools: the diagnostic libraries under tools/lib/)
        self.lib = load_library(lib_path) if lib_path else load_library(CHECKED_LIB_PATH) if self.checked else load_library()
        self.device = device
        self.range_word: Optional[torch.Tensor] = None
        if self.checked:
            if self.lib.atmvfi_range_checked() != 1:
                raise HipLibraryMissing(f"{CHECKED_LIB_PATH} is not a checked build (make -C atm-vfi_amd/csrc checked)")
            self.range_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.profile: Optional[List] = None      # when a list: (name, meta, start_evt, end_evt) per launch
        # "f16x3": 3x3/s1 convs run split-precision on the 16-bit MFMA (hi*hi + hi*lo + lo*hi, fp32 accumulate);
        # "f32": everything on the exact-fp32 MFMA.
        self.precision = "f16x3"
        self.attention_f16x3 = True        # window attention on the f16x3 MFMA path too (False: the exact-fp32 kernel)
        # per-call kernel-instance overrides (parity tests, sweeps): (schedule, wn) of the fp32-input 3x3 kernel, tile width of the
        # fp32-input f16x3 GEMM; None / 0 = the library's cost model
        self.conv3_instance = None
        self.gemm_tile_wn = 0
        self.warp_tiles = True             # planar warps with LDS-staged source tiles (False: the direct gathers; bit-identical)
        self.recording: Optional[LaunchPlan] = None     # when set: every launch is also appended to this plan
        # Lanes: independent branches of a forward on side streams (HipOps.branch / join; include/atmvfi.h "LANES").  lane 0 = the
        # caller's current stream; lanes 1.. = side streams of this object, created on first use.
        self.lane = 0
        self._lane0: Optional[torch.cuda.Stream] = None      # lane 0's stream while a branch body runs (see branch())
        self._side_streams: List[torch.cuda.Stream] = []
        self._lane_events: List[torch.cuda.Event] = []
        self._next_event = 0
        # split-K scratch for the plane-input GEMM: a callable floats -> fp32 tensor (Network hands out workspace memory); None: never split
        self.gemm_workspace = None

    # ------------------------------------------------------------------ launch plans
    def begin_plan(self, inputs) -> LaunchPlan:
        if self.profile is not None:
            raise PlanUnsupported("per-launch profiling is on")
        self.recording = LaunchPlan(self.lib, inputs)
        return self.recording

    def end_plan(self, result) -> LaunchPlan:
        plan, self.recording = self.recording, None
        return plan.finish(result)

    def abort_plan(self):
        self.recording = None

    # ------------------------------------------------------------------ utils
    def _lane_stream(self, lane: int) -> "torch.cuda.Stream":
        if lane == 0:
            # (inside a branch torch's current stream IS the side stream: lane 0 stays the stream the branch was entered from)
            return self._lane0 if self._lane0 is not None else torch.cuda.current_stream(self.device)
        while len(self._side_streams) < lane:
            self._side_streams.append(torch.cuda.Stream(device=self.device))
        return self._side_streams[lane - 1]

    def _stream(self):
        return ctypes.c_void_p(self._lane_stream(self.lane).cuda_stream)

    def _sync_event(self) -> int:
        """The next synchronisation event of this forward (events are reused from call to call; a plan records their indices)."""
        k = self._next_event
        self._next_event += 1
        while len(self._lane_events) <= k:
            ev = torch.cuda.Event()
            ev.record(self._lane_stream(0))          # (materialises the handle)
            self._lane_events.append(ev)
        return k

    def begin_forward(self):
        """Called by Network at the top of every forward: lane 0, event numbering from 0."""
        self.lane = 0
        self._lane0 = None
        self._next_event = 0
        if self.checked:
            self._check(self.lib.atmvfi_range_word_set(self.range_word.data_ptr(), self._stream()), "range_word_set")

    def _order(self, first: int, then: int):
        """Everything issued so far on lane ``first`` happens before whatever lane ``then`` is given from now on."""
        k = self._sync_event()
        ev = self._lane_events[k]
        ev.record(self._lane_stream(first))
        self._lane_stream(then).wait_event(ev)
        if self.recording is not None:
            self.recording.add_sync(PLAN_RECORD, k, first)
            self.recording.add_sync(PLAN_WAIT, k, then)

    def branch(self, lane: int):
        """``with ops.branch(k): ...`` -- the launches of the block go to lane ``k`` (a side stream), ordered after everything issued so
        far on the current lane; afterwards the current lane continues WITHOUT waiting for them.  ``ops.join(k)`` orders the current
        lane after lane ``k``.  The caller guarantees that, until the join, the two sides touch disjoint memory (workspace scratch
        included: Network hands out per-lane split-K buffers)."""
        ops = self

        class _Branch:
            def __enter__(self_inner):
                self_inner.prev = ops.lane
                if lane != ops.lane:
                    ops._order(ops.lane, lane)
                ops.lane = lane

            def __exit__(self_inner, *exc):
                ops.lane = self_inner.prev
                return False
        return _Branch()

    def join(self, lane: int):
        if lane != self.lane:
            self._order(lane, self.lane)

    def lane_handles(self):
        """(stream handles of lanes 1.., event handles) for ``LaunchPlan.run``."""
        return ([ctypes.c_void_p(s.cuda_stream) for s in self._side_streams], [ctypes.c_void_p(e.cuda_event) for e in self._lane_events])

    def _check(self, rc: int, name: str):
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {self.lib.atmvfi_last_error().decode()}")

    def _run(self, name: str, meta: dict, fn, *args):
        if self.recording is not None:
            self.recording.add_op(fn, args, self.lane)
        if self.profile is None:
            self._check(fn(*args), name)
            return
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        st = self._lane_stream(self.lane)
        s.record(st)
        self._check(fn(*args), name)
        e.record(st)
        self.profile.append((name, meta, s, e))

    def empty(self, *shape) -> torch.Tensor:
        t = torch.empty(*shape, dtype=torch.float32, device=self.device)
        if self.recording is not None:
            self.recording.add_output(t)
        return t

    def to_device_int(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(device=self.device, dtype=torch.int32).contiguous()

    # ---------------------------------------------------------------- weights
    def pack_weight(self, mode: int, w: torch.Tensor) -> PackedWeight:
        _chk(w, "pack_weight")
        w = w.detach().contiguous()
        if mode == GEMM_DECONV:
            cin, cout, kh, kw = w.shape
        elif mode == GEMM_LINEAR:
            cout, cin = w.shape[0], w.shape[1]
            kh = kw = 1
        else:
            cout, cin, kh, kw = w.shape
        n = self.lib.atmvfi_packed_weight_floats(mode, cout, cin, kh, kw)
        dst = self.empty(n)
        self._check(self.lib.atmvfi_pack_weight(mode, _ptr(w), _ptr(dst), cout, cin, kh, kw, self._stream()), "pack_weight")
        pw = PackedWeight(mode, cout, cin, kh, kw, w, dst)
        if cin >= 16:       # split-precision planes for the f16x3 engines (tiny-K layers stay on the fp32 MFMA)
            nh = self.lib.atmvfi_split_weight_halves(mode, cout, cin, kh, kw)
            pw.hi = torch.empty(nh, dtype=torch.float16, device=self.device)
            pw.lo = torch.empty(nh, dtype=torch.float16, device=self.device)
            self._check(self.lib.atmvfi_pack_weight_split(mode, _ptr(w), _ptr(pw.hi), _ptr(pw.lo), cout, cin, kh, kw, self._stream()),
                        "pack_weight_split")
            if mode == GEMM_CONV and kh == 3 and kw == 3:
                n3 = self.lib.atmvfi_conv3x3_weight_halves(cout, cin)
                pw.hi3 = torch.empty(n3, dtype=torch.float16, device=self.device)
                pw.lo3 = torch.empty(n3, dtype=torch.float16, device=self.device)
                self._check(self.lib.atmvfi_pack_weight_conv3x3(_ptr(w), _ptr(pw.hi3), _ptr(pw.lo3), cout, cin, self._stream()),
                            "pack_weight_conv3x3")
        return pw

    def pack_dw_weight(self, w: torch.Tensor) -> torch.Tensor:
        _chk(w, "pack_dw_weight")
        w = w.detach().contiguous()
        c = w.shape[0]
        dst = self.empty(9 * c)
        self._check(self.lib.atmvfi_pack_dw_weight(_ptr(w), _ptr(dst), c, self._stream()), "pack_dw_weight")
        return dst

    def _prec(self, w: PackedWeight):
        """(precision, hi, lo) fields of GemmParams for this weight under the current precision mode."""
        if self.precision == "f16x3" and w.hi is not None:
            return 1, w.hi.data_ptr(), w.lo.data_ptr()
        return 0, None, None

    def pad_channels(self, v: torch.Tensor, mult: int = 32) -> torch.Tensor:
        """Per-channel vector padded with zeros to a multiple of ``mult`` (in_prelu contract)."""
        c = v.shape[0]
        out = torch.zeros((c + mult - 1) // mult * mult, dtype=torch.float32, device=self.device)
        out[:c] = v.detach()
        return out

    # ------------------------------------------------------------------ GEMMs
    @staticmethod
    def _gemm_sink(p: "GemmParams", sink: Optional[Planes], rows: int, cout: int, c0: int, gc: int, groups: int, what: str):
        """Fill the plane-sink fields of ``p``: ``rows`` output rows (per group) x ``cout`` channels at channel offset ``c0``
        (+ ``gc`` per row group) of ``sink``."""
        if sink is None:
            return
        if p.precision != 1:
            raise ValueError(f"{what}: a plane sink needs the f16x3 engine")
        top = c0 + (groups - 1) * gc + cout
        if sink.rows != rows or c0 % 4 or gc % 4 or top > sink.chunks * 32:
            raise ValueError(f"{what}: plane sink [{sink.rows},{sink.c}] cannot take {rows} rows x {cout} channels at offset {c0} (+{gc} per group)")
        p.out_hi, p.out_lo = sink.t[0].data_ptr(), sink.t[1].data_ptr()
        p.out_plane_rows, p.out_plane_c0, p.out_plane_gc = sink.ld_rows, c0, gc

    def conv(self, x, w: PackedWeight, out, stride=1, pad=1, dil=1, bias=None, prelu=None, in_prelu=None,
             planes: Optional[Planes] = None, planes_prelu=None, planes_c0: int = 0, out_shape=None):
        """``planes``: also write the result as split planes, at channel offset ``planes_c0``; through ``planes_prelu`` (3x3 / stride 1
        f16x3 kernel only).  ``out`` may be None (planes only; generic GEMM path) when ``out_shape`` = (N, Ho, Wo, Cout) is given."""
        ld, n, h, wd, cin = nhwc_view(x, "conv.in")
        if out is None:
            if planes is None or out_shape is None:
                raise ValueError("conv: no output")
            old, (on, oh, ow, cout) = 0, out_shape
        else:
            old, on, oh, ow, cout = nhwc_view(out, "conv.out")
        if cin != w.cin or cout != w.cout or on != n or w.mode != GEMM_CONV:
            raise ValueError(f"conv: shape mismatch in {tuple(x.shape)} w ({w.cout},{w.cin},{w.kh},{w.kw}) out {(on, oh, ow, cout)}")
        meta = {"flops": 2.0 * n * oh * ow * cout * cin * w.kh * w.kw,
                "bytes": 4.0 * (n * h * wd * cin + n * oh * ow * cout + cout * cin * w.kh * w.kw),
                "shape": f"M{n * oh * ow} N{cout} K{cin * w.kh * w.kw}"}
        if (self.precision == "f16x3" and w.hi3 is not None and w.kh == 3 and stride == 1 and pad == 1 and dil == 1
                and in_prelu is None and out is not None and planes_c0 == 0):
            if planes is not None and (planes.rows < n * h * wd or planes.c != cout):
                raise ValueError(f"conv: plane sink [{planes.rows},{planes.c}] does not match {n * h * wd} pixels x {cout} channels")
            if planes is not None and planes_prelu is not None and planes_prelu.numel() < (cout + 31) // 32 * 32:
                raise ValueError("conv: planes_prelu must be padded to a multiple of 32 channels")
            self._run("conv3x3_f16x3", meta, self.lib.atmvfi_conv3x3_f16x3, _ptr(x), ld, n, h, wd, cin, _ptr(w.hi3), _ptr(w.lo3),
                      cout, _ptr(out), old, _ptr(bias), _ptr(prelu),
                      _ptr(planes.t[0]) if planes is not None else None, _ptr(planes.t[1]) if planes is not None else None,
                      planes.ld_rows if planes is not None else 0, _ptr(planes_prelu) if planes is not None else None,
                      *(self.conv3_instance or (-1, 0)), self._stream())
            return
        if planes is not None and planes_prelu is not None:
            raise ValueError("conv: planes_prelu needs the 3x3 / stride-1 f16x3 kernel")
        p = GemmParams(mode=GEMM_CONV, in_=x.data_ptr(), in_ld=ld, N=n, H=h, W=wd, Cin=cin, in_gstride=0, in_rpg=0,
                       weight=w.packed.data_ptr(), Cout=cout, kh=w.kh, kw=w.kw, stride=stride, pad=pad, dil=dil,
                       Ho=oh, Wo=ow, M=n * oh * ow, out=None if out is None else out.data_ptr(), out_ld=old, out_gstride=0, out_rpg=0,
                       out_row_map=None, bias=_ptr(bias), prelu=_ptr(prelu), in_prelu=_ptr(in_prelu), residual=None, res_ld=0)
        p.precision, p.weight_hi, p.weight_lo = self._prec(w)
        p.tile_wn = self.gemm_tile_wn
        self._gemm_sink(p, planes, n * oh * ow, cout, planes_c0, 0, 1, "conv")
        p._srcs = (x, out, bias, prelu, in_prelu)
        self._run("conv2d_f16x3" if p.precision else "conv2d", meta, self.lib.atmvfi_conv2d, ctypes.byref(p), self._stream())

    def _gemm_splitk(self, p: "GemmParams", m: int, ngemm: int, ksteps: int):
        """Give the launch split-K scratch when the launcher would use it (under-filled grid, long K); returns the tensor (kept alive by
        the caller's workspace) or None."""
        if self.gemm_workspace is None or p.precision != 1 or not p.in_hi or p.tile_wn != 0:
            return None
        need = int(self.lib.atmvfi_gemm_workspace_floats(m, ngemm, ksteps))
        if not need:
            return None
        ws = self.gemm_workspace(need)
        if ws is None:
            return None
        p.workspace, p.workspace_floats = ws.data_ptr(), ws.numel()
        return ws

    def pack_stem(self, w1, b1, p1, w2, b2, p2, w3, b3, p3) -> "StemWeights":
        """The operand layouts of ``atmvfi_stem_fused`` (include/atmvfi.h) from the nine parameters of feat_extracts.0.0 / 0.1 / 1.0
        (weight, bias, PReLU slope each): fp16 hi / lo' planes [k-step][cout][32] with k = (ky * 3 + kx) * Cin + c (layer 1: one
        k-step of 27 values).  Host-side torch ops, once per parameter version (like the q / kv stacking in ``_prepare``)."""
        c0, c1 = w2.shape[0], w3.shape[0]
        if tuple(w1.shape) != (c0, 3, 3, 3) or tuple(w2.shape) != (c0, c0, 3, 3) or tuple(w3.shape) != (c1, c0, 3, 3):
            raise ValueError("pack_stem: expected 3 -> C0 -> C0 -> C1 3x3 weights")
        ns = (9 * c0 + 31) // 32

        def planes_of(w, rows):
            k = w.detach().float().permute(0, 2, 3, 1).reshape(w.shape[0], 9 * c0)          # [cout][(ky, kx, c)]
            full = torch.zeros(rows, ns * 32, dtype=torch.float32, device=self.device)
            full[:w.shape[0], :9 * c0] = k
            full = full.reshape(rows, ns, 32).permute(1, 0, 2).contiguous()                  # [k-step][cout][32]
            hi = full.half()
            lo = ((full - hi.float()) * 1024.0).half()
            return hi, lo

        def padded(v, rows, fill):
            out = torch.full((rows,), fill, dtype=torch.float32, device=self.device)
            out[:v.shape[0]] = v.detach().float()
            return out
        r2 = (c0 + 15) // 16 * 16
        w2h, w2l = planes_of(w2, r2)
        w3h, w3l = planes_of(w3, c1)
        k1 = torch.zeros(r2, 32, dtype=torch.float32, device=self.device)                   # [cout][(ky, kx, ci)], 27 values
        k1[:c0, :27] = w1.detach().float().permute(0, 2, 3, 1).reshape(c0, 27)
        w1h = k1.half()
        w1l = ((k1 - w1h.float()) * 1024.0).half()
        return StemWeights(c0, c1, w1h, w1l, padded(b1, r2, 0.0), padded(p1, r2, 1.0), w2h, w2l, padded(b2, r2, 0.0), padded(p2, r2, 1.0),
                           w3h, w3l, b3.detach().float().contiguous(), p3.detach().float().contiguous())

    def stem_fused(self, x, w: "StemWeights", out: Planes):
        """feat_extracts.0.0 -> 0.1 -> 1.0 (network_base.py:99-110) in one launch: NHWC4 frames [F,H,W,4] -> split planes of the
        half-resolution C1-channel map (rows = pixels (f, y, x))."""
        _chk(x, "stem_fused.x")
        if x.dim() != 4 or x.shape[3] != 4 or not x.is_contiguous():
            raise ValueError(f"stem_fused: expected contiguous NHWC4 frames [F,H,W,4], got {tuple(x.shape)}")
        f, h, wd, _ = x.shape
        if h % 2 or wd % 2 or out.rows != f * (h // 2) * (wd // 2) or out.c != w.c1:
            raise ValueError(f"stem_fused: planes [{out.rows},{out.c}] do not hold {f}x{h // 2}x{wd // 2} pixels x {w.c1} channels")
        if self.precision != "f16x3":
            raise ValueError("stem_fused: needs the f16x3 engine")
        flops = 2.0 * f * h * wd * (27 * w.c0 + 9 * w.c0 * w.c0) + 2.0 * f * (h // 2) * (wd // 2) * 9 * w.c0 * w.c1
        meta = {"flops": flops, "bytes": 16.0 * f * h * wd + 4.0 * out.rows * out.chunks * 32, "shape": f"{f}x{h}x{wd} 3>{w.c0}>{w.c0}>{w.c1}s2"}
        self._run("stem_fused", meta, self.lib.atmvfi_stem_fused, _ptr(x), f, h, wd, w.c0, w.c1, _ptr(w.w1h), _ptr(w.w1l), _ptr(w.b1), _ptr(w.p1),
                  _ptr(w.w2h), _ptr(w.w2l), _ptr(w.b2), _ptr(w.p2), _ptr(w.w3h), _ptr(w.w3l), _ptr(w.b3), _ptr(w.p3),
                  _ptr(out.t[0]), _ptr(out.t[1]), out.ld_rows, self._stream())

    def conv3x3_planes(self, x: Planes, n: int, h: int, wd: int, w: PackedWeight, out=None, bias=None, prelu=None,
                       planes: Optional[Planes] = None, planes_c0: int = 0, planes_prelu=None, in_chunk0: int = 0, cin: Optional[int] = None,
                       wn: int = 0, out_cmin: int = 0, planes2: Optional[Planes] = None, planes2_c0: int = 0,
                       workspace: Optional[torch.Tensor] = None):
        """3x3 / stride 1 / pad 1 conv (+bias, PReLU) on split-plane input ``x`` (rows = pixels of an [n,h,wd] map; channels
        ``32*in_chunk0 .. +cin``).  ``out``: fp32 NHWC view or None; ``planes``: plane sink written at channel offset ``planes_c0``
        (its own ``planes_prelu`` applied to that copy only); ``planes2``: a second, raw plane sink.  ``out_cmin``: only channels >= it are stored in ``out``.  Needs the spare
        zero row of ``Planes.alloc``.  ``workspace``: fp32 scratch of at least ``conv3x3_workspace_floats(...)`` elements: lets the launcher
        split K over idle CUs on under-filled grids (split-K with a fixed-order reduce; small frames)."""
        cin = (x.c - 32 * in_chunk0) if cin is None else cin
        if self.precision != "f16x3" or w.hi3 is None:
            raise ValueError("conv3x3_planes: needs the f16x3 engine and the conv3x3 weight planes")
        if w.mode != GEMM_CONV or w.kh != 3 or w.kw != 3 or w.cin != cin:
            raise ValueError(f"conv3x3_planes: weight ({w.cout},{w.cin},{w.kh},{w.kw}) does not match Cin {cin}")
        if x.rows != n * h * wd or x.ld_rows <= x.rows or 32 * in_chunk0 + cin > x.chunks * 32:
            raise ValueError(f"conv3x3_planes: input planes [{x.rows} (+{x.ld_rows - x.rows} spare), {x.c}] do not fit {n}x{h}x{wd} pixels x {cin} channels "
                             "(one spare zero row is required)")
        cout = w.cout
        old = 0
        out_ptr = _ptr(out)
        if out is not None:
            old, on, oh, ow, oc = nhwc_view(out, "conv3x3_planes.out")
            if (on, oh, ow) == (n, h, wd) and oc == cout - out_cmin and out_cmin > 0:
                # compact view of just the stored channels [out_cmin, cout): the kernel never touches columns below out_cmin, so the
                # base may point that many floats in front of the buffer (out_cmin % 4 == 0 keeps the 16-byte alignment)
                out_ptr = _ptr(out, -4 * out_cmin)       # a pointer IN FRONT of its buffer: attributed to `out` by the TPtr, never by its address
            elif (on, oh, ow, oc) != (n, h, wd, cout):
                raise ValueError(f"conv3x3_planes: out {tuple(out.shape)} != [{n},{h},{wd},{cout}] (or {cout - out_cmin} channels: compact)")
        elif planes is None:
            raise ValueError("conv3x3_planes: no output")
        if planes is not None:
            if planes.rows != n * h * wd or planes_c0 % 8 or planes_c0 + cout > planes.chunks * 32:
                raise ValueError(f"conv3x3_planes: plane sink [{planes.rows},{planes.c}] cannot take {cout} channels at offset {planes_c0}")
            if planes_prelu is not None and planes_prelu.numel() < (cout + 31) // 32 * 32:
                raise ValueError("conv3x3_planes: planes_prelu must be padded to a multiple of 32 channels")
        if planes2 is not None and (planes is None or planes2.rows != n * h * wd or planes2_c0 % 8 or planes2_c0 + cout > planes2.chunks * 32):
            raise ValueError(f"conv3x3_planes: second plane sink [{planes2.rows},{planes2.c}] needs the first one and room for {cout} channels at {planes2_c0}")
        meta = {"flops": 2.0 * n * h * wd * cout * cin * 9, "bytes": 4.0 * (n * h * wd * (cin + cout) + cout * cin * 9),
                "shape": f"M{n * h * wd} N{cout} K{cin * 9}"}
        coff = in_chunk0 * x.ld_rows * 32 * 2       # bytes
        if workspace is not None and (workspace.dtype != torch.float32 or not workspace.is_cuda or not workspace.is_contiguous()):
            raise ValueError("conv3x3_planes: the split-K workspace must be a contiguous CUDA fp32 tensor")
        self._run("conv3x3_planes", meta, self.lib.atmvfi_conv3x3_planes3, _ptr(x.t[0], coff), _ptr(x.t[1], coff), x.ld_rows,
                  n, h, wd, cin, _ptr(w.hi3), _ptr(w.lo3), cout, out_ptr, old, _ptr(bias), _ptr(prelu),
                  _ptr(planes.t[0]) if planes is not None else None, _ptr(planes.t[1]) if planes is not None else None,
                  planes.ld_rows if planes is not None else 0, planes_c0, _ptr(planes_prelu) if planes is not None else None,
                  _ptr(planes2.t[0]) if planes2 is not None else None, _ptr(planes2.t[1]) if planes2 is not None else None,
                  planes2.ld_rows if planes2 is not None else 0, planes2_c0, out_cmin, wn, _ptr(workspace),
                  workspace.numel() if workspace is not None else 0, self._stream())

    def pack_readout(self, w: torch.Tensor) -> torch.Tensor:
        """Operand of ``atmvfi_conv3x3_planes_readout``: refine_head.1's weight [3, C, 3, 3] (C = 32 or 64) as the 27 x C matrix
        W2[(tap, o)][c], hi / lo' split, in the producing kernel's register order: fp16 [plane][row tile 2][k-step C / 32][lane][8] with
        lane = 16 g + r -> row 16 t + r, and the lane's eight k values = its four channels of n-tile 2 kk and of n-tile 2 kk + 1
        (channels 32 kk + cb(g) + e and 32 kk + 16 + cb(g) + e, cb = {0, 8, 4, 12}[g]).  Host-side torch ops, once per parameter version."""
        o3, c, kh, kw = w.shape
        if (o3, kh, kw) != (3, 3, 3) or c not in (32, 64):
            raise ValueError(f"pack_readout: expected a [3, 32 | 64, 3, 3] weight, got {tuple(w.shape)}")
        w2 = torch.zeros(32, c, dtype=torch.float32, device=self.device)
        w2[:27] = w.detach().float().permute(2, 3, 0, 1).reshape(27, c)            # row = (ky * 3 + kx) * 3 + o
        kk_n = c // 32
        lane = torch.arange(64, device=self.device)
        r, g = lane & 15, lane >> 4
        cb = 8 * (g & 1) + 4 * (g >> 1)
        e = torch.arange(8, device=self.device)
        ch = cb[:, None] + torch.where(e < 4, e, 16 + e - 4)[None, :]            # [lane, 8] channel inside a 32-channel k-step
        full = torch.empty(2, kk_n, 64, 8, dtype=torch.float32, device=self.device)
        for t in range(2):
            for kk in range(kk_n):
                full[t, kk] = w2[(16 * t + r)[:, None].expand(64, 8), 32 * kk + ch]
        hi = full.half()
        lo = ((full - hi.float()) * 1024.0).half()
        return torch.stack([hi, lo], 0).contiguous()

    def conv3x3_planes_readout(self, x: Planes, n: int, h: int, wd: int, w: PackedWeight, bias, prelu, w2: torch.Tensor, contrib: torch.Tensor):
        """refine_head.0 (3x3 conv + PReLU on split-plane input, Cout 32 or 64) with refine_head.1's 27 tap contributions per pixel
        computed in its epilogue: contrib [27, n*h*wd] planar fp32 (include/atmvfi.h)."""
        if self.precision != "f16x3" or w.hi3 is None or w.kh != 3 or w.cin > x.chunks * 32 or x.rows != n * h * wd or x.ld_rows <= x.rows:
            raise ValueError("conv3x3_planes_readout: needs the f16x3 engine, conv3x3 weight planes and input planes with a spare zero row")
        if w.cout not in (32, 64) or tuple(w2.shape) != (2, 2, w.cout // 32, 64, 8) or w2.dtype != torch.float16:
            raise ValueError("conv3x3_planes_readout: Cout must be 32 or 64 and w2 the pack_readout() operand of that width")
        if contrib.dtype != torch.float32 or contrib.dim() != 2 or contrib.shape[0] != 27 or contrib.shape[1] < n * h * wd or not contrib.is_contiguous():
            raise ValueError("conv3x3_planes_readout: contrib must be contiguous fp32 [27, >= N*H*W]")
        meta = {"flops": 2.0 * n * h * wd * w.cout * (w.cin * 9 + 27), "bytes": 4.0 * n * h * wd * (w.cin + 27), "shape": f"M{n * h * wd} N{w.cout} K{w.cin * 9} +27"}
        self._run("conv3x3_planes", meta, self.lib.atmvfi_conv3x3_planes_readout, _ptr(x.t[0]), _ptr(x.t[1]), x.ld_rows, n, h, wd, w.cin,
                  _ptr(w.hi3), _ptr(w.lo3), w.cout, _ptr(bias), _ptr(prelu), _ptr(w2), _ptr(contrib), contrib.shape[1], self._stream())

    def refine_tail(self, contrib, bias, slope, it, it_sum, it_clamped):
        """out = clamp(it + 2 * sigmoid(PReLU(bias + sum of the nine shifted tap contributions)) - 1): refine_head.1 + the residual."""
        _planar(it, 3, "refine_tail.it"); _planar(it_sum, 3, "refine_tail.sum"); _planar(it_clamped, 3, "refine_tail.clamped")
        b, _, h, w = it.shape
        if contrib.dim() != 2 or contrib.shape[0] != 27 or contrib.shape[1] < b * h * w or not contrib.is_contiguous():
            raise ValueError("refine_tail: contrib must be contiguous fp32 [27, >= B*H*W]")
        self._run("refine_tail", {"bytes": 4.0 * b * h * w * (27 + 9)}, self.lib.atmvfi_refine_tail, _ptr(contrib), contrib.shape[1], _ptr(bias),
                  _ptr(slope), _ptr(it), _ptr(it_sum), _ptr(it_clamped), b, h, w, self._stream())

    def conv3x3_workspace_floats(self, n: int, h: int, wd: int, cin: int, cout: int) -> int:
        """fp32 elements of split-K scratch ``conv3x3_planes`` wants for this shape on this device (0: it would not split)."""
        return int(self.lib.atmvfi_conv3x3_planes_workspace_floats(n, h, wd, cin, cout))

    def head1x1_planes(self, x: Planes, n: int, h: int, wd: int, w: PackedWeight, out, bias=None):
        """nn.Conv2d(Cin, Cout <= 8, 1) on split-plane input (the 5-channel read-out of a motion MLP): one lane per pixel row, fp32 FMAs."""
        old, on, oh, ow, oc = nhwc_view(out, "head1x1_planes.out")
        if w.mode != GEMM_CONV or w.kh != 1 or w.cout > 8 or (on, oh, ow, oc) != (n, h, wd, w.cout) or x.rows != n * h * wd or w.cin > x.chunks * 32:
            raise ValueError(f"head1x1_planes: weight ({w.cout},{w.cin},{w.kh},{w.kw}) / out {tuple(out.shape)} / planes [{x.rows},{x.c}] mismatch")
        wt = w.orig.detach()
        if not wt.is_contiguous():
            wt = wt.contiguous()
        meta = {"flops": 2.0 * n * h * wd * w.cout * w.cin, "bytes": 4.0 * n * h * wd * (w.cin + w.cout), "shape": f"M{n * h * wd} N{w.cout} K{w.cin}"}
        self._run("head1x1_planes", meta, self.lib.atmvfi_head1x1_planes, _ptr(x.t[0]), _ptr(x.t[1]), x.ld_rows, n * h * wd, w.cin,
                  _ptr(wt), _ptr(bias), w.cout, _ptr(out), old, self._stream())

    def conv_planes(self, x: Planes, n: int, h: int, wd: int, w: PackedWeight, out=None, stride=1, pad=1, dil=1, bias=None, prelu=None,
                    sink: Optional[Planes] = None, sink_c0: int = 0, in_chunk0: int = 0, x2: Optional[Planes] = None, x2_chunk0: int = 0,
                    split_chunks: int = 0):
        """Conv2d (k 1 or 3, any stride / dilation; + bias, PReLU) on split-plane input through the LDS-DMA GEMM's CONV mode: rows of
        ``x`` = pixels of an [n,h,wd] map, channels from 32-channel chunk ``in_chunk0`` on.  With ``x2`` the first ``split_chunks``
        chunks come from ``x`` and the rest from ``x2`` (chunk ``x2_chunk0`` on): a channel concat that never materialises.
        ``out``: fp32 NHWC view or None; ``sink``: plane sink at channel ``sink_c0``.  Needs the spare zero row of ``Planes.alloc``."""
        if self.precision != "f16x3" or w.hi is None or w.mode != GEMM_CONV:
            raise ValueError("conv_planes: needs the f16x3 engine and split conv weights")
        cin, cout = w.cin, w.cout
        oh = (h + 2 * pad - dil * (w.kh - 1) - 1) // stride + 1
        ow = (wd + 2 * pad - dil * (w.kw - 1) - 1) // stride + 1
        c1 = 32 * split_chunks if x2 is not None else cin
        for t, c0, cc, what in ((x, in_chunk0, c1, "x"),) + (((x2, x2_chunk0, cin - c1, "x2"),) if x2 is not None else ()):
            if t.rows != n * h * wd or t.ld_rows <= t.rows or 32 * c0 + cc > t.chunks * 32:
                raise ValueError(f"conv_planes: {what} [{t.rows} (+{t.ld_rows - t.rows} spare), {t.c}] does not hold {n}x{h}x{wd} pixels x "
                                 f"{cc} channels from chunk {c0} (one spare zero row is required)")
        if x2 is not None and not 0 < 32 * split_chunks < cin:
            raise ValueError("conv_planes: split_chunks must cut the input channels at a 32-channel boundary inside (0, Cin)")
        old = 0
        if out is not None:
            old, on, ooh, oow, oc = nhwc_view(out, "conv_planes.out")
            if (on, ooh, oow, oc) != (n, oh, ow, cout):
                raise ValueError(f"conv_planes: out {tuple(out.shape)} != [{n},{oh},{ow},{cout}]")
        elif sink is None:
            raise ValueError("conv_planes: no output")
        p = GemmParams(mode=GEMM_CONV, in_=None, in_ld=x.ld_rows, N=n, H=h, W=wd, Cin=cin, in_gstride=0, in_rpg=0,
                       weight=w.packed.data_ptr(), Cout=cout, kh=w.kh, kw=w.kw, stride=stride, pad=pad, dil=dil,
                       Ho=oh, Wo=ow, M=n * oh * ow, out=None if out is None else out.data_ptr(), out_ld=old, out_gstride=0, out_rpg=0,
                       out_row_map=None, bias=_ptr(bias), prelu=_ptr(prelu), in_prelu=None, residual=None, res_ld=0)
        p.precision, p.weight_hi, p.weight_lo = 1, w.hi.data_ptr(), w.lo.data_ptr()
        off = in_chunk0 * x.ld_rows * 64
        p.in_hi, p.in_lo = x.t[0].data_ptr() + off, x.t[1].data_ptr() + off
        if x2 is not None:
            off2 = x2_chunk0 * x2.ld_rows * 64
            p.in_hi2, p.in_lo2, p.in_ld2, p.in_split_chunks = x2.t[0].data_ptr() + off2, x2.t[1].data_ptr() + off2, x2.ld_rows, split_chunks
        p.tile_wn = self.gemm_tile_wn if self.gemm_tile_wn < 0 else 0        # -1: the reference schedule (gemm_split.hip)
        self._gemm_sink(p, sink, n * oh * ow, cout, sink_c0, 0, 1, "conv_planes")
        meta = {"flops": 2.0 * n * oh * ow * cout * cin * w.kh * w.kw, "bytes": 4.0 * (n * h * wd * cin + n * oh * ow * cout + cout * cin * w.kh * w.kw),
                "shape": f"M{n * oh * ow} N{cout} K{cin * w.kh * w.kw}"}
        ws = self._gemm_splitk(p, n * oh * ow, cout, w.kh * w.kw * ((cin + 31) // 32))
        p._srcs = (out, bias, prelu, ws)
        self._run("conv2d_split", meta, self.lib.atmvfi_conv2d, ctypes.byref(p), self._stream())

    def deconv(self, x, w: PackedWeight, out, bias=None, prelu=None, in_prelu=None, planes: Optional[Planes] = None,
               sink: Optional[Planes] = None, sink_c0: int = 0, in_shape=None):
        """``planes``: the input rows [N*H*W, Cin] again in split-plane form (already through in_prelu): the LDS-DMA GEMM
        then replaces the fp32-input engine (``x`` may then be None with ``in_shape`` = (N, H, W, Cin)).  ``sink``: also write the
        result as split planes at channel offset ``sink_c0`` (``out`` may then be None)."""
        if x is None:
            if planes is None or in_shape is None:
                raise ValueError("deconv: no input")
            ld, (n, h, wd, cin) = 0, in_shape
        else:
            ld, n, h, wd, cin = nhwc_view(x, "deconv.in")
        cout = w.cout
        if out is None:
            if sink is None:
                raise ValueError("deconv: no output")
            old, on, oh, ow = 0, n, 2 * h, 2 * wd
        else:
            old, on, oh, ow, oc = nhwc_view(out, "deconv.out")
            if oc != cout:
                raise ValueError(f"deconv: out has {oc} channels, weight {cout}")
        if cin != w.cin or on != n or w.mode != GEMM_DECONV or oh != 2 * h or ow != 2 * wd:
            raise ValueError(f"deconv: shape mismatch in {(n, h, wd, cin)} w ({w.cin},{w.cout}) out {(on, oh, ow, cout)}")
        p = GemmParams(mode=GEMM_DECONV, in_=None if x is None else x.data_ptr(), in_ld=ld, N=n, H=h, W=wd, Cin=cin, in_gstride=0, in_rpg=0,
                       weight=w.packed.data_ptr(), Cout=cout, kh=2, kw=2, stride=2, pad=0, dil=1, Ho=oh, Wo=ow,
                       M=n * h * wd, out=None if out is None else out.data_ptr(), out_ld=old, out_gstride=0, out_rpg=0, out_row_map=None,
                       bias=_ptr(bias), prelu=_ptr(prelu), in_prelu=_ptr(in_prelu), residual=None, res_ld=0)
        p.precision, p.weight_hi, p.weight_lo = self._prec(w)
        p.tile_wn = self.gemm_tile_wn
        self._gemm_sink(p, sink, n * oh * ow, cout, sink_c0, 0, 1, "deconv")
        use_planes = planes is not None and p.precision == 1
        if x is None and not use_planes:
            raise ValueError("deconv: plane input needs the f16x3 engine")
        if use_planes:
            if planes.rows < n * h * wd or planes.c != cin:
                raise ValueError("deconv: planes do not match the input rows")
            p.in_, p.in_ld, p.in_prelu = None, planes.ld_rows, None
            p.in_hi, p.in_lo = planes.t[0].data_ptr(), planes.t[1].data_ptr()
        meta = {"flops": 2.0 * n * h * wd * 4 * cout * cin, "bytes": 4.0 * (n * h * wd * cin + n * oh * ow * cout + 4 * cout * cin),
                "shape": f"M{n * h * wd} N{4 * cout} K{cin}"}
        ws = self._gemm_splitk(p, n * h * wd, 4 * ((cout + 3) // 4 * 4), (cin + 31) // 32) if use_planes else None
        p._srcs = (x, out, bias, prelu, in_prelu, ws)
        self._run("deconv2x2_split" if use_planes else "deconv2x2_f16x3" if p.precision else "deconv2x2", meta, self.lib.atmvfi_deconv2x2,
                  ctypes.byref(p), self._stream())

    split_planes_ok = True        # this backend has the split-plane sinks and the LDS-DMA GEMM
    pyramid_packs = True          # image_pyramid(pack=...) does pack_frames' work in the same launch
    motion_head_sink = True       # ... and the motion head's plane sink (no split pass for the motion MLP's eight motion channels)

    def split_planes(self, x, out: Planes, prelu=None, c0: Optional[int] = None):
        """fp32 rows -> split planes, optionally through a per-channel PReLU first.  ``c0``: write the channels at this offset
        (multiple of 8) of wider planes and leave everything else alone (zero fill only up to the next multiple of 8)."""
        ld, m, c, gs, _ = rows_view(x, "split_planes.in")
        if gs != 0 or m != out.rows or (c0 is None and c != out.c) or (c0 is not None and (c0 % 8 or c0 + c > out.chunks * 32)):
            raise ValueError("split_planes: expected a plain [rows,C] view matching the planes")
        if prelu is not None and prelu.numel() < c:
            raise ValueError("split_planes: prelu needs one slope per channel")
        meta = {"bytes": 4.0 * m * c + 4.0 * m * ((c + 31) // 32 * 32 if c0 is None else (c + 7) // 8 * 8)}
        if c0 is None:
            self._run("split_planes", meta, self.lib.atmvfi_split_planes, _ptr(x), ld, m, c, _ptr(prelu), _ptr(out.t[0]),
                      _ptr(out.t[1]), out.ld_rows, self._stream())
        else:
            self._run("split_planes", meta, self.lib.atmvfi_split_planes_at, _ptr(x), ld, m, c, _ptr(prelu), _ptr(out.t[0]),
                      _ptr(out.t[1]), out.ld_rows, c0, 8, self._stream())

    def linear(self, x, w: PackedWeight, out, bias=None, residual=None, out_row_map=None, sink: Optional[Planes] = None,
               sink_c0: int = 0, sink_gc: int = 0):
        """``sink``: also write the result as split planes: output row r (after ``out_row_map``; of its row group when ``out`` is a
        grouped [G,R,C] view) at channel ``sink_c0`` (+ ``sink_gc`` per group)."""
        planes = x if isinstance(x, Planes) else None
        if planes is not None:
            if self.precision != "f16x3" or w.hi is None:
                raise ValueError("linear: split-plane input needs the f16x3 engine and split weights")
            ld, m, cin, gs, rpg = planes.ld_rows, planes.rows, planes.c, 0, 0
        else:
            ld, m, cin, gs, rpg = rows_view(x, "linear.in")
        old, mo, cout, ogs, orpg = rows_view(out, "linear.out")
        if cin != w.cin or cout != w.cout or w.kh != 1 or w.mode == GEMM_DECONV:
            raise ValueError(f"linear: shape mismatch in {tuple(x.shape)} w ({w.cout},{w.cin}) out {tuple(out.shape)}")
        if out_row_map is None and mo != m:
            raise ValueError("linear: row count mismatch")
        if out_row_map is not None and (out_row_map.numel() != m or out_row_map.dtype != torch.int32 or not out_row_map.is_cuda):
            raise ValueError("linear: out_row_map must be a CUDA int32 tensor with one entry per GEMM row")
        res_ld = 0
        if residual is not None:
            res_ld, rm, rc, rgs, _ = rows_view(residual, "linear.residual")
            if rm != m or rc != cout or rgs != 0:
                raise ValueError("linear: residual must be a plain [M,Cout] view")
        p = GemmParams(mode=GEMM_LINEAR, in_=None if planes is not None else x.data_ptr(), in_ld=ld, N=1, H=1, W=1, Cin=cin, in_gstride=gs, in_rpg=rpg,
                       weight=w.packed.data_ptr(), Cout=cout, kh=1, kw=1, stride=1, pad=0, dil=1, Ho=1, Wo=1, M=m,
                       out=out.data_ptr(), out_ld=old, out_gstride=ogs, out_rpg=orpg, out_row_map=_ptr(out_row_map),
                       bias=_ptr(bias), prelu=None, in_prelu=None, residual=_ptr(residual), res_ld=res_ld)
        p.precision, p.weight_hi, p.weight_lo = self._prec(w)
        p.tile_wn = self.gemm_tile_wn
        if planes is not None:
            p.in_hi, p.in_lo = planes.t[0].data_ptr(), planes.t[1].data_ptr()
        if sink is not None:
            self._gemm_sink(p, sink, orpg if orpg else mo, cout, sink_c0, sink_gc, (mo // orpg) if orpg else 1, "linear")
        meta = {"flops": 2.0 * m * cout * cin, "bytes": 4.0 * (m * cin + m * cout + cout * cin), "shape": f"M{m} N{cout} K{cin}"}
        ws = self._gemm_splitk(p, m, cout, (cin + 31) // 32) if planes is not None else None
        p._srcs = (None if planes is not None else x, out, bias, residual, out_row_map, ws)
        self._run("linear_split" if planes is not None else "linear_f16x3" if p.precision else "linear", meta, self.lib.atmvfi_linear, ctypes.byref(p), self._stream())

    # ------------------------------------------------------------- transformer
    @staticmethod
    def _sink(planes: Optional[Planes], rows: int, c: int, what: str):
        """(hi, lo, ld) arguments of a kernel's split-plane sink (all null when the caller wants fp32 only)."""
        if planes is None:
            return None, None, 0
        if planes.rows != rows or planes.c != c:
            raise ValueError(f"{what}: planes hold {planes.rows} x {planes.c}, kernel writes {rows} x {c}")
        return _ptr(planes.t[0]), _ptr(planes.t[1]), planes.ld_rows

    def layernorm(self, x, out, gamma, beta, src_row_map=None, planes: Optional[Planes] = None):
        """``out`` (fp32 rows) may be None when only the split planes are wanted."""
        ld, m, c, gs, rpg = rows_view(x, "layernorm.in")
        if out is None:
            if planes is None:
                raise ValueError("layernorm: no output")
            old, mo, co, ogs = 0, planes.rows, planes.c, 0
        else:
            old, mo, co, ogs, _ = rows_view(out, "layernorm.out")
        if co != c or ogs != 0:
            raise ValueError("layernorm: output must be a plain [rows,C] view")
        if src_row_map is None and mo != m:
            raise ValueError("layernorm: row count mismatch")
        if src_row_map is not None and src_row_map.numel() != mo:
            raise ValueError("layernorm: src_row_map needs one entry per output row")
        hi, lo, pld = self._sink(planes, mo, c, "layernorm")
        meta = {"bytes": 4.0 * mo * c * (1 + (out is not None) + (planes is not None))}
        self._run("layernorm", meta, self.lib.atmvfi_layernorm, _ptr(x), ld, gs, rpg, _ptr(src_row_map), _ptr(out), old,
                  _ptr(gamma), _ptr(beta), mo, c, hi, lo, pld, self._stream())

    def dwconv_gelu(self, x, out, w9, bias, planes: Optional[Planes] = None):
        ld, n, h, w, c = nhwc_view(x, "dwconv.in")
        old = 0
        if out is not None:
            old, on, oh, ow, oc = nhwc_view(out, "dwconv.out")
            if (on, oh, ow, oc) != (n, h, w, c):
                raise ValueError("dwconv: shape mismatch")
        elif planes is None:
            raise ValueError("dwconv: no output")
        hi, lo, pld = self._sink(planes, n * h * w, c, "dwconv")
        meta = {"bytes": 4.0 * n * h * w * c * (1 + (out is not None) + (planes is not None))}
        self._run("dwconv3x3_gelu", meta, self.lib.atmvfi_dwconv3x3_gelu, _ptr(x), ld, _ptr(out), old, _ptr(w9), _ptr(bias),
                  n, h, w, c, hi, lo, pld, self._stream())

    def window_attention(self, qkv, out, motion, labels, bw, nw, ws, heads, hd, kv_shift, planes: Optional[Planes] = None):
        _chk(qkv, "attn.qkv")
        n = ws * ws
        c = heads * hd
        if out is None and planes is None:
            raise ValueError("window_attention: no output")
        if out is not None:
            _chk(out, "attn.out")
        if tuple(qkv.shape) != (bw * n, 3 * c) or not qkv.is_contiguous() or (
                out is not None and (tuple(out.shape) != (bw * n, c) or not out.is_contiguous())):
            raise ValueError(f"window_attention: qkv {tuple(qkv.shape)} / out do not match Bw {bw} N {n} C {c}")
        if motion is not None and (tuple(motion.shape) != (bw * n, heads, 2) or not motion.is_contiguous()):
            raise ValueError("window_attention: motion must be contiguous [Bw*N, heads, 2]")
        if labels is not None and (tuple(labels.shape) != (nw, n) or labels.dtype != torch.int32 or not labels.is_cuda):
            raise ValueError("window_attention: labels must be CUDA int32 [nW, N]")
        meta = {"flops": 4.0 * bw * heads * n * n * hd, "bytes": 4.0 * bw * n * 4 * c}
        hi, lo, pld = self._sink(planes, bw * n, c, "window_attention")
        fn = self.lib.atmvfi_window_attention_f16x3 if self.precision == "f16x3" and self.attention_f16x3 else self.lib.atmvfi_window_attention
        self._run("window_attention", meta, fn, _ptr(qkv), _ptr(out), _ptr(motion), _ptr(labels),
                  bw, nw, ws, heads, hd, kv_shift, hi, lo, pld, self._stream())

    def motion_head(self, motion, row_map, w0, b0, w1, b1, out, planes: Optional[Planes] = None, planes_c0: int = 0, planes_gc: int = 0):
        """``planes``: also write the two values as split planes -- channel ``planes_c0`` (+ ``planes_gc`` per row group of ``out``),
        plane row = the row inside its group."""
        old, mo, co, ogs, orpg = rows_view(out, "motion_head.out")
        rows = motion.shape[0]
        if co != 2:
            raise ValueError("motion_head: output view must have 2 channels")
        if planes is not None:
            groups = (mo // orpg) if orpg else 1
            if planes.rows != (orpg if orpg else mo) or planes_c0 % 2 or planes_gc % 2 or planes_c0 + (groups - 1) * planes_gc + 2 > planes.chunks * 32:
                raise ValueError("motion_head: the plane sink must hold one row per row of a group and even channel offsets inside its chunks")
        self._run("motion_head", {"bytes": 4.0 * rows * 18}, self.lib.atmvfi_motion_head_planes, _ptr(motion), _ptr(row_map), _ptr(w0),
                  _ptr(b0), _ptr(w1), _ptr(b1), _ptr(out), old, ogs, orpg, rows, motion.shape[1],
                  _ptr(planes.t[0]) if planes is not None else None, _ptr(planes.t[1]) if planes is not None else None,
                  planes.ld_rows if planes is not None else 0, planes_c0, planes_gc, self._stream())

    # ------------------------------------------------------------------ warps
    def _tiled_warp_ok(self, w, *planes) -> bool:
        """The LDS-staged warps (`atmvfi_flow_warp_tiled`, `atmvfi_warp_blend_tiled`; bit-identical to the direct ones) take rows by
        16-byte loads: W a multiple of 4, planes 16-byte aligned.  ``warp_tiles = False`` is the A/B switch."""
        return self.warp_tiles and w % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in planes)

    def flow_warp(self, src, flow, dst):
        _chk(src, "flow_warp.src"); _chk(dst, "flow_warp.dst")
        b, c, h, w = src.shape
        if not src.is_contiguous() or not dst.is_contiguous() or tuple(dst.shape) != (b, c, h, w):
            raise ValueError("flow_warp: src/dst must be contiguous NCHW of equal shape")
        bs, ps, cs = flow_view(flow, h, w, "flow_warp.flow")
        meta = {"bytes": 4.0 * b * h * w * (2 * c + 2)}
        fn = self.lib.atmvfi_flow_warp_tiled if self._tiled_warp_ok(w, src) else self.lib.atmvfi_flow_warp
        self._run("flow_warp", meta, fn, _ptr(src), _ptr(flow), bs, ps, cs, _ptr(dst), b, c, h, w, self._stream())

    PADDING_MODES = {"zeros": 0, "border": 1, "reflection": 2}

    def flow_warp_ex(self, src, flow, dst, mask=None, padding_mode="zeros"):
        """flow_warp(feature, flow, mask=..., padding_mode=...) of the reference in its non-default forms (flow_warp.py:50-60):
        contiguous planar src / dst [B,C,H,W], flow [B,2,H,W]; ``mask``: a [B,H,W] bool / uint8 tensor to fill, or None."""
        _chk(src, "flow_warp_ex.src"); _chk(dst, "flow_warp_ex.dst"); _chk(flow, "flow_warp_ex.flow")
        b, c, h, w = src.shape
        if padding_mode not in self.PADDING_MODES:
            raise ValueError(f"flow_warp: padding_mode must be one of {sorted(self.PADDING_MODES)}, got {padding_mode!r}")
        if (not all(t.is_contiguous() for t in (src, flow, dst)) or tuple(dst.shape) != (b, c, h, w) or tuple(flow.shape) != (b, 2, h, w)):
            raise ValueError("flow_warp_ex: contiguous src/dst [B,C,H,W] and flow [B,2,H,W] expected")
        if mask is not None and (mask.device != src.device or not mask.is_contiguous() or tuple(mask.shape) != (b, h, w) or mask.element_size() != 1):
            raise ValueError("flow_warp_ex: mask must be a contiguous [B,H,W] bool / uint8 tensor on the source's device")
        meta = {"bytes": 4.0 * b * h * w * (2 * c + 2)}
        self._run("flow_warp_ex", meta, self.lib.atmvfi_flow_warp_ex, _ptr(src), _ptr(flow), _ptr(dst), 0 if mask is None else mask.data_ptr(),
                  b, c, h, w, self.PADDING_MODES[padding_mode], self._stream())

    def flow_warp_up2(self, src, flow, dst, flow_up):
        """flow_warp(src, flow) -> dst and the flow up-sampled x2 (values doubled) -> flow_up, one launch; all contiguous planar."""
        b, c, h, w = src.shape
        if (not all(t.is_contiguous() for t in (src, flow, dst, flow_up)) or tuple(dst.shape) != (b, c, h, w) or tuple(flow.shape) != (b, 2, h, w)
                or tuple(flow_up.shape) != (b, 2, 2 * h, 2 * w)):
            raise ValueError("flow_warp_up2: contiguous src/dst [B,C,H,W], flow [B,2,H,W], flow_up [B,2,2H,2W] expected")
        meta = {"bytes": 4.0 * b * h * w * (2 * c + 2 + 8)}
        fn = self.lib.atmvfi_flow_warp_up2_tiled if self._tiled_warp_ok(w, src) else self.lib.atmvfi_flow_warp_up2
        self._run("flow_warp_up2", meta, fn, _ptr(src), _ptr(flow), _ptr(dst), _ptr(flow_up), b, c, h, w, self._stream())

    def flow_warp_nhwc(self, src, flow, dst):
        ld, b, h, w, c = nhwc_view(src, "flow_warp_nhwc.src")
        old, ob, oh, ow, oc = nhwc_view(dst, "flow_warp_nhwc.dst")
        if (ob, oh, ow, oc) != (b, h, w, c):
            raise ValueError("flow_warp_nhwc: shape mismatch")
        bs, ps, cs = flow_view(flow, h, w, "flow_warp_nhwc.flow")
        meta = {"bytes": 4.0 * b * h * w * (2 * c + 2)}
        self._run("flow_warp_nhwc", meta, self.lib.atmvfi_flow_warp_nhwc, _ptr(src), ld, src.stride(0), _ptr(flow), bs, ps, cs,
                  _ptr(dst), old, dst.stride(0), b, c, h, w, self._stream())

    def warp_blend(self, im0, im1, motion, i0w, i1w, it, flow0=None, flow1=None, mask1=None, mask2=None,
                   orig0=None, orig1=None, pack15=None, pack_planes: Optional[Planes] = None, pack_c0: int = 0):
        _planar(im0, 3, "warp_blend.im0"); _planar(im1, 3, "warp_blend.im1")
        b, _, h, w = im0.shape
        mld, mb, mh, mw, mc = nhwc_view(motion, "warp_blend.motion")
        if (mb, mh, mw, mc) != (b, h, w, 5):
            raise ValueError(f"warp_blend: motion view {tuple(motion.shape)} != [{b},{h},{w},5]")
        for t in (i0w, i1w, it):
            _planar(t, 3, "warp_blend.out")
        pld = 0
        if pack15 is not None:
            pld, pb, ph, pw, pc = nhwc_view(pack15, "warp_blend.pack15")
            if (pb, ph, pw, pc) != (b, h, w, 15):
                raise ValueError("warp_blend: pack15 view must be [B,H,W,15]")
            _planar(orig0, 3, "warp_blend.orig0"); _planar(orig1, 3, "warp_blend.orig1")
        # algorithmic bytes per pixel: two 3-channel sources + 5 motion channels read, three 3-channel frames written; the finest
        # level also writes the two flows and two masks (6 floats) and -- for the refiner -- reads the two original frames (6) and
        # writes the 15-channel pack (fp32 NHWC, or 16 channels of split planes at the same 4 bytes per element)
        per_px = 6 + 5 + 9 + (6 if flow0 is not None else 0) + ((6 + 15) if pack15 is not None else 0) + ((6 + 16) if pack_planes is not None else 0)
        meta = {"bytes": 4.0 * b * h * w * per_px}
        if pack_planes is not None:
            if pack_planes.rows != b * h * w or pack_c0 % 4 or pack_c0 + 16 > pack_planes.chunks * 32:
                raise ValueError("warp_blend: the plane sink must hold B*H*W rows and 16 channels at the offset")
            _planar(orig0, 3, "warp_blend.orig0"); _planar(orig1, 3, "warp_blend.orig1")
        fn = self.lib.atmvfi_warp_blend_tiled if self._tiled_warp_ok(w, im0, im1) else self.lib.atmvfi_warp_blend_planes
        self._run("warp_blend", meta, fn, _ptr(im0), _ptr(im1), _ptr(motion), mld, motion.stride(0),
                  _ptr(i0w), _ptr(i1w), _ptr(it), _ptr(flow0), _ptr(flow1), _ptr(mask1), _ptr(mask2), _ptr(orig0), _ptr(orig1),
                  _ptr(pack15), pld, _ptr(pack_planes.t[0]) if pack_planes is not None else None,
                  _ptr(pack_planes.t[1]) if pack_planes is not None else None,
                  pack_planes.ld_rows if pack_planes is not None else 0, pack_c0, b, h, w, self._stream())

    def resize(self, src, dst, value_scale=1.0):
        """src: any [B,C,Hi,Wi] strided view; dst: contiguous planar [B,C,Ho,Wo]."""
        _chk(src, "resize.src"); _chk(dst, "resize.dst")
        b, c, hi, wi = src.shape
        if not dst.is_contiguous() or dst.shape[0] != b or dst.shape[1] != c:
            raise ValueError("resize: dst must be contiguous [B,C,Ho,Wo]")
        meta = {"bytes": 4.0 * b * c * (hi * wi + dst.shape[2] * dst.shape[3])}
        self._run("resize_bilinear_ac", meta, self.lib.atmvfi_resize_bilinear_ac, _ptr(src), src.stride(0), src.stride(1),
                  src.stride(2), src.stride(3), _ptr(dst), b, c, hi, wi, dst.shape[2], dst.shape[3], float(value_scale), self._stream())

    def image_pyramid(self, im0, im1, l1, l2, l3, pack=None):
        """The x0.5 pyramid levels 1..3 of both frames in one launch: im0 / im1 contiguous [B,3,H,W]; l1..l3 contiguous [2B,3,H>>l,W>>l].
        ``pack``: also write torch.cat([im0, im1], 0) as NHWC4 [2B,H,W,4] (``pack_frames``' output) in the same launch."""
        b, c, h, w = im0.shape
        if pack is not None and (tuple(pack.shape) != (2 * b, h, w, 4) or not pack.is_contiguous()):
            raise ValueError("image_pyramid: pack must be contiguous [2B,H,W,4]")
        for t, l in ((l1, 1), (l2, 2), (l3, 3)):
            if not t.is_contiguous() or tuple(t.shape) != (2 * b, 3, h >> l, w >> l):
                raise ValueError(f"image_pyramid: level {l} must be contiguous [{2 * b},3,{h >> l},{w >> l}], got {tuple(t.shape)}")
        if c != 3 or im1.shape != im0.shape or not im0.is_contiguous() or not im1.is_contiguous():
            raise ValueError("image_pyramid: frames must be contiguous [B,3,H,W]")
        meta = {"bytes": 4.0 * 2 * b * 3 * h * w * (1 + 0.25 + 0.0625 + 0.015625) + (4.0 * 2 * b * h * w * 7 if pack is not None else 0.0)}
        self._run("image_pyramid", meta, self.lib.atmvfi_image_pyramid_pack, _ptr(im0), _ptr(im1), _ptr(l1), _ptr(l2), _ptr(l3), _ptr(pack),
                  b, h, w, self._stream())

    def frame_u8_to_f32(self, src_u8, dst, pad_top: int, pad_left: int, bgr: bool):
        """uint8 [H,W,3] device tensor -> fp32 planar [3,Hp,Wp] (x / 255, replicate padding, optional BGR -> RGB)."""
        if src_u8.dtype != torch.uint8 or src_u8.dim() != 3 or src_u8.shape[2] != 3 or not src_u8.is_contiguous() or not src_u8.is_cuda:
            raise ValueError("frame_u8_to_f32: source must be a contiguous CUDA uint8 [H,W,3] tensor")
        if dst.dtype != torch.float32 or dst.dim() != 3 or dst.shape[0] != 3 or not dst.is_contiguous() or not dst.is_cuda:
            raise ValueError("frame_u8_to_f32: destination must be a contiguous CUDA fp32 [3,Hp,Wp] tensor")
        h, w = src_u8.shape[:2]
        self._run("frame_u8_to_f32", {"bytes": 3.0 * h * w + 12.0 * dst.shape[1] * dst.shape[2]}, self.lib.atmvfi_frame_u8_to_f32,
                  _ptr(src_u8), h, w, int(bgr), _ptr(dst), dst.shape[1], dst.shape[2], pad_top, pad_left, self._stream())

    def frame_f32_to_u8(self, src, dst_u8, pad_top: int, pad_left: int, bgr: bool):
        """fp32 planar [3,Hp,Wp] -> crop -> np.round(x * 255) -> uint8 [H,W,3] device tensor (optional RGB -> BGR)."""
        if src.dtype != torch.float32 or src.dim() != 3 or src.shape[0] != 3 or not src.is_contiguous() or not src.is_cuda:
            raise ValueError("frame_f32_to_u8: source must be a contiguous CUDA fp32 [3,Hp,Wp] tensor")
        if dst_u8.dtype != torch.uint8 or dst_u8.dim() != 3 or dst_u8.shape[2] != 3 or not dst_u8.is_contiguous() or not dst_u8.is_cuda:
            raise ValueError("frame_f32_to_u8: destination must be a contiguous CUDA uint8 [H,W,3] tensor")
        h, w = dst_u8.shape[:2]
        self._run("frame_f32_to_u8", {"bytes": 3.0 * h * w + 12.0 * h * w}, self.lib.atmvfi_frame_f32_to_u8, _ptr(src), src.shape[1],
                  src.shape[2], pad_top, pad_left, _ptr(dst_u8), h, w, int(bgr), self._stream())

    def pack_frames(self, im0, im1, dst):
        _planar(im0, 3, "pack_frames.im0"); _planar(im1, 3, "pack_frames.im1")
        b, _, h, w = im0.shape
        if tuple(dst.shape) != (2 * b, h, w, 4) or not dst.is_contiguous():
            raise ValueError("pack_frames: dst must be contiguous [2B,H,W,4]")
        self._run("pack_frames", {"bytes": 4.0 * 2 * b * h * w * 7}, self.lib.atmvfi_pack_frames, _ptr(im0), _ptr(im1), _ptr(dst),
                  b, h, w, self._stream())

    def final_residual(self, it, r, it_sum, it_clamped):
        _planar(it, 3, "final_residual.it"); _planar(it_sum, 3, "final_residual.sum"); _planar(it_clamped, 3, "final_residual.clamped")
        b, _, h, w = it.shape
        rld, rb, rh, rw, rc = nhwc_view(r, "final_residual.r")
        if (rb, rh, rw, rc) != (b, h, w, 3):
            raise ValueError("final_residual: r must be a [B,H,W,3] view")
        self._run("final_residual", {"bytes": 4.0 * b * h * w * 12}, self.lib.atmvfi_final_residual, _ptr(it), _ptr(r), rld,
                  _ptr(it_sum), _ptr(it_clamped), b, h, w, self._stream())

    def l1_mean_workspace_floats(self, n: int, per_sample: int) -> int:
        return int(self.lib.atmvfi_l1_mean_workspace_floats(n, per_sample))

    def l1_mean(self, a, b, out, workspace=None):
        """mean |a - b| per sample, fixed summation order (run-to-run bit-identical).  ``workspace``: fp32 scratch of at least
        ``l1_mean_workspace_floats`` elements (workspace memory of the caller).  Without one (direct callers, tests) the scratch is
        kept on this object per (samples, elements): a temporary would be handed back to torch's allocator while the two passes may
        still be in flight on a side lane, and would turn into a stale pointer inside a recorded plan."""
        _chk(a, "l1_mean.a"); _chk(b, "l1_mean.b")
        if not a.is_contiguous() or not b.is_contiguous() or a.shape != b.shape:
            raise ValueError("l1_mean: inputs must be contiguous and of equal shape")
        n = a.shape[0]
        if workspace is None:
            cache = self.__dict__.setdefault("_l1_scratch", {})
            key = (n, a.numel() // n)
            workspace = cache.get(key)
            if workspace is None:
                if self.recording is not None:
                    raise PlanUnsupported("l1_mean scratch created while recording")
                workspace = cache[key] = torch.empty(self.l1_mean_workspace_floats(*key), dtype=torch.float32, device=a.device)
        self._run("l1_mean", {"bytes": 8.0 * a.numel()}, self.lib.atmvfi_l1_mean, _ptr(a), _ptr(b), _ptr(out), n, a.numel() // n,
                  _ptr(workspace), workspace.numel(), self._stream())

    def ensemble_select(self, losses, cands, out0, out1):
        """Per sample the candidate flow pair of the level with the smallest loss (first on ties): losses = three [B] tensors,
        cands = three (flow0, flow1) pairs of contiguous [B,2,h,w] tensors, out0 / out1 contiguous [B,2,h,w]."""
        b = out0.shape[0]
        per = out0.numel() // b
        ts = [t for pair in cands for t in pair] + [out0, out1]
        if len(losses) != 3 or len(cands) != 3 or any(t.shape != out0.shape or not t.is_contiguous() for t in ts) or any(l.numel() != b for l in losses):
            raise ValueError("ensemble_select: three [B] losses, three pairs of contiguous [B,2,h,w] candidates and two outputs of that shape expected")
        for t in ts + list(losses):
            _chk(t, "ensemble_select")
        self._run("ensemble_select", {"bytes": 4.0 * out0.numel() * 4}, self.lib.atmvfi_ensemble_select, _ptr(losses[0]), _ptr(losses[1]),
                  _ptr(losses[2]), _ptr(cands[0][0]), _ptr(cands[0][1]), _ptr(cands[1][0]), _ptr(cands[1][1]), _ptr(cands[2][0]),
                  _ptr(cands[2][1]), _ptr(out0), _ptr(out1), b, per, self._stream())
