#!/usr/bin/env python3
"""Symbolic execution of the straight-line fp32 motion chain in the gfx950 ISA of window_attn_x3_kernel<9> (attention.hip of commit
40b1371, built with the SLP vectorizer: the build that fails test_window_attention[ws12_hd84_12x20_s6_x-f16x3] on MI355X).

    python tools/probes/slp_isa_symexec.py k9_on.s <first line> <last line>

Every VGPR holds an expression tree; registers live into the region are symbols ``v<N>@in``.  At the end the two stored values
(motion x, motion y, before the cross-lane sums) are expanded into sums of products and printed term by term:
    x: P * (KEY - floor((KEY + 0.5) * INVWS) * WS + NEGQX)        y: P * (floor((KEY + 0.5) * INVWS) - QY)
For a correct build the multiset of (P, KEY) pairs of x equals that of y -- every probability meets the coordinates of its own key."""
import re
import sys


def parse_operand(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return ("vr", int(m.group(1)), int(m.group(2)))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return ("v", int(m.group(1)))
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return ("sr", int(m.group(1)), int(m.group(2)))
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return ("s", int(m.group(1)))
    if tok.startswith("-") and parse_operand(tok[1:])[0] in ("v",):
        return ("neg",) + parse_operand(tok[1:])
    return ("c", tok)


class M:
    def __init__(self):
        self.v = {}

    def rd(self, op, lane=0):
        k = op[0]
        if k == "v":
            return self.v.get(op[1], f"v{op[1]}@in")
        if k == "vr":
            return self.v.get(op[1] + lane, f"v{op[1] + lane}@in")
        if k == "s":
            return f"s{op[1]}"
        if k == "sr":
            return f"s{op[1] + lane}"
        if k == "neg":
            return ("neg", self.rd(op[1:]))
        return op[1]

    def wr(self, op, val, lane=0):
        self.v[(op[1] + lane) if op[0] == "vr" else op[1]] = val


def mods(rest):
    d = {}
    for name in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
        m = re.search(name + r":\[(\d),(\d)(?:,(\d))?\]", rest)
        if m:
            d[name] = [int(x) for x in m.groups() if x is not None]
    return d


def run(lines):
    m = M()
    for ln in lines:
        ln = ln.split(";")[0].strip()
        if not ln or ln.startswith(".") or ln.endswith(":"):
            continue
        parts = ln.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        md = mods(rest)
        rest_ops = re.split(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi|dst_sel|src0_sel)", rest)[0]
        ops = [parse_operand(t) for t in re.split(r",\s*(?![^\[]*\])", rest_ops) if t.strip()]
        if op in ("v_add_f32_e32", "v_sub_f32_e32", "v_mul_f32_e32"):
            a, b = m.rd(ops[1]), m.rd(ops[2])
            m.wr(ops[0], ({"v_add_f32_e32": "add", "v_sub_f32_e32": "sub", "v_mul_f32_e32": "mul"}[op], a, b))
        elif op == "v_floor_f32_e32":
            m.wr(ops[0], ("floor", m.rd(ops[1])))
        elif op in ("v_cvt_f32_ubyte0_e32", "v_cvt_f32_i32_e32", "v_cvt_f32_u32_e32"):
            m.wr(ops[0], ("float", m.rd(ops[1])))
        elif op == "v_mov_b32_e32":
            m.wr(ops[0], m.rd(ops[1]))
        elif op in ("v_pk_add_f32", "v_pk_mul_f32"):
            osel = md.get("op_sel", [0, 0])
            oselh = md.get("op_sel_hi", [1, 1])
            nlo, nhi = md.get("neg_lo", [0, 0]), md.get("neg_hi", [0, 0])
            res = []
            for half, sel, neg in ((0, osel, nlo), (1, oselh, nhi)):
                srcs = []
                for j in (0, 1):
                    o = ops[1 + j]
                    val = m.rd(o, sel[j]) if o[0] in ("vr", "sr") else m.rd(o)
                    srcs.append(("neg", val) if neg[j] else val)
                res.append(("add" if op == "v_pk_add_f32" else "mul", srcs[0], srcs[1]))
            m.wr(ops[0], res[0], 0)
            m.wr(ops[0], res[1], 1)
        elif op.startswith(("v_div_", "v_rcp", "v_fma", "v_fmac")):
            m.wr(ops[0], ("opaque", op, ln))
        elif op.startswith(("s_", "ds_bpermute", "v_cmp", "global_", "v_lshl", "v_mad", "v_mul_lo", "v_add3", "v_and", "v_or", "v_bfe", "v_add_u32", "ds_", "v_xor", "v_bitop", "v_lshr", "v_cndmask", "v_mul_hi", "v_sub_u32", "v_ashr")):
            if op.startswith(("v_",)) and ops and ops[0][0] in ("v", "vr"):
                m.wr(ops[0], ("int", ln))
        else:
            raise SystemExit(f"unhandled instruction: {ln}")
    return m


def show(e, depth=0):
    if isinstance(e, str):
        return e
    if e[0] in ("add", "sub", "mul"):
        return "(" + show(e[1]) + {"add": " + ", "sub": " - ", "mul": " * "}[e[0]] + show(e[2]) + ")"
    if e[0] == "neg":
        return "-" + show(e[1])
    if e[0] in ("floor", "float"):
        return e[0] + "(" + show(e[1]) + ")"
    return "<" + str(e[1])[:40] + ">"


def terms(e, sign=1):
    """Flatten a sum into [(sign, factor expression)]."""
    if isinstance(e, tuple) and e[0] == "add":
        return terms(e[1], sign) + terms(e[2], sign)
    if isinstance(e, tuple) and e[0] == "sub":
        return terms(e[1], sign) + terms(e[2], -sign)
    if isinstance(e, tuple) and e[0] == "neg":
        return terms(e[1], -sign)
    return [(sign, e)]


if __name__ == "__main__":
    fn, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lines = open(fn).read().split("\n")[a - 1:b]
    m = run(lines)
    for name, reg in (("x", int(sys.argv[4])), ("y", int(sys.argv[5]))):
        e = m.v[reg]
        ts = terms(e)
        print(f"== motion {name} (v{reg}): {len(ts)} terms")
        for sg, t in ts:
            print("  ", "+" if sg > 0 else "-", show(t)[:400])
