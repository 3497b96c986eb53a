"""Guard for the hand-counted LDS waits of conv3x3_f16x3_row.hip: inside a stage's MFMA stream there must be no scalar memory
load (s_load / s_buffer_load share lgkmcnt with the LDS and return out of order, which would void a counted wait), every
fragment read must be followed by a wait that covers it before the stage ends, and the counted waits must be there at all."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "atm-vfi_amd", "csrc")


def kernels(asm_text):
    cur, out = None, {}
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w*conv3x3_f16x3_row_kernelILi(\d+)ELi(\d+)ELi(\d+)E\w*):", line)
        if m:
            cur = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
            out[cur] = []
        elif cur is not None:
            out[cur].append(line)
            if "s_endpgm" in line:
                cur = None
    return out


def check(asm_text):
    problems = []
    ks = kernels(asm_text)
    if len(ks) != 16:
        problems.append(f"expected 16 instances of the row kernel, found {len(ks)}")
    for (wn, nwv, taps), lines in sorted(ks.items()):
        idx = [i for i, l in enumerate(lines) if "v_mfma_f32_16x16x32_f16" in l]
        if not idx:
            problems.append(f"<{wn},{nwv},{taps}>: no MFMA found")
            continue
        body = lines[idx[0]:idx[-1] + 1]
        if any(re.search(r"\bs_(buffer_)?load_", l) for l in body):
            problems.append(f"<{wn},{nwv},{taps}>: scalar memory load inside the MFMA stream")
        if wn >= 3:
            counted = [l for l in body if re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", l)]
            if not counted:
                problems.append(f"<{wn},{nwv},{taps}>: no counted lgkmcnt wait in the pipelined loop")
            tail = lines[idx[-1] - 8:idx[-1]]
            if not any("s_waitcnt lgkmcnt(0)" in l for l in tail):
                problems.append(f"<{wn},{nwv},{taps}>: the last MFMA group is not preceded by lgkmcnt(0)")
    return problems


def check_planes(asm_text):
    """conv3x3_planes_kernel<WN>: the k-loop alternates read phases and MFMA phases between raw s_barriers; its vmcnt waits are
    immediates computed for a loop in which every wave issues the same DMA instructions in every phase.  Per instance: 9 (first
    chunk) + 9 (chunk loop body) + 3 (tail) k-steps = 21 MFMA phases (s_setprio 1 .. s_setprio 0), each with exactly 6*WN MFMAs,
    nothing that touches memory and no branch; in front of each a read phase (back to the previous barrier) with 4 + 2*WN
    fragment reads, at least one LDS-DMA, lgkmcnt(0), no branch, and -- except in the first LA - 1 k-steps of a tile, which find
    their weights complete -- a counted vmcnt wait; 46 barriers in all; no register spill; and between the first and the last
    MFMA phase no vmcnt wait that the compiler made (the kernel's own come from inline asm)."""
    problems = []
    lines = asm_text.splitlines()
    for wn in range(1, 9):
        starts = [i for i, l in enumerate(lines) if re.match(rf"^_ZN\S*conv3x3_planes_kernelILi{wn}E\S*:", l)]
        if not starts:
            problems.append(f"planes<{wn}>: kernel not found")
            continue
        end = next(i for i in range(starts[0], len(lines)) if "s_endpgm" in lines[i])
        raw = lines[starts[0]:end]
        body = [l.split(";")[0] for l in raw]
        if any("scratch_" in l for l in body):
            problems.append(f"planes<{wn}>: register spill (scratch access)")
        bars = [i for i, l in enumerate(body) if re.search(r"\bs_barrier\b", l)]
        if len(bars) != 46:          # 2 (prologue) + 2 * (9 + 9 + 3) k-steps + 2 (tile boundary: re-align, drop behind)
            problems.append(f"planes<{wn}>: expected 46 s_barrier, found {len(bars)}")
        p1 = [i for i, l in enumerate(body) if re.search(r"\bs_setprio 1\b", l)]
        p0 = [i for i, l in enumerate(body) if re.search(r"\bs_setprio 0\b", l)]
        if len(p1) != 21 or len(p0) != 21:
            problems.append(f"planes<{wn}>: expected 21 MFMA phases, found {len(p1)} / {len(p0)}")
            continue
        nowait = 0
        for k, (a, b) in enumerate(zip(p1, p0)):
            seg = body[a + 1:b]
            n_mfma = sum("v_mfma" in l for l in seg)
            mem = [l for l in seg if re.search(r"\b(ds_|global_|buffer_|flat_|scratch_)", l)]
            if n_mfma != 6 * wn or mem or any(re.search(r"\bs_cbranch|\bs_branch", l) for l in seg):
                problems.append(f"planes<{wn}>: MFMA phase {k} has {n_mfma} MFMAs, {len(mem)} memory instructions (or a branch)")
            bar = max(i for i in bars if i < a)
            prev = max([i for i in bars if i < bar] + [0])
            rd = body[prev + 1:bar]
            waits = [l.strip() for l in rd if "s_waitcnt" in l]
            if sum("ds_read_b128" in l for l in rd) != 4 + 2 * wn or sum("global_load_lds" in l for l in rd) == 0:
                problems.append(f"planes<{wn}>: read phase {k}: unexpected fragment read / DMA count")
            if not any("lgkmcnt(0)" in w for w in waits):
                problems.append(f"planes<{wn}>: read phase {k} lacks lgkmcnt(0)")
            if not any(re.search(r"vmcnt\([1-9]\d*\)", w) for w in waits):
                nowait += 1
        if nowait > 3:               # the first LA - 1 <= 3 k-steps of the first chunk
            problems.append(f"planes<{wn}>: {nowait} read phases without a counted vmcnt wait (at most LA - 1 = 3 expected)")
        for i in range(p1[0], p0[-1]):
            if re.search(r"s_waitcnt.*vmcnt", raw[i]) and "ASMSTART" not in raw[i - 1]:
                problems.append(f"planes<{wn}>: compiler-inserted '{raw[i].strip()}' inside the k-loop region (line {i})")
    return problems


def check_pp(asm_text):
    """gemm_pp_kernel<CONVM>: every MFMA phase (between s_setprio 1 and s_setprio 0) holds exactly 48 MFMAs and nothing that
    touches memory; every read phase in front of one (back to the previous s_barrier) holds the 16 fragment reads; no register is
    spilled (a scratch reload brings a compiler-made s_waitcnt vmcnt(0) with it, which would drain the DMA ring); and the only
    vmcnt waits between the first and the last MFMA phase are the kernel's own (inline asm): vmcnt(6), vmcnt(8), and vmcnt(0) at
    the places the source puts it (last k-step but one of a last tile; the second group before its epilogue)."""
    problems = []
    lines = asm_text.splitlines()
    for tag in ("Lb0", "Lb1"):
        starts = [i for i, l in enumerate(lines) if re.match(rf"^_ZN\S*gemm_pp_kernelI{tag}E\S*:", l)]
        if not starts:
            problems.append(f"gemm_pp<{tag}>: kernel not found")
            continue
        end = next(i for i in range(starts[0], len(lines)) if "s_endpgm" in lines[i])
        body = [l.split(";")[0] for l in lines[starts[0]:end]]
        if any("scratch_" in l for l in body):
            problems.append(f"gemm_pp<{tag}>: register spill (scratch access)")
        p1 = [i for i, l in enumerate(body) if re.search(r"\bs_setprio 1\b", l)]
        p0 = [i for i, l in enumerate(body) if re.search(r"\bs_setprio 0\b", l)]
        if len(p1) != len(p0) or len(p1) < 3:
            problems.append(f"gemm_pp<{tag}>: expected matching s_setprio pairs around >= 3 MFMA phases, found {len(p1)} / {len(p0)}")
            continue
        for a, b in zip(p1, p0):
            seg = body[a + 1:b]
            n_mfma = sum("v_mfma_f32_16x16x32_f16" in l for l in seg)
            mem = [l for l in seg if re.search(r"\b(ds_|global_|buffer_|flat_|scratch_)", l)]
            if n_mfma != 48 or mem or any(re.search(r"\bs_cbranch|\bs_branch", l) for l in seg):
                problems.append(f"gemm_pp<{tag}>: MFMA phase at line {a}: {n_mfma} MFMAs, {len(mem)} memory instructions")
            bar = max(i for i in range(a) if re.search(r"\bs_barrier\b", body[i]))           # the barrier that opens the MFMA phase
            prev = max([i for i in range(bar) if re.search(r"\bs_barrier\b|\bs_setprio 0\b", body[i])] + [0])
            rd = body[prev:bar]
            if sum("ds_read_b128" in l for l in rd) < 16 or not any("lgkmcnt(0)" in l for l in rd):
                problems.append(f"gemm_pp<{tag}>: read phase before line {a}: fewer than 16 fragment reads or no lgkmcnt(0)")
        # compiler-made vmcnt waits inside the k-loop region: the kernel's own come from inline asm (marked ASMSTART / ASMEND)
        raw = lines[starts[0]:end]
        first, last = p1[0], p0[-1]
        for i in range(first, last):
            if re.search(r"s_waitcnt.*vmcnt", raw[i]) and "ASMSTART" not in raw[i - 1]:
                problems.append(f"gemm_pp<{tag}>: compiler-inserted '{raw[i].strip()}' inside the k-loop region (line {i})")
    return problems


def check_dw_dma(asm_text):
    """dwconv_gelu_dma_kernel<RS> (pointwise.hip): every wave fills a private LDS ring by LDS-DMA and waits for input row j with a vmcnt
    immediate that counts what it has issued behind that row's DMA -- two DMA instructions per row, two plane stores per output row, in
    that order.  Per instance: no spill; the prologue puts DW_RING rows in flight; then two copies of the body (image-edge x-groups / the
    others), each RS + 2 steps of {counted wait, three ds_read_b128, lgkmcnt(0), [2 DMA], [2 stores]}; replaying the instruction stream,
    the DMA of row j must be exactly N + 1 .. N + 2 operations back at its `s_waitcnt vmcnt(N)`; no other vmcnt wait inside a body."""
    problems = []
    lines = asm_text.splitlines()
    # the ring depth comes from the kernel source, not from a second hard-coded copy here
    m = re.search(r"constexpr\s+int\s+DW_RING\s*=\s*(\d+)\s*;", open(os.path.join(CSRC, "pointwise.hip")).read())
    if not m:
        return ["dw_dma: constexpr int DW_RING not found in pointwise.hip"]
    ring = int(m.group(1))
    for rs in (8, 16, 17):
        starts = [i for i, l in enumerate(lines) if re.match(rf"^_ZN\S*dwconv_gelu_dma_kernelILi{rs}E\S*:", l)]
        if not starts:
            problems.append(f"dw_dma<{rs}>: kernel not found")
            continue
        end = next(i for i in range(starts[0], len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
        body = lines[starts[0]:end]
        if any(re.search(r"scratch_(load|store)", l) for l in body):
            problems.append(f"dw_dma<{rs}>: register spill")
        ev = []
        for l in body:
            t = l.strip()
            if t.startswith("global_load_lds_dwordx4"):
                ev.append("dma")
            elif t.startswith("global_store_dwordx2"):
                ev.append("store")
            elif t.startswith("global_store") or t.startswith("global_load") or t.startswith("buffer_") or t.startswith("flat_"):
                ev.append("other")
            elif t.startswith("s_waitcnt") and "vmcnt(" in t:
                ev.append(int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        rows = rs + 2
        # prologue: the weight loads ("other"), 2 * ring DMAs, hipcc's own wait for the weights
        k = 0
        while k < len(ev) and ev[k] == "other":
            k += 1
        if ev[k:k + 2 * ring] != ["dma"] * (2 * ring):
            problems.append(f"dw_dma<{rs}>: the prologue does not issue {2 * ring} DMAs behind the weight loads")
            continue
        k += 2 * ring
        # hipcc's own wait for the weight loads: at most ONE, and it must leave the ring's DMAs in flight is not required of it (the
        # compiler counts only the builtin DMAs), but a second wait here, or one counted beyond what has been issued, is a change
        # of schedule this guard should see
        nwait = 0
        while k < len(ev) and isinstance(ev[k], int) and nwait < 1 and ev[k] <= 2 * ring:
            k += 1
            nwait += 1
        for copy in range(2):
            queue = [("dma", j) for j in range(ring) for _ in range(2)]          # what is in flight, oldest first (upper bound)
            for j in range(rows):
                if k >= len(ev) or not isinstance(ev[k], int):
                    problems.append(f"dw_dma<{rs}> body {copy}, step {j}: expected a counted vmcnt wait, found {ev[k] if k < len(ev) else 'the end'}")
                    break
                last = max(i for i, q in enumerate(queue) if q == ("dma", j))
                behind = len(queue) - 1 - last
                if ev[k] != behind:
                    problems.append(f"dw_dma<{rs}> body {copy}, step {j}: vmcnt({ev[k]}) but {behind} operations were issued behind row {j}'s DMA")
                k += 1
                want = (["dma", "dma"] if j + ring < rows else []) + (["store", "store"] if j >= 2 else [])
                if ev[k:k + len(want)] != want:
                    problems.append(f"dw_dma<{rs}> body {copy}, step {j}: expected {want}, found {ev[k:k + len(want)]}")
                    break
                k += len(want)
                queue += [("dma", j + ring)] * (2 if j + ring < rows else 0) + [("store", j - 2)] * (2 if j >= 2 else 0)
        if k != len(ev):
            problems.append(f"dw_dma<{rs}>: {len(ev) - k} unexpected vector-memory operations / waits after the two bodies: {ev[k:k + 6]}")
    return problems


def check_no_slp_pairs(asm_text, what):
    """No packed fp32 instruction with a LOW-half operand select (v_pk_*_f32 ... op_sel:[..]): the library's own f32x2 code only ever
    broadcasts through op_sel_hi; op_sel:[..] on fp32 pairs is the shape the SLP vectorizer builds when it pairs (x, y) chains, and
    the one build that had it was wrong on the hardware (csrc/Makefile).  The library is compiled with -fno-slp-vectorize."""
    bad = [l.strip() for l in asm_text.splitlines() if re.search(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[", l)]
    return [f"{what}: {len(bad)} packed fp32 instruction(s) with a low-half operand select, e.g. '{bad[0]}'"] if bad else []


def makefile_flags():
    """The compiler flags the library is built with (csrc/Makefile CXXFLAGS), so that this guard looks at the code that ships."""
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith("CXXFLAGS"):
            fl = line.split("=", 1)[1].split()
            return [f for f in fl if f.startswith("-f") and f != "-fPIC"]
    raise SystemExit("check_isa: CXXFLAGS not found in csrc/Makefile")


def file_flags(src):
    """Per-file additions (FLAGS_<file> in the Makefile)."""
    stem = os.path.splitext(src)[0]
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith(f"FLAGS_{stem} "):
            return line.split("=", 1)[1].split()
    return []


def _kernel_body(lines, pattern):
    starts = [i for i, l in enumerate(lines) if re.match(pattern, l)]
    if not starts:
        return None
    end = next(i for i in range(starts[0], len(lines)) if "s_endpgm" in lines[i])
    return [l.split(";")[0].rstrip() for l in lines[starts[0]:end + 1]]


def _blocks(body, first):
    """Basic blocks of body[first:]: (label or None, [instructions]); a block ends at a label or behind a branch."""
    out, cur, name = [], [], None
    for l in body[first:]:
        t = l.strip()
        if not t or t.startswith(".") and not re.match(r"^\.LBB\S+:", t):
            continue
        if re.match(r"^\.LBB\S+:", t):
            if cur:
                out.append((name, cur))
            name, cur = t[:-1], []
            continue
        cur.append(t)
        if re.match(r"s_(c?branch|endpgm)", t):
            out.append((name, cur))
            name, cur = None, []
    if cur:
        out.append((name, cur))
    return out


def _count(insts):
    c = {"valu": 0, "pk": 0, "cvt_split": 0, "swap": 0, "vmax": 0, "cndmask": 0, "store": 0, "store_b128": 0, "ds": 0, "salu": 0, "mfma": 0, "waits": 0}
    for t in insts:
        t = t.strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            c["pk"] += op.startswith("v_pk_")
            c["cvt_split"] += op.startswith("v_cvt_pk") or op.startswith("v_fma_mix")
            c["swap"] += op.startswith("v_permlane")
            c["vmax"] += op.startswith("v_max_f32")
            c["cndmask"] += op.startswith("v_cndmask")
        elif op.startswith("global_store") or op.startswith("buffer_store"):
            c["store"] += 1
            c["store_b128"] += op.endswith("dwordx4")
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op == "s_waitcnt" or op == "s_nop":
            c["waits"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def epilogue_budget(planes_asm, pp_asm):
    """the vector-instruction budget of the contraction kernels' epilogues, counted on the ISA that ships.
    Static: the instructions behind the k-loop's last MFMA phase, per basic block (the epilogue forms are uniform branches: fold +
    bias + PReLU in its three forms, fp32 rows, each plane sink in its forms), classified by what they contain; per block the
    counts and, for the blocks that handle a whole tile's values, instructions per output element (a lane of conv3x3_planes<WN>
    owns 2 pixels x 16 WN channels / 4 lane groups = 8 WN outputs of a tile; a lane of gemm_pp 64 x 64 / 64 = 64)."""
    rows = ["# Epilogue instruction budget (static count on the shipped ISA; tools/check_isa.py --budget)", ""]
    lines = planes_asm.splitlines()
    for wn in (4, 7, 8):
        body = _kernel_body(lines, rf"^_ZN\S*conv3x3_planes_kernelILi{wn}E\S*:")
        if body is None:
            continue
        p0 = [i for i, l in enumerate(body) if re.search(r"\bs_setprio 0\b", l)]
        outs = 8 * wn
        kloop = _count(body[:p0[-1]])
        rows.append(f"## conv3x3_planes_kernel<{wn}>: {outs} outputs per lane and tile; k-loop region: {kloop['mfma']} MFMAs in 21 unrolled k-steps "
                    f"({6 * wn} per k-step), {kloop['valu']} other vector instructions, {kloop['ds']} LDS, {kloop['salu']} scalar")
        rows.append("| stage (all basic blocks of that kind) | blocks | VALU | of which packed | split (cvt_pk / fma_mix) | permlane swap | v_max | v_cndmask | stores | copies of the stage in the code | VALU per output and copy |")
        rows.append("|---|---|---|---|---|---|---|---|---|---|---|")
        groups = {}
        for name, insts in _blocks(body, p0[-1] + 1):
            c = _count(insts)
            if c["valu"] + c["store"] == 0:
                continue
            if c["mfma"]:
                kind = "fused read-out (refine_head.1): split + MFMAs + stores"
            elif c["swap"] and c["vmax"]:
                kind = "plane sink with its own PReLU, max form (split, swap, 16-byte stores)"
            elif c["swap"] and c["cndmask"]:
                kind = "plane sink with its own PReLU, select form"
            elif c["swap"]:
                kind = "plane sink, no activation (split, swap, 16-byte stores)"
            elif c["store"]:
                kind = "fp32 rows (and the stores of masked sink lanes)"
            elif c["vmax"] >= outs // 2:
                kind = "fold (acc + cor / 1024) + bias + PReLU, max form"
            elif c["cndmask"] >= outs // 2:
                kind = "fold + bias + PReLU, select form"
            elif c["pk"] >= outs // 2:
                kind = "fold + bias, no activation"
            else:
                kind = "addresses, predicates, slope-range check, next-tile bookkeeping"
            g = groups.setdefault(kind, dict(blocks=0, valu=0, pk=0, cvt_split=0, swap=0, vmax=0, cndmask=0, store=0))
            g["blocks"] += 1
            for k in ("valu", "pk", "cvt_split", "swap", "vmax", "cndmask", "store"):
                g[k] += c[k]
        npair = (wn + 1) // 2
        for kind, g in groups.items():
            copies = 1
            if kind.startswith("plane sink"):
                copies = max(1, round(g["swap"] / (8 * npair)))          # 4 swaps per (n-tile pair, pixel row): 8 npair per copy
            per = f"{g['valu'] / (copies * outs):.2f}" if not kind.startswith(("addresses", "fp32", "fused")) else "-"
            rows.append(f"| {kind} | {g['blocks']} | {g['valu']} | {g['pk']} | {g['cvt_split']} | {g['swap']} | {g['vmax']} | {g['cndmask']} | {g['store']} | {copies} | {per} |")
        rows.append("")
    lines = pp_asm.splitlines()
    for tag, nm in (("Lb0", "LINEAR / DECONV"), ("Lb1", "CONV")):
        body = _kernel_body(lines, rf"^_ZN\S*gemm_pp_kernelI{tag}E\S*:")
        if body is None:
            continue
        p0 = [i for i, l in enumerate(body) if re.search(r"\bs_setprio 0\b", l)]
        rows.append(f"## gemm_pp_kernel<{nm}>: 64 outputs per lane and tile (16 vectors of 4); every epilogue form is a separate copy of the row loop")
        rows.append("| stage (all basic blocks of that kind) | blocks | VALU | of which packed | split (cvt_pk / fma_mix) | v_cndmask | stores | LDS | VALU per output (by the stores: 4 outputs per 16-byte fp32 store, 2 per 8-byte plane store) |")
        rows.append("|---|---|---|---|---|---|---|---|---|")
        groups = {}
        for name, insts in _blocks(body, p0[-1] + 1):
            c = _count(insts)
            if c["valu"] + c["store"] + c["ds"] == 0:
                continue
            if c["ds"] >= 8 and not c["store"]:
                kind = "fold + transposition through LDS (ds_write_b128 / ds_read_b128)"
            elif c["cvt_split"] and c["store"]:
                kind = "plane-sink rows (split + 8-byte stores): strided / 1x1 convs, deconvs, fc2's sink -- all copies"
            elif c["store"]:
                kind = "fp32 rows (bias, PReLU, residual, 16-byte stores) -- all copies"
            else:
                kind = "constants, row maps, residual batch, addresses, next-tile bookkeeping"
            g = groups.setdefault(kind, dict(blocks=0, valu=0, pk=0, cvt_split=0, cndmask=0, store=0, ds=0))
            g["blocks"] += 1
            for k in ("valu", "pk", "cvt_split", "cndmask", "store", "ds"):
                g[k] += c[k]
        for kind, g in groups.items():
            per = (f"{g['valu'] / (4 * g['store']):.2f}" if kind.startswith("fp32") else f"{g['valu'] / (2 * g['store']):.2f}" if kind.startswith("plane") else
                   f"{g['valu'] / 64:.2f}" if kind.startswith("fold") else "-")
            rows.append(f"| {kind} | {g['blocks']} | {g['valu']} | {g['pk']} | {g['cvt_split']} | {g['cndmask']} | {g['store']} | {g['ds']} | {per} |")
        rows.append("")
    return "\n".join(rows)


def compile_asm(src, out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", *makefile_flags(), *file_flags(src), f"-I{ROOT}/include",
           "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    if "--budget" in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as td:
            print(epilogue_budget(compile_asm("conv3x3_planes.hip", os.path.join(td, "planes.s")), compile_asm("gemm_pp.hip", os.path.join(td, "pp.s"))))
        return 0
    with tempfile.TemporaryDirectory() as td:
        problems = check(compile_asm("conv3x3_f16x3_row.hip", os.path.join(td, "row.s")))
        planes = compile_asm("conv3x3_planes.hip", os.path.join(td, "planes.s"))
        problems += check_planes(planes)
        problems += check_no_slp_pairs(planes, "conv3x3_planes.hip")
        pp = compile_asm("gemm_pp.hip", os.path.join(td, "pp.s"))
        problems += check_pp(pp)
        if "-fno-slp-vectorize" not in makefile_flags():
            problems.append("csrc/Makefile: CXXFLAGS lost -fno-slp-vectorize")
        problems += check_dw_dma(compile_asm("pointwise.hip", os.path.join(td, "pointwise.s")))
        for name, text in (("gemm_pp.hip", pp), ("attention.hip", compile_asm("attention.hip", os.path.join(td, "attn.s"))),
                           ("stem.hip", compile_asm("stem.hip", os.path.join(td, "stem.s")))):
            problems += check_no_slp_pairs(text, name)
            if name == "stem.hip" and re.search(r"scratch_(load|store)", text):
                problems.append("stem.hip: register spill (a scratch reload is a vector-memory operation: its wait covers the next tile's patch loads)")
    for p in problems:
        print("ISA check:", p)
    print("ISA check: ok" if not problems else f"ISA check: {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
