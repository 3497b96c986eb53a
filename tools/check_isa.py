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


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "row.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT}/include",
               "-S", "--cuda-device-only", os.path.join(CSRC, "conv3x3_f16x3_row.hip"), "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        problems = check(open(out).read())
    for p in problems:
        print("ISA check:", p)
    print("ISA check: ok" if not problems else f"ISA check: {len(problems)} problem(s)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
