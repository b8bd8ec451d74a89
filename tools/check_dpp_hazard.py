"""The spectral kernels issue v_fmac_f32_dpp through inline assembly (csrc/fft_device.h: dpp_butterfly, dpp_butterfly_raw), which
the compiler's hazard recogniser does not see: a DPP read of a VGPR needs two wait states after a VALU write of that VGPR.  This
script compiles the spectral translation units to assembly and checks every DPP instruction against the instructions in front of it -- a
register-allocator copy or a spill reload landing inside a block of butterflies would otherwise corrupt results silently.
python tools/check_dpp_hazard.py  (exit code 1 and a listing on a violation; needs hipcc, no GPU)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "formula-vad_amd", "csrc", f) for f in ("kernels_fft.hip", "kernels_stft.hip", "kernels_vadfft.hip", "kernels_fftgen.hip")]


def regs(tok):
    """VGPR numbers named by an operand like v12 or v[4:7]"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(asm_text):
    bad, n_dpp = [], 0
    window = []  # the last instructions: (wait states it provides, VGPRs it writes, text)
    for line in asm_text.split("\n"):
        t = line.strip()
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            if t.endswith(":") and not t.startswith(";"):
                window = []  # a label: a jump target, judged conservatively below (nothing known in front)
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if "dpp" in op or "row_" in t or "quad_perm" in t:
            n_dpp += 1
            src = regs(ops[1].split()[0]) if len(ops) > 1 else set()
            states = 0
            for ws, writes, text in reversed(window):
                if states >= 2:
                    break
                if writes & src:
                    bad.append((text, t))
                    break
                states += ws
            if not window:
                bad.append(("<label directly in front>", t))
        if op.startswith("s_nop"):
            window.append((int(ops[0], 0) + 1 if ops else 1, set(), t))
        elif op.startswith("v_"):
            window.append((1, regs(ops[0].split()[0]) if ops else set(), t))
        else:
            window.append((1, set(), t))
        window = window[-4:]
    return n_dpp, bad


def main():
    n_all, bad_all = 0, []
    with tempfile.TemporaryDirectory() as d:
        for src in SRCS:
            out = os.path.join(d, "k.s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                   "-I" + os.path.join(ROOT, "include"), "-x", "hip", "--cuda-device-only", "-S", src, "-o", out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                print(r.stderr[-2000:])
                return 2
            n, bad = check(open(out).read())
            n_all += n
            bad_all += bad
    print(f"{n_all} DPP instructions, {len(bad_all)} without two wait states behind a VALU write of their source")
    for w, t in bad_all[:20]:
        print("   ", w, "->", t)
    return 1 if bad_all or n_all == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
