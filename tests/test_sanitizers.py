"""AddressSanitizer + UBSan over the host-only part of the library (GPU sanitizers are not available on the
pool; the host code is where untrusted bytes enter: ONNX protobuf, WAV headers, label files).  The driver
tests/sanitize/host_san.cpp is built from the product's own host sources and fed valid files plus a few
hundred truncated / bit-flipped mutations of them: parse errors are fine, memory errors are not."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from test_host import _make_onnx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "formula-vad_amd", "csrc")


@pytest.fixture(scope="module")
def san_bin(tmp_path_factory):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++ for the sanitizer build")
    out = tmp_path_factory.mktemp("san") / "host_san"
    srcs = [os.path.join(CSRC, f) for f in ("host_vad.cpp", "host_stats.cpp", "host_io.cpp", "tables_weights.cpp")]
    srcs.append(os.path.join(ROOT, "tests", "sanitize", "host_san.cpp"))
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-ffp-contract=off", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           "-I" + os.path.join(ROOT, "include"), *srcs, "-o", str(out), "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build failed: " + r.stderr[-400:])
    return str(out)


def _run(san_bin, mode, args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([san_bin, mode, *args], capture_output=True, text=True, env=env, timeout=600)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return r.stdout


def _mutations(blob, rng, n, tmp, stem, ext):
    paths = []
    for i in range(n):
        b = bytearray(blob)
        kind = i % 4
        if kind == 0:                      # truncate
            b = b[: int(rng.integers(0, len(b)))]
        elif kind == 1:                    # flip bytes, mostly in the header / structure part
            for _ in range(int(rng.integers(1, 8))):
                pos = int(rng.integers(0, min(len(b), 4096) if rng.random() < 0.7 else len(b)))
                b[pos] = int(rng.integers(0, 256))
        elif kind == 2:                    # overwrite a varint / length field with a huge value
            pos = int(rng.integers(0, min(len(b), 2048)))
            b[pos:pos + 5] = b"\xff\xff\xff\xff\x7f"
        else:                              # splice a random chunk
            pos = int(rng.integers(0, len(b)))
            b[pos:pos] = rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8).tobytes()
        p = tmp / f"{stem}_{i}.{ext}"
        p.write_bytes(bytes(b))
        paths.append(str(p))
    return paths


def test_onnx_reader_under_sanitizers(san_bin, weights7, tmp_path):
    rng = np.random.default_rng(5)
    files = []
    for gemm in (False, True):
        blob = _make_onnx(weights7, gemm)
        p = tmp_path / f"valid_{int(gemm)}.onnx"
        p.write_bytes(blob)
        files.append(str(p))
        files += _mutations(blob, rng, 120, tmp_path, f"m{int(gemm)}", "onnx")
    out = _run(san_bin, "onnx", files)
    ok = int(out.split("ok=")[1].split()[0])
    assert ok >= 2                                   # both valid files parse (some mutations may too)


def _wav(pcm, fmt):
    n_ch, n = pcm.shape
    if fmt == "f32":
        data = pcm.T.astype("<f4").tobytes()
        tag, bits = 3, 32
    else:
        data = (np.clip(pcm.T, -1, 1) * 32767).astype("<i2").tobytes()
        tag, bits = 1, 16
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(data), b"WAVE", b"fmt ", 16, tag, n_ch, 48000,
                      48000 * n_ch * bits // 8, n_ch * bits // 8, bits, b"data", len(data))
    return hdr + data


def test_wav_and_label_parsers_under_sanitizers(san_bin, tmp_path):
    rng = np.random.default_rng(6)
    files = []
    for fmt in ("f32", "pcm16"):
        blob = _wav(rng.uniform(-1, 1, (2, 4000)).astype(np.float32), fmt)
        p = tmp_path / f"valid_{fmt}.wav"
        p.write_bytes(blob)
        files.append(str(p))
        files += _mutations(blob, rng, 100, tmp_path, fmt, "wav")
    out = _run(san_bin, "wav", files)
    assert int(out.split("ok=")[1].split()[0]) >= 2
    txt = "".join(f"{a:.4f}\t{a + 1.5:.4f}\tspeech\r\n" for a in np.arange(0, 50, 2.5)).encode()
    lab = tmp_path / "valid.txt"
    lab.write_bytes(txt)
    labs = [str(lab)] + _mutations(txt, rng, 100, tmp_path, "lab", "txt")
    out = _run(san_bin, "audacity", labs)
    assert int(out.split("ok=")[1].split()[0]) >= 1


def test_vad_and_stats_under_sanitizers(san_bin):
    out = _run(san_bin, "vad", ["3"])
    assert "ok=5" in out
