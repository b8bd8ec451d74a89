"""bench.py's `extra.pcie_inclusive_*` on their own: fvad_engine_run on 128 streams x 64 s of host audio (pageable f32, pageable
PCM16, page-locked f32; with and without the denoised audio copied back), best of 3.  Context options as name=value arguments
(copy_threads=16, no_pipeline=1, run_groups=1,3,4,8 ...): python tools/pcie_run.py [option=value ...]
FVAD_TRACE_RUN=1 in the environment makes the library print a timeline of every call on stderr (when each lane group is
staged, enqueued, finished, drained)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
pkg = load_package(); fv = pkg.binding
L = fv.lib()
ctx = fv.Context(0); ctx.load_synth(7)
for a in sys.argv[1:]:
    k, v = a.split("=")
    ctx.set_option(k, v)
CHUNK = 24000
n_l, n_s = 128, 64
src = [pkg.synth.make_stream(n_s + 0.5, seed=500 + i)[0][0][: n_s * 48000].copy() for i in range(4)]
host_pcm = [src[i % 4].copy() for i in range(n_l)]
n_ch = n_s * 2
cap = (n_ch * CHUNK + 1024) // 1024 + 1
h_b = np.ones((n_l, cap), np.float32); h_r = np.ones((n_l, n_ch), np.float32)
h_d = np.ones((n_l, n_ch * CHUNK), np.float32)
fr = n_l * n_s * 100


def run(name, fill):
    arr = (fv.Lane * n_l)()
    for i in range(n_l):
        a = arr[i]
        a.band_sum = fv.fptr(h_b[i]); a.band_sum_capacity = cap
        a.chunk_rms = fv.fptr(h_r[i]); a.chunk_rms_capacity = n_ch
        fill(a, i)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        fv.check(L.fvad_engine_run(ctx.h, arr, n_l, None), "fvad_engine_run", ctx.h)
        best = min(best, time.perf_counter() - t0)
    print(f"{name:44s} {best * 1e3:7.2f} ms  {fr / best / 1e6:6.2f} M frames/s", flush=True)


def f32(den):
    def fill(a, i):
        a.pcm = fv.fptr(host_pcm[i]); a.n_samples = host_pcm[i].shape[0]
        a.denoised = fv.fptr(h_d[i]) if den else None
    return fill


run("pageable f32, no denoised D2H", f32(False))
run("pageable f32, denoised f32 D2H", f32(True))
pcm16 = [np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16) for x in src]
host16 = [pcm16[i % 4].copy() for i in range(n_l)]
h_q = np.ones((n_l, n_ch * CHUNK), np.int16)


def i16(den):
    def fill(a, i):
        a.pcm = None
        a.pcm_i16 = host16[i].ctypes.data_as(C.POINTER(C.c_int16)); a.n_samples = host16[i].shape[0]
        a.denoised_i16 = h_q[i].ctypes.data_as(C.POINTER(C.c_int16)) if den else None
    return fill


run("pageable PCM16, no denoised D2H", i16(False))
run("pageable PCM16, denoised PCM16 D2H", i16(True))
n_samp = n_s * 48000
pin_in = C.c_void_p()
fv.check(L.fvad_host_alloc(ctx.h, n_l * n_samp * 4, C.byref(pin_in)), "fvad_host_alloc", ctx.h)
a_in = np.ctypeslib.as_array(C.cast(pin_in, C.POINTER(C.c_float)), shape=(n_l, n_samp))
for i in range(n_l):
    a_in[i] = host_pcm[i]


def pinned(a, i):
    a.pcm = fv.fptr(a_in[i]); a.n_samples = n_samp; a.denoised = None


run("page-locked f32, no denoised D2H", pinned)
print("last NSNet2 path:", ctx.last_nn_path())
