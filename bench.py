#!/usr/bin/env python3
"""bench.py -- end-to-end VAD pipeline throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of synthetic 48 kHz mono streams that
are already resident in HBM: chunk RMS -> /3 decimation -> sqrt-Hann STFT-320 -> log-power features
-> NSNet2 (fp32 MFMA) -> gain -> inverse STFT overlap-add -> x3 upsample -> 1024-point Hann rFFT ->
500-2000 Hz band sum (all on the GPU), then the band sums go back to the host and the reference's
sequential VAD state machine (exact f64 order) turns them into speech segments.  The host stage
of step i overlaps the GPU stage of step i+1; all K steps complete inside the timed region.

Unit: 1 frame = one NSNet2 STFT frame of one channel = 10 ms of audio (SURVEY.md section 8d;
BASELINE.json calls it a "20 ms frame" after its window length).  Streams shard across ranks with
no data-path collective (weak scaling: every rank processes the same amount of audio); the only
collective is the final all_gather of per-stream Evaluator statistics (RCCL), outside the steps.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/F16 MFMA, dense (the f16x3 kernels issue v_mfma_f32_16x16x32_f16)
HBM_PEAK_GBPS = 8000.0          # spec
FRAMES_PER_CHUNK = 50
CHUNK = 24000
GRU_FLOP_PER_CHUNK_LAUNCH = 53 * 2 * 1200 * 400   # one GRU layer's recurrence: 53 steps with h != 0
# the same recurrence as the f16x3 kernel executes it: three f16 MFMAs per product, K padded 400 -> 416
GRU_H3_MFMA_FLOP_PER_CHUNK_LAUNCH = 53 * 2 * 1200 * 416 * 3
# the network as the reference runs it (fc1 separate, no padding): "effective" FLOPs
NSNET2_FLOP_PER_CHUNK = 2 * (54 * (161 * 400 + 2 * 1200 * 400) + 53 * 2 * 1200 * 400
                             + 50 * (400 * 600 + 600 * 600 + 600 * 161))
# what the kernels execute: fc1 folded into GRU1's input projection (K 161 -> 176), fc2/fc3 columns padded
# to 39 tiles, fc3/fc4 K padded to 608, fc4 columns to 176
NSNET2_EXECUTED_FLOP_PER_CHUNK = 2 * (54 * (176 * 1200 + 400 * 1200) + 53 * 2 * 1200 * 400
                                      + 50 * (400 * 624 + 608 * 624 + 608 * 176))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=384, help="streams per GPU per step")
    ap.add_argument("--seconds", type=int, default=64, help="audio seconds per stream per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg4-strong", action="store_true", help="N > 1: skip the strong-scaling block (config 4's plan on the same ranks)")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-2 / config-3 side measurements")
    ap.add_argument("--vad-threads", type=int, default=0)
    ap.add_argument("--nn-math", default="f32", choices=("f32", "f16x3", "bf16x3"),
                    help="arithmetic of the NSNet2 matrix products of the HEADLINE (fvad_ctx_set_nn_math): f32 = "
                         "v_mfma_f32_16x16x4_f32, the reference's arithmetic (default); f16x3 = the emulation (three f16 "
                         "MFMAs on split f32 operands, 22-bit operands); bf16x3 = the dense layers as six bf16 MFMAs on "
                         "exact three-piece splits (24-bit operands), recurrences on f32 MFMA.  With f32 both emulations "
                         "are timed too, over the same number of steps, and reported in the `emulated` / `emulated24` blocks")
    ap.add_argument("--no-emulated", action="store_true", help="skip the `emulated` block (the f16x3 path)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL; one rank per GPU) or gloo (rehearsal: every rank on cuda:0)")
    ap.add_argument("--config", default="headline", choices=("headline", "cfg4"),
                    help="headline: BASELINE config 3's pipeline at a saturating batch (the metric's workload); "
                         "cfg4: BASELINE config 4, 21 Miami-race-sized streams dealt round-robin to the ranks")
    ap.add_argument("--cfg4-streams", type=int, default=21)
    ap.add_argument("--cfg4-seconds", type=int, default=7200, help="seconds per stream (a multiple of 600)")
    ap.add_argument("--cfg4-one-call", action="store_true",
                    help="config 4: one device-resident call per step and the host VAD behind it (rounds 1-4) instead of time slices with the host VAD beside the GPU")
    return ap.parse_args()


def make_inputs(pkg, rank, lanes, seconds):
    """lanes x seconds of synthetic 48 kHz mono, every lane different: 32 seeded 64 s patterns per rank, lane l =
    pattern l % 32 rotated by (l // 32) * 4801 samples (a rotation is a different stream: other chunk phases,
    other burst times) and tiled to the requested length"""
    base_sec = min(64, seconds)
    out = np.empty((lanes, seconds * 48000), np.float32)
    labels = []
    cache = {}
    reps = (seconds + base_sec - 1) // base_sec
    for lane in range(lanes):
        key = lane % 32
        if key not in cache:
            cache[key] = pkg.synth.make_stream(float(base_sec), seed=1000 * rank + key)
        pcm, lab = cache[key]
        shift = 4801 * (lane // 32)
        out[lane] = np.tile(np.roll(pcm[0], shift), reps)[: seconds * 48000]
        labels.append([(a, b) for a, b in roll_labels(lab, shift / 48000.0, float(base_sec), reps) if b <= seconds])
    return out, labels


def cpu_baseline(pkg, fv, weights, n_threads):
    """The oracle (a from-scratch C restatement of the reference's algorithm; the Zig + kissfft +
    onnxruntime reference cannot be built here) timed on this box's host cores, one thread per
    stream like src/simulator.zig:221-232, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    native = orc.lib(native=True)  # -O3 -march=native build of the same source, made on this box
    seconds = 150.0
    streams = [pkg.synth.make_stream(seconds, seed=5000 + i)[0] for i in range(n_threads)]

    import ctypes as C
    w, keep = orc.make_weights_struct(weights)

    def work(pcm):
        cfg = orc.PipelineConfig()
        native.orc_pipeline_config_default(C.byref(cfg))
        err = C.c_int(0)
        h = native.orc_pipeline_create(C.byref(cfg), C.byref(w), C.byref(err))
        ptrs = (orc.c_float_p * 1)(orc.fptr(pcm[0]))
        native.orc_pipeline_push_samples(h, ptrs, pcm.shape[1])
        native.orc_pipeline_destroy(h)

    work(streams[0][:, : 24000 * 2])  # warm
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(s,)) for s in streams]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    frames = n_threads * int(seconds * 48000) // CHUNK * FRAMES_PER_CHUNK
    return {"value": frames / dt, "unit": "frames/s", "cpu_model": cpu_model(), "cores": n_threads, "kind": "port",
            "sample": f"{n_threads} streams x {seconds:.0f} s mono through the C oracle pipeline "
                      f"(oracle/, -O3 -march=native), one thread per stream; {dt:.1f} s wall"}


def check_against_oracle(ctx, weights, host_pcm, lanes, d_den, n_samp, band, segments):
    """The checker (oracle/, test infrastructure) on a few lanes of the batch the benchmark has just run.
    Tolerances are the parity tests': denoised <= 1e-4 of peak, band sums <= 1e-4 relative, segments exact."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    out = {"lanes": list(lanes), "max_err": {"denoised_of_peak": 0.0, "band_sum_rel": 0.0}, "segments_identical": True,
           "n_segments": 0, "checker": "oracle/ (CPU restatement), orc.Pipeline on the same samples"}
    den = np.empty(n_samp // CHUNK * CHUNK, np.float32)
    for lane in lanes:
        p = orc.Pipeline(weights, n_channels=1, keep_denoised=True)
        p.push(host_pcm[lane][None])
        ref_den, ref_band = p.denoised()[0], p.band_volumes()[:, 0]
        ctx.to_host(den, d_den + lane * den.nbytes)
        e_den = float(np.abs(den.astype(np.float64) - ref_den).max() / np.abs(ref_den).max())
        e_band = float((np.abs(band[lane].astype(np.float64) - ref_band) / np.abs(ref_band)).max())
        out["max_err"]["denoised_of_peak"] = max(out["max_err"]["denoised_of_peak"], e_den)
        out["max_err"]["band_sum_rel"] = max(out["max_err"]["band_sum_rel"], e_band)
        ref_segs = [(s[0], s[1]) for s in p.segments()]
        got_segs = [(s[0], s[1]) for s in segments[lane]]
        out["segments_identical"] = out["segments_identical"] and got_segs == ref_segs
        out["n_segments"] += len(ref_segs)
    out["ok"] = bool(out["max_err"]["denoised_of_peak"] <= 1e-4 and out["max_err"]["band_sum_rel"] <= 1e-4
                     and out["segments_identical"] and np.isfinite(out["max_err"]["denoised_of_peak"]))
    return out


def gather_all_stats(pkg, fv, ctx, dist, world, args, cdev, local_ids, local_stats, n_streams):
    """Per-stream SingleStats of every rank, in plan order.  One rank per GPU: the library's own RCCL
    all-gather (fvad_stats_allgather, no torch on the path); rehearsal ranks that share a GPU, or a failure of
    the native path, use the same exchange over torch.distributed.  Returns (stats, description, ranks the RCCL
    communicator saw or None)."""
    if world == 1:
        return pkg.shard.gather_stats(local_ids, [fv.single_stats_to_array(s) for s in local_stats], n_streams), "none", None
    if args.dist_backend == "nccl":
        try:
            comm = pkg.shard.native_comm(ctx, dist)
            seen = int(fv.lib().fvad_comm_world(comm.h))
            out = pkg.shard.gather_stats_native(comm, local_ids, local_stats, n_streams)
            comm.close()
            return out, "fvad_stats_allgather (ncclAllGather via librccl, C ABI)", seen
        except Exception as e:  # keep the run alive; the JSON line says what happened
            note = f"torch.distributed all_gather(nccl) after native path failed: {e!r}"
    else:
        note = f"torch.distributed all_gather({args.dist_backend})"
    out = pkg.shard.gather_stats(local_ids, [fv.single_stats_to_array(s) for s in local_stats], n_streams,
                                 dist=dist, device=cdev)
    return out, note, None


def stats_digest(allst, agg):
    """sha256 over the plan-order per-stream statistics (11 f32 each) and the aggregate struct built from them"""
    import hashlib
    h = hashlib.sha256()
    for a in allst:
        h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    h.update(bytes(agg))
    return h.hexdigest()


def roll_labels(labels, shift_s, period_s, reps):
    """labels of np.tile(np.roll(x, shift), reps): every burst moves by shift_s modulo the period (split where it
    wraps) and repeats every period"""
    out = []
    for a, b in labels:
        a2, b2 = (a + shift_s) % period_s, (b + shift_s) % period_s
        parts = [(a2, b2)] if a2 < b2 else [(a2, period_s), (0.0, b2)]
        for r in range(reps):
            out += [(x + r * period_s, y + r * period_s) for x, y in parts if y - x > 1e-6]
    return sorted(out)


def run_cfg4(args, pkg, fv, ctx, torch, dist, rank, world, cdev, steps=None, warmup=None, emit=True):
    """(emit=False: return rank 0's record instead of printing it -- the `cfg4_strong` block of a multi-GPU headline run.)
    BASELINE config 4: 21 independent streams (Miami-race sized) dealt round-robin to the ranks
    (shard.streams_for_rank: 3,3,3,3,3,2,2,2 over 8), every rank runs the whole path for its streams -- GPU
    kernels, host VAD, Evaluator statistics against the streams' labels -- then ONE all-gather of the per-stream
    SingleStats and the plan-order aggregate (statistics.zig:116-172).  A step = the whole plan."""
    L = fv.lib()
    n_streams, seconds = args.cfg4_streams, args.cfg4_seconds
    period = min(600, seconds)
    reps = seconds // period
    seconds = reps * period
    mine = pkg.shard.streams_for_rank(n_streams, rank, world)
    n_l = len(mine)
    n_samp = seconds * 48000
    n_chunks = n_samp // CHUNK
    n_frames_fft = n_chunks * CHUNK // 1024
    base, base_labels = pkg.synth.make_stream(float(period) + 0.5, seed=900)
    base = base[0][: period * 48000].copy()
    labels = {}
    d_pcm = ctx.device_alloc(max(n_l, 1) * n_samp * 4)
    for j, sid in enumerate(mine):                       # inputs resident in HBM before timing
        x = np.tile(np.roll(base, 4801 * sid), reps)
        ctx.to_device(d_pcm + j * n_samp * 4, x)
        labels[sid] = roll_labels(base_labels, 4801 * sid / 48000.0, float(period), reps)
        del x
    d_band = ctx.device_alloc(max(n_l, 1) * n_frames_fft * 4)
    d_rms = ctx.device_alloc(max(n_l, 1) * n_chunks * 4)
    h_band = np.empty((n_l, n_frames_fft), np.float32)
    h_rms = np.empty((n_l, n_chunks), np.float32)
    stat_cfg = {"ignore_shorter_than_sec": 0.7, "extrude_start": 5.0, "extrude_end": 10.0, "fill_gaps": 5.0}
    vad_threads = args.vad_threads or min(max(n_l, 1), 16)
    vb = fv.VadBatch(max(n_l, 1))

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comm = None
    collective = "none"
    if world > 1 and args.dist_backend == "nccl":
        try:
            comm = pkg.shard.native_comm(ctx, dist)
            collective = "fvad_stats_allgather (ncclAllGather via librccl, C ABI)"
        except Exception as e:
            collective = f"torch.distributed all_gather(nccl) after native bootstrap failed: {e!r}"
    elif world > 1:
        collective = f"torch.distributed all_gather({args.dist_backend})"
    timing = {}

    # The rank's streams as time slices, the host VAD of slice k beside the GPU's slice k + 1 (shard.run_sliced_with_vad): a slice
    # is one launch of at most 49152 chunks, and at least four slices even for two or three streams (their VAD is 36 ms per
    # two-hour stream on one core: a third of a rank's step if it ran behind the kernels)

    def step():
        t0 = time.perf_counter()
        all_segs = []
        if n_l and not args.cfg4_one_call:
            all_segs, info = pkg.shard.run_sliced_with_vad(ctx, d_pcm, n_l, n_samp, n_chunks, vb, n_threads=vad_threads)
            t1 = t0 + info["gpu_s"]
        elif n_l:
            fv.check(L.fvad_engine_enqueue_device(ctx.h, d_pcm, n_l, n_samp, n_samp, None, d_band, d_rms, None), "cfg4 enqueue", ctx.h)
            fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_band.ctypes.data, d_band, h_band.nbytes), "cfg4 band", ctx.h)
            fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_rms.ctypes.data, d_rms, h_rms.nbytes), "cfg4 rms", ctx.h)
            ctx.synchronize()
            t1 = time.perf_counter()
            all_segs = vb.run(h_band, h_rms, n_threads=vad_threads)
        else:
            t1 = time.perf_counter()
        local = []
        n_seg = 0
        for segs, sid in zip(all_segs, mine):
            n_seg += len(segs)
            secs = [(np.float32(s_[0]) / np.float32(48000), np.float32(s_[1]) / np.float32(48000)) for s_ in segs]
            local.append(fv.stats_from_segments(secs, labels[sid], stat_cfg))
        t2 = time.perf_counter()
        if comm is not None:
            allst = pkg.shard.gather_stats_native(comm, mine, local, n_streams)
        else:
            allst = pkg.shard.gather_stats(mine, [fv.single_stats_to_array(s_) for s_ in local], n_streams,
                                           dist=dist if world > 1 else None, device=cdev if world > 1 else None)
        agg = fv.stats_aggregate([fv.array_to_single_stats(a) for a in allst])   # plan order, every rank
        t3 = time.perf_counter()
        timing.update(gpu=(t1 - t0), host=(t2 - t1), gather_aggregate=(t3 - t2), segments=n_seg)
        return allst, agg

    n_steps = args.steps if steps is None else steps
    for _ in range(args.warmup if warmup is None else warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        allst, agg = step()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0])
    rccl_ranks = comm.world if comm is not None else None
    if comm is not None:
        comm.close()
    vb.close()
    for d in (d_pcm, d_band, d_rms):
        ctx.device_free(d)
    if rank == 0:
        frames = n_streams * n_chunks * FRAMES_PER_CHUNK
        out = {
            "metric": "20ms audio frames/sec end-to-end VAD pipeline", "value": frames * n_steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": n_steps, "warmup": args.warmup if warmup is None else warmup, "ms_per_step": elapsed / n_steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic 48 kHz mono (one 600 s seeded pattern, rolled per stream and tiled), random-init NSNet2 weights seed 7",
            "config": {"workload": f"BASELINE config 4: {n_streams} streams x {seconds} s, whole streams dealt round-robin to "
                                   f"{world} rank(s) ({[len(pkg.shard.streams_for_rank(n_streams, r, world)) for r in range(world)]}), "
                                   "per-rank GPU path + host VAD + Evaluator statistics, one all-gather, plan-order aggregate",
                       "streams": n_streams, "seconds_per_stream": seconds,
                       "parallelism": f"streams sharded over {world} rank(s), no data-path collective"},
            "audio_seconds_per_s": frames * n_steps / elapsed / 100.0,
            "aggregate": {"n_streams": n_streams, "tpr": agg.true_positive_rate.overall, "ppv": agg.precision.overall,
                          "collective": collective, "rccl_ranks": rccl_ranks,
                          # the gathered per-stream SingleStats in plan order and the aggregate formed from them, as bytes:
                          # equal digests from 1 rank and from N ranks = "the same report" (statistics.zig:116-172 sums in slice order)
                          "stats_sha256": stats_digest(allst, agg)},
            "rank0_step_s": timing,
        }
        if not emit:
            return out
        print(json.dumps(out))
    return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _code_only(text):
    """the source without // comments, indentation and blank lines: a comment edit is not a kernel change"""
    out = []
    for line in text.splitlines():
        i = line.find("//")
        if i >= 0 and line[:i].count('"') % 2 == 0:
            line = line[:i]
        line = line.strip()
        if line:
            out.append(line)
    return "\n".join(out).encode()


def kernel_source_digest():
    """sha256 over the kernel sources (code only: _code_only): profiles/*_pmc_summary.json records the digest of the build
    its counters were taken from, so a stale profile is visible in the line instead of being quoted silently"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "formula-vad_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(_code_only(open(os.path.join(d, name), "r", errors="replace").read()))
    return h.hexdigest()[:16]


def pmc_traffic(chunks_per_launch, nn_math="f32"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_summary.json; counters cannot be read from inside the benchmark) together with where they
    come from: file, the commit and kernel-source digest recorded when the passes were summarised, and whether
    that digest is the current build's.  None when no profile holds this kernel at this launch size."""
    want = "gru_rec_h3_kernel" if nn_math == "f16x3" else "gru_rec3_kernel"
    ts3 = nn_math == "bf16x3"            # its recurrence is the instance that also writes three-piece fragments
    try:
        pdir = os.path.join(ROOT, "profiles")
        files = sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_summary.json")),
                       key=lambda f: os.path.getmtime(os.path.join(pdir, f)))
        cands = []
        for name in files:
            d = json.load(open(os.path.join(pdir, name)))
            if d["chunks_per_launch"] != min(chunks_per_launch, 49152):
                continue
            for k, v in d["kernels"].items():
                if k.startswith(want) and (nn_math == "f16x3" or k.rstrip(">").endswith("true") == ts3):
                    cands.append((d.get("recorded_at", ""), name, k, v, d))
        if not cands:
            return None
        # the newest summary by its own timestamp (file mtimes do not survive a checkout); unstamped ones are oldest
        _, name, k, v, d = sorted(cands, key=lambda c: (c[0], c[1]))[-1]
        digest = d.get("kernel_source_digest")
        return {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                "hbm_bytes_basis": v.get("hbm_bytes_basis", "2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction)"),
                "hbm_bytes_by_request_size": v.get("hbm_bytes_by_request_size"),
                "hbm_bytes_fetch_x2_plus_write": v.get("hbm_bytes_fetch_x2_plus_write", v["hbm_bytes_per_launch"]),
                "hbm_bytes_per_launch_raw_counters": v.get("hbm_bytes_per_launch_raw_counters"),
                "mfma_busy_frac": v.get("mfma_busy_frac"), "clock_GHz": v.get("clock_GHz"),
                "kernel": k, "profile": "profiles/" + name, "profile_commit": d.get("source_commit"),
                "profile_kernel_source_digest": digest, "current_kernel_source_digest": kernel_source_digest(),
                "stale": None if digest is None else digest != kernel_source_digest()}
    except Exception as e:
        return {"error": repr(e)}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run,
    one rank per GPU over RCCL; this process never touches a GPU), relay their output -- rank 0's JSON line included
    -- and exit with the launcher's code.  One worker per stream / file joined before the report is the reference's
    own parallelism (src/simulator.zig:221-232)."""
    import socket
    import subprocess
    if args.dist_backend == "nccl":
        import torch
        have = torch.cuda.device_count()           # counts devices without initialising the runtime
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node shows {have} "
                  f"(use --dist-backend gloo to rehearse several ranks on one GPU)", file=sys.stderr)
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"bench.py: launching {args.gpus} ranks: {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))               # before anything touches the GPU in this process
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    import torch
    import torch.distributed as dist
    rehearsal = args.dist_backend == "gloo"
    if rehearsal:
        local_rank = 0                       # several ranks share the one GPU of a rehearsal box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cdev = torch.device("cpu") if rehearsal else dev  # where the tiny collectives' tensors live

    pkg = load_package()
    fv = pkg.binding
    import ctypes as C
    L = fv.lib()
    ctx = fv.Context(local_rank)
    ctx.load_synth(7)
    ctx.set_nn_math(args.nn_math)
    nn_math = ctx.nn_math_effective()            # what the context really uses (FVAD_NN_MATH at create overrides the setting)
    weights = ctx.weights()

    if args.config == "cfg4":
        run_cfg4(args, pkg, fv, ctx, torch, dist, rank, world, cdev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        ctx.close()
        return

    lanes, seconds = args.lanes, args.seconds
    n_chunks = seconds * 48000 // CHUNK
    n_frames_fft = (n_chunks * CHUNK) // 1024
    frames_per_step = lanes * n_chunks * FRAMES_PER_CHUNK
    host_pcm, labels = make_inputs(pkg, rank, lanes, seconds)
    # device buffers come from the library's own allocator (fvad_device_alloc): no torch on the data path
    n_samp = seconds * 48000
    d_pcm = ctx.device_alloc(lanes * n_samp * 4)
    for lane in range(lanes):                            # inputs resident in HBM before timing
        ctx.to_device(d_pcm + lane * n_samp * 4, host_pcm[lane])
    d_den = ctx.device_alloc(lanes * n_chunks * CHUNK * 4)
    d_band = [ctx.device_alloc(lanes * n_frames_fft * 4) for _ in range(2)]
    d_rms = [ctx.device_alloc(lanes * n_chunks * 4) for _ in range(2)]
    # page-locked (fvad_host_alloc): a device -> host copy into pageable memory would block the enqueueing thread
    # until everything queued before it has run
    h_band = [ctx.host_alloc(lanes * n_frames_fft).reshape(lanes, n_frames_fft) for _ in range(2)]
    h_rms = [ctx.host_alloc(lanes * n_chunks).reshape(lanes, n_chunks) for _ in range(2)]
    # a 1-GPU box gives this process a 16-CPU share whatever os.cpu_count() says
    vad_threads = args.vad_threads or min(lanes, 16, max(1, (os.cpu_count() or 2) - 1))

    results = {}
    host_ms = []

    vad_batch = fv.VadBatch(lanes)

    def host_stage(step, slot):
        """band sums + chunk RMS -> per-frame volume ratio -> VAD state machines -> segments: the library's
        batched host stage (fvad_vad_batch_run: VADMetadata chain + VADMachine.run per frame, streams dealt to
        threads), straight from the buffers the GPU stage filled"""
        t_h0 = time.perf_counter()
        results[step] = vad_batch.run(h_band[slot], h_rms[slot], n_threads=vad_threads)
        host_ms.append((time.perf_counter() - t_h0) * 1e3)

    eopts = fv.EngineOpts()
    L.fvad_engine_opts_default(C.byref(eopts))
    eopts.no_wait = 1          # return when queued: the next step is enqueued behind the running one

    def gpu_stage(slot):
        rc = L.fvad_engine_enqueue_device(ctx.h, d_pcm, lanes, n_samp, n_samp, d_den, d_band[slot], d_rms[slot], C.byref(eopts))
        fv.check(rc, "fvad_engine_enqueue_device", ctx.h)
        fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_band[slot].ctypes.data, d_band[slot], h_band[slot].nbytes),
                 "copy band sums", ctx.h)
        fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_rms[slot].ctypes.data, d_rms[slot], h_rms[slot].nbytes),
                 "copy rms", ctx.h)

    gpu_wall_ms = []
    join_ms = []
    # HIP events on the context's stream (fvad_ctx_stream): step i+1 is enqueued while step i is still running, so
    # the GPU does not idle between steps; the host waits for step i's event, not for the whole stream
    hip = C.CDLL("libamdhip64.so")
    L.fvad_ctx_stream.restype = C.c_void_p
    stream = C.c_void_p(L.fvad_ctx_stream(ctx.h))
    step_ev = [C.c_void_p(), C.c_void_p()]
    for e in step_ev:
        if hip.hipEventCreateWithFlags(C.byref(e), 0x2) != 0:      # hipEventDisableTiming
            raise RuntimeError("hipEventCreate failed")

    def run_steps(k, tag):
        """K steps; step i = GPU stage (kernels + D2H of band sums / RMS into slot i & 1) then the host stage on
        those buffers.  Order per iteration: wait for the host stage of step i-1 (it frees slot (i+1) & 1),
        enqueue step i+1, wait for step i's event, start its host stage."""
        worker = None
        t_g = time.perf_counter()
        if k > 0:
            gpu_stage(0)
            hip.hipEventRecord(step_ev[0], stream)
        for i in range(k):
            slot = i & 1
            t_j = time.perf_counter()
            if worker is not None:
                worker.join()                    # host stage of step i-1 (overlapped the GPU stage of i)
            join_ms.append((time.perf_counter() - t_j) * 1e3)
            if i + 1 < k:
                gpu_stage(slot ^ 1)
                hip.hipEventRecord(step_ev[slot ^ 1], stream)
            if hip.hipEventSynchronize(step_ev[slot]) != 0:      # band sums of step i are on the host
                raise RuntimeError("hipEventSynchronize failed")
            now = time.perf_counter()
            gpu_wall_ms.append((now - t_g) * 1e3)    # completion to completion
            t_g = now
            worker = threading.Thread(target=host_stage, args=((tag, i), slot))
            worker.start()
        if worker is not None:
            worker.join()

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(math, tag):
        """W warm-up steps, then EXACTLY K timed steps (barrier + synchronize on both sides) of the whole path with the
        context's arithmetic set to `math`; HIP-event kernel times over the timed steps; the self-check of the last
        timed step; then the device-only rate of the same work (no host stage), for the record."""
        ctx.set_nn_math(math)
        eff = ctx.nn_math_effective()
        del gpu_wall_ms[:], join_ms[:], host_ms[:]
        run_steps(args.warmup, tag + "-warm")
        ctx.enable_timing(True)
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps, tag)
        barrier()
        elapsed = time.perf_counter() - t0
        ktimes = ctx.kernel_times()              # sums over the K timed steps, HIP events on ctx's stream
        ctx.enable_timing(False)
        path = ctx.last_nn_path()
        # self-check of the configuration that was just timed: two lanes of the LAST timed step (the first and
        # the last of the batch) against the CPU oracle -- denoised audio, band sums, segment boundaries
        check = None
        if rank == 0:
            check = check_against_oracle(ctx, weights, host_pcm, [0, lanes - 1], d_den, n_samp,
                                         h_band[(args.steps - 1) & 1], results[(tag, args.steps - 1)])
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            gpu_stage(i & 1)
        barrier()
        dev_elapsed = time.perf_counter() - t1
        t = torch.tensor([elapsed, dev_elapsed], dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {"nn_math": eff, "elapsed": float(t[0]), "dev_elapsed": float(t[1]), "rank_elapsed": elapsed, "ktimes": ktimes,
                "self_check": check, "nn_path": path, "gpu_wall_ms": float(np.mean(gpu_wall_ms[args.warmup:])),
                "join_ms": [round(j, 1) for j in join_ms], "host_ms": float(np.mean(host_ms)) if host_ms else 0.0}

    def roofline_of(m):
        """dominant kernel: the GRU recurrence (two launches per step).  achieved = the MFMA FLOPs one launch executes /
        its mean HIP-event time; peak = the dense MFMA peak of the instruction it issues"""
        h3 = m["nn_math"] == "f16x3"
        kt = m["ktimes"]
        gru_ms = (kt.get("gru1_rec", 0.0) + kt.get("gru2_rec", 0.0)) / (2 * args.steps)
        gru_flop = lanes * n_chunks * GRU_FLOP_PER_CHUNK_LAUNCH
        gru_mfma_flop = lanes * n_chunks * (GRU_H3_MFMA_FLOP_PER_CHUNK_LAUNCH if h3 else GRU_FLOP_PER_CHUNK_LAUNCH)
        peak = F16_MFMA_PEAK_TFLOPS if h3 else FP32_MFMA_PEAK_TFLOPS
        achieved = gru_mfma_flop / (gru_ms * 1e-3) / 1e12 if gru_ms > 0 else 0.0
        # gi read once; h written and read back once: as f32 rows (f32 kernels) or as split fragments (f16x3: 416 slots x 4 B)
        hbm = lanes * n_chunks * (54 * 1200 * 4 + 2 * 54 * (416 if h3 else 400) * 4)
        pmc = pmc_traffic(lanes * n_chunks, m["nn_math"])
        r = {"bound": "mfma",
             "kernel": ("gru_rec_h3_kernel<12, 1> (v_mfma_f32_16x16x32_f16, three per f32 product)" if h3 else
                        "gru_rec3_kernel<12, 2, true> (fp32 v_mfma_f32_16x16x4_f32; also writes h as three bf16 pieces)"
                        if m["nn_math"] == "bf16x3" else "gru_rec3_kernel<12, 2, false> (fp32 v_mfma_f32_16x16x4_f32)"),
             "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
             # HBM bytes per launch from the PMC passes.  Three figures, and which one is believed: `traffic` is the sum of request
             # counts x request sizes (TCC_EA0_RDREQ_32B / 64B / 128B, WRREQ_64B: no correction) when the profile has those passes;
             # 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction) agrees with it to three digits on every kernel here,
             # because they issue 128-byte read requests almost exclusively and FETCH_SIZE tallies each at 64; the raw FETCH_SIZE +
             # WRITE_SIZE under-counts the reads by half (tools/fetch_calib.hip under the same passes: known byte counts)
             "traffic": pmc.get("hbm_bytes_per_launch") if pmc else None,
             "traffic_basis": pmc.get("hbm_bytes_basis") if pmc else None,
             "traffic_by_request_size": pmc.get("hbm_bytes_by_request_size") if pmc else None,
             "traffic_fetch_x2_plus_write": pmc.get("hbm_bytes_fetch_x2_plus_write") if pmc else None,
             "traffic_uncorrected_counters": pmc.get("hbm_bytes_per_launch_raw_counters") if pmc else None,
             # plain scalars beside the figure, so a record that keeps only `roofline`'s top level still says where the bytes
             # come from and whether the kernels have changed since they were counted
             "traffic_profile": pmc.get("profile") if pmc else None,
             "traffic_stale": pmc.get("stale") if pmc else None,
             "traffic_source": pmc,
             "algorithmic_hbm_bytes_per_launch": hbm,
             "algorithmic_hbm_frac_of_8TBps": hbm / (gru_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if gru_ms > 0 else 0.0,
             "launch_ms": gru_ms, "flop_per_launch": gru_mfma_flop}
        if h3:
            # useful (algorithmic) FLOPs of the same launch: 2 x 400 x 1200 per step and sequence
            r["algorithmic_tflops"] = gru_flop / (gru_ms * 1e-3) / 1e12 if gru_ms > 0 else 0.0
            r["algorithmic_frac_of_f16_peak"] = r["algorithmic_tflops"] / F16_MFMA_PEAK_TFLOPS
            r["f32_equivalent_frac_of_f32_mfma_peak"] = r["algorithmic_tflops"] / FP32_MFMA_PEAK_TFLOPS
        return r

    def pipeline_of(m):
        kt = m["ktimes"]
        dev_ms_step = sum(kt.values()) / args.steps
        nn_ms = sum(v for k, v in kt.items() if "gemm" in k or "gru" in k) / args.steps
        c = lanes * n_chunks
        return {"nn_math": m["nn_math"], "nn_path": m["nn_path"],
                "nsnet2_executed_f32_equivalent_tflops": c * NSNET2_EXECUTED_FLOP_PER_CHUNK / (nn_ms * 1e-3) / 1e12 if nn_ms else 0.0,
                "nsnet2_f32_equivalent_frac_of_f32_mfma_peak": c * NSNET2_EXECUTED_FLOP_PER_CHUNK / (nn_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS if nn_ms else 0.0,
                "nsnet2_effective_tflops_unfolded_network": c * NSNET2_FLOP_PER_CHUNK / (nn_ms * 1e-3) / 1e12 if nn_ms else 0.0,
                "hbm_algorithmic_GBps": frames_per_step * 3840 / (dev_ms_step * 1e-3) / 1e9 if dev_ms_step else 0.0,
                "hbm_frac_of_8TBps": frames_per_step * 3840 / (dev_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS if dev_ms_step else 0.0,
                "kernel_ms_sum": dev_ms_step}

    head = measure(nn_math, "timed")
    if head["nn_math"] != nn_math:
        raise RuntimeError(f"asked for {nn_math}, the context runs {head['nn_math']}")
    # ---- the emulated arithmetic on the same batch, first-class: the same W + K steps end to end, its own roofline
    # and self-check.  It is NOT the headline: f16x3 carries 22 significand bits per operand, the reference's f32 24.
    emulated = emulated24 = None
    if nn_math == "f32" and not args.no_emulated:
        em = measure("f16x3", "emulated")
        if em["nn_math"] == "f16x3":
            emulated = {"nn_math": "f16x3", "note": "three v_mfma_f32_16x16x32_f16 per product on (hi, lo) f16 pieces of power-of-two "
                        "scaled f32 operands, f32 accumulation: 22 significand bits per operand (f32: 24) -- narrower than the "
                        "reference's arithmetic, hence not `value`; opt-in through fvad_ctx_set_nn_math",
                        "value": frames_per_step * args.steps * world / em["elapsed"], "unit": "frames/s",
                        "ms_per_step": em["elapsed"] / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                        "device_only_frames_per_s": frames_per_step * args.steps * world / em["dev_elapsed"],
                        "roofline": roofline_of(em), "roofline_pipeline": pipeline_of(em),
                        "kernel_ms_per_step": {k: v / args.steps for k, v in em["ktimes"].items()},
                        "self_check": em["self_check"], "host_stage_ms": em["host_ms"]}
        # ... and the emulation that is NOT narrower than f32: the five dense layers as six bf16 MFMAs per product on exact
        # three-piece splits of both operands (24 significand bits, f32's exponent range), the recurrences on f32 MFMA
        em24 = measure("bf16x3", "emulated24")
        if em24["nn_math"] == "bf16x3":
            emulated24 = {"nn_math": "bf16x3", "note": "dense layers: six v_mfma_f32_16x16x32_bf16 per product on exact three-piece bf16 splits of "
                          "both f32 operands (x = h + m + l: all 24 significand bits, no scales; the three dropped cross terms are <= 2^-23 "
                          "of a product, one f32 rounding), f32 accumulation; GRU recurrences on v_mfma_f32_16x16x4_f32.  Opt-in through "
                          "fvad_ctx_set_nn_math; reported beside the headline, which stays on plain f32",
                          "value": frames_per_step * args.steps * world / em24["elapsed"], "unit": "frames/s",
                          "ms_per_step": em24["elapsed"] / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
                          "device_only_frames_per_s": frames_per_step * args.steps * world / em24["dev_elapsed"],
                          "roofline": roofline_of(em24), "roofline_pipeline": pipeline_of(em24),
                          "kernel_ms_per_step": {k: v / args.steps for k, v in em24["ktimes"].items()},
                          "self_check": em24["self_check"], "host_stage_ms": em24["host_ms"]}
        ctx.set_nn_math(nn_math)
    elapsed, dev_elapsed, ktimes, self_check = head["elapsed"], head["dev_elapsed"], head["ktimes"], head["self_check"]

    # ---- the same batch read as 192 STEREO streams (the reference's real corpus is stereo: channel_vol_ratio):
    # the GPU work is identical (a lane is a channel), the host stage runs 2-channel state machines on
    # min-over-channels band sums and the per-chunk RMS ratio (BufferedVolumeAnalyzer.zig:48-69)
    stereo = None
    if rank == 0 and lanes % 2 == 0:
        t_s0 = time.perf_counter()
        gpu_stage(0)
        ctx.synchronize()
        t_s1 = time.perf_counter()
        vb2 = fv.VadBatch(lanes // 2, n_channels=2)     # lane = stream * 2 + channel: the same buffers, read as pairs
        n_seg2 = sum(len(x) for x in vb2.run(h_band[0], h_rms[0], n_threads=vad_threads))
        vb2.close()
        t_s2 = time.perf_counter()
        stereo = {"streams": lanes // 2, "channels": 2, "gpu_stage_ms": (t_s1 - t_s0) * 1e3, "host_stage_ms": (t_s2 - t_s1) * 1e3,
                  "frames_per_s_not_overlapped": frames_per_step / (t_s2 - t_s0), "segments": n_seg2,
                  "note": "one step, host stage after the GPU stage (in the timed loop they overlap); frames = channel-frames"}

    # ---- final Evaluator aggregate: per-stream SingleStats -> all_gather (RCCL) -> ordered aggregate
    stat_cfg = {"ignore_shorter_than_sec": 0.7, "extrude_start": 5.0, "extrude_end": 10.0, "fill_gaps": 5.0}
    last = results[("timed", args.steps - 1)]
    local_ids = [rank + world * lane for lane in range(lanes)]   # round-robin plan order
    local_stats = []
    for lane in range(lanes):
        segs = [(np.float32(s[0]) / np.float32(48000), np.float32(s[1]) / np.float32(48000)) for s in last[lane]]
        local_stats.append(fv.stats_from_segments(segs, labels[lane], stat_cfg))
    ta = time.perf_counter()
    allst, collective, rccl_ranks = gather_all_stats(pkg, fv, ctx, dist, world, args, cdev, local_ids, local_stats, lanes * world)
    agg = fv.stats_aggregate([fv.array_to_single_stats(a) for a in allst])
    agg_ms = (time.perf_counter() - ta) * 1e3

    # per-rank step times and the communicator's size, for the record of an N-rank run
    rank_ms = [None] * world
    if world > 1:
        dist.all_gather_object(rank_ms, head["rank_elapsed"] / args.steps * 1e3)
    else:
        rank_ms = [head["rank_elapsed"] / args.steps * 1e3]

    # ---- N > 1: the same ranks also run BASELINE config 4's plan (21 x 7200 s streams dealt round-robin, the RCCL
    # statistics gather, the plan-order aggregate) -- a FIXED amount of work, so one driver invocation per N records weak
    # scaling (the headline: every rank the same batch) and strong scaling (`cfg4_strong`) side by side
    cfg4_strong = None
    if world > 1 and not args.no_cfg4_strong:
        try:
            cfg4_strong = run_cfg4(args, pkg, fv, ctx, torch, dist, rank, world, cdev, steps=2, warmup=1, emit=False)
        except Exception as e:  # every rank takes the same path: an exception here is an allocation or a library error
            cfg4_strong = {"error": repr(e)}

    if rank == 0:
        total_frames = frames_per_step * args.steps * world
        value = total_frames / elapsed
        h3 = nn_math == "f16x3"
        out = {
            "metric": "20ms audio frames/sec end-to-end VAD pipeline",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f16x3 EMULATION of f32 (22-bit operands: narrower than the reference's f32)" if h3 else
                      "bf16x3 emulation of the f32 GEMMs (24-bit operands, six cross terms) + f32 recurrences" if nn_math == "bf16x3" else "f32"),
            "nn_math_effective": head["nn_math"], "nn_path": head["nn_path"],
            "data": "synthetic 48 kHz mono (seeded noise floor + 500-2000 Hz harmonic bursts; every lane a different stream), random-init NSNet2 weights seed 7",
            "config": {"workload": f"full pipeline (window->STFT->NSNet2->iSTFT->FFT1024 band->VAD decision), "
                                   f"{lanes} streams x {seconds} s per GPU per step = {lanes * n_chunks} chunks = "
                                   f"{frames_per_step} frames; BASELINE config 3 pipeline at a saturating batch "
                                   f"(config 3's own 82-chunk batch is in extra.cfg3)",
                       "streams_per_gpu": lanes, "seconds_per_stream": seconds,
                       "frame": "10 ms hop / 20 ms window @16 kHz (NSNet2.zig:12-13)",
                       "parallelism": f"streams sharded over {world} GPU(s), no data-path collective"},
            "ranks": {"launched": world, "backend": args.dist_backend if world > 1 else "none", "rccl_ranks": rccl_ranks,
                      "ms_per_step_per_rank": rank_ms},
            "audio_seconds_per_s": value / 100.0,
            "device_only_frames_per_s": total_frames / dev_elapsed,
            "roofline": roofline_of(head),
            "roofline_pipeline": pipeline_of(head),
            "kernel_ms_per_step": {k: v / args.steps for k, v in ktimes.items()},
            "aggregate": {"ms": agg_ms, "n_streams": lanes * world, "tpr": agg.true_positive_rate.overall,
                          "ppv": agg.precision.overall, "collective": collective, "stats_sha256": stats_digest(allst, agg)},
            "self_check": self_check,
            "parity_note": "parity unpinned: the checker is oracle/, a CPU restatement of the reference; the reference holds no "
                           "golden vector for FFT / NSNet2 / band sums / segments and cannot be built here (DESIGN.md section 4)",
            "emulated": emulated,
            "emulated24": emulated24,
            "headline_as_stereo": stereo,
            "host_vad_threads": vad_threads,
            "gpu_stage_wall_ms": head["gpu_wall_ms"],
            "kernel_ms_sum": float(sum(ktimes.values()) / args.steps),
            "join_wait_ms": head["join_ms"],
            "host_stage_ms": head["host_ms"],
        }
        if cfg4_strong is not None:
            out["cfg4_strong"] = cfg4_strong
        # the CPU baseline and the side measurements belong to the single-GPU run only
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, fv, weights, min(os.cpu_count() or 1, 16))
        if not args.no_extra and world == 1:
            out["extra"] = side_measurements(pkg, fv, ctx, torch, dev)
        print(json.dumps(out))
        if not self_check or not self_check["ok"] or (emulated and not emulated["self_check"]["ok"]) or \
                (emulated24 and not emulated24["self_check"]["ok"]):
            print("bench.py: SELF-CHECK FAILED: the timed configuration does not match the oracle", file=sys.stderr)
            sys.exit(1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    torch.cuda.synchronize()
    ctx.close()


# (5120 and 20480 chunks sit just above a size that fills the chip with one round of workgroups: the launch planner's cases)
BATCH_CURVE_POINTS = ((1, 1), (8, 1), (2, 41), (16, 16), (64, 16), (128, 16), (128, 32), (128, 40), (128, 64), (128, 96), (128, 128), (128, 160),
                      (384, 128))


def batch_curve(fv, ctx, host_pcm, points=BATCH_CURVE_POINTS, budget_s=0.4):
    """frames/s of the device-resident path (kernels + D2H of band sums and RMS, f32, denoised audio materialised) against
    the batch of one call: (lanes, chunks per lane) per point, calls back to back, the mean over enough calls to fill
    `budget_s`; `nn_path` says which kernels ran.  The smallest batch that sustains 1e7 frames/s is reported beside it."""
    import ctypes as C
    L = fv.lib()
    out = []
    src = np.concatenate([np.asarray(x, np.float32) for x in host_pcm[:2]])
    for n_l, n_ch in points:
        n = n_ch * CHUNK
        nfr = max(1, n // 1024)
        lane = np.resize(src, n).astype(np.float32)
        d_in = ctx.device_alloc(n_l * n * 4)
        d_den = ctx.device_alloc(n_l * n * 4)
        d_band = ctx.device_alloc(n_l * nfr * 4)
        d_rms = ctx.device_alloc(n_l * n_ch * 4)
        h_band = np.empty((n_l, nfr), np.float32)
        h_rms = np.empty((n_l, n_ch), np.float32)
        try:
            for i in range(n_l):
                ctx.to_device(d_in + i * n * 4, np.roll(lane, 977 * i))

            def call():
                fv.check(L.fvad_engine_enqueue_device(ctx.h, C.c_void_p(d_in), n_l, n, n, C.c_void_p(d_den), C.c_void_p(d_band), C.c_void_p(d_rms), None),
                         "batch curve", ctx.h)
                fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_band.ctypes.data, C.c_void_p(d_band), h_band.nbytes), "batch curve band", ctx.h)
                fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_rms.ctypes.data, C.c_void_p(d_rms), h_rms.nbytes), "batch curve rms", ctx.h)
                ctx.synchronize()
            call()
            t0 = time.perf_counter()
            call()
            one = time.perf_counter() - t0
            reps = int(max(3, min(200, budget_s / max(one, 1e-6))))
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            dt = (time.perf_counter() - t0) / reps
            chunks = n_l * n_ch
            out.append({"chunks": chunks, "lanes": n_l, "frames": chunks * FRAMES_PER_CHUNK, "ms": dt * 1e3,
                        "frames_per_s": chunks * FRAMES_PER_CHUNK / dt, "nn_path": ctx.last_nn_path(), "calls": reps})
        finally:
            for d in (d_in, d_den, d_band, d_rms):
                ctx.device_free(d)
    ok = [p["chunks"] for p in out if p["frames_per_s"] >= 1e7]
    # "sustains": this point and every larger one measured is at or above 1e7
    sustained = None
    for p in sorted(out, key=lambda p: -p["chunks"]):
        if p["frames_per_s"] >= 1e7:
            sustained = p["chunks"]
        else:
            break
    return {"points": out, "smallest_batch_at_1e7_frames_per_s": min(ok) if ok else None,
            "smallest_batch_from_which_every_larger_point_is_at_1e7": sustained,
            "note": "per call of fvad_engine_enqueue_device + D2H of band sums / RMS + synchronize, calls back to back (host-paired wall time)"}


def side_measurements(pkg, fv, ctx, torch, dev):
    note = lambda m: print(f"bench.py: side measurement: {m}", file=sys.stderr, flush=True)  # noqa: E731
    """BASELINE config 2 (FFT isolation, 1024 frames and 2^20 frames) and config 3 at its literal
    batch (82 chunks = 4100 frames), device-resident, timed with HIP events."""
    import ctypes as C
    L = fv.lib()
    extra = {}
    f = fv.FFT(ctx, 320, 16000)
    win = torch.from_numpy(np.ascontiguousarray(__import__("numpy").sqrt(
        0.5 - 0.5 * np.cos(2 * np.pi * np.arange(320) / 319)).astype(np.float32))).to(dev)
    note("config 2 (FFT isolation)")
    for n in (1024, 1 << 20):
        x = torch.rand((n, 320), device=dev) * 2 - 1
        mag = torch.empty((n, 161), device=dev)
        torch.cuda.synchronize()
        reps = 200 if n == 1024 else 20
        L.fvad_fft_forward_batch(f.h, x.data_ptr(), n, win.data_ptr(), None, mag.data_ptr(), 1)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            L.fvad_fft_forward_batch(f.h, x.data_ptr(), n, win.data_ptr(), None, mag.data_ptr(), 1)
        ctx.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / reps   # back-to-back launches on the context's stream
        del x, mag
        extra[f"cfg2_fft320_{n}_frames"] = {"us_per_launch": ms * 1e3, "frames_per_s": n / (ms * 1e-3),
                                            "hbm_GBps": n * 1924 / (ms * 1e-3) / 1e9,
                                            "hbm_frac_of_8TBps": n * 1924 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    f.close()
    # config 3: 82 chunks (2 lanes x 41)
    pcm = np.stack([pkg.synth.make_stream(20.5, seed=30 + i)[0][0][: 41 * CHUNK] for i in range(2)])
    d = torch.from_numpy(pcm).to(dev)
    band = torch.empty((2, 41 * CHUNK // 1024), device=dev)
    rms = torch.empty((2, 41), device=dev)
    torch.cuda.synchronize()
    # 60 untimed calls (25 ms: the section in front of this one is the 2^20-frame FFT, another power and clock state -- with 4
    # warm-up calls the same 100 calls measured 1.5 % slower than when repeated a second later), then 200 back to back
    CFG3_WARM, CFG3_CALLS = 60, 200
    for it in range(CFG3_WARM + CFG3_CALLS):
        if it == CFG3_WARM:
            ctx.synchronize()
            t0 = time.perf_counter()
        L.fvad_engine_enqueue_device(ctx.h, d.data_ptr(), 2, d.stride(0), 41 * CHUNK, None, band.data_ptr(), rms.data_ptr(), None)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / CFG3_CALLS
    cfg3_path = ctx.last_nn_path()
    cfg3_graph_ms = None
    try: # the same 100 calls replayed as a hipGraph (fvad_engine_opts.use_graph)
        go = fv.EngineOpts()
        L.fvad_engine_opts_default(C.byref(go))
        go.use_graph = 1
        for it in range(CFG3_WARM + CFG3_CALLS):
            if it == CFG3_WARM:
                ctx.synchronize()
                t0 = time.perf_counter()
            fv.check(L.fvad_engine_enqueue_device(ctx.h, d.data_ptr(), 2, d.stride(0), 41 * CHUNK, None, band.data_ptr(), rms.data_ptr(), C.byref(go)),
                     "config 3 graph replay", ctx.h)
        ctx.synchronize()
        cfg3_graph_ms = (time.perf_counter() - t0) / CFG3_CALLS * 1e3
    except Exception as e:
        cfg3_graph_ms = repr(e)
    # the same 100 calls after the context option ws2_calibrate measured the first-poll waits on this device (include/fvad.h)
    cfg3_cal = None
    try:
        ctx.set_option("ws2_calibrate", 1)
        for it in range(CFG3_WARM + CFG3_CALLS):
            if it == CFG3_WARM:
                ctx.synchronize()
                t0 = time.perf_counter()
            L.fvad_engine_enqueue_device(ctx.h, d.data_ptr(), 2, d.stride(0), 41 * CHUNK, None, band.data_ptr(), rms.data_ptr(), None)
        ctx.synchronize()
        cfg3_cal = {"ms": (time.perf_counter() - t0) / CFG3_CALLS * 1e3,
                    "waits_10ns_ticks": {"groups_25_25": list(ctx.ws2_waits(1)), "groups_13_25_gi1_in_kernel": list(ctx.ws2_waits(3))}}
        ctx.set_option("ws2_calibrate", 0)
        cfg3_cal["table"] = {"groups_25_25": list(ctx.ws2_waits(1)), "groups_13_25_gi1_in_kernel": list(ctx.ws2_waits(3))}
    except Exception as e:
        cfg3_cal = {"error": repr(e)}
    # host-buffer entry point (what AudioPipeline.pushSamples hands over): H2D of the 48 kHz input, the
    # kernels, D2H of band sums / RMS (and of the denoised audio in the second figure).  Pageable numpy
    # buffers, staged by the library; never `value`.
    note("host-buffer entry point (PCIe-inclusive)")
    n_l, n_s = 128, 64
    import ctypes as C
    src = [pkg.synth.make_stream(n_s + 0.5, seed=500 + i)[0][0][: n_s * 48000].copy() for i in range(4)]
    host_pcm = [src[i % 4].copy() for i in range(n_l)]          # distinct, touched buffers like a caller's
    n_ch = n_s * 2
    cap = (n_ch * CHUNK + 1024) // 1024 + 1
    h_b = np.ones((n_l, cap), np.float32)                       # outputs allocated and touched once, reused
    h_r = np.ones((n_l, n_ch), np.float32)
    h_d = np.ones((n_l, n_ch * CHUNK), np.float32)
    for name, den in (("pcie_inclusive_no_denoised_d2h", False), ("pcie_inclusive_with_denoised_d2h", True)):
        arr = (fv.Lane * n_l)()
        for i in range(n_l):
            a = arr[i]
            a.pcm = fv.fptr(host_pcm[i]); a.n_samples = host_pcm[i].shape[0]; a.state = None
            a.denoised = fv.fptr(h_d[i]) if den else None
            a.band_sum = fv.fptr(h_b[i]); a.band_sum_capacity = cap
            a.chunk_rms = fv.fptr(h_r[i]); a.chunk_rms_capacity = n_ch
            a.fft_bins = None
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            fv.check(L.fvad_engine_run(ctx.h, arr, n_l, None), "fvad_engine_run", ctx.h)
            dt_p = time.perf_counter() - t0
            best = dt_p if best is None else min(best, dt_p)
        fr = n_l * n_s * 100
        extra[name] = {"frames_per_s": fr / best, "ms": best * 1e3, "streams": n_l, "seconds_per_stream": n_s,
                       "host_link_GBps": fr * (3840 if den else 1920) / best / 1e9,
                       "note": "fvad_engine_run on pageable host buffers (best of 3): staged H2D, kernels and D2H pipelined over lane groups (sizes by the formats' rates: csrc/engine_run.cpp plan_groups)"}
    # 16-bit transport: the same streams handed over as PCM16 (converted by the kernel that reads them) and, in the
    # second figure, the denoised audio taken back as PCM16 too
    try:
        pcm16 = [np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16) for x in src]
        host16 = [pcm16[i % 4].copy() for i in range(n_l)]
        h_q = np.ones((n_l, n_ch * CHUNK), np.int16)
        for name, den in (("pcie_inclusive_i16_no_denoised_d2h", False), ("pcie_inclusive_i16_with_denoised_i16_d2h", True)):
            arr = (fv.Lane * n_l)()
            for i in range(n_l):
                a = arr[i]
                a.pcm = None
                a.pcm_i16 = host16[i].ctypes.data_as(C.POINTER(C.c_int16)); a.n_samples = host16[i].shape[0]; a.state = None
                a.denoised_i16 = h_q[i].ctypes.data_as(C.POINTER(C.c_int16)) if den else None
                a.band_sum = fv.fptr(h_b[i]); a.band_sum_capacity = cap
                a.chunk_rms = fv.fptr(h_r[i]); a.chunk_rms_capacity = n_ch
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                fv.check(L.fvad_engine_run(ctx.h, arr, n_l, None), "fvad_engine_run i16", ctx.h)
                dt_p = time.perf_counter() - t0
                best = dt_p if best is None else min(best, dt_p)
            fr = n_l * n_s * 100
            extra[name] = {"frames_per_s": fr / best, "ms": best * 1e3, "host_link_GBps": fr * (1920 if den else 960) / best / 1e9,
                           "note": "fvad_engine_run on pageable PCM16 buffers (best of 3): 960 B per frame in, 960 B per frame of denoised PCM16 back"}
        del host16, h_q
    except Exception as e:
        extra["pcie_inclusive_i16"] = {"error": repr(e)}
    # the same call on page-locked buffers from fvad_host_alloc: no staging, the DMA engine works in place
    try:
        n_samp = n_s * 48000
        pin_in, pin_den = C.c_void_p(), C.c_void_p()
        fv.check(L.fvad_host_alloc(ctx.h, n_l * n_samp * 4, C.byref(pin_in)), "fvad_host_alloc", ctx.h)
        fv.check(L.fvad_host_alloc(ctx.h, n_l * n_samp * 4, C.byref(pin_den)), "fvad_host_alloc", ctx.h)
        a_in = np.ctypeslib.as_array(C.cast(pin_in, C.POINTER(C.c_float)), shape=(n_l, n_samp))
        a_den = np.ctypeslib.as_array(C.cast(pin_den, C.POINTER(C.c_float)), shape=(n_l, n_samp))
        for i in range(n_l):
            a_in[i] = host_pcm[i]
        for name, den in (("pcie_inclusive_pinned_no_denoised_d2h", False), ("pcie_inclusive_pinned_with_denoised_d2h", True)):
            arr = (fv.Lane * n_l)()
            for i in range(n_l):
                a = arr[i]
                a.pcm = fv.fptr(a_in[i]); a.n_samples = n_samp; a.state = None
                a.denoised = fv.fptr(a_den[i]) if den else None
                a.band_sum = fv.fptr(h_b[i]); a.band_sum_capacity = cap
                a.chunk_rms = fv.fptr(h_r[i]); a.chunk_rms_capacity = n_ch
                a.fft_bins = None
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                fv.check(L.fvad_engine_run(ctx.h, arr, n_l, None), "fvad_engine_run", ctx.h)
                dt_p = time.perf_counter() - t0
                best = dt_p if best is None else min(best, dt_p)
            fr = n_l * n_s * 100
            extra[name] = {"frames_per_s": fr / best, "ms": best * 1e3, "host_link_GBps": fr * (3840 if den else 1920) / best / 1e9,
                           "note": "fvad_engine_run on fvad_host_alloc buffers (best of 3)"}
        del a_in, a_den
        L.fvad_host_free(ctx.h, pin_in)
        L.fvad_host_free(ctx.h, pin_den)
    except Exception as e:
        extra["pcie_inclusive_pinned"] = {"error": repr(e)}
    # BASELINE config 4's shape on one GPU: 21 long streams (Miami-race sized, 7200 s each) end to end,
    # input resident in HBM: kernels, D2H of band sums / RMS, host VAD for all 21 streams
    note("config 4's corpus shape on one GPU")
    try:
        n_st, n_sec = 21, 7200
        base = torch.from_numpy(pkg.synth.make_stream(600.5, seed=900)[0][0][: 600 * 48000].copy()).to(dev)
        big = torch.empty((n_st, n_sec * 48000), dtype=torch.float32, device=dev)
        for i in range(n_st):
            big[i] = torch.roll(base, 4801 * i).repeat(n_sec // 600)
        nch = n_sec * 2
        nfr = nch * CHUNK // 1024
        b4 = torch.empty((n_st, nfr), dtype=torch.float32, device=dev)
        r4 = torch.empty((n_st, nch), dtype=torch.float32, device=dev)
        hb = np.empty((n_st, nfr), np.float32)
        hr = np.empty((n_st, nch), np.float32)
        # untimed first pass: lets the library grow its 29 GB denoised-audio scratch
        fv.check(L.fvad_engine_enqueue_device(ctx.h, big.data_ptr(), n_st, big.stride(0), n_sec * 48000, None,
                                              b4.data_ptr(), r4.data_ptr(), None), "cfg4 warm-up", ctx.h)
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fv.check(L.fvad_engine_enqueue_device(ctx.h, big.data_ptr(), n_st, big.stride(0), n_sec * 48000, None,
                                              b4.data_ptr(), r4.data_ptr(), None), "cfg4 enqueue", ctx.h)
        fv.check(L.fvad_ctx_copy_to_host(ctx.h, hb.ctypes.data, b4.data_ptr(), hb.nbytes), "cfg4 band", ctx.h)
        fv.check(L.fvad_ctx_copy_to_host(ctx.h, hr.ctypes.data, r4.data_ptr(), hr.nbytes), "cfg4 rms", ctx.h)
        ctx.synchronize()
        t_gpu = time.perf_counter() - t0
        fs = np.arange(nfr) * 1024
        c0, c1 = fs // CHUNK, (fs + 1023) // CHUNK
        ww0 = (np.minimum((c0 + 1) * CHUNK, fs + 1024) - fs).astype(np.float32)
        ww1 = np.float32(1024) - ww0
        rc = np.where(hr > 0, np.where(hr < 1, 1.0, 1.0 / np.maximum(hr, 1e-30)), 0.0).astype(np.float32)
        rat = ((rc[:, c0] * ww0 + np.where(ww1 > 0, rc[:, c1] * ww1, np.float32(0))) / (ww0 + ww1)).astype(np.float32)
        ms4 = [fv.VadMachine() for _ in range(n_st)]
        fv.vad_run_many(ms4, [hb[i][:, None] for i in range(n_st)], [rat[i] for i in range(n_st)], n_threads=16)
        n_seg = sum(len(m.segments()) for m in ms4)
        stats = [m.lazy_stats() for m in ms4]
        for m in ms4:
            m.close()
        t_all = time.perf_counter() - t0
        extra["cfg4_shape_21_streams_x_7200s_one_gpu"] = {
            "frames_per_s": n_st * n_sec * 100 / t_all, "s_total": t_all, "s_gpu_incl_d2h": t_gpu,
            "s_host_vad_and_metadata": t_all - t_gpu, "segments": n_seg,
            "long_term_chain_evaluations": int(sum(a for a, _ in stats)), "long_term_pushes": int(sum(b for _, b in stats)),
            "note": "21 streams x 2 h on one GPU (the 8-GPU config gives each GPU 2-3 of them); host stage not overlapped"}
        # the same corpus as time slices with the host stage of slice k (fvad_vad_batch_run_part) beside the GPU's slice k + 1
        # (shard.run_sliced_with_vad: a slice is one 49152-chunk launch, 16 chunks of halo per lane and slice)
        try:
            vbs = fv.VadBatch(n_st)
            t0 = time.perf_counter()
            segs_ov, info = pkg.shard.run_sliced_with_vad(ctx, big.data_ptr(), n_st, big.stride(0), nch, vbs, n_threads=16)
            t_ov = time.perf_counter() - t0
            vbs.close()
            extra["cfg4_shape_21_streams_x_7200s_one_gpu"]["host_stage_overlapped"] = {
                "frames_per_s": n_st * n_sec * 100 / t_ov, "s_total": t_ov, "s_until_last_slice_on_host": info["gpu_s"],
                "s_host_tail": info["host_tail_s"], "slices": info["slices"], "chunks_per_slice_and_lane": info["slice_chunks"],
                "segments": sum(len(x) for x in segs_ov), "same_segment_count": sum(len(x) for x in segs_ov) == n_seg}
        except Exception as e:
            extra["cfg4_shape_21_streams_x_7200s_one_gpu"]["host_stage_overlapped"] = {"error": repr(e)}
        del big, b4, r4, base
    except Exception as e:  # e.g. not enough free HBM next to other tenants
        extra["cfg4_shape_21_streams_x_7200s_one_gpu"] = {"error": repr(e)}
    # BASELINE config 5's "hipGraph-captured steady-state loop": the same device-resident call launched
    # directly and replayed from a captured hipGraph (fvad_engine_opts.use_graph), at a large and at the smallest shape
    note("hipGraph replay")
    try:
        for tag, n_l2, n_ch2, reps in (("16384_chunks", 128, 128, 6), ("2_chunks", 2, 1, 40)):
            xg = torch.from_numpy(np.stack([host_pcm[i % len(host_pcm)][: n_ch2 * CHUNK] for i in range(n_l2)])).to(dev)
            bg = torch.empty((n_l2, max(1, n_ch2 * CHUNK // 1024)), device=dev)
            rg = torch.empty((n_l2, n_ch2), device=dev)
            dg = torch.empty((n_l2, n_ch2 * CHUNK), device=dev)
            res = {}
            for mode in ("direct", "graph"):
                go = fv.EngineOpts()
                L.fvad_engine_opts_default(C.byref(go))
                go.use_graph = 1 if mode == "graph" else 0
                for it in range(reps + 2):
                    if it == 2:
                        ctx.synchronize()
                        t0 = time.perf_counter()
                    fv.check(L.fvad_engine_enqueue_device(ctx.h, xg.data_ptr(), n_l2, xg.stride(0), n_ch2 * CHUNK, dg.data_ptr(),
                                                          bg.data_ptr(), rg.data_ptr(), C.byref(go)), "graph bench", ctx.h)
                ctx.synchronize()
                res[mode] = (time.perf_counter() - t0) / reps * 1e3
            extra[f"hipgraph_replay_{tag}"] = {"direct_ms": res["direct"], "graph_ms": res["graph"],
                                               "note": "per call of fvad_engine_enqueue_device; 9-12 kernel launches per launch batch"}
            del xg, bg, rg, dg
    except Exception as e:
        extra["hipgraph_replay"] = {"error": repr(e)}
    note("throughput against batch size")
    try:
        extra["batch_curve"] = batch_curve(fv, ctx, host_pcm)
    except Exception as e:
        extra["batch_curve"] = {"error": repr(e)}
    extra["cfg3_82_chunks_4100_frames"] = {"ms": dt * 1e3, "frames_per_s": 4100 / dt, "ms_as_hipgraph_replay": cfg3_graph_ms,
                                           "after_ws2_calibrate": cfg3_cal, "nn_path": cfg3_path,
                                           "note": "mean of 200 calls back to back after 60 untimed ones; latency-bound: 55 dependent, exchange-bound steps over only 82 sequences (gru_ws2k_kernel: both GRU layers in one launch, layer 2 a step behind layer 1, recurrent weights stationary in registers across 228 workgroups of 16 wavefronts, h exchanged per step)"}
    return extra


if __name__ == "__main__":
    main()
