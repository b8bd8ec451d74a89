"""Multi-GPU sharding of independent streams.

Streams never interact (one pipeline per file in the reference, src/simulator.zig:225-231), so
whole streams are dealt round-robin to ranks and no audio crosses GPUs.  The only exchange is the
final gather of the per-stream Evaluator statistics (11 f32 each, statistics.zig:8-37) to rank 0,
where statistics.aggregate runs in PLAN order to keep the reference's f32 summation order
(statistics.zig:124-129).  On GPUs the gather is the library's own ncclAllGather (fvad_stats_allgather,
csrc/comm.cpp: no PyTorch on that path, `native_comm` only borrows torch.distributed to hand the 128-byte
bootstrap id around); `gather_stats` is the same exchange over torch.distributed, used with gloo for CPU
tests and for rehearsals of several ranks on one GPU (RCCL refuses two ranks on one device)."""
import numpy as np

N_STAT = 11  # floats in SingleStats as laid out in include/fvad.h (f_score_beta included)


def streams_for_rank(n_streams, rank, world):
    """Round-robin: stream i -> rank i % world (21 streams over 8 GPUs -> 3,3,3,3,3,2,2,2)"""
    return [i for i in range(n_streams) if i % world == rank]


def gather_stats(local_ids, local_stats, n_streams, dist=None, device=None):
    """local_stats: [len(local_ids)][N_STAT] float32.  Returns [n_streams][N_STAT] in plan order on
    every rank (all_gather of a fixed-size block per rank; payload <= world*max_per_rank*44 B)."""
    import torch
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    max_per_rank = (n_streams + world - 1) // world
    block = torch.zeros((max_per_rank, N_STAT + 1), dtype=torch.float32)
    block[:, 0] = -1.0
    for j, (sid, st) in enumerate(zip(local_ids, local_stats)):
        block[j, 0] = float(sid)
        block[j, 1:] = torch.from_numpy(np.asarray(st, dtype=np.float32))
    if world == 1:
        blocks = [block]
    else:
        if device is not None:
            block = block.to(device)
        blocks = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(blocks, block)
        blocks = [b.cpu() for b in blocks]
    out = np.zeros((n_streams, N_STAT), np.float32)
    seen = np.zeros(n_streams, bool)
    for b in blocks:
        for row in b.numpy():
            sid = int(row[0])
            if sid >= 0:
                out[sid] = row[1:]
                seen[sid] = True
    assert seen.all(), "a stream's statistics never arrived"
    return out


def native_comm(ctx, dist):
    """One fvad Comm (RCCL) per rank of an initialised torch.distributed job: rank 0 creates the bootstrap
    id, broadcast_object_list carries it (any backend), every rank joins with its context's device."""
    from . import binding as fv
    rank, world = dist.get_rank(), dist.get_world_size()
    # communicator creation is collective: make sure EVERY rank can open RCCL before any rank enters it (a rank
    # that raised while the others wait in ncclCommInitRank would hang the job)
    try:
        my_id, err = fv.comm_unique_id(), None
    except Exception as e:  # librccl missing / not loadable on this rank
        my_id, err = None, repr(e)
    oks = [None] * world
    dist.all_gather_object(oks, err)
    bad = [(r, e) for r, e in enumerate(oks) if e is not None]
    if bad:
        raise RuntimeError(f"RCCL unavailable on rank(s) {bad}")
    box = [my_id if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return fv.Comm(ctx, box[0], world, rank)


def gather_stats_native(comm, local_ids, local_stats, n_streams):
    """local_stats: list of binding.SingleStats -> [n_streams][N_STAT] float32 in plan order (every rank)"""
    from . import binding as fv
    allst = comm.allgather_stats(list(local_ids), list(local_stats), n_streams)
    return np.stack([fv.single_stats_to_array(s) for s in allst]) if allst else np.zeros((0, N_STAT), np.float32)


# ------------------------------------------------------------------ time-split sharding of one long stream
CHUNK = 24000
FFT = 1024
HALO_CHUNKS = 2   # warm-up chunks in front of a rank's range (see fvad_lane_state_seek in include/fvad.h)


def split_stream(n_chunks, world):
    """Contiguous, balanced chunk ranges [(c0, c1)] of one stream for `world` ranks (BASELINE config 5: one long
    corpus over 8 GPUs)."""
    base, extra = divmod(n_chunks, world)
    out, c = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((c, c + n))
        c += n
    return out


def time_split_job(n_chunks, c0, c1):
    """What rank (c0, c1) has to process: chunks [start, stop) with start two chunks early (exact state from c0 on:
    the NSNet2 GRU is reset per chunk, src/NSNet2.zig:71-112,188-203) and one chunk late (the 1024-sample VAD
    frame that straddles the range's end).  Returns (start, stop)."""
    return max(c0 - HALO_CHUNKS, 0), min(c1 + 1, n_chunks)


def run_time_split_rank(ctx, pcm, c0, c1, want_denoised=True):
    """One rank's share of a single-channel stream `pcm` (host array, float32 or int16): returns the denoised
    audio and chunk RMS of chunks [c0, c1) and the band sums of the FFT frames that START inside them, each
    bit-identical to the unsplit run."""
    from . import binding as fv
    n_chunks = pcm.shape[0] // CHUNK
    start, stop = time_split_job(n_chunks, c0, c1)
    st = ctx.lane_state()
    try:
        fv.check(fv.lib().fvad_lane_state_seek(st, start * CHUNK, FFT), "fvad_lane_state_seek")
        o = ctx.engine_run([pcm[start * CHUNK: stop * CHUNK]], states=[st], want_denoised=want_denoised)[0]
    finally:
        fv.lib().fvad_lane_state_destroy(st)
    first = o["first_frame_index"]                         # absolute sample index of the lane's first frame
    last_rank = c1 == n_chunks
    k_lo = -((first - c0 * CHUNK) // FFT) if first < c0 * CHUNK else 0          # first frame starting at >= 24000 c0
    k_hi = o["n_fft_frames"] if last_rank else -((first - c1 * CHUNK) // FFT)   # first frame starting at >= 24000 c1
    k_lo, k_hi = max(k_lo, 0), min(k_hi, o["n_fft_frames"])
    den = o["denoised"][(c0 - start) * CHUNK: (c1 - start) * CHUNK] if want_denoised else None
    return {"denoised": den, "chunk_rms": o["chunk_rms"][c0 - start: c1 - start],
            "band_sum": o["band_sum"][k_lo:k_hi], "first_frame_index": first + k_lo * FFT}


# ------------------------------------------------------------------ config 5's steady-state loop: fixed-shape hipGraph replays
ALIGN_CHUNKS = 16   # lcm(24000, 1024) = 384000 samples = 16 chunks = 375 FFT frames: windows that start on a multiple of
                    # 16 chunks keep the VAD FFT's frame grid anchored at absolute sample 0 (BufferedFFT.zig:149)


def run_time_split_rank_graph(ctx, pcm, c0, c1, window=496, lanes=4, want_denoised=True, use_graph=True):
    """One rank's share [c0, c1) of a long single-channel stream as a loop of IDENTICAL device-resident launches
    (BASELINE config 5: "hipGraph-captured steady-state frame loop"): every replay processes `lanes` windows of
    ALIGN_CHUNKS + window chunks through fvad_engine_enqueue_device with fvad_engine_opts.use_graph = 1 -- captured on
    the first call, replayed afterwards while only the CONTENTS of the input buffer change.  A window starts
    ALIGN_CHUNKS chunks early from zero history: two chunks are what NSNet2's cross-chunk state needs
    (src/NSNet2.zig:188-203; `time_split_job`), sixteen keep the 1024-sample frame grid where the unsplit stream has it,
    and because a window's length is a multiple of 16 chunks no frame straddles two windows.  The stream's very first
    window starts at chunk 0 and needs no halo (zero history IS its state, NSNet2.zig:77-79).
    The REPLAY's rate follows the batch curve of one launch (bench.py `extra.batch_curve`): the default replay is 4 x 512 = 2048
    chunks (1.9e7 frames/s on one MI355X); `lanes=8, window=1008` makes it 8192 chunks (2.2e7, and half the halo overhead) at
    0.8 GB per buffer.  This Python loop around it is a correctness harness, not the fast path: it packs the lanes on the host
    and copies synchronously each replay, so its own rate is host-bound (a few 1e6 frames/s); a production host keeps the
    corpus on the device (fvad_engine_enqueue_device with no_wait) or goes through fvad_engine_run's pipelined staging.
    pcm: host float32 array holding the stream at least up to chunk c1 (a prefix is enough).  Returns what
    run_time_split_rank returns, bit-identical to the unsplit run in `reproducible` mode."""
    H = ALIGN_CHUNKS
    assert window % ALIGN_CHUNKS == 0 and window > 0 and lanes > 0
    n_have = pcm.shape[0] // CHUNK
    assert 0 <= c0 < c1 <= n_have
    L = H + window                                   # chunks per lane
    fpl = L * CHUNK // FFT                           # frames per lane (a whole number: L is a multiple of 16)
    a0 = c0 // ALIGN_CHUNKS * ALIGN_CHUNKS
    starts, s = [], (0 if a0 == 0 else a0 - H)       # lane start chunks
    while True:
        starts.append(s)
        out_hi = s + L
        if out_hi >= c1:
            break
        s = out_hi - H
    n_samp = L * CHUNK
    h_in = np.zeros((lanes, n_samp), np.float32)
    h_band = np.empty((lanes, fpl), np.float32)
    h_rms = np.empty((lanes, L), np.float32)
    h_den = np.empty((lanes, n_samp), np.float32) if want_denoised else None
    den = np.empty((c1 - c0) * CHUNK, np.float32) if want_denoised else None
    rms = np.empty(c1 - c0, np.float32)
    f_lo = -(-(c0 * CHUNK) // FFT)                   # first frame starting at >= 24000 c0
    f_hi = -(-(c1 * CHUNK) // FFT) if c1 < n_have else (n_have * CHUNK) // FFT
    band = np.empty(f_hi - f_lo, np.float32)
    replays = 0
    bufs = []                                        # freed in `finally`, however many of the allocations succeeded
    try:
        d_pcm, d_den, d_band, d_rms = (bufs.append(ctx.device_alloc(n)) or bufs[-1]
                                       for n in (lanes * n_samp * 4, lanes * n_samp * 4, lanes * fpl * 4, lanes * L * 4))
        for i in range(0, len(starts), lanes):
            grp = starts[i:i + lanes]
            for j in range(lanes):                   # lanes past the share's end (and audio past the prefix) are silence:
                if j < len(grp):                     # only the tail a lane does not overwrite is zeroed (up to 0.8 GB otherwise)
                    seg = pcm[grp[j] * CHUNK: min(grp[j] + L, n_have) * CHUNK]
                    h_in[j, : seg.shape[0]] = seg
                    h_in[j, seg.shape[0]:] = 0.0
                else:
                    h_in[j] = 0.0
            ctx.to_device(d_pcm, h_in)
            ctx.enqueue_device(d_pcm, lanes, n_samp, n_samp, d_den, d_band, d_rms, use_graph=use_graph)
            replays += 1
            ctx.to_host(h_band, d_band)
            ctx.to_host(h_rms, d_rms)
            if want_denoised:
                ctx.to_host(h_den, d_den)
            for j, s in enumerate(grp):
                lo, hi = max(s if s == 0 else s + H, c0), min(s + L, c1)   # chunks this lane is the authority for
                if lo >= hi:
                    continue
                rms[lo - c0: hi - c0] = h_rms[j, lo - s: hi - s]
                if want_denoised:
                    den[(lo - c0) * CHUNK: (hi - c0) * CHUNK] = h_den[j, (lo - s) * CHUNK: (hi - s) * CHUNK]
                g0 = max(-(-(lo * CHUNK) // FFT), f_lo)                     # frames STARTING inside [lo, hi)
                g1 = min(-(-(hi * CHUNK) // FFT), f_hi)
                if g0 < g1:
                    base = s * CHUNK // FFT                                   # lane's first frame (s is a multiple of 16)
                    band[g0 - f_lo: g1 - f_lo] = h_band[j, g0 - base: g1 - base]
    finally:
        for d in bufs:
            ctx.device_free(d)
    return {"denoised": den, "chunk_rms": rms, "band_sum": band, "first_frame_index": f_lo * FFT, "replays": replays,
            "lane_starts": starts}


# ------------------------------------------------------------------ one long batch in time slices, the host VAD beside the GPU
def run_sliced_with_vad(ctx, d_pcm, n_lanes, lane_stride, n_chunks, vad_batch, slice_chunks=None, n_threads=16,
                        pcm_i16=False, max_launch=49152):
    """A device-resident batch of `n_lanes` long lanes (`n_chunks` chunks each, lane l at d_pcm + l * lane_stride samples) as a
    loop of time slices, with the host stage of slice k -- frame metadata + VAD state machines, `vad_batch.run_part`
    (fvad_vad_batch_run_part) -- running beside the GPU's slice k + 1.  In one call the host stage comes after the last
    kernel (a two-hour stream: 20 ms of GPU, then 36 ms of VADMachine on one core); here only the last slice's is exposed.

    A slice is one fvad_engine_enqueue_device call (no_wait) over chunks [s0 - 16, s1) of every lane: it starts 16 chunks early
    from zero history -- two chunks are what NSNet2's cross-chunk state needs (src/NSNet2.zig:188-203), sixteen keep the
    1024-sample frame grid where the unsplit stream has it (lcm(24000, 1024) = 16 chunks = 375 frames; BufferedFFT.zig:149) --
    and the halo's frames and chunks are dropped.  Within one kernel selection the band sums and RMS are the unsplit run's bit
    for bit; by default a slice is one launch of the size the unsplit run's launches have, and there are at least four slices.
    Returns (segments per stream, {"slices", "slice_chunks", "gpu_s", "host_tail_s"})."""
    import ctypes as C
    import threading
    import time
    from . import binding as fv
    L = fv.lib()
    H = ALIGN_CHUNKS
    if slice_chunks is None: # one launch of the largest size per slice, but at least four slices: a few long lanes would otherwise be
        # ONE slice with all of their VAD behind it (two-hour streams: 36 ms each on one core against ~30 ms of GPU per stream)
        slice_chunks = max(H, min((max_launch // n_lanes - H) // H * H, -(-n_chunks // (4 * H)) * H))
    assert slice_chunks % H == 0 and slice_chunks > 0
    S = min(slice_chunks, (n_chunks + H - 1) // H * H)
    slices = [(s0, min(s0 + S, n_chunks)) for s0 in range(0, n_chunks, S)]
    len_max = min(S + H, n_chunks)
    fr_max = len_max * CHUNK // FFT
    hip = C.CDLL("libamdhip64.so")
    L.fvad_ctx_stream.restype = C.c_void_p
    stream = C.c_void_p(L.fvad_ctx_stream(ctx.h))
    opts = fv.EngineOpts()
    L.fvad_engine_opts_default(C.byref(opts))
    opts.no_wait = 1
    enqueue = L.fvad_engine_enqueue_device_i16 if pcm_i16 else L.fvad_engine_enqueue_device
    bytes_per_sample = 2 if pcm_i16 else 4
    d_band, d_rms, h_band, h_rms, ev = [], [], [], [], []   # released in `finally`, however many of them were made
    errs = []

    def gpu_stage(k):
        s0, s1 = slices[k]
        start = max(s0 - H, 0)
        n, slot = s1 - start, k & 1
        fv.check(enqueue(ctx.h, C.c_void_p(d_pcm + start * CHUNK * bytes_per_sample), n_lanes, lane_stride, n * CHUNK, None,
                         C.c_void_p(d_band[slot]), C.c_void_p(d_rms[slot]), C.byref(opts)), "fvad_engine_enqueue_device", ctx.h)
        fr = n * CHUNK // FFT
        fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_band[slot].ctypes.data, C.c_void_p(d_band[slot]), n_lanes * fr * 4), "copy band sums", ctx.h)
        fv.check(L.fvad_ctx_copy_to_host(ctx.h, h_rms[slot].ctypes.data, C.c_void_p(d_rms[slot]), n_lanes * n * 4), "copy rms", ctx.h)
        hip.hipEventRecord(ev[slot], stream)

    def host_stage(k):
        try:
            s0, s1 = slices[k]
            start = max(s0 - H, 0)
            n, slot = s1 - start, k & 1
            fr = n * CHUNK // FFT
            f0 = (s0 - start) * CHUNK // FFT                     # the halo's frames
            f1 = fr if s1 == n_chunks else f0 + (s1 - s0) * CHUNK // FFT
            band = h_band[slot][: n_lanes * fr].reshape(n_lanes, fr)[:, f0:f1]
            rms = h_rms[slot][: n_lanes * n].reshape(n_lanes, n)[:, s0 - start:]
            vad_batch.run_part(band, rms, s0 * CHUNK // FFT, n_threads=n_threads, want_segments=False)
        except Exception as e:  # re-raised below, in the caller's thread
            errs.append(e)

    worker = None
    try:
        for _ in range(2):
            d_band.append(ctx.device_alloc(n_lanes * fr_max * 4))
            d_rms.append(ctx.device_alloc(n_lanes * len_max * 4))
            h_band.append(ctx.host_alloc(n_lanes * fr_max))
            h_rms.append(ctx.host_alloc(n_lanes * len_max))
            e = C.c_void_p()
            if hip.hipEventCreateWithFlags(C.byref(e), 0x2) != 0:   # hipEventDisableTiming
                raise RuntimeError("hipEventCreate failed")
            ev.append(e)
        t0 = time.perf_counter()
        worker = None
        gpu_stage(0)
        for k in range(len(slices)):
            if worker is not None:
                worker.join()                                # host stage of slice k - 1: it frees slot (k + 1) & 1
            if errs:
                raise errs[0]
            if k + 1 < len(slices):
                gpu_stage(k + 1)
            if hip.hipEventSynchronize(ev[k & 1]) != 0:
                raise RuntimeError("hipEventSynchronize failed")
            worker = threading.Thread(target=host_stage, args=(k,))
            worker.start()
        t_gpu = time.perf_counter() - t0
        worker.join()
        if errs:
            raise errs[0]
        t_all = time.perf_counter() - t0
        segs = vad_batch._segments()
    finally:
        if worker is not None:
            worker.join()
        ctx.synchronize()
        for e in ev:
            hip.hipEventDestroy(e)
        for d in d_band + d_rms:
            ctx.device_free(d)
        for h in h_band + h_rms:
            ctx.host_free(h)
    return segs, {"slices": len(slices), "slice_chunks": S, "gpu_s": t_gpu, "host_tail_s": t_all - t_gpu}
