// engine_run.cpp -- fvad_engine_run: the host-buffer form of the batched engine (what fvad_pipeline_push_samples and
// fvad_nsnet2_denoise call): staging through page-locked rings, lane groups pipelined against the kernels, lane states.
// Part of libfvad_hip.so.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include "internal.h"

using namespace fvad;

// ---- large host <-> device transfers
// hipMemcpyAsync from / to pageable memory moves ~20 GB/s up and only ~4-8 GB/s down on this platform.
// Transfers above a few MB go through a pinned ring instead: worker threads copy user memory <-> pinned
// slots while the DMA engine moves the other half of the ring, so the rate is the slower of the
// parallel memcpy and the PCIe DMA rather than their sum.
namespace {
constexpr size_t kPinSlotBytes = 8u << 20;
constexpr size_t kPinSmallBytes = 4u << 20; // transfers below this total go through the small bounce buffers
constexpr int kPinSlots = 16; // per half
struct CopySeg { void* host; void* dev; size_t bytes; };

int ensure_pin(fvad_ctx* ctx, Workspace::PinRing& ring)
{
    if (ring.base) return FVAD_OK;
    FVAD_HIP(ctx, hipHostMalloc((void**)&ring.base, 2 * kPinSlots * kPinSlotBytes, hipHostMallocDefault));
    for (hipEvent_t& e : ring.ev) FVAD_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return FVAD_OK;
}

// host -> device (to_device) or device -> host, ordered on ctx->stream; returns after the last DMA has
// been enqueued (to_device) or after the data is in user memory (!to_device)
int staged_copy(fvad_ctx* ctx, const std::vector<CopySeg>& segs, bool to_device, hipStream_t st)
{
    Workspace::PinRing& ring = to_device ? ctx->ws.ring_in : ctx->ws.ring_out;
    size_t total_padded = 0;
    for (const CopySeg& s : segs) total_padded += (s.bytes + 63) & ~(size_t)63;
    if (total_padded <= kPinSmallBytes) {
        // Small transfers -- every live push: hipMemcpyAsync to or from pageable memory blocks the calling thread (a
        // device -> host copy until everything queued before it has run: two of them in a row cost a push ~25 us of
        // tail), so the bytes go through one page-locked bounce buffer per direction: host -> device = memcpy + async
        // copies that return at once; device -> host = async copies, ONE wait, memcpy.
        Workspace::PinSmall& b = to_device ? ctx->ws.small_in : ctx->ws.small_out;
        if (!b.base) {
            // the event first: a buffer without its event would make every later call wait on a null event
            if (!b.ev) FVAD_HIP(ctx, hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
            FVAD_HIP(ctx, hipHostMalloc((void**)&b.base, kPinSmallBytes, hipHostMallocDefault));
        } else if (to_device) {
            FVAD_HIP(ctx, hipEventSynchronize(b.ev)); // the previous use's copies have left the buffer
        }
        size_t off = 0;
        for (const CopySeg& s : segs) {
            if (!s.bytes) continue;
            if (to_device) {
                memcpy(b.base + off, s.host, s.bytes);
                FVAD_HIP(ctx, hipMemcpyAsync(s.dev, b.base + off, s.bytes, hipMemcpyHostToDevice, st));
            } else {
                FVAD_HIP(ctx, hipMemcpyAsync(b.base + off, s.dev, s.bytes, hipMemcpyDeviceToHost, st));
            }
            off += (s.bytes + 63) & ~(size_t)63;
        }
        if (to_device) {
            FVAD_HIP(ctx, hipEventRecord(b.ev, st));
            return FVAD_OK;
        }
        // (an event wait, not a stream wait: a stream wait that lasts as long as the kernels in front of these copies has been
        // seen to stall other threads' enqueues for as long -- fvad_engine_run's drain thread, below)
        FVAD_HIP(ctx, hipEventRecord(b.ev, st));
        FVAD_HIP(ctx, hipEventSynchronize(b.ev));
        off = 0;
        for (const CopySeg& s : segs) {
            if (!s.bytes) continue;
            memcpy(s.host, b.base + off, s.bytes);
            off += (s.bytes + 63) & ~(size_t)63;
        }
        return FVAD_OK;
    }
    std::vector<CopySeg> blocks;
    for (const CopySeg& s : segs) {
        bool direct = s.bytes < (256u << 10); // small pieces (band sums, RMS) would waste ring slots
        if (!direct) {
            // page-locked user memory (fvad_host_alloc, hipHostMalloc, hipHostRegister): the DMA engine reads
            // or writes it in place
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, s.host) == hipSuccess && attr.type == hipMemoryTypeHost) direct = true;
            else (void)hipGetLastError(); // an unknown (pageable) pointer is reported as an error: clear it
        }
        if (direct) {
            if (s.bytes) FVAD_HIP(ctx, to_device ? hipMemcpyAsync(s.dev, s.host, s.bytes, hipMemcpyHostToDevice, st)
                                                 : hipMemcpyAsync(s.host, s.dev, s.bytes, hipMemcpyDeviceToHost, st));
            continue;
        }
        for (size_t o = 0; o < s.bytes; o += kPinSlotBytes)
            blocks.push_back({(char*)s.host + o, (char*)s.dev + o, std::min(kPinSlotBytes, s.bytes - o)});
    }
    if (blocks.empty()) return FVAD_OK;
    int rc = ensure_pin(ctx, ring);
    if (rc) return rc;
    // Block by block through the ring's 32 slots: copy threads move user memory <-> slots (block i on thread i mod T, so
    // blocks finish about in order), the calling thread enqueues the DMAs in order, an event per slot says when its DMA is done.
    // What is not overlapped is ONE block's host copy at the start (host -> device) or at the end (device -> host) -- the
    // wave-by-wave form before it (16 slots copied, then their 16 DMAs) exposed a whole wave: 1.3 ms per call of ~100 MB and
    // more, i.e. per lane group of fvad_engine_run (tools/pcie_run.py with FVAD_TRACE_RUN=1).
    const size_t nb = blocks.size();
    constexpr size_t NS = 2 * kPinSlots;
    const int T = std::max(1, std::min<int>(ctx->tune.copy_threads, (int)nb));
    std::unique_ptr<std::atomic<unsigned char>[]> copied(new std::atomic<unsigned char>[nb]), queued(new std::atomic<unsigned char>[nb]);
    for (size_t i = 0; i < nb; ++i) { copied[i].store(0, std::memory_order_relaxed); queued[i].store(0, std::memory_order_relaxed); }
    std::atomic<bool> failed{false};
    auto slot = [&](size_t i) { return ring.base + (i % NS) * kPinSlotBytes; };
    auto wait_flag = [&](std::atomic<unsigned char>& f) {
        // (a wait is normally a fraction of a block's copy; on a host whose cores are busy elsewhere -- the GPU boxes are shared --
        // a yield loop only takes time slices from the threads it waits for, so it backs off to short sleeps)
        for (unsigned spins = 0; !f.load(std::memory_order_acquire); ++spins) {
            if (failed.load(std::memory_order_relaxed)) return false;
            if (spins < 256) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(30));
        }
        return true;
    };
    hipError_t herr = hipSuccess;
    std::vector<std::thread> th;
    if (to_device) {
        auto worker = [&](size_t t) {
            hipSetDevice(ctx->device);
            for (size_t i = t; i < nb; i += (size_t)T) {
                if (i >= NS) { // the slot's previous DMA (block i - NS) must have left it
                    if (!wait_flag(queued[i - NS])) return;
                    if (hipEventSynchronize(ring.ev[i % NS]) != hipSuccess) { failed.store(true); return; }
                }
                memcpy(slot(i), blocks[i].host, blocks[i].bytes);
                copied[i].store(1, std::memory_order_release);
            }
        };
        try {
            for (int t = 0; t < T; ++t) th.emplace_back(worker, (size_t)t);
        } catch (...) { // no thread to be had: the ones that started leave at their next wait
            failed.store(true);
            for (auto& x : th) x.join();
            return set_err(ctx, FVAD_ERR_HIP, "staged copy: cannot start a copy thread");
        }
        for (size_t i = 0; i < nb && !failed.load(); ++i) {
            if (!wait_flag(copied[i])) break;
            if ((herr = hipMemcpyAsync(blocks[i].dev, slot(i), blocks[i].bytes, hipMemcpyHostToDevice, st)) != hipSuccess ||
                (herr = hipEventRecord(ring.ev[i % NS], st)) != hipSuccess) { failed.store(true); break; }
            queued[i].store(1, std::memory_order_release);
        }
        for (auto& x : th) x.join();
        if (failed.load()) return set_err(ctx, FVAD_ERR_HIP, herr != hipSuccess ? hipGetErrorString(herr) : "staged host -> device copy failed");
        // the ring may be reused by a later call: the last DMA (they run in order) must have left the host
        FVAD_HIP(ctx, hipEventSynchronize(ring.ev[(nb - 1) % NS]));
    } else {
        auto worker = [&](size_t t) {
            hipSetDevice(ctx->device);
            for (size_t i = t; i < nb; i += (size_t)T) {
                if (!wait_flag(queued[i])) return;
                if (hipEventSynchronize(ring.ev[i % NS]) != hipSuccess) { failed.store(true); return; }
                memcpy(blocks[i].host, slot(i), blocks[i].bytes);
                copied[i].store(1, std::memory_order_release);
            }
        };
        try {
            for (int t = 0; t < T; ++t) th.emplace_back(worker, (size_t)t);
        } catch (...) { // no thread to be had: the ones that started leave at their next wait
            failed.store(true);
            for (auto& x : th) x.join();
            return set_err(ctx, FVAD_ERR_HIP, "staged copy: cannot start a copy thread");
        }
        for (size_t i = 0; i < nb && !failed.load(); ++i) {
            if (i >= NS && !wait_flag(copied[i - NS])) break; // the slot has been emptied into user memory
            if ((herr = hipMemcpyAsync(slot(i), blocks[i].dev, blocks[i].bytes, hipMemcpyDeviceToHost, st)) != hipSuccess ||
                (herr = hipEventRecord(ring.ev[i % NS], st)) != hipSuccess) { failed.store(true); break; }
            queued[i].store(1, std::memory_order_release);
        }
        for (auto& x : th) x.join();
        if (failed.load()) return set_err(ctx, FVAD_ERR_HIP, herr != hipSuccess ? hipGetErrorString(herr) : "staged device -> host copy failed");
    }
    return FVAD_OK;
}

// ---- lane groups of a pipelined call (host buffers): which sizes.
// While the GPU runs group g the calling thread stages group g + 1 and a second thread drains group g - 1, so a call costs the
// staging of its FIRST group + the kernels of all groups + the drain of its LAST group -- unless staging cannot keep up, and
// less per chunk the larger a group's launch is.  Four equal groups (rounds 2-4) expose a quarter of the input and, with
// denoised audio going back, a quarter of the output.  Measured on 128 streams x 64 s (tools/pcie_sweep.sh, fourteen schedules
// x five format combinations, one MI355X box; ms per call, the equal split first):
//   input staged faster than ~2 x the kernels' rate (PCM16: 850 chunks / ms against 400):  1,3,4,8     45.6 -> 41.1
//   the same with denoised PCM16 going back (a small last group too):                      1,3,4,4,3,1 50.1 -> 45.8
//   input at 1.3 x the kernels' rate (page-locked f32, DMA'd in place: 530 / ms):          2,2,4,4,4   48.4 -> 44.6
//   input at the kernels' rate (pageable f32 through the ring: 410 / ms):                  4,4,4,4     50.0 (every other one slower:
//                                      a group cannot start before it has arrived, and small launches run the spin kernels of
//                                      the small-batch family, beside which the DMA itself slows down)
// A model of staging / kernel / drain rates was tried first and picked worse schedules than the equal split (it prices neither the
// launch sizes that fill the chip badly -- 5120 chunks cost what 8192 do -- nor the DMA beside the spin kernels): the table is
// what was measured.  parts: sixteenths of the call's chunks per group.
void plan_groups(double stage_rate, bool big_drain, std::vector<int>& parts)
{
    if (stage_rate >= 800.0) parts = big_drain ? std::vector<int>{1, 3, 4, 4, 3, 1} : std::vector<int>{1, 3, 4, 8};
    else if (stage_rate >= 500.0) parts = big_drain ? std::vector<int>{2, 4, 4, 4, 2} : std::vector<int>{2, 2, 4, 4, 4};
    else parts = {4, 4, 4, 4};
}
} // namespace

extern "C" {

int fvad_engine_run(fvad_ctx* ctx, fvad_lane* lanes, size_t n_lanes, const fvad_engine_opts* opts_in)
{
    if (!ctx || (n_lanes && !lanes)) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_engine_opts opts;
    if (opts_in) opts = *opts_in; else fvad_engine_opts_default(&opts);
    const size_t F = opts.fft_size ? (size_t)opts.fft_size : (size_t)kVadFft; // VAD FFT frame length
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "NSNet2 weights not loaded");
    hipSetDevice(ctx->device);
    VadFftPlan plan;
    {
        const int prc = get_vad_plan(ctx, F, &plan);
        if (prc) return prc;
    }
    const size_t NB = F / 2 + 1;
    if (opts.min_bin < 0 || opts.max_bin > (int)(F / 2) || opts.max_bin < opts.min_bin) return set_err(ctx, FVAD_ERR_OUT_OF_RANGE, "band bins out of range");
    Workspace& ws = ctx->ws;
    hipStream_t st = ctx->stream;

    // ---- sizes and staging layout (every lane region 64-float aligned)
    size_t in_total = 0, den_total = 0, den16_total = 0, frames_total = 0, chunks_total = 0;
    std::vector<size_t> in_off(n_lanes), den_off(n_lanes), den16_off(n_lanes), band_off(n_lanes), rms_off(n_lanes), n_rem(n_lanes);
    bool want_bins = false;
    for (size_t l = 0; l < n_lanes; ++l) {
        fvad_lane& L = lanes[l];
        if (!L.pcm && !L.pcm_i16 && L.n_samples) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "lane without pcm");
        if (opts.on_device && ((uintptr_t)L.pcm_i16 | (uintptr_t)L.denoised_i16) % 16)
            return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "device PCM16 buffers must be 16-byte aligned");
        L.n_chunks = L.n_samples / kChunk48;
        if (L.state && L.state->fft_size != F) {
            if (L.state->n_rem || L.state->samples_consumed) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "lane state was used with another fft_size");
            L.state->fft_size = F;
        }
        n_rem[l] = L.state ? L.state->n_rem : 0;
        const size_t n_den = n_rem[l] + L.n_chunks * kChunk48;
        L.n_fft_frames = n_den / F;
        L.first_frame_index = L.state ? L.state->next_frame_index : 0;
        if (L.n_fft_frames > L.band_sum_capacity || L.n_chunks > L.chunk_rms_capacity)
            return set_err(ctx, FVAD_ERR_BUFFER_TOO_SMALL, "band_sum / chunk_rms capacity too small");
        if (L.fft_bins) want_bins = true;
        in_off[l] = in_total;
        // staging slots are counted in floats; a PCM16 lane needs half of them
        in_total += ((L.pcm ? L.n_chunks * kChunk48 : L.n_chunks * kChunk48 / 2) + 63) / 64 * 64;
        den16_off[l] = den16_total;
        if (L.denoised_i16 && !opts.on_device) den16_total += (L.n_chunks * kChunk48 / 2 + 63) / 64 * 64;
        den_off[l] = den_total;
        den_total += (kVadFftMax + L.n_chunks * kChunk48 + 63) / 64 * 64;
        band_off[l] = frames_total;
        frames_total += L.n_fft_frames;
        rms_off[l] = chunks_total;
        chunks_total += L.n_chunks;
    }
    int rc;
    if (!opts.on_device && (rc = grow(ctx, &ws.in, &ws.in_cap, in_total))) return rc;
    if ((rc = grow(ctx, &ws.den, &ws.den_cap, den_total))) return rc;
    if (den16_total && (rc = grow(ctx, &ws.den16, &ws.den16_cap, den16_total))) return rc;
    if ((rc = grow(ctx, &ws.band, &ws.band_cap, frames_total + chunks_total + 64))) return rc;
    if (want_bins && (rc = grow(ctx, &ws.bins, &ws.bins_cap, frames_total * NB))) return rc;
    // scratch carries for stateless lanes
    size_t n_scratch = 0;
    for (size_t l = 0; l < n_lanes; ++l) if (!lanes[l].state) n_scratch += 2;
    if (n_scratch * sizeof(LaneCarry) > ws.carries_cap) {
        hipStreamSynchronize(st);
        if (ws.carries) hipFree(ws.carries);
        ws.carries = nullptr; ws.carries_cap = 0;
        FVAD_HIP(ctx, hipMalloc((void**)&ws.carries, n_scratch * sizeof(LaneCarry)));
        ws.carries_cap = n_scratch * sizeof(LaneCarry);
        ws.generation++;
    }
    if (n_scratch) FVAD_HIP(ctx, hipMemsetAsync(ws.carries, 0, n_scratch * sizeof(LaneCarry), st));
    ws.carries_clean = 0;      // this call's launches write them
    ws.jobs_mirror.clear();    // ... and the K4 job table

    // the host-side lane state (remainder length, current carry, counters) is committed only if the whole call
    // succeeds: a caller that retries after an error must not feed the same audio to an advanced state
    struct StateGuard {
        struct Snap { fvad_lane_state* s; int cur; size_t n_rem; uint64_t consumed, next_index; };
        std::vector<Snap> snaps;
        bool commit = false;
        ~StateGuard()
        {
            if (commit) return;
            for (const Snap& x : snaps) { x.s->cur = x.cur; x.s->n_rem = x.n_rem; x.s->samples_consumed = x.consumed; x.s->next_frame_index = x.next_index; }
        }
    } guard;
    for (size_t l = 0; l < n_lanes; ++l)
        if (lanes[l].state) guard.snaps.push_back({lanes[l].state, lanes[l].state->cur, lanes[l].state->n_rem,
                                                   lanes[l].state->samples_consumed, lanes[l].state->next_frame_index});
    float* d_rms = ws.band + frames_total;
    std::vector<LaneJob> jobs(n_lanes);
    struct Restore { float* dst; const float* src; size_t bytes; };
    std::vector<Restore> restores; // the previous call's FFT remainder of every lane, to be put in front of its new audio
    std::vector<CopySeg> h2d;
    size_t scratch_i = 0;
    for (size_t l = 0; l < n_lanes; ++l) {
        fvad_lane& L = lanes[l];
        LaneJob& j = jobs[l];
        const size_t n_in = L.n_chunks * kChunk48;
        const bool pcm16 = !L.pcm;
        if (opts.on_device) { j.d_in = L.pcm; j.d_in16 = pcm16 ? L.pcm_i16 : nullptr; }
        else if (pcm16) {
            if (n_in) h2d.push_back({(void*)L.pcm_i16, ws.in + in_off[l], n_in * sizeof(int16_t)});
            j.d_in = nullptr;
            j.d_in16 = reinterpret_cast<const int16_t*>(ws.in + in_off[l]);
        } else {
            if (n_in) h2d.push_back({(void*)L.pcm, ws.in + in_off[l], n_in * sizeof(float)});
            j.d_in = ws.in + in_off[l];
        }
        if (L.denoised_i16) j.d_den16 = opts.on_device ? L.denoised_i16 : reinterpret_cast<int16_t*>(ws.den16 + den16_off[l]);
        // denoised region: [1024-float prefix | chunks]; the not-yet-FFT'd remainder of the previous
        // call sits right in front of the new audio so that K4 sees one contiguous signal
        float* den_base = ws.den + den_off[l] + kVadFftMax;
        j.d_den = den_base;
        j.n_chunks = L.n_chunks;
        j.d_rms = d_rms + rms_off[l];
        j.h_spec = L.spectrogram;
        j.h_feat = L.features;
        if (L.state) {
            j.carry[0] = L.state->carry[0]; j.carry[1] = L.state->carry[1]; j.cur = L.state->cur;
            // (queued behind the first group's kernels, in front of its K4: nothing earlier reads it, and the GPU
            // idles until K1 is launched -- every host call in front of that launch is latency of a live push)
            if (n_rem[l]) restores.push_back({den_base - n_rem[l], L.state->den_rem, n_rem[l] * sizeof(float)});
        } else {
            j.carry[0] = ws.carries + scratch_i; j.carry[1] = ws.carries + scratch_i + 1; j.cur = 0;
            scratch_i += 2;
        }
    }
    // ---- lane groups.  With host buffers and enough work the call is pipelined over up to four groups
    // of lanes: while the GPU runs group g, the host stages group g+1's input into the pinned ring and
    // drains group g-1's output (copies on their own streams, ordered by events).  Staging pageable
    // memory moves ~25 GB/s on the host side whatever the method, so hiding it behind compute is what
    // is left to gain.
    size_t h2d_bytes = 0;
    for (const CopySeg& c : h2d) h2d_bytes += c.bytes;
    int G = 1;
    std::vector<int> parts;
    // (two lanes are enough: a group is at least one lane, and a few long lanes -- four two-hour files -- are the case where
    // staging everything first costs most; the planned sixteenths merge into as many groups as the lanes allow)
    if (!opts.on_device && n_lanes >= 2 && h2d_bytes >= (64u << 20) && !ctx->tune.no_pipeline) {
        // rates of this call's formats, chunks per ms (one chunk: 96 000 bytes as f32, 48 000 as PCM16): pageable memory through
        // the page-locked ring ~41 GB/s in, ~37 GB/s out; page-locked user memory at the link's 51 GB/s (tools/drain_rate.hip)
        bool in16 = false, in_pinned = false, out32 = false, out16 = false, out_pinned = false;
        auto pinned = [](const void* p) {
            hipPointerAttribute_t attr;
            if (p && hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost) return true;
            (void)hipGetLastError();
            return false;
        };
        for (size_t l = 0; l < n_lanes; ++l) {
            const fvad_lane& L = lanes[l];
            if (!L.n_chunks) continue;
            if (!L.pcm) in16 = true;
            if (L.denoised) out32 = true; else if (L.denoised_i16) out16 = true;
        }
        for (size_t l = 0; l < n_lanes; ++l)
            if (lanes[l].n_chunks) {
                in_pinned = pinned(lanes[l].pcm ? (const void*)lanes[l].pcm : (const void*)lanes[l].pcm_i16);
                out_pinned = pinned(lanes[l].denoised ? (const void*)lanes[l].denoised : (const void*)lanes[l].denoised_i16);
                break;
            }
        const double in_gbps = in_pinned ? 51.0 : 41.0;
        (void)out_pinned;
        const double stage_rate = in_gbps * 1e6 / (in16 ? 48000.0 : 96000.0);
        if (!ctx->tune.run_groups.empty()) { // context option run_groups: measurements, tests
            int cur = 0;
            for (char c : ctx->tune.run_groups + ",") {
                if (c == ',') { parts.push_back(cur); cur = 0; } else cur = cur * 10 + (c - '0');
            }
        } else if (chunks_total >= 8192) plan_groups(stage_rate, out32 || out16, parts); // (a sixteenth of less is a launch of a few hundred chunks)
        else parts = {4, 4, 4, 4};
        G = (int)parts.size();
    }
    // group boundaries: contiguous lanes, chunk counts as planned (a group is at least one lane; a plan's group that no lane
    // boundary falls into is merged with its neighbour)
    std::vector<size_t> gb(1, 0);
    if (G > 1) {
        size_t acc = 0;
        int g = 0, cum = parts[0];
        for (size_t l = 0; l < n_lanes && g + 1 < G; ++l) {
            acc += lanes[l].n_chunks;
            while (g + 1 < G && acc * 16 >= chunks_total * (size_t)cum) {
                if (gb.back() != l + 1 && l + 1 < n_lanes) gb.push_back(l + 1);
                cum += parts[++g];
            }
        }
    }
    gb.push_back(n_lanes);
    G = (int)gb.size() - 1;
    if (G > 1) {
        if (!ws.copy_in) FVAD_HIP(ctx, hipStreamCreateWithFlags(&ws.copy_in, hipStreamNonBlocking));
        if (!ws.copy_out) FVAD_HIP(ctx, hipStreamCreateWithFlags(&ws.copy_out, hipStreamNonBlocking));
    }
    hipStream_t s_in = G > 1 ? ws.copy_in : st, s_out = G > 1 ? ws.copy_out : st;
    if (G > 1) {
        for (int g = 0; g < G; ++g) {
            if (!ws.grp_in[g]) FVAD_HIP(ctx, hipEventCreateWithFlags(&ws.grp_in[g], hipEventDisableTiming));
            if (!ws.grp_k[g]) FVAD_HIP(ctx, hipEventCreateWithFlags(&ws.grp_k[g], hipEventDisableTiming));
        }
    }
    // K4 job table for every lane (pointers are known up front; uploaded in front of the first K4 launch)
    long max_frames = 0;
    const VadFftJob* jobs_upload = nullptr;
    int jobs_upload_slot = 0;
    {
        if (ws.fft_jobs_cap < n_lanes) {
            hipStreamSynchronize(st);
            if (ws.fft_jobs) hipFree(ws.fft_jobs);
            if (ws.h_fft_jobs) hipHostFree(ws.h_fft_jobs);
            ws.fft_jobs = nullptr; ws.h_fft_jobs = nullptr; ws.fft_jobs_cap = 0;
            FVAD_HIP(ctx, hipMalloc((void**)&ws.fft_jobs, n_lanes * sizeof(VadFftJob)));
            FVAD_HIP(ctx, hipHostMalloc((void**)&ws.h_fft_jobs, 2 * n_lanes * sizeof(VadFftJob), hipHostMallocDefault));
            ws.fft_jobs_cap = n_lanes;
            ws.generation++;
        }
        // the pinned table has two slots (shared with fvad_engine_enqueue_device*, whose no_wait calls may still have
        // an upload pending): a slot is rewritten only after its previous upload has left the host
        const int js = ws.jobs_slot;
        ws.jobs_slot ^= 1;
        if (!ws.jobs_ev[js]) FVAD_HIP(ctx, hipEventCreateWithFlags(&ws.jobs_ev[js], hipEventDisableTiming));
        else FVAD_HIP(ctx, hipEventSynchronize(ws.jobs_ev[js]));
        VadFftJob* hj = ws.h_fft_jobs + (size_t)js * ws.fft_jobs_cap;
        for (size_t l = 0; l < n_lanes; ++l) {
            const fvad_lane& L = lanes[l];
            hj[l] = {jobs[l].d_den - n_rem[l], ws.band + band_off[l],
                     L.fft_bins ? ws.bins + band_off[l] * NB : nullptr, (long)L.n_fft_frames};
            max_frames = std::max(max_frames, (long)L.n_fft_frames);
        }
        jobs_upload = hj;
        jobs_upload_slot = js;
    }

    auto outputs_of = [&](size_t l0, size_t l1) -> int {
        std::vector<CopySeg> d2h;
        // Band sums and RMS of consecutive lanes are contiguous in the workspace: they leave as TWO copies into the page-locked
        // bounce buffer and are dealt to the lanes' arrays from there -- not two per lane (a small copy is a blit KERNEL, which
        // finds no compute unit while a persistent GEMM or recurrence of a later group holds every register of the chip: 64 of
        // them per group took up to 7 ms; tools/pcie_run.py with FVAD_TRACE_RUN=1)
        size_t band_n = 0, rms_n = 0;
        for (size_t l = l0; l < l1; ++l) { band_n += lanes[l].n_fft_frames; rms_n += lanes[l].n_chunks; }
        const size_t rms_at = (band_n * sizeof(float) + 63) & ~(size_t)63;
        const bool gathered = l1 > l0 && rms_at + rms_n * sizeof(float) <= kPinSmallBytes;
        if (gathered && band_n + rms_n) {
            Workspace::PinSmall& b = ctx->ws.small_out;
            if (!b.base) {
                if (!b.ev) FVAD_HIP(ctx, hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
                FVAD_HIP(ctx, hipHostMalloc((void**)&b.base, kPinSmallBytes, hipHostMallocDefault));
            }
            if (band_n) FVAD_HIP(ctx, hipMemcpyAsync(b.base, ws.band + band_off[l0], band_n * sizeof(float), hipMemcpyDeviceToHost, s_out));
            if (rms_n) FVAD_HIP(ctx, hipMemcpyAsync(b.base + rms_at, d_rms + rms_off[l0], rms_n * sizeof(float), hipMemcpyDeviceToHost, s_out));
            FVAD_HIP(ctx, hipEventRecord(b.ev, s_out));
            FVAD_HIP(ctx, hipEventSynchronize(b.ev));
            const float* hb = reinterpret_cast<const float*>(b.base);
            const float* hr = reinterpret_cast<const float*>(b.base + rms_at);
            for (size_t l = l0; l < l1; ++l) {
                fvad_lane& L = lanes[l];
                if (L.n_fft_frames) memcpy(L.band_sum, hb + (band_off[l] - band_off[l0]), L.n_fft_frames * sizeof(float));
                if (L.n_chunks) memcpy(L.chunk_rms, hr + (rms_off[l] - rms_off[l0]), L.n_chunks * sizeof(float));
            }
        }
        for (size_t l = l0; l < l1; ++l) {
            fvad_lane& L = lanes[l];
            if (L.n_fft_frames) {
                if (!gathered) d2h.push_back({L.band_sum, ws.band + band_off[l], L.n_fft_frames * sizeof(float)});
                if (L.fft_bins) d2h.push_back({L.fft_bins, ws.bins + band_off[l] * NB, L.n_fft_frames * NB * sizeof(float)});
            }
            if (L.n_chunks) {
                if (!gathered) d2h.push_back({L.chunk_rms, d_rms + rms_off[l], L.n_chunks * sizeof(float)});
                if (L.denoised && !opts.on_device) d2h.push_back({L.denoised, jobs[l].d_den, L.n_chunks * kChunk48 * sizeof(float)});
                if (L.denoised_i16 && !opts.on_device) d2h.push_back({L.denoised_i16, jobs[l].d_den16, L.n_chunks * kChunk48 * sizeof(int16_t)});
            }
        }
        if (d2h.empty()) return FVAD_OK;
        return staged_copy(ctx, d2h, false, s_out);
    };

    const bool trace_run = ctx->tune.trace_run; // timeline of the call on stderr (context option trace_run / FVAD_TRACE_RUN=1 at fvad_ctx_create; tools/pcie_run.py)
    const auto t_start = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what, int g) {
        if (trace_run) fprintf(stderr, "[run] %8.3f ms  %s %d\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what, g);
    };
    if (trace_run && G > 1) {
        fprintf(stderr, "[run] groups:");
        for (int g = 0; g < G; ++g) { size_t n = 0; for (size_t l = gb[g]; l < gb[g + 1]; ++l) n += lanes[l].n_chunks; fprintf(stderr, " %zu", n); }
        fprintf(stderr, " chunks\n");
    }
    // a second host thread drains group g's outputs (its own pinned ring and stream) while this one stages
    // group g+1's input: both are memcpy-bound host work
    std::atomic<int> groups_recorded{0};
    std::atomic<bool> abort_out{false};
    int rc_out = FVAD_OK;
    std::thread out_thread;
    if (G > 1)
        out_thread = std::thread([&] {
            hipSetDevice(ctx->device);
            for (int g = 0; g < G; ++g) {
                while (groups_recorded.load(std::memory_order_acquire) <= g) {
                    if (abort_out.load()) return;
                    std::this_thread::yield();
                }
                if (hipStreamWaitEvent(s_out, ws.grp_k[g], 0) != hipSuccess) { rc_out = FVAD_ERR_HIP; return; }
                // The host waits for the group's kernels in an EVENT wait.  A stream wait of that length (hipStreamSynchronize on
                // s_out, which the copies below end in) can stall the calling thread's own enqueues for as long as it lasts: the
                // staging of group g + 1 then starts only when group g is drained and the call is serialised -- 76 ms instead of
                // 50, observed in bench.py (after its 16-thread CPU leg, never in a fresh process; FVAD_TRACE_RUN=1 shows it)
                if (hipEventSynchronize(ws.grp_k[g]) != hipSuccess) { rc_out = FVAD_ERR_HIP; return; }
                stamp("kernels done, group", g);
                if ((rc_out = outputs_of(gb[g], gb[g + 1]))) return;
                stamp("drained group", g);
            }
        });
    struct Joiner { std::thread& t; std::atomic<bool>& a; ~Joiner() { if (t.joinable()) { a.store(true); t.join(); } } } joiner{out_thread, abort_out};

    for (int g = 0; g < G; ++g) {
        const size_t l0 = gb[g], l1 = gb[g + 1];
        // input of this group
        std::vector<CopySeg> in_g;
        for (size_t l = l0; l < l1; ++l) {
            const size_t n_in = lanes[l].n_chunks * kChunk48;
            if (!opts.on_device && n_in) {
                if (lanes[l].pcm) in_g.push_back({(void*)lanes[l].pcm, ws.in + in_off[l], n_in * sizeof(float)});
                else in_g.push_back({(void*)lanes[l].pcm_i16, ws.in + in_off[l], n_in * sizeof(int16_t)});
            }
        }
        if ((rc = staged_copy(ctx, in_g, true, s_in))) return rc;
        stamp("staged group", g);
        if (G > 1) {
            FVAD_HIP(ctx, hipEventRecord(ws.grp_in[g], s_in));
            FVAD_HIP(ctx, hipStreamWaitEvent(st, ws.grp_in[g], 0));
        }
        // kernels of this group
        std::vector<LaneJob> jg(jobs.begin() + l0, jobs.begin() + l1);
        if ((rc = run_chunks(ctx, jg, opts.max_chunks_per_launch))) return rc;
        for (size_t l = l0; l < l1; ++l) jobs[l].cur = jg[l - l0].cur;
        long mf = 0;
        for (size_t l = l0; l < l1; ++l) mf = std::max(mf, (long)lanes[l].n_fft_frames);
        if (g == 0) { // what only K4 needs: the lanes' remainders in front of their new audio, the job table
            for (const Restore& r : restores) FVAD_HIP(ctx, hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyDeviceToDevice, st));
            if (max_frames) {
                FVAD_HIP(ctx, hipMemcpyAsync(ws.fft_jobs, jobs_upload, n_lanes * sizeof(VadFftJob), hipMemcpyHostToDevice, st));
                FVAD_HIP(ctx, hipEventRecord(ws.jobs_ev[jobs_upload_slot], st));
            }
        }
        if (mf) {
            time_begin(ctx, "fft1024_bandsum");
            int any_bins = 0;
            for (size_t l = l0; l < l1; ++l) any_bins |= lanes[l].fft_bins != nullptr;
            FVAD_HIP(ctx, (hipError_t)fvad_launch_vadfft_jobs(ws.fft_jobs + l0, (int)(l1 - l0), mf, plan, opts.min_bin, opts.max_bin, st, any_bins, ctx->n_cu, ctx->tune.k4_plain_loads ? 1 : 0));
            time_end(ctx);
        }
        for (size_t l = l0; l < l1; ++l) {
            fvad_lane& L = lanes[l];
            const float* den_start = jobs[l].d_den - n_rem[l];
            if (L.n_chunks && L.denoised && opts.on_device)
                FVAD_HIP(ctx, hipMemcpyAsync(L.denoised, jobs[l].d_den, L.n_chunks * kChunk48 * sizeof(float), hipMemcpyDeviceToDevice, st));
            if (L.state) {
                const size_t n_den = n_rem[l] + L.n_chunks * kChunk48;
                const size_t rem = n_den - L.n_fft_frames * F;
                if (rem) FVAD_HIP(ctx, hipMemcpyAsync(L.state->den_rem, den_start + L.n_fft_frames * F, rem * sizeof(float), hipMemcpyDeviceToDevice, st));
                L.state->n_rem = rem;
                L.state->cur = jobs[l].cur;
                L.state->samples_consumed += L.n_chunks * kChunk48;
                L.state->next_frame_index += L.n_fft_frames * (uint64_t)F;
            }
        }
        if (G > 1) {
            FVAD_HIP(ctx, hipEventRecord(ws.grp_k[g], st));
            groups_recorded.store(g + 1, std::memory_order_release);
        }
        stamp("enqueued group", g);
    }
    if (G > 1) {
        out_thread.join(); // all groups recorded: the worker runs to completion
        if (rc_out) return set_err(ctx, rc_out, "device-to-host output copy failed");
        FVAD_HIP(ctx, hipStreamSynchronize(s_in));
        FVAD_HIP(ctx, hipStreamSynchronize(s_out));
    } else if ((rc = outputs_of(0, n_lanes))) return rc;
    FVAD_HIP(ctx, hipStreamSynchronize(st));
    FVAD_HIP(ctx, hipGetLastError());
    guard.commit = true;
    return FVAD_OK;
}

} // extern "C"
