// nn_dispatch.cpp -- the NSNet2 pass of a launch: workspace, batch padding, the kernel-selection cost model, the launch
// sequences of the three arithmetics (run_nn), launch planning, and K1 -> NSNet2 -> K3 over the chunks of a call (run_chunks).
// Part of libfvad_hip.so (internal.h declares what the rest of the library calls).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "internal.h"

namespace fvad {

void free_workspace_nn(Workspace& ws)
{
    float** bufs[] = {&ws.feat, &ws.spec, &ws.a1, &ws.gi, &ws.h1, &ws.h2, &ws.hs1, &ws.hs2, &ws.f2, &ws.f3, &ws.gains};
    for (float** b : bufs) { if (*b) hipFree(*b); *b = nullptr; }
    if (ws.descs) hipFree(ws.descs);
    if (ws.h_descs) hipHostFree(ws.h_descs);
    ws.descs = nullptr; ws.h_descs = nullptr;
    ws.descs_mirror.clear();
    ws.cap_chunks = 0;
    ws.cap_rows = 0;
    ws.a1_cap_rows = ws.h_cap_rows = ws.hs_cap_rows = 0;
}


int ensure_workspace(fvad_ctx* ctx, long n_chunks, int T, int skip, long n_last)
{
    Workspace& ws = ctx->ws;
    // chunk-count-sized buffers (descriptors, spectrogram): rounded to 768 = lcm of every batch padding (32, 128, 192 -> 384, 256)
    const long need = ((n_chunks + 767) / 768) * 768;
    const DeviceModel& dm = ctx->dm;
    const bool same_widths = ws.w_a1 == dm.w_a1 && ws.w_gi == dm.w_gi && ws.w_h == dm.w_h && ws.w_f == dm.w_f;
    // (capacities are ROWS -- padded sequences x steps: a long sequence and a wide batch need not fit at once)
    // the NSNet2 buffers hold the rows of the padding this launch really uses (a one-sequence call of 14400 steps is 32
    // padded sequences, not 768: 2 GB of gi instead of 53)
    // n_last: the size of a call's short last launch, whose padding need not be below the full launches'
    const size_t need_rows = (size_t)std::max(padded_batch(ctx, n_chunks, T, skip), n_last > 0 ? padded_batch(ctx, n_last, T, skip) : 0L) * (size_t)T;
    int rc;
    if (!(need <= ws.cap_chunks && need_rows <= ws.cap_rows && same_widths)) {
        hipStreamSynchronize(ctx->stream);
        const long G = std::max(need, ws.cap_chunks);
        const size_t rows = std::max(need_rows, ws.cap_rows);
        free_workspace_nn(ws);
        FVAD_HIP(ctx, hipMalloc((void**)&ws.descs, (size_t)G * sizeof(ChunkDesc)));
        FVAD_HIP(ctx, hipHostMalloc((void**)&ws.h_descs, 2 * (size_t)G * sizeof(ChunkDesc), hipHostMallocDefault));
        for (hipEvent_t& e : ws.desc_ev) if (!e) FVAD_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        // what every arithmetic uses.  Zero-filled: padded rows / padded columns are read by the GEMMs and must stay finite
        if ((rc = dev_alloc(ctx, &ws.feat, rows * kFeatStride, true))) return rc;
        if ((rc = dev_alloc(ctx, &ws.spec, (size_t)G * kFramesPerChunk * kNBins * 2, true))) return rc;
        if ((rc = dev_alloc(ctx, &ws.gi, rows * (size_t)dm.w_gi, true))) return rc;
        if ((rc = dev_alloc(ctx, &ws.f2, rows * (size_t)dm.w_f, true))) return rc;
        if ((rc = dev_alloc(ctx, &ws.f3, rows * (size_t)dm.w_f, true))) return rc;
        if ((rc = dev_alloc(ctx, &ws.gains, rows * kFeatStride, true))) return rc;
        ws.w_a1 = dm.w_a1; ws.w_gi = dm.w_gi; ws.w_h = dm.w_h; ws.w_f = dm.w_f;
        ws.cap_chunks = G;
        ws.cap_rows = rows;
        ws.generation++;
    }
    // ---- buffers only ONE arithmetic (or one kernel option) reads: allocated when a context first runs that way, so an
    // f32 context does not carry the emulations' fragment buffers (at 49152 chunks: 8.8 GB of f16x3 fragments, 33 GB of
    // bf16x3 ones) and an f16x3 context does not carry row-major h1 / h2
    const int math = nn_math_effective(ctx);
    auto group = [&](std::initializer_list<float**> bufs, std::initializer_list<size_t> widths, size_t& cap_rows) -> int {
        if (need_rows <= cap_rows) return FVAD_OK;
        hipStreamSynchronize(ctx->stream);
        const size_t rows = std::max(need_rows, ws.cap_rows);
        auto w = widths.begin();
        for (float** b : bufs) {
            if (*b) hipFree(*b);
            *b = nullptr;
            const int r = dev_alloc(ctx, b, rows * *w++, true);
            if (r) { cap_rows = 0; return r; }
        }
        cap_rows = rows;
        ws.generation++;
        return FVAD_OK;
    };
    // fc1's output: generic models, and the baseline model only with the fold switched off (gemm_kernel = v3nofold)
    if (dm.generic || ctx->tune.gemm_kernel.find("nofold") != std::string::npos)
        if ((rc = group({&ws.a1}, {(size_t)dm.w_a1}, ws.a1_cap_rows))) return rc;
    // h1 / h2 row-major f32: the f32 kernels and the bf16x3 mode's f32 recurrences
    if (math != FVAD_NN_MATH_F16X3)
        if ((rc = group({&ws.h1, &ws.h2}, {(size_t)dm.w_h, (size_t)dm.w_h}, ws.h_cap_rows))) return rc;
    // f16x3: h1 / h2 as split f16 fragments, 13 K-steps of 2 KB per 16 rows
    if (math == FVAD_NN_MATH_F16X3)
        if ((rc = group({&ws.hs1, &ws.hs2}, {416, 416}, ws.hs_cap_rows))) return rc;
    // bf16x3: h1 / h2 and the fc2 / fc3 outputs as three-piece fragments (13 / 19 K-steps of 3 KB per 16 rows)
    if (math == FVAD_NN_MATH_BF16X3)
        if ((rc = group({&ws.b3_hs1, &ws.b3_hs2, &ws.b3_f2, &ws.b3_f3}, {624, 624, 912, 912}, ws.b3_cap_rows))) return rc;
    return FVAD_OK;
}

// the workspace for every launch of a plan (plan_launches): the largest launch, and the other launch whose padding is the widest
int ensure_workspace_plan(fvad_ctx* ctx, const std::vector<long>& plan)
{
    long largest = 0, other = 0, other_pad = 0;
    for (long n : plan) largest = std::max(largest, n);
    for (long n : plan) {
        if (n == largest) continue;
        const long pad = padded_batch(ctx, n, kRowsPerChunk, kWarmupRows);
        if (pad > other_pad) { other_pad = pad; other = n; }
    }
    return ensure_workspace(ctx, largest, kRowsPerChunk, kWarmupRows, other);
}

// Large batches: the LDS-DMA kernels with 192 / 128 / 64 sequences per workgroup; small batches keep
// one wavefront (16 sequences) per workgroup so that more CUs take part.
struct GruChoice {
    int version; // 3: gru_rec3 (expects the z/r recurrent biases folded into gi),
                 // 4: gru_lat (16 sequences per workgroup, tiles split over 8 waves),
                 // 5: gru_ws (weights stationary in registers across 25 x G workgroups, kernels_ws.hip)
    int waves;
};

constexpr size_t kWsLocalAt = 520 + 2000; // 2 x 256 flags + error word + ticket; gru_ws2k's step trace (ws2_variant 64); then its XCD-local flags and tickets
constexpr size_t kWsSyncWords = kWsLocalAt + kWs2LocalWords;

// gru_ws launches spin on each other's flags, so two of them must not share the chip half-resident.
// Within a process every such launch waits (on the GPU) for the previous one on the same device; across
// processes the kernel's bounded spins and its gru_lat fallback take over.
static std::mutex g_ws_mu;
static hipEvent_t g_ws_ev[64] = {};

// Measured cycles per time step of one workgroup on MI355X; a launch costs
// ceil(workgroups / CUs) rounds of that.  The workgroup shapes trade sequences per CU against
// wavefronts per SIMD: 192 sequences (12 waves), 128 (8), 64 (4), or the low-latency shape (waves = 0
// here): 16 sequences with the unit tiles of a step split over 8 waves.
// weight-stationary kernel, measured (tools/gru_crossover.py): a step costs 2.6 us of exchange (publish, flag,
// barriers) plus 3.9 us per row tile (25 KB of h from the memory side + 100 MFMAs per gate wavefront), against
// ~40 us for a step of the low-latency kernel: it wins up to ~1900 sequences
static double gru_ws_cost(long n_pad, int n_cu)
{
    int RT = 0, G = 0;
    if (!fvad_gru_ws_shape(n_pad, n_cu, &RT, &G)) return 1e30;
    return 6.2e3 + 9.4e3 * RT;
}

// both layers pipelined in one launch (gru_ws2_kernel): 55 steps instead of 2 x 54 and no input-projection GEMM for
// layer 2; a step costs about what gru_ws's does at the same row tiles per group (fewer groups fit: 26 workgroups each)
static double gru_ws2_cost_both_layers(long n_pad, int T, int n_cu, int variant)
{
    int RT = 0, G = 0;
    if (!fvad_gru_ws2_shape(n_pad, n_cu, &RT, &G) || !fvad_gru_ws2_ok(n_pad, T, n_cu, variant)) return 1e30;
    if (variant & 8) return 55.0 * (6.2e3 + 9.4e3 * RT); // the 8-wavefront kernel (up to 4 row tiles per group)
    // gru_ws2k (one row tile per group: the hand-off chain alone) / gru_ws2m (row tiles streamed: ~5.9k clocks each, MFMA-paced)
    return RT == 1 ? 55.0 * 13e3 : 55.0 * (7e3 + 5.9e3 * RT); // measured: tools/ws2_ab.py, tools/ws2m_ab.py
}

// gru_lat with 1, 2 or 3 row tiles per workgroup on one stream of R (gru_lat_kernel, gru_lat2_kernel, gru_lat3_kernel: the same
// bits): measured 41 / 73.6 / 103 us per step, on the scale of this model 120k / 215k / 301k clocks per round of workgroups
static double gru_lat_cost(long n_pad, int n_cu, int* rt_out)
{
    static const double per_step[4] = {0, 120e3, 215e3, 301e3};
    double best = 1e30;
    int best_rt = 1;
    for (int rt = 1; rt <= 3; ++rt) {
        if (n_pad % (16 * rt)) continue;
        const long wgs = n_pad / (16 * rt);
        const double c = (double)((wgs + n_cu - 1) / n_cu) * per_step[rt];
        if (c < best) { best = c; best_rt = rt; }
    }
    if (rt_out) *rt_out = best_rt;
    return best;
}

static double gru_cost(long n_pad, int waves, int n_cu)
{
    if (waves == 0) return gru_lat_cost(n_pad, n_cu, nullptr);
    const double per_step = waves == 12 ? 25 * 32.3e3 : waves == 8 ? 25 * 23.4e3 : 25 * 13.0e3;
    const long wgs = n_pad / (16 * waves);
    return (double)((wgs + n_cu - 1) / n_cu) * per_step;
}

// Batch padding: the 12-wave recurrence needs a multiple of 192 sequences and the GEMM row panels a
// multiple of 256 rows (of 54 and of 50 rows per sequence), i.e. 384 sequences; the other shapes need
// 128.  Pick whichever padding gives the cheaper recurrence.
// The arithmetic of the NSNet2 matrix products is a property of the context (and of the loaded model), never of
// a launch's size: f16x3 when it was asked for (fvad_ctx_set_nn_math, or FVAD_NN_MATH at fvad_ctx_create), the model
// is eligible (DeviceModel::h3_ok) and no f32 kernel variant is forced; f32 otherwise.
int nn_math_effective(const fvad_ctx* ctx)
{
    const Tuning& tn = ctx->tune;
    const int want = tn.nn_math_force >= 0 ? tn.nn_math_force : ctx->nn_math;
    if (want == FVAD_NN_MATH_F32) return FVAD_NN_MATH_F32;
    if (!tn.gru_kernel.empty() || !tn.gemm_kernel.empty()) return FVAD_NN_MATH_F32;
    if (want == FVAD_NN_MATH_BF16X3) // exact three-piece splits: no bounds to satisfy, only the baseline dimensions
        return (ctx->dm.loaded && ctx->dm.generic) ? FVAD_NN_MATH_F32 : FVAD_NN_MATH_BF16X3;
    if (ctx->dm.loaded && !ctx->dm.h3_ok) return FVAD_NN_MATH_F32;
    return FVAD_NN_MATH_F16X3;
}

long padded_batch(const fvad_ctx* ctx, long n, int T, int skip)
{
    const long a = (n + 383) / 384 * 384, b = (n + 127) / 128 * 128;
    const Tuning& tn = ctx->tune;
    if (ctx->dm.generic) { // run-time-sized kernels: 64-row GEMM workgroups over T n and (T - skip) n rows
        const long g = (n + 31) / 32 * 32;
        return ((g * T) % 64 != 0 || (g * (T - skip)) % 64 != 0) ? (n + 63) / 64 * 64 : g;
    }
    const char* force = tn.gru_kernel.empty() ? nullptr : tn.gru_kernel.c_str();
    const int cu = ctx->n_cu;
    if (nn_math_effective(ctx) == FVAD_NN_MATH_F16X3) {
        // kernels_h3.hip at every batch size: 192- or 128-sequence workgroups (a round of the latter costs 0.76 of a
        // round of the former)
        // the tiled layouts group 16 sequences per time step, and the GEMM panels take 16 such row tiles: both
        // (n_pad / 16) T and (n_pad / 16) (T - skip) must be multiples of 16 (T = 54, skip = 4: any multiple of 128)
        auto fits = [&](long np) { return ((np / 16) * T) % 16 == 0 && ((np / 16) * (T - skip)) % 16 == 0; };
        const double ca = (double)((a / 192 + cu - 1) / cu), cb = 0.76 * (double)((b / 128 + cu - 1) / cu);
        long pick = (a == b || tn.h3_waves == 12) ? a : (tn.h3_waves == 8) ? b : (cb <= ca ? b : a);
        if (!fits(pick)) pick = fits(a) ? a : (pick + 255) / 256 * 256; // 16 row-tile groups: fits for every T
        return pick;
    }
    if (nn_math_effective(ctx) == FVAD_NN_MATH_BF16X3) {
        // kernels_b3.hip GEMMs (16 row tiles per panel) + gru_rec3: a multiple of 128 sequences, 384 when its 12-wave
        // recurrence is cheaper; every launch, small ones too
        auto fits = [&](long np) { return ((np / 16) * T) % 16 == 0 && ((np / 16) * (T - skip)) % 16 == 0; };
        const double cost_a = std::min(std::min(gru_cost(a, 12, cu), gru_cost(a, 8, cu)), gru_cost(a, 4, cu));
        const double cost_b = std::min(gru_cost(b, 8, cu), gru_cost(b, 4, cu));
        long pick = (a == b || cost_a < cost_b) ? a : b;
        if (!fits(pick)) pick = fits(a) ? a : (pick + 255) / 256 * 256;
        return pick;
    }
    // the weight-stationary recurrence and the small-batch GEMMs (64-row workgroups over T n and (T - skip) n rows:
    // 54 n and 50 n) only need a multiple of 32 sequences; an odd sequence length (fvad_nsnet2_forward takes any)
    // one of 64
    long c = (n + 31) / 32 * 32;
    if ((c * T) % 64 != 0 || (c * (T - skip)) % 64 != 0) c = (n + 63) / 64 * 64;
    if (!tn.reproducible && (!force || force[1] == '5' || force[1] == '6') && tn.gemm_kernel.empty() && c < 2048 &&
        std::min(gru_ws_cost(c, cu), gru_ws2_cost_both_layers(c, T, cu, tn.ws2_variant) / 108.0) < std::min(gru_cost(b, 0, cu), gru_cost(b, 4, cu)))
        return c;
    // the persistent GEMM takes 256-row panels of T n and of (T - skip) n rows: any multiple of 128 sequences at the
    // engine's T = 54 / 50, a multiple of 256 for an odd sequence length (fvad_nsnet2_forward takes any); `reproducible`
    // promises ONE kernel family, so there the batch is padded until the panels fit instead of changing family
    auto fits256 = [&](long np) { return (np * T) % 256 == 0 && (np * (T - skip)) % 256 == 0; };
    auto repro = [&](long np) { return (tn.reproducible && !fits256(np)) ? (np + 255) / 256 * 256 : np; };
    if (force || a == b) return repro(a);
    const double cost_a = std::min(std::min(gru_cost(a, 12, cu), gru_cost(a, 8, cu)), std::min(gru_cost(a, 4, cu), gru_cost(a, 0, cu)));
    const double cost_b = std::min(gru_cost(b, 8, cu), std::min(gru_cost(b, 4, cu), gru_cost(b, 0, cu)));
    return repro(cost_b <= cost_a ? b : a);
}

static GruChoice pick_gru(const fvad_ctx* ctx, long n_pad, int T, bool allow_v3)
{
    const char* force = ctx->tune.gru_kernel.empty() ? nullptr : ctx->tune.gru_kernel.c_str(); // "v3w12", "v3w8", "v3w4", "v4w8" (gru_lat), "v5w0" (gru_ws)
    if (force) {
        GruChoice c{force[1] - '0', atoi(force + 3)};
        if (c.version == 3 && !allow_v3) c = {4, 8}; // gru_rec3 needs the folded biases of the large-batch path
        if (c.version == 6 && (allow_v3 || !fvad_gru_ws2_ok(n_pad, T, ctx->n_cu, ctx->tune.ws2_variant))) c = {5, 0}; // the pipelined kernels belong to the small-batch sequence, up to 16 row tiles per group
        return c;
    }
    const int cu = ctx->n_cu;
    if (ctx->tune.reproducible && allow_v3 && n_pad % 64 == 0) {
        // one kernel family at every batch size: gru_rec3 (its 4-, 8- and 12-wave shapes run the same per-row chains)
        int w = 4;
        if (n_pad % 128 == 0 && gru_cost(n_pad, 8, cu) < gru_cost(n_pad, w, cu)) w = 8;
        if (n_pad % 192 == 0 && gru_cost(n_pad, 12, cu) < gru_cost(n_pad, w, cu)) w = 12;
        return {3, w};
    }
    int best = 0; // low-latency shape
    if (n_pad % 64 == 0 && gru_cost(n_pad, 4, cu) < gru_cost(n_pad, best, cu)) best = 4;
    if (n_pad % 128 == 0 && gru_cost(n_pad, 8, cu) < gru_cost(n_pad, best, cu)) best = 8;
    if (n_pad % 192 == 0 && gru_cost(n_pad, 12, cu) < gru_cost(n_pad, best, cu)) best = 12;
    if (!allow_v3) { // small-batch GEMM path only
        // per layer: 54 steps of gru_ws (+ layer 2's share of its input-projection GEMM, ~1.5k cycles a step)
        const double ws = 54.0 * gru_ws_cost(n_pad, cu), ws2 = gru_ws2_cost_both_layers(n_pad, T, cu, ctx->tune.ws2_variant);
        const double other = 54.0 * gru_cost(n_pad, best, cu);
        if (ws2 < 2.0 * std::min(ws, other) + 54.0 * 1.5e3) return {6, 0};
        if (ws < other) return {5, 0};
    }
    if (best == 0 || !allow_v3) return {4, 8};
    return {3, best};
}

// Deadline of one spin wait of a weight-stationary launch, in 100 MHz ticks: every wait of such a launch ends within the
// launch's own duration when its workgroups are co-resident, so 20 x the cost model's estimate of the whole launch (cycles at
// ~2.1 GHz = 21 cycles per tick), at least 2 ms, tells "not making progress" from "slow" with a wide margin -- and a launch
// that shares the GPU with another process's kernels gives up after milliseconds and takes the fallback, where a fixed
// 0.25 s deadline stalled a 0.4 ms push for 250 ms.
static unsigned long long ws_spin_deadline(const fvad_ctx* ctx, double est_cycles)
{
    if (!ctx->tune.ws_spin_auto) return ctx->tune.ws_spin_ticks;
    const double ticks = 20.0 * est_cycles / 21.0;
    return (unsigned long long)std::max(200000.0, std::min(ticks, 25000000.0));
}

// buffers of the weight-stationary recurrence, sized once for the largest batch that kernel takes (2560
// sequences: 8 MB of h exchange) so that nothing is allocated inside a stream capture
int ensure_gru_ws(fvad_ctx* ctx)
{
    Workspace& ws = ctx->ws;
    const size_t need = std::max(fvad_gru_ws_exchange_floats(2560), fvad_gru_ws2_exchange_floats(2304));
    if (!ws.hx) {
        FVAD_HIP(ctx, hipMalloc((void**)&ws.hx, need * sizeof(float)));
        ws.hx_cap = need;
        ws.generation++;
    }
    if (!ws.ws_sync) {
        FVAD_HIP(ctx, hipMalloc((void**)&ws.ws_sync, kWsSyncWords * sizeof(unsigned)));
        ws.generation++;
    }
    if (!ws.ws_fallbacks) {
        FVAD_HIP(ctx, hipMalloc((void**)&ws.ws_fallbacks, sizeof(unsigned long long)));
        FVAD_HIP(ctx, hipMemsetAsync(ws.ws_fallbacks, 0, sizeof(unsigned long long), ctx->stream));
        ws.generation++;
    }
    return FVAD_OK;
}

// the polled words (flags of both GRU layers, error word) are zeroed once per network pass -- by a kernel, not
// a memset: the launch sequence may be under capture, and a captured graph holds kernel nodes only
static int prepare_gru_ws(fvad_ctx* ctx, long n_pad)
{
    int rc = ensure_gru_ws(ctx);
    if (rc) return rc;
    if (std::max(fvad_gru_ws_exchange_floats(n_pad), fvad_gru_ws2_exchange_floats(n_pad)) > ctx->ws.hx_cap)
        return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "batch too large for gru_ws");
    // the pipelined recurrence's fallback launch leaves the words zeroed (sync_clean); a pass of gru_ws_kernel, a failed
    // pass, or a sequence under capture (a graph must not depend on what ran before it) starts from a reset
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &cap);
    if (!ctx->ws.sync_clean || cap != hipStreamCaptureStatusNone)
        fvad_launch_zero_words(ctx->ws.ws_sync, (int)kWsSyncWords, ctx->stream);
    ctx->ws.sync_clean = false;
    return FVAD_OK;
}

// Launches of the weight-stationary kernels spin on each other's flags, so two of them must not share the chip
// half-resident: within a process every such launch waits (on the GPU) for the previous one on the same device
template <class F> static int ws_serialised(fvad_ctx* ctx, F&& launch)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ctx->stream, &cap);
    const bool serialise = cap == hipStreamCaptureStatusNone && ctx->device >= 0 && ctx->device < 64;
    std::unique_lock<std::mutex> lk(g_ws_mu, std::defer_lock);
    if (serialise) {
        lk.lock();
        hipEvent_t& ev = g_ws_ev[ctx->device];
        if (!ev) { if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return -1; }
        else if (hipStreamWaitEvent(ctx->stream, ev, 0) != hipSuccess) return -1;
    }
    const int rc = launch();
    if (serialise && hipEventRecord(g_ws_ev[ctx->device], ctx->stream) != hipSuccess) return -1;
    return rc;
}

static int launch_gru(fvad_ctx* ctx, GruChoice c, const float* gi, const DevBuf& r_v2, const float* bR,
                      float* hout, long n_pad, int T, int layer, int tile_major)
{
    if (c.version == 4) {
        // more 16-sequence tiles than CUs: two or three row tiles per workgroup share one stream of R in fewer rounds of
        // workgroups (same bits; 8192 sequences: 4.03 against 4.43 ms per layer, 12288: 5.6 against gru_rec3<4>'s 6.7)
        int rt = 1;
        (void)gru_lat_cost(n_pad, ctx->n_cu, &rt);
        if (ctx->tune.gru_lat_tiles > 0) rt = (n_pad % (16 * ctx->tune.gru_lat_tiles) == 0) ? ctx->tune.gru_lat_tiles : 1;
        return fvad_launch_gru_lat(gi, r_v2.p, bR, hout, n_pad, T, nullptr, tile_major, ctx->stream, rt);
    }
    if (c.version == 5) {
        Workspace& ws = ctx->ws;
        unsigned* err = ws.ws_sync + 512;
        int rc = ws_serialised(ctx, [&] {
            return fvad_launch_gru_ws(gi, r_v2.p, bR, hout, ws.hx, ws.ws_sync + 256 * layer, err, n_pad, T, ctx->n_cu, tile_major,
                                      ws_spin_deadline(ctx, (double)T * gru_ws_cost(n_pad, ctx->n_cu)), ctx->stream);
        });
        if (rc) return rc;
        // fallback behind it: returns at once unless a workgroup of the launch above gave up waiting; the last layer
        // of a pass adds the error word to the context's fallback counter (fvad_ctx_ws_fallbacks)
        rc = fvad_launch_gru_lat(gi, r_v2.p, bR, hout, n_pad, T, err, tile_major, ctx->stream);
        if (rc == 0 && layer == 1) fvad_launch_count_word(ws.ws_fallbacks, err, ctx->stream);
        return rc;
    }
    if (c.waves <= 0 || n_pad % (16 * c.waves)) return -1;
    if (c.version == 3) return fvad_launch_gru_rec3(gi, r_v2.p, bR, hout, n_pad, T, c.waves, ctx->stream);
    return -1;
}

// NSNet2 of any dimensions (DeviceModel::generic): fc1 -> gi1 -> GRU1 -> gi2 -> GRU2 -> fc2 -> fc3 -> fc4 on the
// run-time-sized kernels; one kernel family, f32 MFMA throughout
static int run_nn_generic(fvad_ctx* ctx, long n_pad, int T, int skip)
{
    Workspace& ws = ctx->ws;
    const DeviceModel& m = ctx->dm;
    const DeviceModel::GenDims& g = m.gd;
    hipStream_t st = ctx->stream;
    const long rows = n_pad * T, rows_out = n_pad * (T - skip);
    if (n_pad % 32) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "batch not padded to 32 sequences");
    if ((size_t)rows > ws.cap_rows || (size_t)rows > ws.a1_cap_rows || (size_t)rows > ws.h_cap_rows)
        return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "NSNet2 workspace not allocated for this batch");
    auto S = [](int K) { return (K + 15) / 16; };
    int rc = 0;
    ctx->last_nn_path = "f32: panel_gemm<8> + gru_gen (model dims " + std::to_string(g.F1) + "/" + std::to_string(g.H) + "/" +
                        std::to_string(g.N2) + "/" + std::to_string(g.N3) + ")";
    time_begin(ctx, "fc1_gemm");
    rc |= fvad_launch_panel_gemm(ws.feat, kFeatStride, m.g_fc1_w.p, m.g_fc1_b.p, ws.a1, g.F1p, rows, 8, g.F1p / 128, S(161), FVAD_ACT_NONE, 0, 0, st);
    time_end(ctx);
    time_begin(ctx, "gru1_in_gemm");
    rc |= fvad_launch_panel_gemm(ws.a1, g.F1p, m.g_gi1_w.p, m.g_gi1_b.p, ws.gi, g.Gp, rows, 8, g.Gp / 128, S(g.F1), FVAD_ACT_NONE, 0, 0, st);
    time_end(ctx);
    time_begin(ctx, "gru1_rec");
    rc |= fvad_launch_gru_gen(ws.gi, g.Gp, m.g_r1.p, m.g_br1.p, ws.h1, g.Hp, n_pad, T, g.J, st);
    time_end(ctx);
    time_begin(ctx, "gru2_in_gemm");
    rc |= fvad_launch_panel_gemm(ws.h1, g.Hp, m.g_gi2_w.p, m.g_gi2_b.p, ws.gi, g.Gp, rows, 8, g.Gp / 128, g.J, FVAD_ACT_NONE, 0, 0, st);
    time_end(ctx);
    time_begin(ctx, "gru2_rec");
    rc |= fvad_launch_gru_gen(ws.gi, g.Gp, m.g_r2.p, m.g_br2.p, ws.h2, g.Hp, n_pad, T, g.J, st);
    time_end(ctx);
    time_begin(ctx, "fc2_gemm");
    rc |= fvad_launch_panel_gemm(ws.h2, g.Hp, m.g_fc2_w.p, m.g_fc2_b.p, ws.f2, g.N2p, rows_out, 8, g.N2p / 128, g.J, FVAD_ACT_RELU, skip ? T : 0, skip, st);
    time_end(ctx);
    time_begin(ctx, "fc3_gemm");
    rc |= fvad_launch_panel_gemm(ws.f2, g.N2p, m.g_fc3_w.p, m.g_fc3_b.p, ws.f3, g.N3p, rows_out, 8, g.N3p / 128, S(g.N2), FVAD_ACT_RELU, 0, 0, st);
    time_end(ctx);
    time_begin(ctx, "fc4_gemm");
    rc |= fvad_launch_panel_gemm(ws.f3, g.N3p, m.g_fc4_w.p, m.g_fc4_b.p, ws.gains, kFeatStride, rows_out, 11, 1, S(g.N3), FVAD_ACT_SIGMOID, 0, 0, st);
    time_end(ctx);
    if (rc) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "no kernel instance for this layer shape");
    FVAD_HIP(ctx, hipGetLastError());
    return FVAD_OK;
}

int run_nn(fvad_ctx* ctx, long n_pad, int T, int skip, long n_real)
{
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "NSNet2 weights not loaded");
    if (n_real <= 0 || n_real > n_pad) n_real = n_pad;
    if (ctx->dm.generic) return run_nn_generic(ctx, n_pad, T, skip);
    Workspace& ws = ctx->ws;
    const DeviceModel& m = ctx->dm;
    hipStream_t st = ctx->stream;
    const long rows = n_pad * T;
    const long rows_out = n_pad * (T - skip);
    int rc = 0;
    const Tuning& tn = ctx->tune;
    const char* force = tn.gemm_kernel.empty() ? nullptr : tn.gemm_kernel.c_str(); // "v1" (small-batch GEMM) / "v3" / "v3nofold"
    const int math = nn_math_effective(ctx);
    const bool h3 = math == FVAD_NN_MATH_F16X3, b3 = math == FVAD_NN_MATH_BF16X3;
    if ((h3 || b3) && n_pad % 128) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "the emulated kernels need a batch padded to 128 sequences");
    {   // the buffers this arithmetic writes were sized by ensure_workspace for this very launch; a mismatch is a bug, not a reason to write past them
        const size_t r = (size_t)rows;
        const bool nofold = force && strstr(force, "nofold");
        if (r > ws.cap_rows || (h3 ? r > ws.hs_cap_rows : r > ws.h_cap_rows) || (nofold && r > ws.a1_cap_rows))
            return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "NSNet2 workspace not allocated for this arithmetic / batch");
    }
    if (b3) {
        // bf16x3: the five dense layers as six bf16 MFMAs per product on exact three-piece splits (kernels_b3.hip), the
        // two recurrences on the f32 matrix cores (gru_rec3, which writes h a second time as three-piece fragments)
        if (!ws.b3_hs1 || (size_t)n_pad * (size_t)T > ws.b3_cap_rows) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "bf16x3 workspace not allocated");
        auto gemm_b3 = [&](const float* A, int in_ts, int a_ld, const DevBuf& W, const float* b, float* Cc, int out, int c_ld, int seq_T,
                           long row_tiles, int nt, int nblk, int K, int act, int valid, int mT, int mskip) {
            return fvad_launch_panel_gemm_b3(A, in_ts, a_ld, W.p, b, Cc, out, c_ld, seq_T, row_tiles, nt, nblk, K, act, valid, mT, mskip, ctx->n_cu, st);
        };
        const int waves = n_pad % 192 == 0 ? 12 : 8;
        const long G = n_pad / 16;
        ctx->last_nn_path = std::string("bf16x3: panel_gemm_b3 (fc1 folded) + gru_rec3<") + std::to_string(waves) + "> (f32 recurrences)";
        time_begin(ctx, "gru1_in_gemm_fc1folded");
        rc |= gemm_b3(ws.feat, 0, kFeatStride, m.gi1f_b3, m.gi1f_bzr.p, ws.gi, 0, 1200, T, G * T, 15, 5, 161, FVAD_ACT_NONE, 75, 0, 0);
        time_end(ctx);
        time_begin(ctx, "gru1_rec");
        rc |= fvad_launch_gru_rec3(ws.gi, m.r1v2.p, m.br1.p, ws.h1, n_pad, T, waves, st, ws.b3_hs1);
        time_end(ctx);
        time_begin(ctx, "gru2_in_gemm");
        rc |= gemm_b3(ws.b3_hs1, 1, 13, m.gi2_b3, m.gi2_bzr.p, ws.gi, 0, 1200, T, G * T, 15, 5, 400, FVAD_ACT_NONE, 75, 0, 0);
        time_end(ctx);
        time_begin(ctx, "gru2_rec");
        rc |= fvad_launch_gru_rec3(ws.gi, m.r2v2.p, m.br2.p, ws.h2, n_pad, T, waves, st, ws.b3_hs2);
        time_end(ctx);
        time_begin(ctx, "fc2_gemm");
        rc |= gemm_b3(ws.b3_hs2, 1, 13, m.fc2_b3, m.fc2h3_b.p, ws.b3_f2, 2, 19, T - skip, G * (T - skip), 10, 4, 400, FVAD_ACT_RELU, 38, skip ? T : 0, skip);
        time_end(ctx);
        time_begin(ctx, "fc3_gemm");
        rc |= gemm_b3(ws.b3_f2, 1, 19, m.fc3_b3, m.fc3h3_b.p, ws.b3_f3, 2, 19, T - skip, G * (T - skip), 10, 4, 600, FVAD_ACT_RELU, 38, 0, 0);
        time_end(ctx);
        time_begin(ctx, "fc4_gemm");
        rc |= gemm_b3(ws.b3_f3, 1, 19, m.fc4_b3, m.fc4h3_b.p, ws.gains, 0, kFeatStride, T - skip, G * (T - skip), 12, 1, 600, FVAD_ACT_SIGMOID, 11, 0, 0);
        time_end(ctx);
        if (rc) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "no kernel instance for this layer shape");
        FVAD_HIP(ctx, hipGetLastError());
        return FVAD_OK;
    }
    const bool big = h3 || (force ? force[1] != '1' : (tn.reproducible || n_pad >= 2048));
    if (tn.reproducible && !force && (rows % 256 || rows_out % 256))
        return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "reproducible: batch not padded to the persistent GEMM's 256-row panels");
    if (big && rows % 256 == 0 && rows_out % 256 == 0) {
        // The persistent kernel: one workgroup per CU walking all (row panel, column block) items; 15-, 13- and
        // 11-tile column blocks.  K is the true reduction length (S super-steps of 16 cover it, zero-padded).
        auto gemm = [&](const float* A, int lda, const float* W, const float* b, float* Cc, int ldc, long r, int nt,
                        int nblk, int S, int K, int act, int valid, int mT, int mskip) {
            return fvad_launch_panel_gemm3(A, lda, W, b, Cc, ldc, r, nt, nblk, S, K, act, valid, mT, mskip, ctx->n_cu, st);
        };
        // f16x3 path (kernels_h3.hip): its intermediates (gi, h1, h2, f2, f3) are in the tiled layout, row tiles of
        // 16 sequences at one time step; the features come in and the gains go out row-major
        auto gemm_h3 = [&](const float* A, int in_ts, int a_ld, const DevBuf& W, const DeviceModel::H3Scale& sc, const float* b,
                           float* Cc, int out, int c_ld, int seq_T, long row_tiles, int nt, int nblk, int K, int act,
                           int valid, int mT, int mskip, float out_sx) {
            return fvad_launch_panel_gemm_h3(A, in_ts, a_ld, W.p, b, Cc, out, c_ld, seq_T, row_tiles, nt, nblk, K, act, valid,
                                             mT, mskip, sc.sx, sc.sw, out_sx, ctx->n_cu, st);
        };
        if (h3) {
            int waves = n_pad % 192 == 0 ? 12 : 8;
            if ((tn.h3_waves == 8 || tn.h3_waves == 12) && n_pad % (16 * tn.h3_waves) == 0) waves = tn.h3_waves;
            ctx->last_nn_path = std::string("f16x3: panel_gemm_h3 + gru_rec_h3<") + std::to_string(waves) + ">";
            const long G = n_pad / 16;
            // gi: tiled f32; hs1 / hs2 / f2 / f3: split tiled, scaled for the layer that reads them
            time_begin(ctx, "gru1_in_gemm_fc1folded");
            rc |= gemm_h3(ws.feat, 0, kFeatStride, m.gi1f_h3, m.h3_gi1f, m.gi1f_bzr.p, ws.gi, 1, 75, T, G * T, 15, 5, 161, FVAD_ACT_NONE, 75, 0, 0, 1.0f);
            time_end(ctx);
            time_begin(ctx, "gru1_rec");
            rc |= fvad_launch_gru_rec_h3(ws.gi, m.r1_h3.p, m.br1.p, ws.hs1, n_pad, T, waves, m.h3_r1.sx, m.h3_r1.sw, st);
            time_end(ctx);
            time_begin(ctx, "gru2_in_gemm");
            rc |= gemm_h3(ws.hs1, 1, 13, m.gi2_h3, m.h3_gi2, m.gi2_bzr.p, ws.gi, 1, 75, T, G * T, 15, 5, 400, FVAD_ACT_NONE, 75, 0, 0, 1.0f);
            time_end(ctx);
            time_begin(ctx, "gru2_rec");
            rc |= fvad_launch_gru_rec_h3(ws.gi, m.r2_h3.p, m.br2.p, ws.hs2, n_pad, T, waves, m.h3_r2.sx, m.h3_r2.sw, st);
            time_end(ctx);
            time_begin(ctx, "fc2_gemm");
            rc |= gemm_h3(ws.hs2, 1, 13, m.fc2_h3, m.h3_fc2, m.fc2h3_b.p, ws.f2, 2, 19, T - skip, G * (T - skip), 10, 4, 400, FVAD_ACT_RELU, 38, skip ? T : 0, skip, m.h3_fc3.sx);
            time_end(ctx);
            time_begin(ctx, "fc3_gemm");
            rc |= gemm_h3(ws.f2, 1, 19, m.fc3_h3, m.h3_fc3, m.fc3h3_b.p, ws.f3, 2, 19, T - skip, G * (T - skip), 10, 4, 600, FVAD_ACT_RELU, 38, 0, 0, m.h3_fc4.sx);
            time_end(ctx);
            time_begin(ctx, "fc4_gemm");
            rc |= gemm_h3(ws.f3, 1, 19, m.fc4_h3, m.h3_fc4, m.fc4h3_b.p, ws.gains, 0, kFeatStride, T - skip, G * (T - skip), 12, 1, 600, FVAD_ACT_SIGMOID, 11, 0, 0, 1.0f);
            time_end(ctx);
            if (rc) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "no kernel instance for this layer shape");
            FVAD_HIP(ctx, hipGetLastError());
            return FVAD_OK;
        }
        const bool fold = !(force && strstr(force, "nofold"));
        const GruChoice gc = pick_gru(ctx, n_pad, T, fold);
        ctx->last_nn_path = std::string("f32: panel_gemm3") + (fold ? " (fc1 folded)" : "") + " + " +
                            (gc.version == 3 ? "gru_rec3<" + std::to_string(gc.waves) + ">" : gc.version == 5 ? std::string("gru_ws") : std::string("gru_lat"));
        if (gc.version == 5 && (rc = prepare_gru_ws(ctx, n_pad))) return rc; // only when forced: tuning / tests
        const bool bzr = gc.version == 3;
        if (fold) {
            time_begin(ctx, "gru1_in_gemm_fc1folded");
            rc |= gemm(ws.feat, kFeatStride, m.gi1f_w.p, bzr ? m.gi1f_bzr.p : m.gi1f_b.p, ws.gi, 1200, rows, 15, 5, 11, 161, FVAD_ACT_NONE, 75, 0, 0);
            time_end(ctx);
        } else {
            time_begin(ctx, "fc1_gemm");
            rc |= fvad_launch_panel_gemm(ws.feat, kFeatStride, m.fc1_w.p, m.fc1_b.p, ws.a1, 400, rows, 25, 1, 11, FVAD_ACT_NONE, 0, 0, st);
            time_end(ctx);
            time_begin(ctx, "gru1_in_gemm");
            rc |= gemm(ws.a1, 400, m.gi1v2_w.p, m.gi1_btm.p, ws.gi, 1200, rows, 15, 5, 25, 400, FVAD_ACT_NONE, 75, 0, 0);
            time_end(ctx);
        }
        time_begin(ctx, "gru1_rec");
        rc |= launch_gru(ctx, gc, ws.gi, m.r1v2, m.br1.p, ws.h1, n_pad, T, 0, 1);
        time_end(ctx);
        time_begin(ctx, "gru2_in_gemm");
        rc |= gemm(ws.h1, 400, m.gi2v2_w.p, bzr ? m.gi2_bzr.p : m.gi2_btm.p, ws.gi, 1200, rows, 15, 5, 25, 400, FVAD_ACT_NONE, 75, 0, 0);
        time_end(ctx);
        time_begin(ctx, "gru2_rec");
        rc |= launch_gru(ctx, gc, ws.gi, m.r2v2, m.br2.p, ws.h2, n_pad, T, 1, 1);
        time_end(ctx);
        time_begin(ctx, "fc2_gemm");
        rc |= gemm(ws.h2, 400, m.fc2v3_w.p, m.fc2v3_b.p, ws.f2, 608, rows_out, 13, 3, 25, 400, FVAD_ACT_RELU, 38, skip ? T : 0, skip);
        time_end(ctx);
        time_begin(ctx, "fc3_gemm");
        rc |= gemm(ws.f2, 608, m.fc3v3_w.p, m.fc3v3_b.p, ws.f3, 608, rows_out, 13, 3, 38, 600, FVAD_ACT_RELU, 38, 0, 0);
        time_end(ctx);
        time_begin(ctx, "fc4_gemm");
        rc |= gemm(ws.f3, 608, m.fc4_w.p, m.fc4_b.p, ws.gains, kFeatStride, rows_out, 11, 1, 38, 600, FVAD_ACT_SIGMOID, 11, 0, 0);
        time_end(ctx);
        if (rc) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "no kernel instance for this layer shape");
        FVAD_HIP(ctx, hipGetLastError());
        return FVAD_OK;
    }
    // ---- small batches: a handful of 64-row panels per launch, so every layer is cut into narrow column blocks
    // (2 tiles up to 2048 rows, 4 above: the same arithmetic, more and lighter workgroups) with loads several phases
    // ahead (panel_gemm_s_kernel); fc1 is folded into the first GRU's input projection like in the large-batch
    // family; gi rows are tile-major
    const int fam = rows > 2048 ? 1 : 0, snt = fam ? 4 : 2;
    const int nb_gi = (75 + snt - 1) / snt, nb_fc = (38 + snt - 1) / snt; // (fc4 runs in 2-tile blocks whatever the family: below)
    const GruChoice gcs = pick_gru(ctx, n_pad, T, false);
    // one row tile per group (up to 96 sequences: BASELINE config 3's 82 chunks, every live push): the pipelined kernel computes
    // layer 1's input projection too, and the GEMM launch in front of it disappears
    const bool gi1_in_kernel = gcs.version == 6 && fvad_gru_ws2_gi1_in_kernel(n_pad, T, ctx->n_cu, tn.ws2_variant);
    if (!gi1_in_kernel) {
        time_begin(ctx, "gru1_in_gemm_fc1folded");
        rc |= fvad_launch_panel_gemm_s(ws.feat, kFeatStride, m.s_gi1f_w[fam].p, m.gi1f_b.p, ws.gi, 1200, rows, snt, nb_gi, 11, FVAD_ACT_NONE, 0, 0, st, 75);
        time_end(ctx);
    }
    if (gcs.version >= 5 && (rc = prepare_gru_ws(ctx, n_pad))) return rc;
    ctx->last_nn_path = std::string("f32: panel_gemm (fc1 folded) + ") + (gcs.version == 6 ? fvad_gru_ws2_kernel_name(n_pad, T, ctx->n_cu, tn.ws2_variant) :
                        gcs.version == 5 ? "gru_ws" : "gru_lat");
    if (gcs.version == 6) {
        // both GRU layers in one launch, layer 2 a step behind layer 1, its input projection computed inside
        unsigned* err = ws.ws_sync + 512;
        time_begin(ctx, "gru12_rec_pipelined");
        rc |= ws_serialised(ctx, [&] {
            return fvad_launch_gru_ws2(ws.gi, ws.feat, m.s_w1frag.p, m.gi1f_b.p, m.r1v2.p, m.br1.p, m.s_w2frag.p, m.s_bw2.p, m.r2v2.p, m.br2.p, ws.h2,
                                       ws.hx, ws.ws_sync, err, n_pad, T, ctx->n_cu, ws_spin_deadline(ctx, gru_ws2_cost_both_layers(n_pad, T, ctx->n_cu, tn.ws2_variant) * T / 55.0), tn.ws2_variant,
                                       tn.ws2_waits ? tn.ws2_waits : tn.ws2_waits_cal[fvad_gru_ws2_wait_class(n_pad, T, ctx->n_cu, tn.ws2_variant)], st,
                                       ws.ws_sync + kWsLocalAt);
        });
        // one launch behind it: the whole fallback (layer 1, layer 2's input projection, layer 2 -- run only if the
        // error word was raised), the pass count, and the reset of the polled words for the next pass
        rc |= fvad_launch_gru_ws2_fallback(ws.gi, gi1_in_kernel ? ws.feat : nullptr, m.s_gi1f_w[0].p, m.gi1f_b.p, m.r1v2.p, m.br1.p, m.s_gi2_w[0].p,
                                           m.gi2_btm.p, m.r2v2.p, m.br2.p, ws.h1, ws.h2, n_pad, T, ws.ws_sync, ws.ws_fallbacks, st, (int)kWsLocalAt, kWs2LocalWords);
        {   // the launch above leaves the words zeroed -- once it has RUN: a sequence under capture has not
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(st, &cap);
            if (rc == 0 && cap == hipStreamCaptureStatusNone) ws.sync_clean = true;
        }
        time_end(ctx);
    } else {
        time_begin(ctx, "gru1_rec");
        rc |= launch_gru(ctx, gcs, ws.gi, m.r1v2, m.br1.p, ws.h1, n_pad, T, 0, 1);
        time_end(ctx);
        time_begin(ctx, "gru2_in_gemm");
        rc |= fvad_launch_panel_gemm_s(ws.h1, 400, m.s_gi2_w[fam].p, m.gi2_btm.p, ws.gi, 1200, rows, snt, nb_gi, 25, FVAD_ACT_NONE, 0, 0, st, 75);
        time_end(ctx);
        time_begin(ctx, "gru2_rec");
        rc |= launch_gru(ctx, gcs, ws.gi, m.r2v2, m.br2.p, ws.h2, n_pad, T, 1, 1);
        time_end(ctx);
    }
    // fc2 .. fc4 over the rows of the REAL sequences only (nobody reads a padded batch's other rows: K3 and the callers of
    // fvad_nsnet2_forward take the real ones): 82 chunks = 4100 rows in 65 panels instead of 4800 in 75, a one-chunk push one
    // panel instead of 25.  (Measured: nothing at 82 chunks -- 650 or 750 workgroups are three per CU on the fullest CUs either
    // way -- but a one-chunk push's three launches lose their 450 idle workgroups each.)
    GemmRowMap r2{}, r3{};
    r2.L = r3.L = T - skip;
    r2.n_rows = r3.n_rows = (int)(n_real * (T - skip));
    r2.in_T = T; r2.in_t0 = skip; r2.out_T = T - skip; r2.out_t0 = 0; // fc2 reads h2 rows [sequence][T], writes rows [sequence][T - skip]
    r3.in_T = r3.out_T = T - skip;
    (void)rows_out;
    time_begin(ctx, "fc2_gemm");
    rc |= fvad_launch_panel_gemm_s_rows(ws.h2, 400, m.s_fc2_w[fam].p, m.fc2_b.p, ws.f2, 640, r2, snt, nb_fc, 25, FVAD_ACT_RELU, st);
    time_end(ctx);
    time_begin(ctx, "fc3_gemm");
    rc |= fvad_launch_panel_gemm_s_rows(ws.f2, 640, m.s_fc3_w[fam].p, m.fc3_b.p, ws.f3, 640, r3, snt, nb_fc, 38, FVAD_ACT_RELU, st);
    time_end(ctx);
    time_begin(ctx, "fc4_gemm");
    // fc4 has 11 tiles: in 4-tile blocks that is 3 workgroups per panel, each a chain of 608 MFMAs per wavefront on a chip that
    // is otherwise idle (82 chunks: 195 workgroups); 2-tile blocks are twice the workgroups at half the chain (15 -> 10 us there)
    rc |= fvad_launch_panel_gemm_s_rows(ws.f3, 640, m.s_fc4_w[0].p, m.s_fc4_b.p, ws.gains, kFeatStride, r3, 2, 6, 38, FVAD_ACT_SIGMOID, st, 11);
    time_end(ctx);
    if (rc) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "no kernel instance for this layer shape");
    FVAD_HIP(ctx, hipGetLastError());
    return FVAD_OK;
}

// Context option ws2_calibrate: gru_ws2k's first-poll waits measured on THIS device.  The built-in table was swept on one box
// (tools/ws2_delay.py); what the right wait is depends on how long a flag takes to cross the fabric, which is a property of
// the part and its clocks.  A 5 x 5 grid around the table's entry (both layers' waits, +-0.8 us in steps of 0.4: the optimum is
// a narrow diagonal valley, which a search along one axis at a time misses), then 3 x 3 in steps of 0.2 around its best; the network pass (run_nn: the product path, on whatever the workspace holds) timed five times per candidate,
// fastest run kept.  A candidate replaces the table's entry only if it is more than 1.5 % faster -- run-to-run noise is about
// 1 %.  Timing only: results do not depend on the waits (tests/test_gpu.py checks bits with and without).
// About 80 ms per class; classes measured: the one a one-chunk push falls in and the one BASELINE config 3's 82 chunks fall in.
int calibrate_ws2_waits(fvad_ctx* ctx)
{
    Tuning& tn = ctx->tune;
    if (!ctx->dm.loaded || ctx->dm.generic)
        return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "ws2_calibrate: load an NSNet2 model of the baseline shape first");
    const int T = kRowsPerChunk;
    int rc = ensure_workspace(ctx, 96, T, kWarmupRows);
    if (rc) return rc;
    if ((rc = ensure_gru_ws(ctx))) return rc;
    Workspace& ws = ctx->ws;
    FVAD_HIP(ctx, hipMemsetAsync(ws.feat, 0, (size_t)96 * T * kFeatStride * sizeof(float), ctx->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FVAD_HIP(ctx, hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return set_err(ctx, FVAD_ERR_HIP, "hipEventCreate"); }
    const unsigned saved = tn.ws2_waits;
    const std::string saved_path = ctx->last_nn_path;
    auto time_pass = [&](long n_pad, unsigned waits, float* best) -> int {
        tn.ws2_waits = waits;
        *best = 1e30f;
        for (int i = 0; i < 6; i++) { // the first is a warm-up
            if (hipEventRecord(e0, ctx->stream) != hipSuccess) return FVAD_ERR_HIP;
            const int r = run_nn(ctx, n_pad, T, kWarmupRows);
            if (r) return r;
            if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return FVAD_ERR_HIP;
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (i) *best = std::min(*best, ms);
        }
        return FVAD_OK;
    };
    bool done[4] = {false, false, false, false};
    for (long n : {1L, 82L}) {
        const long n_pad = padded_batch(ctx, n, T, kWarmupRows);
        const int cls = fvad_gru_ws2_wait_class(n_pad, T, ctx->n_cu, tn.ws2_variant);
        if (cls == 0 || done[cls] || pick_gru(ctx, n_pad, T, false).version != 6) continue;
        done[cls] = true;
        const unsigned base = fvad_gru_ws2_builtin_waits(cls, fvad_gru_ws2_local_layer1(n_pad, T, ctx->n_cu, tn.ws2_variant));
        float t_base = 0;
        if ((rc = time_pass(n_pad, base, &t_base))) break;
        unsigned best = base;
        float t_best = t_base;
        for (int d1 = -80; d1 <= 80 && !rc; d1 += 40)
            for (int d2 = -80; d2 <= 80 && !rc; d2 += 40) {
                const int l1 = (int)(base & 0xFFFFu) + d1, l2 = (int)(base >> 16) + d2;
                if ((d1 == 0 && d2 == 0) || l1 < 0 || l2 < 0) continue;
                float t = 0;
                rc = time_pass(n_pad, (unsigned)l1 | ((unsigned)l2 << 16), &t);
                if (!rc && t < t_best) { t_best = t; best = (unsigned)l1 | ((unsigned)l2 << 16); }
            }
        {   // a finer 3 x 3 (+-0.2 us) around the coarse grid's best
            const unsigned centre = best;
            for (int d1 = -20; d1 <= 20 && !rc; d1 += 20)
                for (int d2 = -20; d2 <= 20 && !rc; d2 += 20) {
                    const int l1 = (int)(centre & 0xFFFFu) + d1, l2 = (int)(centre >> 16) + d2;
                    if ((d1 == 0 && d2 == 0) || l1 < 0 || l2 < 0) continue;
                    float t = 0;
                    rc = time_pass(n_pad, (unsigned)l1 | ((unsigned)l2 << 16), &t);
                    if (!rc && t < t_best) { t_best = t; best = (unsigned)l1 | ((unsigned)l2 << 16); }
                }
        }
        if (!rc && best != base) { // the winner against the table once more, back to back: keep it only on a clear margin
            float t0 = 0, t1 = 0;
            if (!(rc = time_pass(n_pad, base, &t0)) && !(rc = time_pass(n_pad, best, &t1)) && t1 < 0.985f * t0) tn.ws2_waits_cal[cls] = best;
        }
    }
    tn.ws2_waits = saved;
    ctx->last_nn_path = saved_path;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

// Chunks per launch of a call over `total` chunks; max_chunks <= 0: the caller leaves it to the engine.
// Launch planning (only then, f32, default kernel selection): between the largest batch the pipelined recurrence takes
// (1536 chunks: 16 row tiles per group) and ~3400 chunks ONE launch would fall to the large-batch family's low-latency
// recurrence, which keeps 128-210 of the 256 CUs busy (2048 chunks: 7.4 ms = 13.8 M frames/s); two or three equal launches of
// at most 1536 chunks stay on the pipelined kernels (2 x 2.77 ms = 18.5 M frames/s).  Measured crossover (bench.py's batch
// curve): 4.86 ms + 1.26 us per chunk against 2.69 us per chunk.
// ... and above that the launch sizes that fill the chip are few: 4096 chunks are one round of gru_lat's 256 workgroups, 8192 of
// gru_lat2's, 12288 of gru_lat3's, 16384 of gru_rec3<4>'s, 32768 of <8>'s, 49152 of <12>'s -- a launch just above one of them pays
// a whole further round of the recurrence (tools/batch_sizes.py, ms per call: 4096 chunks 9.98, 5120 14.9, 8192 18.6, 9216 23.3,
// 12288 26.8, 13312 30.4, 16384 34.1, 20480 50.8, 32768 65.8).  A call whose size the caller leaves to the engine is therefore cut
// into launches of such a size and a remainder, when the measured curve says that pays by more than 4 %: 5120 = 4096 + 1024
// (12.8 ms for 14.9), 9216 = 8192 + 1024, 20480 = 16384 + 4096 (44.1 for 50.8).
static double launch_ms(double n) // one launch of n chunks, device-resident, default kernel selection (one MI355X box)
{
    struct Seg { double n0, t0, n1, t1; };
    static const Seg segs[] = {{0, 0.30, 1536, 3.97},          // pipelined weight-stationary recurrence
                               {1536, 6.7, 4096, 9.98},        // gru_lat, 97..256 workgroups: the recurrence costs what 4096 chunks cost
                               {4096, 13.9, 8192, 18.59},      // gru_lat2
                               {8192, 22.2, 12288, 26.77},     // gru_lat3
                               {12288, 29.1, 16384, 34.1},     // gru_rec3<4>
                               {16384, 45.9, 32768, 65.78},    // gru_rec3<8>
                               {32768, 77.1, 49152, 97.3}};    // gru_rec3<12> (its lower end is an estimate)
    for (const Seg& g : segs)
        if (n <= g.n1) return g.t0 + (g.t1 - g.t0) * (n - g.n0) / (g.n1 - g.n0);
    return 97.3 * n / 49152.0;
}
// cost of `total` chunks under plan_launches' rules, and the plan itself (sizes in launch order, the remainder last)
static double plan_cost(long total, long limit, std::vector<long>* plan)
{
    if (total <= 0) return 0.0;
    if (total <= 1536 && total <= limit) { if (plan) plan->push_back(total); return launch_ms((double)total); }
    if (total <= 3400 && limit >= 1536) { // two or three equal launches on the pipelined kernels (the crossover above)
        const long k = (total + 1535) / 1536, each = ((total + k - 1) / k + 15) / 16 * 16;
        double c = 0;
        for (long left = total; left > 0; left -= each) { const long n = std::min(left, each); c += launch_ms((double)n); if (plan) plan->push_back(n); }
        return c;
    }
    static const long sizes[] = {49152, 32768, 16384, 12288, 8192, 4096};
    double best = total <= limit ? launch_ms((double)total) : 1e30;
    long best_size = total <= limit ? total : 0;
    for (long sz : sizes) {
        if (sz > limit || sz >= total) continue;
        const double c = (double)(total / sz) * launch_ms((double)sz) + plan_cost(total % sz, limit, nullptr);
        if (c < (best_size == total ? 0.96 * best : best)) { best = c; best_size = sz; }
    }
    if (best_size == 0) { // the caller's limit is below every size that fills the chip: launches of the limit
        best_size = limit;
        best = (double)(total / limit) * launch_ms((double)limit) + plan_cost(total % limit, limit, nullptr);
    }
    if (plan) {
        if (best_size == total) plan->push_back(total);
        else {
            for (long i = 0; i < total / best_size; ++i) plan->push_back(best_size);
            plan_cost(total % best_size, limit, plan);
        }
    }
    return best;
}

// the launches of a call of `total` chunks: sizes in order.  max_chunks > 0: the caller's (or a test's) limit -- launches of that
// size and a remainder, as ever; otherwise the plan above (default f32 kernel selection only: `reproducible`, forced kernels,
// the emulations and run-time-sized models keep launches of the context's max_chunks)
void plan_launches(const fvad_ctx* ctx, long total, long max_chunks, std::vector<long>& plan)
{
    plan.clear();
    if (total <= 0) return;
    const Tuning& tn = ctx->tune;
    const bool planned = max_chunks <= 0 && !ctx->dm.generic && nn_math_effective(ctx) == FVAD_NN_MATH_F32 && !tn.reproducible &&
                         tn.gru_kernel.empty() && tn.gemm_kernel.empty();
    const long limit = max_chunks > 0 ? max_chunks : tn.max_chunks;
    if (planned) { plan_cost(total, limit, &plan); return; }
    for (long left = total; left > 0; left -= limit) plan.push_back(std::min(left, limit));
}

long planned_max_chunks(const fvad_ctx* ctx, long total, long max_chunks) // the largest launch of that plan
{
    std::vector<long> plan;
    plan_launches(ctx, total, max_chunks, plan);
    long m = 0;
    for (long n : plan) m = std::max(m, n);
    return m > 0 ? m : (max_chunks > 0 ? max_chunks : ctx->tune.max_chunks);
}

// K1 -> NSNet2 -> K3 over every chunk of every job, in launches of <= max_chunks chunks.
int run_chunks(fvad_ctx* ctx, std::vector<LaneJob>& jobs, long max_chunks, ChunkDesc* capture_descs, ChunkDesc* capture_dev, long* n_launches)
{
    if (n_launches) *n_launches = 0;
    // capture_descs != nullptr: the call is being captured into a hipGraph.  Every launch gets its own
    // region of the graph's private descriptor table (host copy capture_descs, device copy capture_dev,
    // uploaded once by the caller after the capture): the graph holds no copy node and does not depend on
    // the workspace's shared table, which direct calls overwrite.  No event is waited for or recorded.
    size_t capture_off = 0;
    long total = 0;
    for (auto& j : jobs) total += (long)j.n_chunks;
    if (total == 0) return FVAD_OK;
    std::vector<long> plan;
    plan_launches(ctx, total, max_chunks, plan);
    int rc = ensure_workspace_plan(ctx, plan);
    if (rc) return rc;
    Workspace& ws = ctx->ws;
    size_t launch_i = 0;

    size_t job = 0, chunk_in_job = 0;
    while (job < jobs.size()) {
        // fill one launch, lane-contiguous
        const long cap = std::min<long>(launch_i < plan.size() ? plan[launch_i] : plan.back(), ws.cap_chunks);
        ++launch_i;
        long n = 0;
        std::vector<size_t> touched;
        struct Tap { size_t job, chunk0, count; long batch0; };
        std::vector<Tap> taps;
        // pinned descriptor table: two slots, so the host can build the next launch while the GPU still
        // runs this one; a slot is free once its (tiny) upload has been consumed
        const int slot = ws.desc_slot;
        ChunkDesc* hd;
        if (capture_descs) hd = capture_descs + capture_off;
        else {
            ws.desc_slot ^= 1;
            FVAD_HIP(ctx, hipEventSynchronize(ws.desc_ev[slot]));
            hd = ws.h_descs + (size_t)slot * (size_t)ws.cap_chunks;
        }
        size_t j = job, c = chunk_in_job;
        while (j < jobs.size() && n < cap) {
            LaneJob& lj = jobs[j];
            if (lj.n_chunks == 0) { ++j; c = 0; continue; }
            const size_t take = std::min<size_t>(lj.n_chunks - c, (size_t)(cap - n));
            if (lj.h_spec || lj.h_feat) taps.push_back({j, c, take, n});
            for (size_t k = 0; k < take; ++k) {
                ChunkDesc& d = hd[n + (long)k];
                d.in = lj.d_in ? lj.d_in + (c + k) * (size_t)kChunk48 : nullptr;
                d.in16 = lj.d_in16 ? lj.d_in16 + (c + k) * (size_t)kChunk48 : nullptr;
                d.den = lj.d_den + (c + k) * (size_t)kChunk48;
                d.den16 = lj.d_den16 ? lj.d_den16 + (c + k) * (size_t)kChunk48 : nullptr;
                d.carry_in = lj.carry[lj.cur];
                d.carry_out = lj.carry[lj.cur ^ 1];
                d.first = (k == 0);
                d.last = (k + 1 == take);
                d.rms = lj.d_rms ? lj.d_rms + (c + k) : nullptr;
            }
            touched.push_back(j);
            n += (long)take;
            c += take;
            if (c == lj.n_chunks) { ++j; c = 0; }
        }
        const ChunkDesc* dd = ws.descs;
        if (capture_descs) { dd = capture_dev + capture_off; capture_off += (size_t)n; }
        else if (ws.descs_mirror.size() < (size_t)n || memcmp(ws.descs_mirror.data(), hd, (size_t)n * sizeof(ChunkDesc)) != 0) {
            // the stream orders this copy after the previous launch's kernels
            FVAD_HIP(ctx, hipMemcpyAsync(ws.descs, hd, (size_t)n * sizeof(ChunkDesc), hipMemcpyHostToDevice, ctx->stream));
            FVAD_HIP(ctx, hipEventRecord(ws.desc_ev[slot], ctx->stream));
            ws.descs_mirror.assign(hd, hd + n);
        } // else: the device table already holds exactly these descriptors (the previous launch's: a steady-state loop)
        time_begin(ctx, "stft320_logpow");
        // a launch of a few chunks leaves most CUs idle and a chunk's frames are a latency chain on one workgroup:
        // cut them over 2 or 3 workgroups per chunk (the same instructions per frame: the same bits)
        const int fft_parts = n <= 85 ? 3 : (n <= 128 ? 2 : 1);
        fvad_launch_stft(dd, (int)n, ctx->tb, ws.feat, ws.spec, ctx->stream, fft_parts);
        time_end(ctx);
        // parity taps: K1's own outputs (the buffers the network and K3 read), straight to the caller
        for (const Tap& t : taps) {
            const LaneJob& lj = jobs[t.job];
            if (lj.h_spec)
                FVAD_HIP(ctx, hipMemcpyAsync(lj.h_spec + t.chunk0 * (size_t)(kFramesPerChunk * kNBins * 2),
                                             ws.spec + (size_t)t.batch0 * (kFramesPerChunk * kNBins * 2),
                                             t.count * (size_t)(kFramesPerChunk * kNBins * 2) * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
            if (lj.h_feat)
                FVAD_HIP(ctx, hipMemcpy2DAsync(lj.h_feat + t.chunk0 * (size_t)(kRowsPerChunk * kNBins), kNBins * sizeof(float),
                                               ws.feat + (size_t)t.batch0 * (kRowsPerChunk * kFeatStride), kFeatStride * sizeof(float),
                                               kNBins * sizeof(float), t.count * (size_t)kRowsPerChunk, hipMemcpyDeviceToHost, ctx->stream));
        }
        const long n_pad = padded_batch(ctx, n, kRowsPerChunk, kWarmupRows);
        rc = run_nn(ctx, n_pad, kRowsPerChunk, kWarmupRows, n);
        if (rc) return rc;
        time_begin(ctx, "istft320_ola_up3");
        fvad_launch_istft(dd, (int)n, ctx->tb, ws.spec, ws.gains, kFramesPerChunk, 0, ctx->stream, fft_parts);
        time_end(ctx);
        for (size_t t : touched) jobs[t].cur ^= 1;
        if (n_launches) ++*n_launches;
        job = j;
        chunk_in_job = c;
    }
    FVAD_HIP(ctx, hipGetLastError());
    return FVAD_OK;
}

} // namespace fvad
