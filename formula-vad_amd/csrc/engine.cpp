// engine.cpp -- context, device model, workspace and the batched engine of libfvad_hip.so.
//
// Batch formulation: the NSNet2 GRU state is reset for every 0.5 s chunk
// and every other stage is feed-forward, so all chunks of all lanes (lane = one channel of one
// stream) are processed together; cross-chunk effects (160-sample input hop, 4 warm-up feature
// rows, overlap-add tail, upsampler's last sample: src/NSNet2.zig:27-33,175-203) are either
// recomputed from the contiguous lane audio or, at the first chunk of a launch, read from the
// lane's LaneCarry.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>

#include "internal.h"

#ifndef FVAD_DIAG
#define FVAD_DIAG 0 // diagnostics build: see kernels_ws.hip
#endif

namespace fvad {

int set_err(const fvad_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
}
int hip_fail(const fvad_ctx* ctx, hipError_t e, const char* what)
{
    return set_err(ctx, FVAD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int dev_alloc(fvad_ctx* ctx, float** p, size_t n_floats, bool zero)
{
    FVAD_HIP(ctx, hipMalloc((void**)p, n_floats * sizeof(float)));
    if (zero) FVAD_HIP(ctx, hipMemsetAsync(*p, 0, n_floats * sizeof(float), ctx->stream));
    return FVAD_OK;
}

// option trace_kernels (debugging aid): name every stage on stderr and wait for it, so that a faulting kernel is
// the last one named
static const char* g_trace_name = nullptr;
void time_begin(fvad_ctx* ctx, const char* name)
{
    if (ctx->tune.trace_kernels) { g_trace_name = name; fprintf(stderr, "fvad: %s ...", name); fflush(stderr); }
    if (!ctx->timing) return;
    KernelTime kt;
    kt.name = name;
    hipEventCreate(&kt.e0);
    hipEventCreate(&kt.e1);
    hipEventRecord(kt.e0, ctx->stream);
    ctx->times.push_back(kt);
}
void time_end(fvad_ctx* ctx)
{
    if (g_trace_name) {
        const hipError_t e = hipStreamSynchronize(ctx->stream);
        fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
        fflush(stderr);
        g_trace_name = nullptr;
    }
    if (!ctx->timing) return;
    hipEventRecord(ctx->times.back().e1, ctx->stream);
}

// BufferedFFT.init's window and norm (BufferedFFT.zig:95-99; window_fn.zig:22-28,8-16) plus kissfft's tables for an
// n-point real transform, evaluated on the host in double like kissfft does, uploaded once per context and size
bool fvad_fft_size_ok(size_t n) { return n >= 4 && n % 2 == 0 && n <= (size_t)kVadFftMax; }

int get_vad_plan(fvad_ctx* ctx, size_t n, VadFftPlan* out, bool force_generic)
{
    // any even size kissfft would factor (FFT.zig:35-60), up to the generic kernel's LDS: 2 x (n / 2 + 1) complex values
    if (!fvad_fft_size_ok(n))
        return set_err(ctx, FVAD_ERR_INVALID_FFT_SIZE, "fft_size must be even, at least 4 and at most 16384 (FFT.zig:41-43; the generic kernel's LDS)");
    const bool wave_size = n == 512 || n == 1024 || n == 2048; // one wavefront per frame; every other size: one workgroup per frame
    const bool generic = force_generic || !wave_size;
    const int key = (int)n | (generic && wave_size ? 1 << 20 : 0);
    auto it = ctx->vad_plans.find(key);
    if (it == ctx->vad_plans.end()) {
        std::vector<float> win(n), tw, st, all;
        hann_window_periodic(win.data(), n);
        make_twiddles((int)n / 2, tw);
        make_super_twiddles((int)n / 2, st);
        auto put = [&](const std::vector<float>& v) { const size_t o = all.size(); all.insert(all.end(), v.begin(), v.end()); all.resize((all.size() + 63) / 64 * 64); return o; };
        const size_t o_w = put(win), o_tw = put(tw), o_st = put(st);
        fvad_ctx::VadPlanDev pd;
        FVAD_HIP(ctx, hipMalloc((void**)&pd.d, std::max<size_t>(all.size(), 64) * sizeof(float)));
        FVAD_HIP(ctx, hipMemcpy(pd.d, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
        pd.plan = VadFftPlan{(int)n, pd.d + o_w, pd.d + o_tw, pd.d + o_st, window_norm_factor(win.data(), n) / (float)n, generic ? 1 : 0, 0, {}};
        if (generic) {
            // the radices of the complex transform of length n / 2, kissfft's kf_factor order: 4s first, then 2, 3, 5, 7, ...
            int m = (int)n / 2, p = 4, nf = 0;
            while (m > 1) {
                while (m % p) {
                    switch (p) { case 4: p = 2; break; case 2: p = 3; break; default: p += 2; break; }
                    if ((long long)p * p > m) p = m; // no more factors: m is prime
                }
                m /= p;
                if (nf >= 14) { hipFree(pd.d); return set_err(ctx, FVAD_ERR_INVALID_FFT_SIZE, "fft_size has too many prime factors"); }
                pd.plan.fac[nf++] = p;
            }
            pd.plan.n_fac = nf;
        }
        it = ctx->vad_plans.emplace(key, pd).first;
        ctx->ws.generation++;
    }
    *out = it->second.plan;
    return FVAD_OK;
}

// value == nullptr or "": back to the default
static int apply_option(fvad_ctx* ctx, const std::string& name, const char* value)
{
    Tuning& tn = ctx->tune;
    const Tuning def;
    const bool unset = !value || !*value;
    const std::string v = unset ? "" : value;
    auto to_long = [&](long& out) { char* end = nullptr; out = strtol(v.c_str(), &end, 10); return end && *end == 0; };
    auto to_bool = [&](bool& out) { if (unset || v == "0") { out = false; return true; } if (v == "1") { out = true; return true; } return false; };
    if (name == "nn_math") {
        if (unset) tn.nn_math_force = -1;
        else if (v == "f32") tn.nn_math_force = FVAD_NN_MATH_F32;
        else if (v == "f16x3") tn.nn_math_force = FVAD_NN_MATH_F16X3;
        else if (v == "bf16x3") tn.nn_math_force = FVAD_NN_MATH_BF16X3;
        else return FVAD_ERR_INVALID_ARGUMENT;
    } else if (name == "gru_kernel") {
        if (!unset && v != "v3w12" && v != "v3w8" && v != "v3w4" && v != "v4w8" && v != "v5w0" && v != "v6w0") return FVAD_ERR_INVALID_ARGUMENT;
        tn.gru_kernel = v;
    } else if (name == "gemm_kernel") {
        if (!unset && v != "v1" && v != "v3" && v != "v3nofold") return FVAD_ERR_INVALID_ARGUMENT;
        tn.gemm_kernel = v;
    } else if (name == "h3_waves") {
        long w = 0;
        if (!unset && (!to_long(w) || (w != 0 && w != 8 && w != 12))) return FVAD_ERR_INVALID_ARGUMENT;
        tn.h3_waves = (int)w;
    } else if (name == "max_chunks") {
        long c = def.max_chunks;
        if (!unset && (!to_long(c) || c < 1)) return FVAD_ERR_INVALID_ARGUMENT;
        tn.max_chunks = c;
    } else if (name == "copy_threads") {
        long c = def.copy_threads;
        if (!unset && (!to_long(c) || c < 1 || c > 256)) return FVAD_ERR_INVALID_ARGUMENT;
        tn.copy_threads = (int)c;
    } else if (name == "ws_spin_ticks") {
        if (unset) { tn.ws_spin_ticks = def.ws_spin_ticks; tn.ws_spin_auto = true; }
        else {
            char* end = nullptr;
            tn.ws_spin_ticks = strtoull(v.c_str(), &end, 10);
            if (!end || *end) return FVAD_ERR_INVALID_ARGUMENT;
            tn.ws_spin_auto = false;
        }
    } else if (name == "ws2_variant") { // shape / timing knobs of the pipelined recurrence (tools/ws2_variants.py, ws2_delay.py)
        long c = 0;
        if (!unset && (!to_long(c) || c < 0 || c >= (1 << 24))) return FVAD_ERR_INVALID_ARGUMENT;
#if !FVAD_DIAG
        // the timing-only bits (1, 2, 4, 32: WRONG results) and the step trace (64) exist in the diagnostics build only
        // (make diag -> libfvad_hip_diag.so); the shipping library has no way to ask for wrong results
        if (c & (1 | 2 | 4 | 32 | 64 | 256 | 512 | 4096 | 8192 | 16384)) return FVAD_ERR_INVALID_ARGUMENT;
#endif
        tn.ws2_variant = (int)c;
    } else if (name == "ws2_waits") { // gru_ws2k's first-poll waits: layer 1 | layer 2 << 16, in 10 ns ticks; unset / 0 = built in (or calibrated)
        long c = 0;
        if (!unset && (!to_long(c) || c < 0 || c > 0x7FFFFFFFL)) return FVAD_ERR_INVALID_ARGUMENT;
        tn.ws2_waits = (unsigned)c;
    } else if (name == "ws2_calibrate") { // 1: measure those waits on this device now (needs the model); unset / 0: forget the measurement
        bool on = false;
        if (!to_bool(on)) return FVAD_ERR_INVALID_ARGUMENT;
        for (unsigned& w : tn.ws2_waits_cal) w = 0;
        if (on) { const int rc = calibrate_ws2_waits(ctx); if (rc) return rc; }
    } else if (name == "gru_lat_tiles") { // same bits whatever the value
        long c = 0;
        if (!unset && (!to_long(c) || c < 1 || c > 3)) return FVAD_ERR_INVALID_ARGUMENT;
        tn.gru_lat_tiles = (int)c;
    } else if (name == "k4_plain_loads") { if (!to_bool(tn.k4_plain_loads)) return FVAD_ERR_INVALID_ARGUMENT; }
    else if (name == "no_pipeline") { if (!to_bool(tn.no_pipeline)) return FVAD_ERR_INVALID_ARGUMENT; }
    else if (name == "trace_run") { if (!to_bool(tn.trace_run)) return FVAD_ERR_INVALID_ARGUMENT; }
    else if (name == "run_groups") { // "a,b,c": sixteenths per lane group of fvad_engine_run (sum 16, at most 7 groups); unset = planned
        if (!unset) {
            int sum = 0, n = 0, cur = 0;
            bool digit = false;
            for (char c : v + ",") {
                if (c >= '0' && c <= '9') { cur = cur * 10 + (c - '0'); digit = true; if (cur > 16) return FVAD_ERR_INVALID_ARGUMENT; }
                else if (c == ',' && digit && cur > 0) { sum += cur; ++n; cur = 0; digit = false; }
                else return FVAD_ERR_INVALID_ARGUMENT;
            }
            if (sum != 16 || n > 7) return FVAD_ERR_INVALID_ARGUMENT;
        }
        tn.run_groups = v;
    }
    else if (name == "trace_kernels") { if (!to_bool(tn.trace_kernels)) return FVAD_ERR_INVALID_ARGUMENT; }
    else if (name == "reproducible") { if (!to_bool(tn.reproducible)) return FVAD_ERR_INVALID_ARGUMENT; }
    else return FVAD_ERR_INVALID_ARGUMENT;
    ctx->ws.generation++; // a captured launch sequence holds the kernels of the old selection
    return FVAD_OK;
}

// a scratch buffer of the workspace grown to `need` floats (contents are not kept)
int grow(fvad_ctx* ctx, float** p, size_t* cap, size_t need)
{
    if (need <= *cap) return FVAD_OK;
    hipStreamSynchronize(ctx->stream);
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    FVAD_HIP(ctx, hipMalloc((void**)p, need * sizeof(float)));
    *cap = need;
    ctx->ws.generation++;
    return FVAD_OK;
}

} // namespace fvad

using namespace fvad;

// ------------------------------------------------------------------ C ABI: context + model
extern "C" {

int fvad_abi_version(void) { return FVAD_ABI_VERSION; }

const char* fvad_status_name(int s)
{
    switch (s) {
    case FVAD_OK: return "Ok";
    case FVAD_ERR_INVALID_FFT_SIZE: return "InvalidFFTSize";
    case FVAD_ERR_INVALID_SAMPLES_LENGTH: return "InvalidSamplesLength";
    case FVAD_ERR_INVALID_WINDOW_LENGTH: return "InvalidWindowLength";
    case FVAD_ERR_INVALID_RESULT_LENGTH: return "InvalidResultLength";
    case FVAD_ERR_INVALID_BINS_LENGTH: return "InvalidBinsLength";
    case FVAD_ERR_OUT_OF_RANGE: return "OutOfRange";
    case FVAD_ERR_NEGATIVE_FREQUENCY: return "NegativeFrequency";
    case FVAD_ERR_INVALID_INPUT_LENGTH: return "InvalidInputLength";
    case FVAD_ERR_INVALID_SAMPLE_RATE: return "InvalidSampleRate";
    case FVAD_ERR_CHANNEL_COUNT_MISMATCH: return "ChannelCountMismatch";
    case FVAD_ERR_ALLOC_FAILED: return "OutOfMemory";
    case FVAD_ERR_INVALID_ARGUMENT: return "InvalidArgument";
    case FVAD_ERR_NO_DEVICE: return "NoDevice";
    case FVAD_ERR_HIP: return "HipError";
    case FVAD_ERR_NO_MODEL: return "NoModel";
    case FVAD_ERR_MODEL_FORMAT: return "ModelFormat";
    case FVAD_ERR_IO: return "IoError";
    case FVAD_ERR_BUFFER_TOO_SMALL: return "BufferTooSmall";
    default: return "Unknown";
    }
}

int fvad_ctx_create(int device, fvad_ctx** out)
{
    if (!out) return FVAD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) return FVAD_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FVAD_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FVAD_ERR_NO_DEVICE; // code objects are gfx950-only
    if (hipSetDevice(device) != hipSuccess) return FVAD_ERR_NO_DEVICE;
    const int n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    auto* ctx = new (std::nothrow) fvad_ctx();
    if (!ctx) return FVAD_ERR_ALLOC_FAILED;
    ctx->device = device;
    ctx->n_cu = n_cu;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return FVAD_ERR_HIP; }

    // constant tables: one device allocation, sub-ranges 64-float aligned
    std::vector<float> win320(320), win320n(320), tw160, st320;
    nsnet2_window(win320.data());
    const float vol_norm_factor = 1 / (float)kNFft; // NSNet2.zig:323
    for (int i = 0; i < 320; ++i) win320n[i] = win320[i] * vol_norm_factor;
    make_twiddles(160, tw160);
    make_super_twiddles(160, st320);
    ctx->h_win320 = win320;
    std::vector<float> all;
    auto put = [&](const std::vector<float>& v) { const size_t o = all.size(); all.insert(all.end(), v.begin(), v.end()); all.resize((all.size() + 63) / 64 * 64); return o; };
    const size_t o_w320 = put(win320), o_w320n = put(win320n), o_tw160 = put(tw160), o_st320 = put(st320);
    if (hipMalloc((void**)&ctx->d_tables, all.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(ctx->d_tables, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        fvad_ctx_destroy(ctx);
        return FVAD_ERR_HIP;
    }
    ctx->tb.win320 = ctx->d_tables + o_w320;
    ctx->tb.win320n = ctx->d_tables + o_w320n;
    ctx->tb.tw160 = ctx->d_tables + o_tw160;
    ctx->tb.st320 = ctx->d_tables + o_st320;
    VadFftPlan pl;
    if (get_vad_plan(ctx, kVadFft, &pl) != FVAD_OK) { fvad_ctx_destroy(ctx); return FVAD_ERR_HIP; }
    // the tuning variables FVAD_<NAME> are read here, once; a bad value fails the creation rather than being ignored
    for (const char* opt : {"nn_math", "gru_kernel", "gemm_kernel", "h3_waves", "max_chunks", "copy_threads", "ws_spin_ticks",
                            "no_pipeline", "run_groups", "trace_run", "trace_kernels", "reproducible"}) {
        std::string env = std::string("FVAD_") + opt;
        for (char& c : env) c = (char)toupper((unsigned char)c);
        const char* v = getenv(env.c_str());
        if (v && *v && apply_option(ctx, opt, v) != FVAD_OK) {
            fprintf(stderr, "fvad: bad value in environment: %s=%s\n", env.c_str(), v);
            fvad_ctx_destroy(ctx);
            return FVAD_ERR_INVALID_ARGUMENT;
        }
    }
    *out = ctx;
    return FVAD_OK;
}

void fvad_ctx_destroy(fvad_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->vad_plans) if (kv.second.d) hipFree(kv.second.d);
    free_workspace_nn(ctx->ws);
    Workspace& ws = ctx->ws;
    if (ws.in) hipFree(ws.in);
    if (ws.den) hipFree(ws.den);
    if (ws.den16) hipFree(ws.den16);
    if (ws.band) hipFree(ws.band);
    if (ws.bins) hipFree(ws.bins);
    if (ws.carries) hipFree(ws.carries);
    if (ws.hx) hipFree(ws.hx);
    if (ws.ws_sync) hipFree(ws.ws_sync);
    if (ws.ws_fallbacks) hipFree(ws.ws_fallbacks);
    for (float* b : {ws.b3_hs1, ws.b3_hs2, ws.b3_f2, ws.b3_f3}) if (b) hipFree(b);
    for (Workspace::PinRing* r : {&ws.ring_in, &ws.ring_out}) {
        if (r->base) hipHostFree(r->base);
        for (hipEvent_t& e : r->ev) if (e) hipEventDestroy(e);
    }
    for (Workspace::PinSmall* r : {&ws.small_in, &ws.small_out}) {
        if (r->base) hipHostFree(r->base);
        if (r->ev) hipEventDestroy(r->ev);
    }
    if (ws.graph.exec) hipGraphExecDestroy(ws.graph.exec);
    if (ws.graph.graph) hipGraphDestroy(ws.graph.graph);
    if (ws.graph.h_descs) hipHostFree(ws.graph.h_descs);
    if (ws.graph.h_jobs) hipHostFree(ws.graph.h_jobs);
    if (ws.graph.d_descs) hipFree(ws.graph.d_descs);
    if (ws.graph.d_jobs) hipFree(ws.graph.d_jobs);
    for (hipEvent_t& e : ws.grp_in) if (e) hipEventDestroy(e);
    for (hipEvent_t& e : ws.grp_k) if (e) hipEventDestroy(e);
    if (ws.copy_in) hipStreamDestroy(ws.copy_in);
    if (ws.copy_out) hipStreamDestroy(ws.copy_out);
    for (hipEvent_t& e : ws.desc_ev) if (e) hipEventDestroy(e);
    for (hipEvent_t& e : ws.jobs_ev) if (e) hipEventDestroy(e);
    if (ws.fft_jobs) hipFree(ws.fft_jobs);
    if (ws.h_fft_jobs) hipHostFree(ws.h_fft_jobs);
    DeviceModel& m = ctx->dm;
    DevBuf* gbufs[] = {&m.g_fc1_w, &m.g_fc1_b, &m.g_gi1_w, &m.g_gi1_b, &m.g_r1, &m.g_br1, &m.g_gi2_w, &m.g_gi2_b, &m.g_r2, &m.g_br2,
                       &m.g_fc2_w, &m.g_fc2_b, &m.g_fc3_w, &m.g_fc3_b, &m.g_fc4_w, &m.g_fc4_b,
                       &m.gi1f_b3, &m.gi2_b3, &m.fc2_b3, &m.fc3_b3, &m.fc4_b3,
                       &m.gi1f_h3, &m.gi2_h3, &m.fc2_h3, &m.fc3_h3, &m.fc4_h3, &m.fc2h3_b, &m.fc3h3_b, &m.fc4h3_b, &m.r1_h3, &m.r2_h3};
    for (DevBuf* b : gbufs) if (b->p) hipFree(b->p);
    DevBuf* bufs[] = {&m.fc1_w, &m.fc1_b, &m.s_gi1f_w[0], &m.s_gi1f_w[1], &m.s_gi2_w[0], &m.s_gi2_w[1], &m.s_fc2_w[0], &m.s_fc2_w[1],
                      &m.s_fc3_w[0], &m.s_fc3_w[1], &m.s_fc4_w[0], &m.s_fc4_w[1], &m.s_fc4_b, &m.s_w2frag, &m.s_bw2, &m.s_w1frag, &m.br1, &m.br2,
                      &m.fc2_b, &m.fc3_b, &m.fc4_w, &m.fc4_b, &m.r1v2, &m.r2v2, &m.gi1f_w, &m.gi1f_b, &m.gi1v2_w, &m.gi2v2_w, &m.gi1f_bzr, &m.gi2_bzr, &m.gi1_btm, &m.gi2_btm, &m.fc2v3_w, &m.fc3v3_w, &m.fc2v3_b, &m.fc3v3_b};
    for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
    if (ctx->d_tables) hipFree(ctx->d_tables);
    for (auto& kt : ctx->times) { hipEventDestroy(kt.e0); hipEventDestroy(kt.e1); }
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* fvad_last_error(const fvad_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int fvad_ctx_synchronize(fvad_ctx* ctx)
{
    if (!ctx) return FVAD_ERR_INVALID_ARGUMENT;
    FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FVAD_OK;
}
void* fvad_ctx_stream(fvad_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int fvad_host_alloc(fvad_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out || bytes == 0) return FVAD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    hipSetDevice(ctx->device);
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { *out = nullptr; return set_err(ctx, FVAD_ERR_ALLOC_FAILED, "hipHostMalloc failed"); }
    return FVAD_OK;
}

void fvad_host_free(fvad_ctx* ctx, void* p)
{
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    hipHostFree(p);
}

int fvad_device_alloc(fvad_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out || bytes == 0) return FVAD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    hipSetDevice(ctx->device);
    if (hipMalloc(out, bytes) != hipSuccess) { *out = nullptr; (void)hipGetLastError(); return set_err(ctx, FVAD_ERR_ALLOC_FAILED, "hipMalloc failed"); }
    return FVAD_OK;
}

void fvad_device_free(fvad_ctx* ctx, void* p)
{
    if (!ctx || !p) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipFree(p);
}

int fvad_ctx_copy_to_device(fvad_ctx* ctx, void* dst_device, const void* src_host, size_t bytes)
{
    if (!ctx || (bytes && (!dst_device || !src_host))) return FVAD_ERR_INVALID_ARGUMENT;
    if (bytes) FVAD_HIP(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return FVAD_OK;
}

int fvad_ctx_copy_to_host(fvad_ctx* ctx, void* dst_host, const void* src_device, size_t bytes)
{
    if (!ctx || (bytes && (!dst_host || !src_device))) return FVAD_ERR_INVALID_ARGUMENT;
    if (bytes) FVAD_HIP(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return FVAD_OK;
}

int fvad_ctx_set_nn_math(fvad_ctx* ctx, int mode)
{
    if (!ctx || (mode != FVAD_NN_MATH_F32 && mode != FVAD_NN_MATH_F16X3 && mode != FVAD_NN_MATH_BF16X3)) return FVAD_ERR_INVALID_ARGUMENT;
    const int prev = ctx->nn_math;
    if (prev != mode) ctx->ws.generation++; // a captured launch sequence holds the other kernels
    ctx->nn_math = mode;
    return prev;
}

int fvad_ctx_nn_math_effective(const fvad_ctx* ctx)
{
    if (!ctx) return FVAD_ERR_INVALID_ARGUMENT;
    return nn_math_effective(ctx);
}

const char* fvad_ctx_last_nn_path(const fvad_ctx* ctx) { return ctx ? ctx->last_nn_path.c_str() : ""; }

// diagnostics for the tests, not part of the ABI in include/fvad.h: the launches a call of `total` chunks is cut into
// (plan_launches, nn_dispatch.cpp) under the context's current options; returns their number (at most `cap` are written)
int fvad_debug_plan_launches(fvad_ctx* ctx, long total, long max_chunks, long* out, int cap)
{
    if (!ctx || (cap > 0 && !out) || total < 0) return FVAD_ERR_INVALID_ARGUMENT;
    std::vector<long> plan;
    plan_launches(ctx, total, max_chunks, plan);
    for (int i = 0; i < cap && i < (int)plan.size(); ++i) out[i] = plan[(size_t)i];
    return (int)plan.size();
}

// diagnostics for tools/ws2_trace.py, not part of the ABI in include/fvad.h: the step trace gru_ws2k_kernel leaves behind
// the polled words when the context option ws2_variant has bit 64 set (2 x 1000 shader-clock stamps)
int fvad_debug_ws_trace(fvad_ctx* ctx, uint32_t* out, int n_words)
{
#if !FVAD_DIAG
    (void)out; (void)n_words;
    return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "the step trace exists in the diagnostics build only (make -C formula-vad_amd/csrc diag)");
#endif
    if (!ctx || !out || n_words < 0 || n_words > 2000) return FVAD_ERR_INVALID_ARGUMENT;
    if (!ctx->ws.ws_sync) return FVAD_ERR_INVALID_ARGUMENT;
    hipSetDevice(ctx->device);
    FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FVAD_HIP(ctx, hipMemcpy(out, ctx->ws.ws_sync + 520, (size_t)n_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return FVAD_OK;
}

uint32_t fvad_ctx_ws2_waits(const fvad_ctx* ctx, int wait_class)
{
    if (!ctx || wait_class < 1 || wait_class > 3) return 0;
    const Tuning& tn = ctx->tune;
    return tn.ws2_waits ? tn.ws2_waits : tn.ws2_waits_cal[wait_class] ? tn.ws2_waits_cal[wait_class] : fvad_gru_ws2_builtin_waits(wait_class, ctx->n_cu == 256 && !(tn.ws2_variant & (8 | 64 | 2048)));
}

int fvad_ctx_ws_fallbacks(fvad_ctx* ctx, uint64_t* n)
{
    if (!ctx || !n) return FVAD_ERR_INVALID_ARGUMENT;
    *n = 0;
    if (!ctx->ws.ws_fallbacks) return FVAD_OK; // the weight-stationary recurrence never ran on this context
    hipSetDevice(ctx->device);
    unsigned long long v = 0;
    FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FVAD_HIP(ctx, hipMemcpy(&v, ctx->ws.ws_fallbacks, sizeof(v), hipMemcpyDeviceToHost));
    *n = (uint64_t)v;
    return FVAD_OK;
}

int fvad_ctx_set_option(fvad_ctx* ctx, const char* name, const char* value)
{
    if (!ctx || !name) return FVAD_ERR_INVALID_ARGUMENT;
    const int rc = apply_option(ctx, name, value);
    if (rc && !strcmp(name, "ws2_calibrate") && value && !strcmp(value, "1")) return rc; // the measurement failed: its own message stands
    if (rc) return set_err(ctx, rc, std::string("fvad_ctx_set_option: unknown option or bad value: ") + name + "=" + (value ? value : ""));
    return FVAD_OK;
}

int fvad_ctx_enable_timing(fvad_ctx* ctx, int on)
{
    if (!ctx) return FVAD_ERR_INVALID_ARGUMENT;
    ctx->timing = on != 0;
    return FVAD_OK;
}

int fvad_ctx_kernel_times(fvad_ctx* ctx, const char** names, float* ms, size_t cap, size_t* n)
{
    if (!ctx || !n) return FVAD_ERR_INVALID_ARGUMENT;
    FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // fold repeated names (several launches per call) into one sum per kernel, in first-seen order
    ctx->time_names.clear();
    ctx->time_ms.clear();
    for (auto& kt : ctx->times) {
        float t = 0;
        hipEventElapsedTime(&t, kt.e0, kt.e1);
        size_t i = 0;
        for (; i < ctx->time_names.size(); ++i) if (ctx->time_names[i] == kt.name) break;
        if (i == ctx->time_names.size()) { ctx->time_names.push_back(kt.name); ctx->time_ms.push_back(0); }
        ctx->time_ms[i] += t;
        hipEventDestroy(kt.e0);
        hipEventDestroy(kt.e1);
    }
    ctx->times.clear();
    *n = ctx->time_names.size();
    for (size_t i = 0; i < *n && i < cap; ++i) {
        if (names) names[i] = ctx->time_names[i].c_str();
        if (ms) ms[i] = ctx->time_ms[i];
    }
    return FVAD_OK;
}

int fvad_load_nsnet2_weights(fvad_ctx* ctx, const fvad_nsnet2_weights* w)
{
    if (!ctx) return FVAD_ERR_INVALID_ARGUMENT;
    std::string err;
    if (!ctx->hw.from_view(w, err)) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, err);
    hipSetDevice(ctx->device);
    return upload_model(ctx);
}

int fvad_load_nsnet2_synth(fvad_ctx* ctx, uint64_t seed)
{
    if (!ctx) return FVAD_ERR_INVALID_ARGUMENT;
    synth_weights(seed, ctx->hw);
    hipSetDevice(ctx->device);
    return upload_model(ctx);
}

int fvad_load_nsnet2_onnx(fvad_ctx* ctx, const char* path)
{
    if (!ctx || !path) return FVAD_ERR_INVALID_ARGUMENT;
    std::string err;
    const int rc = read_onnx_nsnet2(path, ctx->hw, err);
    if (rc) return set_err(ctx, rc, err);
    hipSetDevice(ctx->device);
    return upload_model(ctx);
}

int fvad_get_nsnet2_weights(const fvad_ctx* ctx, fvad_nsnet2_weights* out)
{
    if (!ctx || !out) return FVAD_ERR_INVALID_ARGUMENT;
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "NSNet2 weights not loaded");
    ctx->hw.view(out);
    return FVAD_OK;
}

// ------------------------------------------------------------------ NSNet2 graph only
int fvad_nsnet2_forward(fvad_ctx* ctx, const float* features, size_t n_seq, size_t T, float* gains)
{
    if (!ctx || !features || !gains || n_seq == 0 || T == 0) return FVAD_ERR_INVALID_ARGUMENT;
    hipSetDevice(ctx->device);
    int rc = ensure_workspace(ctx, (long)n_seq, (int)T, 0);
    if (rc) return rc;
    Workspace& ws = ctx->ws;
    const long n_pad = padded_batch(ctx, (long)n_seq, (int)T, 0);
    // rows are [n_seq*T][161] on the host, [.][176] on the device
    FVAD_HIP(ctx, hipMemsetAsync(ws.feat, 0, (size_t)n_pad * T * kFeatStride * sizeof(float), ctx->stream));
    FVAD_HIP(ctx, hipMemcpy2DAsync(ws.feat, kFeatStride * sizeof(float), features, kNBins * sizeof(float),
                                   kNBins * sizeof(float), n_seq * T, hipMemcpyHostToDevice, ctx->stream));
    rc = run_nn(ctx, n_pad, (int)T, 0, (long)n_seq);
    if (rc) return rc;
    FVAD_HIP(ctx, hipMemcpy2DAsync(gains, kNBins * sizeof(float), ws.gains, kFeatStride * sizeof(float),
                                   kNBins * sizeof(float), n_seq * T, hipMemcpyDeviceToHost, ctx->stream));
    FVAD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FVAD_OK;
}

// ------------------------------------------------------------------ lane state
int fvad_lane_state_create(fvad_ctx* ctx, fvad_lane_state** out)
{
    if (!ctx || !out) return FVAD_ERR_INVALID_ARGUMENT;
    hipSetDevice(ctx->device);
    auto* s = new (std::nothrow) fvad_lane_state();
    if (!s) return FVAD_ERR_ALLOC_FAILED;
    s->ctx = ctx;
    for (int i = 0; i < 2; ++i)
        if (hipMalloc((void**)&s->carry[i], sizeof(LaneCarry)) != hipSuccess) { fvad_lane_state_destroy(s); return FVAD_ERR_HIP; }
    if (hipMalloc((void**)&s->den_rem, kVadFftMax * sizeof(float)) != hipSuccess) { fvad_lane_state_destroy(s); return FVAD_ERR_HIP; }
    fvad_lane_state_reset(s);
    *out = s;
    return FVAD_OK;
}

void fvad_lane_state_reset(fvad_lane_state* s)
{
    if (!s) return;
    hipSetDevice(s->ctx->device);
    // zero history == the reference's freshly initialised NSNet2 (NSNet2.zig:79,116,120,33)
    for (int i = 0; i < 2; ++i) hipMemsetAsync(s->carry[i], 0, sizeof(LaneCarry), s->ctx->stream);
    hipMemsetAsync(s->den_rem, 0, kVadFftMax * sizeof(float), s->ctx->stream);
    hipStreamSynchronize(s->ctx->stream);
    s->cur = 0;
    s->n_rem = 0;
    s->fft_size = kVadFft;
    s->samples_consumed = 0;
    s->next_frame_index = 0;
}

int fvad_lane_state_seek(fvad_lane_state* s, uint64_t sample_index, size_t fft_size)
{
    if (fft_size == 0) fft_size = kVadFft;
    if (!s || sample_index % kChunk48 || !fvad_fft_size_ok(fft_size)) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_lane_state_reset(s);
    // zero history, positioned mid-stream: the VAD FFT's frame grid stays anchored at absolute sample 0, so the
    // first sample_index % fft_size positions of the first frame are (zero) remainder
    s->samples_consumed = sample_index;
    s->fft_size = fft_size;
    s->n_rem = (size_t)(sample_index % fft_size);
    s->next_frame_index = sample_index - s->n_rem;
    return FVAD_OK;
}

void fvad_lane_state_destroy(fvad_lane_state* s)
{
    if (!s) return;
    for (int i = 0; i < 2; ++i) if (s->carry[i]) hipFree(s->carry[i]);
    if (s->den_rem) hipFree(s->den_rem);
    delete s;
}

void fvad_engine_opts_default(fvad_engine_opts* o)
{
    o->on_device = 0;
    o->min_bin = 11; // FFT.freqToBin(500) at 48 kHz / 1024 (FFT.zig:156-167)
    o->max_bin = 43; // FFT.freqToBin(2000)
    o->max_chunks_per_launch = 0;
    o->fft_size = 0; // 1024
    o->no_wait = 0;
    o->use_graph = 0;
}


} // extern "C"

extern "C" {

// device-resident batch: f32 or PCM16 input (exactly one of d_pcm / d_pcm16), optional PCM16 copy of the output
static int enqueue_device_impl(fvad_ctx* ctx, const float* d_pcm, const int16_t* d_pcm16, size_t n_lanes, size_t lane_stride,
                               size_t n_samples, float* d_denoised, int16_t* d_den16, float* d_band_sum, float* d_chunk_rms,
                               const fvad_engine_opts* opts_in)
{
    if (!ctx || (!d_pcm && !d_pcm16) || !d_band_sum || n_lanes == 0) return FVAD_ERR_INVALID_ARGUMENT;
    fvad_engine_opts opts;
    if (opts_in) opts = *opts_in; else fvad_engine_opts_default(&opts);
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "NSNet2 weights not loaded");
    if (lane_stride % (d_pcm ? 4 : 8)) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "lane_stride must be a multiple of 16 bytes");
    if (((uintptr_t)d_pcm | (uintptr_t)d_pcm16 | (uintptr_t)d_denoised | (uintptr_t)d_den16) % 16)
        return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "device audio buffers must be 16-byte aligned");
    hipSetDevice(ctx->device);
    Workspace& ws = ctx->ws;
    hipStream_t st = ctx->stream;
    const size_t n_chunks = n_samples / kChunk48;
    const size_t n_den = n_chunks * kChunk48;
    const size_t F = opts.fft_size ? (size_t)opts.fft_size : (size_t)kVadFft;
    VadFftPlan plan;
    int rc = get_vad_plan(ctx, F, &plan);
    if (rc) return rc;
    if (opts.min_bin < 0 || opts.max_bin > (int)(F / 2) || opts.max_bin < opts.min_bin) return set_err(ctx, FVAD_ERR_OUT_OF_RANGE, "band bins out of range");
    const size_t n_frames = n_den / F;
    if (n_chunks == 0) return FVAD_OK;
    float* den = d_denoised;
    if (!den) {
        if ((rc = grow(ctx, &ws.den, &ws.den_cap, n_lanes * n_den))) return rc;
        den = ws.den;
    }
    const size_t n_scratch = 2 * n_lanes;
    if (n_scratch * sizeof(LaneCarry) > ws.carries_cap) {
        hipStreamSynchronize(st);
        if (ws.carries) hipFree(ws.carries);
        ws.carries = nullptr; ws.carries_cap = 0;
        FVAD_HIP(ctx, hipMalloc((void**)&ws.carries, n_scratch * sizeof(LaneCarry)));
        ws.carries_cap = n_scratch * sizeof(LaneCarry);
        ws.carries_clean = 0;
        ws.generation++;
    }
    if (ws.fft_jobs_cap < n_lanes) {
        hipStreamSynchronize(st);
        if (ws.fft_jobs) hipFree(ws.fft_jobs);
        if (ws.h_fft_jobs) hipHostFree(ws.h_fft_jobs);
        ws.fft_jobs = nullptr; ws.h_fft_jobs = nullptr; ws.fft_jobs_cap = 0;
        FVAD_HIP(ctx, hipMalloc((void**)&ws.fft_jobs, n_lanes * sizeof(VadFftJob)));
        FVAD_HIP(ctx, hipHostMalloc((void**)&ws.h_fft_jobs, 2 * n_lanes * sizeof(VadFftJob), hipHostMallocDefault));
        ws.fft_jobs_cap = n_lanes;
        ws.jobs_mirror.clear();
        ws.generation++;
    }

    // the launch sequence of one call: carries reset, (descriptor upload, K1, NSNet2, K3) per launch, K4
    auto enqueue = [&](ChunkDesc* capture_descs, ChunkDesc* capture_dev, VadFftJob* h_jobs, VadFftJob* d_jobs) -> int {
        // A captured sequence holds kernel nodes only: memset / memcpy nodes replayed after direct launches on
        // the same stream were observed to run with stale parameters (ROCm 7.2), so the carries are zeroed
        // on the stream in front of every hipGraphLaunch and the tables are graph-private device copies.
        // (direct calls: a lane's first chunk reads carry 2 l, its last one writes 2 l + 1; only a call of several launches
        // flips them and writes an even one -- after a single-launch call the carries that are read are still the zeros they were)
        if (!capture_descs && ws.carries_clean < n_scratch) {
            FVAD_HIP(ctx, hipMemsetAsync(ws.carries, 0, n_scratch * sizeof(LaneCarry), st));
            ws.carries_clean = n_scratch;
        }
        std::vector<LaneJob> jobs(n_lanes);
        for (size_t l = 0; l < n_lanes; ++l) {
            jobs[l].d_in = d_pcm ? d_pcm + l * lane_stride : nullptr;
            jobs[l].d_in16 = d_pcm16 ? d_pcm16 + l * lane_stride : nullptr;
            jobs[l].d_den = den + l * n_den;
            jobs[l].d_den16 = d_den16 ? d_den16 + l * n_den : nullptr;
            jobs[l].n_chunks = n_chunks;
            jobs[l].carry[0] = ws.carries + 2 * l;
            jobs[l].carry[1] = ws.carries + 2 * l + 1;
            jobs[l].cur = 0;
            jobs[l].d_rms = d_chunk_rms ? d_chunk_rms + l * n_chunks : nullptr;
        }
        long launches = 0;
        int r = run_chunks(ctx, jobs, opts.max_chunks_per_launch, capture_descs, capture_dev, &launches);
        // several launches: the even carries get written (what run_chunks DID, not what planned_max_chunks predicts; a
        // failed call may have made some of its launches)
        if (launches != 1) ws.carries_clean = 0;
        if (r) return r;
        // one K4 launch for every lane's frames
        for (size_t l = 0; l < n_lanes; ++l) h_jobs[l] = {den + l * n_den, d_band_sum + l * n_frames, nullptr, (long)n_frames};
        if (!capture_descs && (ws.jobs_mirror.size() != n_lanes || memcmp(ws.jobs_mirror.data(), h_jobs, n_lanes * sizeof(VadFftJob)) != 0)) {
            FVAD_HIP(ctx, hipMemcpyAsync(d_jobs, h_jobs, n_lanes * sizeof(VadFftJob), hipMemcpyHostToDevice, st));
            ws.jobs_mirror.assign(h_jobs, h_jobs + n_lanes);
        }
        time_begin(ctx, "fft1024_bandsum");
        FVAD_HIP(ctx, (hipError_t)fvad_launch_vadfft_jobs(d_jobs, (int)n_lanes, (long)n_frames, plan, opts.min_bin, opts.max_bin, st, 0, ctx->n_cu, ctx->tune.k4_plain_loads ? 1 : 0));
        time_end(ctx);
        return FVAD_OK;
    };

    // Opt-in (fvad_engine_opts.use_graph): capture the sequence into a hipGraph once and replay it while the arguments
    // and the workspace stay the same -- BASELINE config 5's "hipGraph-captured steady-state loop".  A step
    // is ~12 launches per 100 ms of GPU work, so this saves well under 1 % (measured in bench.py's extras).
    if (opts.use_graph && !ctx->timing) {
        const long total = (long)(n_lanes * n_chunks);
        const long maxc = planned_max_chunks(ctx, total, opts.max_chunks_per_launch);
        {
            std::vector<long> plan;
            plan_launches(ctx, total, opts.max_chunks_per_launch, plan);
            if ((rc = ensure_workspace_plan(ctx, plan))) return rc; // no allocation while capturing
        }
        if ((rc = ensure_gru_ws(ctx))) return rc;
        Workspace::GraphCache& gc = ws.graph;
        const void* pcm_key = d_pcm ? (const void*)d_pcm : (const void*)d_pcm16;
        const bool hit = gc.valid && gc.pcm == pcm_key && gc.den16 == d_den16 && gc.den == den && gc.band == d_band_sum && gc.rms == d_chunk_rms &&
                         gc.n_lanes == n_lanes && gc.lane_stride == lane_stride && gc.n_samples == n_samples &&
                         gc.min_bin == opts.min_bin && gc.max_bin == opts.max_bin && gc.max_chunks == maxc && gc.fft_size == F &&
                         gc.generation == ws.generation;
        if (!hit) {
            hipStreamSynchronize(st);
            if (gc.exec) hipGraphExecDestroy(gc.exec);
            if (gc.graph) hipGraphDestroy(gc.graph);
            gc.exec = nullptr; gc.graph = nullptr; gc.valid = false;
            if (gc.h_descs_cap < (size_t)total) {
                if (gc.h_descs) hipHostFree(gc.h_descs);
                if (gc.d_descs) hipFree(gc.d_descs);
                gc.h_descs = nullptr; gc.d_descs = nullptr; gc.h_descs_cap = 0;
                FVAD_HIP(ctx, hipHostMalloc((void**)&gc.h_descs, (size_t)total * sizeof(ChunkDesc), hipHostMallocDefault));
                FVAD_HIP(ctx, hipMalloc((void**)&gc.d_descs, (size_t)total * sizeof(ChunkDesc)));
                gc.h_descs_cap = (size_t)total;
            }
            if (gc.h_jobs_cap < n_lanes) {
                if (gc.h_jobs) hipHostFree(gc.h_jobs);
                if (gc.d_jobs) hipFree(gc.d_jobs);
                gc.h_jobs = nullptr; gc.d_jobs = nullptr; gc.h_jobs_cap = 0;
                FVAD_HIP(ctx, hipHostMalloc((void**)&gc.h_jobs, n_lanes * sizeof(VadFftJob), hipHostMallocDefault));
                FVAD_HIP(ctx, hipMalloc((void**)&gc.d_jobs, n_lanes * sizeof(VadFftJob)));
                gc.h_jobs_cap = n_lanes;
            }
            FVAD_HIP(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            rc = enqueue(gc.h_descs, gc.d_descs, gc.h_jobs, gc.d_jobs);
            hipGraph_t g = nullptr;
            const hipError_t ee = hipStreamEndCapture(st, &g);
            if (rc || ee != hipSuccess || !g) {
                if (g) hipGraphDestroy(g);
                return rc ? rc : set_err(ctx, FVAD_ERR_HIP, "hipStreamEndCapture failed");
            }
            gc.graph = g;
            // the graph's private tables: written once, read by every replay
            FVAD_HIP(ctx, hipMemcpy(gc.d_descs, gc.h_descs, (size_t)total * sizeof(ChunkDesc), hipMemcpyHostToDevice));
            FVAD_HIP(ctx, hipMemcpy(gc.d_jobs, gc.h_jobs, n_lanes * sizeof(VadFftJob), hipMemcpyHostToDevice));
            FVAD_HIP(ctx, hipGraphInstantiate(&gc.exec, gc.graph, nullptr, nullptr, 0));
            gc.pcm = pcm_key; gc.den16 = d_den16; gc.den = den; gc.band = d_band_sum; gc.rms = d_chunk_rms;
            gc.n_lanes = n_lanes; gc.lane_stride = lane_stride; gc.n_samples = n_samples;
            gc.min_bin = opts.min_bin; gc.max_bin = opts.max_bin; gc.max_chunks = maxc; gc.fft_size = F;
            gc.generation = ws.generation;
            gc.valid = true;
        }
        FVAD_HIP(ctx, hipMemsetAsync(ws.carries, 0, n_scratch * sizeof(LaneCarry), st));
        ws.carries_clean = 0; // (a replayed sequence of several launches writes the even carries)
        // the replayed sequence may hold a pass of gru_ws_kernel (launches of 385..~1900 chunks), which leaves the polled
        // words counted up: whatever a direct call knew about them is void after a replay
        ws.sync_clean = false;
        FVAD_HIP(ctx, hipGraphLaunch(gc.exec, st));
        FVAD_HIP(ctx, hipStreamSynchronize(st));
        FVAD_HIP(ctx, hipGetLastError());
        return FVAD_OK;
    }

    // the pinned job table has two slots: a slot is rewritten only after its previous upload has left the host
    const int js = ws.jobs_slot;
    ws.jobs_slot ^= 1;
    if (!ws.jobs_ev[js]) FVAD_HIP(ctx, hipEventCreateWithFlags(&ws.jobs_ev[js], hipEventDisableTiming));
    else FVAD_HIP(ctx, hipEventSynchronize(ws.jobs_ev[js]));
    if ((rc = enqueue(nullptr, nullptr, ws.h_fft_jobs + (size_t)js * ws.fft_jobs_cap, ws.fft_jobs))) return rc;
    FVAD_HIP(ctx, hipEventRecord(ws.jobs_ev[js], st));
    if (!opts.no_wait) FVAD_HIP(ctx, hipStreamSynchronize(st));
    FVAD_HIP(ctx, hipGetLastError());
    return FVAD_OK;
}

int fvad_engine_enqueue_device(fvad_ctx* ctx, const float* d_pcm, size_t n_lanes, size_t lane_stride, size_t n_samples,
                               float* d_denoised, float* d_band_sum, float* d_chunk_rms, const fvad_engine_opts* opts_in)
{
    if (!d_pcm) return FVAD_ERR_INVALID_ARGUMENT;
    return enqueue_device_impl(ctx, d_pcm, nullptr, n_lanes, lane_stride, n_samples, d_denoised, nullptr, d_band_sum, d_chunk_rms, opts_in);
}

int fvad_engine_enqueue_device_i16(fvad_ctx* ctx, const int16_t* d_pcm16, size_t n_lanes, size_t lane_stride, size_t n_samples,
                                   int16_t* d_denoised16, float* d_band_sum, float* d_chunk_rms, const fvad_engine_opts* opts_in)
{
    if (!d_pcm16) return FVAD_ERR_INVALID_ARGUMENT;
    return enqueue_device_impl(ctx, nullptr, d_pcm16, n_lanes, lane_stride, n_samples, nullptr, d_denoised16, d_band_sum, d_chunk_rms, opts_in);
}

} // extern "C"
