// internal.h -- shared host-side declarations of libfvad_hip.so (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fvad.h"
#include "kernels.h"

// ------------------------------------------------------------------ host tables / windows
namespace fvad {

void hann_window_periodic(float* result, size_t n);   // window_fn.zig:22-28,51-68
void hann_window_symmetric(float* result, size_t n);  // window_fn.zig:30-41
float window_norm_factor(const float* w, size_t n);   // window_fn.zig:8-16
void nsnet2_window(float* w320);                      // NSNet2.zig:384-396
// exp(-2 pi i j / n) for j < n, evaluated in double, rounded once (kissfft's kf_cexp)
void make_twiddles(int n, std::vector<float>& out);
// exp(-i pi ((k+1)/ncfft + 1/2)) for k < ncfft/2 (kissfft's super_twiddles)
void make_super_twiddles(int ncfft, std::vector<float>& out);

// ------------------------------------------------------------------ weights
struct HostWeights {
    int n_bins = 0, n_fc1 = 0, n_hidden = 0, n_fc2 = 0, n_fc3 = 0;
    std::vector<float> fc1_w, fc1_b, gru1_w, gru1_r, gru1_b, gru2_w, gru2_r, gru2_b, fc2_w, fc2_b,
        fc3_w, fc3_b, fc4_w, fc4_b;
    void view(fvad_nsnet2_weights* out) const;
    bool from_view(const fvad_nsnet2_weights* in, std::string& err);
    bool check_dims(std::string& err) const;  // dims this build can run at all
    bool is_baseline() const { return n_bins == 161 && n_fc1 == 400 && n_hidden == 400 && n_fc2 == 600 && n_fc3 == 600; }
};
void synth_weights(uint64_t seed, HostWeights& w);
int read_onnx_nsnet2(const char* path, HostWeights& w, std::string& err);

// fragment-major packing for the MFMA kernels (see kernels_nn.hip)
void pack_panel(const float* W, int N, int K, int n_blocks, int NT, int S, std::vector<float>& out);
void pack_gru_r2(const float* R, int H, std::vector<float>& out);
void pack_gru_frag(const float* W, int H, int K, std::vector<float>& out); // [H/16][3][ceil(K/16)][64][4], K zero-padded
// f16x3 layouts (kernels_h3.hip): hi/lo f16 pieces of W * sw as bit patterns in float storage
float h3_weight_scale(const float* W, size_t n);      // power of two: max |W| * sw in [2^14, 2^15); 0 if a weight is not finite
float h3_activation_scale(double bound);              // power of two: |x| <= bound -> |x * sx| <= 2^14
void pack_panel_h3(const float* W, int N, int K, int n_blocks, int NT, float sw, std::vector<float>& out);
void pack_gru_r_h3(const float* R, int H, float sw, std::vector<float>& out);
// bf16x3 layout (kernels_b3.hip): three bf16 pieces per weight, no scale
void pack_panel_b3(const float* W, int N, int K, int n_blocks, int NT, std::vector<float>& out);

// ------------------------------------------------------------------ device model
struct DevBuf {
    float* p = nullptr;
    size_t n = 0; // floats
};

struct DeviceModel {
    DevBuf fc1_w, fc1_b, br1, br2, fc2_b, fc3_b, // fc2 / fc3 biases padded to 640
        s_gi1f_w[2], s_gi2_w[2], s_fc2_w[2], s_fc3_w[2], s_fc4_w[2], // small-batch layouts: column blocks of 2 / of 4 tiles
        s_fc4_b, s_w2frag, s_bw2, s_w1frag,
        fc4_w, fc4_b, r1v2, r2v2, gi1f_w, gi1f_b, gi1v2_w, gi2v2_w,
        fc2v3_w, fc3v3_w, fc2v3_b, fc3v3_b, // fc2/fc3 as 3 column blocks of 13 tiles for panel_gemm3
        gi1f_bzr, gi2_bzr, // input-projection biases with the recurrent z/r biases folded in (gru_rec3)
        gi1_btm, gi2_btm;  // plain input-projection biases in tile-major unit order (large-batch GEMM without the fold)
    // f16x3 layouts (kernels_h3.hip): two f16 pieces per weight; sw = weight scale, sx = input scale of the layer
    DevBuf gi1f_h3, gi2_h3, fc2_h3, fc3_h3, fc4_h3, fc2h3_b, fc3h3_b, fc4h3_b, r1_h3, r2_h3;
    struct H3Scale { float sw = 1.0f, sx = 1.0f; };
    H3Scale h3_gi1f, h3_gi2, h3_fc2, h3_fc3, h3_fc4, h3_r1, h3_r2;
    bool h3_ok = false; // every weight and bound finite: the f16x3 kernels may be used
    // bf16x3 layouts of the five dense layers (kernels_b3.hip); biases are shared with the other families
    DevBuf gi1f_b3, gi2_b3, fc2_b3, fc3_b3, fc4_b3;
    bool loaded = false;
    // A model of other dimensions than NSNet2-baseline's 161/400/400/600/600 (NSNet2.init binds whatever file the
    // configuration names, NSNet2.zig:53-112): run by run_nn_generic on kernels that take their sizes at run time.
    // Every width is padded: GEMM outputs to blocks of 8 tiles (128 columns), the hidden size to 16 J.
    bool generic = false;
    struct GenDims {
        int F1 = 0, H = 0, N2 = 0, N3 = 0;            // n_fc1, n_hidden, n_fc2, n_fc3
        int J = 0;                                     // unit tiles of the padded hidden size
        int F1p = 0, Hp = 0, Gp = 0, N2p = 0, N3p = 0; // row strides of a1, h, gi, f2, f3
    } gd;
    DevBuf g_fc1_w, g_fc1_b, g_gi1_w, g_gi1_b, g_r1, g_br1, g_gi2_w, g_gi2_b, g_r2, g_br2,
        g_fc2_w, g_fc2_b, g_fc3_w, g_fc3_b, g_fc4_w, g_fc4_b;
    // floats per row of the workspace buffers for this model (baseline: 400 / 1200 / 400 / 608)
    int w_a1 = 400, w_gi = 1200, w_h = 400, w_f = 608;
};

struct Workspace {
    long cap_chunks = 0; // padded chunk capacity (multiple of 768: covers every batch padding)
    bool sync_clean = false; // the polled words of the weight-stationary recurrences are known to be zero (gru_ws2_fallback_kernel left them so)
    size_t cap_rows = 0; // rows (padded sequences x steps per sequence) the NSNet2 buffers hold
    size_t a1_cap_rows = 0, h_cap_rows = 0, hs_cap_rows = 0; // the same for a1 / h1, h2 / hs1, hs2: allocated for the arithmetic that reads them
    int w_a1 = 0, w_gi = 0, w_h = 0, w_f = 0; // row widths the buffers were allocated for (DeviceModel::w_*)
    // bf16x3 mode only (allocated when a context first runs in it): h1 / h2 and the fc2 / fc3 outputs as three-piece
    // fragments, 13 and 19 K-steps of 3 KB per 16 rows
    float *b3_hs1 = nullptr, *b3_hs2 = nullptr, *b3_f2 = nullptr, *b3_f3 = nullptr;
    size_t b3_cap_rows = 0; // rows (padded sequences x steps) the bf16x3 buffers hold
    ChunkDesc* descs = nullptr;
    // host mirrors of what the device tables hold: a steady-state loop of identical calls (the same buffers, the same shape)
    // builds the same tables again, and their uploads -- two small copies and a memset in front of ~10 kernels -- are skipped
    std::vector<ChunkDesc> descs_mirror;
    std::vector<VadFftJob> jobs_mirror;
    size_t carries_clean = 0; // scratch carries [0, carries_clean) that a launch READS (the even ones: stateless lanes start from them) are known to be zeros
    ChunkDesc* h_descs = nullptr; // pinned, two slots of cap_chunks descriptors
    hipEvent_t desc_ev[2] = {nullptr, nullptr}; // slot's upload has left the host
    int desc_slot = 0;
    float *feat = nullptr, *spec = nullptr, *a1 = nullptr, *gi = nullptr,
          *h1 = nullptr, *h2 = nullptr, *hs1 = nullptr, *hs2 = nullptr, *f2 = nullptr, *f3 = nullptr, *gains = nullptr;
    // generic scratch (engine_run staging, denoised audio, band sums)
    float* in = nullptr;  size_t in_cap = 0;
    float* den = nullptr; size_t den_cap = 0;
    float* den16 = nullptr; size_t den16_cap = 0; // PCM16 denoised staging (int16 pairs in float-sized slots)
    float* band = nullptr; size_t band_cap = 0;
    float* bins = nullptr; size_t bins_cap = 0;
    LaneCarry* carries = nullptr; size_t carries_cap = 0; // scratch carries (2 per lane)
    // weight-stationary recurrence (kernels_ws.hip): h exchange buffer and the block of polled words
    // (256 flags per GRU layer + the error word, zeroed by zero_words_kernel in front of every network pass)
    float* hx = nullptr; size_t hx_cap = 0;
    unsigned* ws_sync = nullptr;
    unsigned long long* ws_fallbacks = nullptr; // device counter: network passes in which gru_ws gave up (gru_lat redid the layers)
    VadFftJob* fft_jobs = nullptr; VadFftJob* h_fft_jobs = nullptr; size_t fft_jobs_cap = 0; // host table: two slots of fft_jobs_cap
    hipEvent_t jobs_ev[2] = {nullptr, nullptr}; int jobs_slot = 0; // slot's upload has left the host (no_wait calls)
    // pinned staging ring for large host <-> device transfers: 32 slots of 8 MB, an event per slot (its DMA is done)
    struct PinRing { char* base = nullptr; hipEvent_t ev[32] = {}; };
    PinRing ring_in, ring_out; // host->device staging / device->host draining (used by different threads)
    // small transfers (a live push: 96 KB in, a few hundred bytes out): one page-locked bounce buffer per direction, so that
    // the copies are truly asynchronous (a copy to or from pageable memory blocks the calling thread)
    struct PinSmall { char* base = nullptr; hipEvent_t ev = nullptr; };
    PinSmall small_in, small_out;
    unsigned generation = 0; // bumped whenever a device buffer of the workspace is reallocated
    // one captured launch sequence of fvad_engine_enqueue_device (opt-in, FVAD_GRAPH=1): replayed while the
    // call's arguments and the workspace are unchanged
    struct GraphCache {
        bool valid = false;
        const void *pcm = nullptr, *den = nullptr, *band = nullptr, *rms = nullptr, *den16 = nullptr;
        size_t n_lanes = 0, lane_stride = 0, n_samples = 0;
        int min_bin = 0, max_bin = 0;
        long max_chunks = 0;
        size_t fft_size = 0;
        unsigned generation = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        // the graph's private descriptor / job tables (host image + device copy): every launch of the
        // sequence has its own region, uploaded once after the capture
        ChunkDesc* h_descs = nullptr; ChunkDesc* d_descs = nullptr;
        size_t h_descs_cap = 0;
        VadFftJob* h_jobs = nullptr; VadFftJob* d_jobs = nullptr;
        size_t h_jobs_cap = 0;
    } graph;
    // copy streams + per-group events of the pipelined host-buffer path (fvad_engine_run)
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    hipEvent_t grp_in[8] = {}, grp_k[8] = {};
};

struct KernelTime {
    std::string name;
    hipEvent_t e0, e1;
};

// Testing / tuning aids of a context (fvad_ctx_set_option).  The environment variables FVAD_<NAME> are read ONCE,
// by fvad_ctx_create, as the initial values; nothing in the data path looks at the environment.
struct Tuning {
    int nn_math_force = -1;      // FVAD_NN_MATH at create: overrides fvad_ctx_set_nn_math (-1: none)
    std::string gru_kernel;      // "" / "v3w12" / "v3w8" / "v3w4" / "v4w8" (gru_lat) / "v5w0" (gru_ws) / "v6w0" (gru_ws2)
    std::string gemm_kernel;     // "" / "v1" (small-batch GEMM) / "v3" / "v3nofold"
    int h3_waves = 0;            // 0 / 8 / 12
    long max_chunks = 49152;     // chunks per launch when the caller passes 0
    bool no_pipeline = false;    // host-buffer path: single lane group
    bool trace_run = false;      // host-buffer path: a timeline of every fvad_engine_run call on stderr (groups staged / enqueued / finished / drained)
    std::string run_groups;      // host-buffer path: the lane groups' sizes in sixteenths of the call ("4,4,4,4"); empty = planned
    int gru_lat_tiles = 0;       // row tiles per workgroup of gru_lat (1, 2, 3: the same bits); 0 = by launch size (the cost model)
    bool k4_plain_loads = false; // vadfft1024_band_kernel: stage frames with plain 8-byte loads even when they are 16-byte aligned (the path
                                 // an unaligned job takes; same arithmetic, same bits: tests)
    int copy_threads = 8;        // memcpy threads of the pinned staging rings
    bool trace_kernels = false;  // name every NSNet2 stage on stderr and wait for it
    bool reproducible = false;   // one kernel family (the large-batch one) at every batch size
    int ws2_variant = 0;         // timing-only variants of gru_ws2_kernel (tools/ws2_variants.py); 0 in production
    unsigned ws2_waits = 0;      // gru_ws2k's first-poll waits for every launch (layer 1 | layer 2 << 16, 10 ns ticks); 0 = per class:
    unsigned ws2_waits_cal[4] = {0, 0, 0, 0}; // what ws2_calibrate measured for fvad_gru_ws2_wait_class 1..3; 0 = the kernel's built-in table
    // spin deadline of the weight-stationary kernels' waits in 100 MHz ticks.  Default (ws_spin_auto): derived per launch from
    // the launch's own expected duration -- 20 x the cost model's estimate, at least 2 ms -- so that a launch that cannot make
    // progress (another process holds the CUs its workgroups need) costs milliseconds, not the 0.25 s of a fixed deadline
    unsigned long long ws_spin_ticks = 0ull;
    bool ws_spin_auto = true;
};

} // namespace fvad

struct fvad_ctx {
    int device = 0;
    int n_cu = 256; // compute units: size of the persistent GEMM grid
    hipStream_t stream = nullptr;
    mutable std::string err;
    // constant tables
    float* d_tables = nullptr;
    FftTables tb{};
    std::vector<float> h_win320;
    // VAD-side FFT tables per size (512 / 1024 / 2048), built on first use
    struct VadPlanDev { float* d = nullptr; VadFftPlan plan{}; };
    std::map<int, VadPlanDev> vad_plans;
    // model
    fvad::HostWeights hw;
    fvad::DeviceModel dm;
    fvad::Workspace ws;
    int nn_math = 0; // FVAD_NN_MATH_F32: the reference's arithmetic (NSNet2.zig:220, ORT CPU EP) at every batch size
    fvad::Tuning tune;
    std::string last_nn_path; // what the last NSNet2 pass ran ("f32: panel_gemm3 + gru_rec3<12>", ...)
    // timing
    bool timing = false;
    std::vector<fvad::KernelTime> times;
    std::vector<std::string> time_names;
    std::vector<float> time_ms;
};

struct fvad_lane_state {
    fvad_ctx* ctx = nullptr;
    LaneCarry* carry[2] = {nullptr, nullptr}; // device, double-buffered
    int cur = 0;
    float* den_rem = nullptr; // device, kVadFftMax floats: denoised samples not yet FFT'd
    size_t n_rem = 0;
    size_t fft_size = kVadFft; // frame length the remainder / frame index refer to
    uint64_t samples_consumed = 0; // raw samples consumed so far (multiple of 24000)
    uint64_t next_frame_index = 0; // absolute index of the next FFT frame's first sample
};

namespace fvad {
int set_err(const fvad_ctx* ctx, int code, const std::string& msg);
int hip_fail(const fvad_ctx* ctx, hipError_t e, const char* what);
#define FVAD_HIP(ctx, call)                                         \
    do {                                                            \
        hipError_t _e = (call);                                     \
        if (_e != hipSuccess) return fvad::hip_fail(ctx, _e, #call); \
    } while (0)

int upload_model(fvad_ctx* ctx);                                              // model.cpp
int dev_alloc(fvad_ctx* ctx, float** p, size_t n_floats, bool zero);          // engine.cpp
int grow(fvad_ctx* ctx, float** p, size_t* cap, size_t need);                 // engine.cpp
void free_workspace_nn(Workspace& ws);                                        // nn_dispatch.cpp
long planned_max_chunks(const fvad_ctx* ctx, long total, long max_chunks);    // nn_dispatch.cpp: the largest launch of a call's plan
void plan_launches(const fvad_ctx* ctx, long total, long max_chunks, std::vector<long>& plan); // the launches of a call, in order
int ensure_workspace_plan(fvad_ctx* ctx, const std::vector<long>& plan);       // the workspace for every launch of a plan
long padded_batch(const fvad_ctx* ctx, long n, int T, int skip);              // nn_dispatch.cpp: sequences a launch of n is padded to
// FVAD_NN_MATH_F32 / FVAD_NN_MATH_F16X3: what run_nn uses on this context with the loaded model, at every batch size
int nn_math_effective(const fvad_ctx* ctx);
int ensure_workspace(fvad_ctx* ctx, long n_chunks, int T, int skip, long n_last = 0);
int ensure_gru_ws(fvad_ctx* ctx);
// tables of the n-point real FFT (any even n up to kVadFftMax; 512 / 1024 / 2048 on the wavefront kernels unless force_generic),
// cached per context
int get_vad_plan(fvad_ctx* ctx, size_t n, VadFftPlan* out, bool force_generic = false);
bool fvad_fft_size_ok(size_t n); // even, 4 .. kVadFftMax
// NSNet2 on ws.feat -> ws.gains for n_chunks sequences of T rows; gains rows skip..T-1 only
// n_real: the sequences of the padded batch that are real (the tail layers of a small launch run on their rows only); 0 = all
int run_nn(fvad_ctx* ctx, long n_chunks_pad, int T, int skip, long n_real = 0);
int calibrate_ws2_waits(fvad_ctx* ctx);

struct LaneJob {
    const float* d_in;   // device, n_chunks * 24000 samples (480 samples of history are read
                         // from the lane's carry, never from before d_in)
    float* d_den;        // device, n_chunks * 24000
    size_t n_chunks;
    LaneCarry* carry[2];
    int cur;             // index of the carry holding the current state
    float* d_rms;        // device, n_chunks
    float* h_spec = nullptr; // host taps (parity / debug): [n_chunks][50][161][2], [n_chunks][54][161]
    float* h_feat = nullptr;
    const int16_t* d_in16 = nullptr; // device, PCM16 input instead of d_in (16-bit transport)
    int16_t* d_den16 = nullptr;      // device, optional PCM16 copy of the denoised audio
};
// n_launches (optional): how many K1 -> NSNet2 -> K3 launches the call was cut into (a lane cut over several launches has its
// even carry written: see Workspace::carries_clean)
int run_chunks(fvad_ctx* ctx, std::vector<LaneJob>& jobs, long max_chunks, ChunkDesc* capture_descs = nullptr,
               ChunkDesc* capture_dev = nullptr, long* n_launches = nullptr);
void time_begin(fvad_ctx* ctx, const char* name);
void time_end(fvad_ctx* ctx);
} // namespace fvad
