// kernels.h -- launch wrappers of the gfx950 kernels (internal; the public ABI is include/fvad.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

enum { FVAD_ACT_NONE = 0, FVAD_ACT_RELU = 1, FVAD_ACT_SIGMOID = 2 };

// ------------------------------------------------------------------ geometry (NSNet2.zig:12-16)
constexpr int kNFft = 320;
constexpr int kNHop = 160;
constexpr int kFramesPerChunk = 50;
constexpr int kWarmupRows = 4;                 // artifact_mitigation_window
constexpr int kRowsPerChunk = kFramesPerChunk + kWarmupRows; // 54
constexpr int kNBins = 161;
constexpr int kFeatStride = 176;               // 161 padded to 11 x 16 (zero-filled tail)
constexpr int kDown = 3;                       // 48 kHz -> 16 kHz
constexpr int kChunk48 = kFramesPerChunk * kNHop * kDown; // 24000
constexpr int kVadFft = 1024;               // VADPipeline.Config.fft_size default (VADPipeline.zig:21)
constexpr int kVadFftMax = 16384;           // any even size up to this: 512 / 1024 / 2048 on the wavefront FFT, the others on the generic kernel

// cross-call carry of one lane (all device floats); mirrors NSNet2.zig:27-33 state
struct LaneCarry {
    float in_tail[kNHop * kDown];          // last 480 raw 48 kHz samples (-> audio_input[0..160])
    float feat_tail[kWarmupRows * kNBins]; // features rows 50..53 of the previous chunk
    float ola_tail[kNHop];                 // audio_output[8000..8160)
    float last_sample;                     // resample carry
    float pad[3];
};

// one 0.5 s chunk of one lane inside a launch
struct ChunkDesc {
    const float* in;    // 24000 samples of this chunk (device)
    float* den;         // 24000 denoised samples out (device)
    const LaneCarry* carry_in; // read by the lane's first chunk of a launch (never null)
    LaneCarry* carry_out;      // written by the lane's last chunk of a launch (!= carry_in)
    uint32_t first;     // first chunk of its lane in this launch -> history comes from carry_in
    uint32_t last;      // last chunk of its lane in this launch -> writes carry_out
    float* rms;         // where this chunk's RMS goes (device), or null
    // 16-bit transport (optional): when in16 != null the chunk's samples are PCM16 and K1 converts them while
    // loading, x = (float)s * (1 / 32768) -- the decode of AudioFileStream.zig:56-102 / host_io.cpp, exact in
    // f32 -- so a 48 kHz stream crosses PCIe and HBM at 2 bytes per sample; den16 != null: K3 also writes the
    // denoised chunk as PCM16, rint(clamp(y * 32768, -32768, 32767))
    const int16_t* in16;
    int16_t* den16;
};

// one lane's worth of 1024-sample frames for K4
struct VadFftJob {
    const float* den;   // first frame's first sample (device, 8-byte aligned)
    float* band_sum;    // [n_frames]
    float* bins;        // [n_frames][n_fft/2 + 1] or null
    long n_frames;
};

// constant tables (device), built on the host in double like kissfft does and rounded once
struct FftTables {
    const float* win320;     // sqrt-Hann, NSNet2.zig:384-396
    const float* win320n;    // win320 * (1/320)  (NSNet2.zig:323,335)
    const float* tw160;      // [160][2] exp(-2 pi i j / 160)
    const float* st320;      // [80][2]  exp(-i pi ((k+1)/160 + 1/2))  real-FFT un-mixing
};

// tables of the VAD-side real FFT of n = 512 / 1024 / 2048 samples (device), built like FftTables
struct VadFftPlan {
    int n;
    const float* win;        // periodic Hann, window_fn.zig:22-28
    const float* tw;         // [n/2][2] exp(-2 pi i j / (n/2))
    const float* st;         // [n/4][2] real-FFT un-mixing
    float norm;              // windowNormFactor / n, BufferedFFT.zig:99
    // any other even size (FFT.init takes whatever kissfft factors, FFT.zig:35-60): the generic mixed-radix kernel, one
    // workgroup per frame, Stockham passes over the radices of n / 2 (4s first, then 2, 3, 5, ... like kissfft's kf_factor)
    int generic;             // 0: one of the wavefront sizes (512 / 1024 / 2048)
    int n_fac;
    int fac[14];
};

int fvad_launch_panel_gemm(const float* A, int lda, const float* Wfrag, const float* bias, float* C,
                           int ldc, long rows, int nt, int n_blocks, int S_steps, int act,
                           int map_T, int map_skip, hipStream_t stream, int n_valid_tiles = 0,
                           const unsigned* guard = nullptr);
// rows of a small-batch GEMM launch (panel_gemm_s_kernel): n_rows compact rows in ceil(n_rows / 64) panels; L > 0: compact row
// r = seq * L + tt is A's row seq * in_T + in_t0 + tt and C's row seq * out_T + out_t0 + tt (steps [t0, t0 + L) of every
// sequence); L == 0: rows as they are
struct GemmRowMap {
    int n_rows = 0, L = 0, in_T = 0, in_t0 = 0, out_T = 0, out_t0 = 0;
};
int fvad_launch_panel_gemm_s_rows(const float* A, int lda, const float* Wfrag, const float* bias, float* C, int ldc,
                                  GemmRowMap rm, int nt, int n_blocks, int S_steps, int act, hipStream_t stream,
                                  int n_valid_tiles = 0, const unsigned* guard = nullptr);
// small batches (kernels_nn.hip: panel_gemm_s_kernel): nt = 2 or 4 tiles per column block, S_steps in {11, 25, 38}
int fvad_launch_panel_gemm_s(const float* A, int lda, const float* Wfrag, const float* bias, float* C, int ldc,
                             long rows, int nt, int n_blocks, int S_steps, int act, int map_T, int map_skip,
                             hipStream_t stream, int n_valid_tiles = 0, const unsigned* guard = nullptr);
int fvad_launch_panel_gemm3(const float* A, int lda, const float* Wfrag, const float* bias, float* C,
                            int ldc, long rows, int nt, int n_blocks, int S_steps, int K, int act,
                            int n_valid_tiles, int map_T, int map_skip, int n_wg, hipStream_t stream);
// f16x3 form (kernels_h3.hip): Wfrag from pack_panel_h3, K = true reduction length, sx / sw = input / weight scales.
// in_ts: A in the split tiled layout (a_ld = K-steps per row tile) or row-major f32 [sequence][seq_T][a_ld];
// out: 0 row-major f32, 1 tiled f32 (c_ld unit tiles per row tile), 2 split tiled (c_ld K-steps, scaled by out_sx);
// row_tiles = output row tiles of 16 rows
int fvad_launch_panel_gemm_h3(const float* A, int in_ts, int a_ld, const float* Wfrag, const float* bias, float* C,
                              int out, int c_ld, int seq_T, long row_tiles, int nt, int n_blocks, int K, int act,
                              int n_valid_tiles, int map_T, int map_skip, float sx, float sw, float out_sx, int n_wg,
                              hipStream_t stream);
// gi: tiled f32; hsplit: h as split fragments (13 K-steps per row tile), the only form h exists in
int fvad_launch_gru_rec_h3(const float* gi, const float* Rfrag, const float* bR, float* hsplit,
                           long n_seq_pad, int T, int waves, float sx, float sw, hipStream_t stream);
// guard != nullptr: the kernel returns at once unless *guard != 0 (fallback behind fvad_launch_gru_ws)
// tile_major: gi rows are [25 J][3 gates][16] (large-batch GEMM) instead of [3 gates][400] (small-batch GEMM)
// the pipelined recurrence's whole fallback + pass count + reset of the polled words in one launch (kernels_nn.hip)
// feat != nullptr: gi does not hold layer 1's input projection yet (the pipelined kernel computed it in-kernel): the fallback
// computes it first from feat, W1frag_nt2 (the small-batch GEMM's 2-tile column blocks of W') and bG1 (tile-major)
int fvad_launch_gru_ws2_fallback(float* gi, const float* feat, const float* W1frag_nt2, const float* bG1, const float* R1frag, const float* bR1,
                                 const float* W2frag_nt2, const float* bW2, const float* R2frag, const float* bR2, float* h1, float* h2,
                                 long n_seq_pad, int T, unsigned* sync, unsigned long long* fallbacks, hipStream_t stream,
                                 int zero_at = 0, int zero_n = 0); // sync[zero_at, zero_at + zero_n): more words the last workgroup zeroes for the next pass
int fvad_launch_gru_lat(const float* gi, const float* R2frag, const float* bR, float* hout,
                        long n_seq_pad, int T, const unsigned* guard, int tile_major, hipStream_t stream, int row_tiles = 1);
// small batches: recurrent weights stationary in registers across 25 x G workgroups, h exchanged per step
// (kernels_ws.hip).  hx: fvad_gru_ws_exchange_floats(n_seq_pad) floats; flags: 256 zeroed words per launch;
// err: one zeroed word shared by the launches of a network pass.  Returns -1 when the batch is too large.
void fvad_launch_zero_words(unsigned* p, int n, hipStream_t stream);
// *counter += (*word != 0): how the context counts the network passes in which gru_ws gave up (a kernel node, so it
// also works inside a captured graph)
void fvad_launch_count_word(unsigned long long* counter, const unsigned* word, hipStream_t stream);
bool fvad_gru_ws_shape(long n_seq_pad, int n_cu, int* RT, int* G);
size_t fvad_gru_ws_exchange_floats(long n_seq_pad);
int fvad_launch_gru_ws(const float* gi, const float* R2frag, const float* bR, float* hout, float* hx, unsigned* flags,
                       unsigned* err, long n_seq_pad, int T, int n_cu, int tile_major, unsigned long long spin_ticks,
                       hipStream_t stream);
// any hidden size (16 J padded units, J <= 64): gi rows of gi_ld floats with gate g of unit tile j at g * 16 J + 16 j,
// bR [3][16 J], R2frag = pack_gru_r2 of the padded matrix, hout rows of h_ld floats
// both GRU layers in one launch, pipelined layer over layer (kernels_ws.hip gru_ws2_kernel): gi1 tile-major rows
// (layer 1's input projection), W2frag = pack_gru_r2 of layer 2's input weights, bW2 = its bias Wb [3][400];
// hx: fvad_gru_ws2_exchange_floats floats; flags: 512 zeroed words; returns -1 when the batch does not fit
bool fvad_gru_ws2_shape(long n_seq_pad, int n_cu, int* RT, int* G);
size_t fvad_gru_ws2_exchange_floats(long n_seq_pad);
// One row tile per group (up to 96 sequences): the 16-wavefront kernel, which computes layer 1's input projection itself from
// feat (rows of kFeatStride floats), W1frag = pack_gru_frag of W' = W_ih W_fc1 and bG1 = its bias, tile-major -- gi1 is not
// read and the GEMM in front is not needed (fvad_gru_ws2_gi1_in_kernel says which case a launch is)
bool fvad_gru_ws2_gi1_in_kernel(long n_seq_pad, int T, int n_cu, int variant);
// whether any of the pipelined kernels takes a launch of this shape (gru_ws2k: one row tile per group; gru_ws2m: 2..16 row tiles
// per group; gru_ws2: the 8-wavefront form, up to 4, which ws2_variant bit 8 forces) and the name of the one that does
bool fvad_gru_ws2_ok(long n_seq_pad, int T, int n_cu, int variant);
const char* fvad_gru_ws2_kernel_name(long n_seq_pad, int T, int n_cu, int variant);
int fvad_launch_gru_ws2(const float* gi1, const float* feat, const float* W1frag, const float* bG1, const float* R1frag, const float* bR1,
                        const float* W2frag, const float* bW2, const float* R2frag, const float* bR2, float* hout2, float* hx,
                        unsigned* flags, unsigned* err, long n_seq_pad, int T, int n_cu, unsigned long long spin_ticks, int variant,
                        unsigned waits, hipStream_t stream, unsigned* lsync = nullptr); // waits: gru_ws2k's first-poll waits (layer 1 | layer 2 << 16, 10 ns ticks); 0 = built in
// lsync: kWs2LocalWords zeroed words (layer 1's XCD-local flags and the placement tickets): gru_ws2k's layer 1 exchanges h1 inside
// one XCD where fvad_gru_ws2_local_layer1 says the launch's shape allows it
constexpr int kWs2LocalWords = 336;
bool fvad_gru_ws2_local_layer1(long n_seq_pad, int T, int n_cu, int variant);
// which table of built-in first-poll waits gru_ws2k uses for this launch: 0 not that kernel, 1 groups of 25 + 25, 2 groups of 13 + 25,
// 3 groups of 13 + 25 with layer 1's input projection in the kernel (what ws2_calibrate measures and overrides)
int fvad_gru_ws2_wait_class(long n_seq_pad, int T, int n_cu, int variant);
unsigned fvad_gru_ws2_builtin_waits(int wait_class, bool local_layer1 = false); // the table's entry, packed like `waits` (local_layer1: the table of launches whose layer 1 exchanges inside one XCD)
int fvad_launch_gru_gen(const float* gi, int gi_ld, const float* R2frag, const float* bR, float* hout, int h_ld,
                        long n_seq_pad, int T, int J, hipStream_t stream);
int fvad_launch_gru_rec3(const float* gi, const float* R2frag, const float* bR, float* hout,
                         long n_seq_pad, int T, int waves, hipStream_t stream, float* hs3 = nullptr);
// bf16x3 form of the dense layers (kernels_b3.hip): Wfrag from pack_panel_b3, K = true reduction length.
// in_ts: A in the three-piece tiled layout TS3 (a_ld = K-steps per row tile) or row-major f32 [sequence][seq_T][a_ld];
// out: 0 row-major f32 [sequence][seq_T][c_ld], 2 TS3 (c_ld K-steps per row tile); row_tiles = output row tiles of 16 rows
int fvad_launch_panel_gemm_b3(const float* A, int in_ts, int a_ld, const float* Wfrag, const float* bias, float* C,
                              int out, int c_ld, int seq_T, long row_tiles, int nt, int n_blocks, int K, int act,
                              int n_valid_tiles, int map_T, int map_skip, int n_wg, hipStream_t stream);

// K1: per chunk: RMS, decimate, STFT-320, log-power features (+ warm-up rows)
// (parts: 1, or 2 / 3 to cut a chunk's frames over several workgroups in launches of a few chunks; same bits)
void fvad_launch_stft(const ChunkDesc* descs, int n_chunks, FftTables tb, float* feat,
                      float* spec, hipStream_t stream, int parts = 1);
// K3: per chunk: gain, inverse STFT, overlap-add, x3 upsample
void fvad_launch_istft(const ChunkDesc* descs, int n_chunks, FftTables tb, const float* spec,
                       const float* gains, int gains_rows_per_chunk, int gains_row0,
                       hipStream_t stream, int parts = 1);
// K4: n-point periodic-Hann rFFT magnitude + band sum over [min_bin, max_bin], n = pl.n
void fvad_launch_vadfft(const float* den, long n_frames, VadFftPlan pl, int min_bin, int max_bin,
                        float* band_sum, float* bins_or_null, hipStream_t stream);
// all lanes in one launch: jobs is a device array of n_jobs entries, max_frames = max n_frames
// any_bins: some job has a `bins` tap (the full-spectrum kernel must run; band sums come from the band kernel either way);
// plain_loads: the band kernel stages its frames with plain 8-byte loads, the path of unaligned jobs (context option k4_plain_loads)
// (returns a hipError_t as int: a failed hipFuncSetAttribute must not leave the caller with stale band sums and FVAD_OK)
int fvad_launch_vadfft_jobs(const VadFftJob* jobs, int n_jobs, long max_frames, VadFftPlan pl,
                            int min_bin, int max_bin, hipStream_t stream, int any_bins, int n_cu, int plain_loads = 0);
// batched FFT.fft for B3 / BASELINE config 2: n_fft in {320, 512, 1024, 2048} (pl is used for n_fft != 320)
int fvad_launch_rfft_batch(const float* frames, long n_frames, int n_fft, const float* window,
                           FftTables tb, VadFftPlan pl, float* bins_or_null, float* mag_or_null,
                           hipStream_t stream); // hipError_t as int
void fvad_launch_irfft_batch(const float* bins, long n_frames, FftTables tb, float* out,
                             hipStream_t stream);
// FFT.invFft for any even size (pl.generic plans; tables of the FORWARD transform, conjugated in the kernel): bins
// [n_frames][n/2 + 1][2] -> out [n_frames][n], unscaled like kiss_fftri
int fvad_launch_irfft_generic(const float* bins, long n_frames, VadFftPlan pl, float* out, hipStream_t stream); // hipError_t as int
