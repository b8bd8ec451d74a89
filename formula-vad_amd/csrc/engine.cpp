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

static int dev_alloc(fvad_ctx* ctx, float** p, size_t n_floats, bool zero)
{
    FVAD_HIP(ctx, hipMalloc((void**)p, n_floats * sizeof(float)));
    if (zero) FVAD_HIP(ctx, hipMemsetAsync(*p, 0, n_floats * sizeof(float), ctx->stream));
    return FVAD_OK;
}

static int upload(fvad_ctx* ctx, DevBuf& b, const std::vector<float>& v)
{
    // a captured launch sequence (Workspace::GraphCache) holds this buffer's address in its kernel nodes
    ctx->ws.generation++;
    if (b.p) { hipStreamSynchronize(ctx->stream); hipFree(b.p); b.p = nullptr; }
    b.n = v.size();
    FVAD_HIP(ctx, hipMalloc((void**)&b.p, v.size() * sizeof(float)));
    FVAD_HIP(ctx, hipMemcpy(b.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return FVAD_OK;
}

static std::vector<float> padded(const float* b, size_t n, size_t n_pad)
{
    std::vector<float> v(n_pad, 0.0f);
    std::copy(b, b + n, v.begin());
    return v;
}

// A model whose dimensions are not NSNet2-baseline's: every layer packed for the run-time-sized kernels
// (panel_gemm_kernel<8, .> column blocks of 128, gru_gen_kernel), widths padded with zero weights and biases.
static int upload_model_generic(fvad_ctx* ctx)
{
    const HostWeights& w = ctx->hw;
    DeviceModel& m = ctx->dm;
    DeviceModel::GenDims& g = m.gd;
    g.F1 = w.n_fc1; g.H = w.n_hidden; g.N2 = w.n_fc2; g.N3 = w.n_fc3;
    g.J = (g.H + 15) / 16;
    g.Hp = 16 * g.J;
    auto pad128 = [](int n) { return (n + 127) / 128 * 128; };
    g.F1p = pad128(g.F1); g.Gp = pad128(3 * g.Hp); g.N2p = pad128(g.N2); g.N3p = pad128(g.N3);
    std::vector<float> f;
    int rc;
    auto dense = [&](const std::vector<float>& W, const std::vector<float>& b, int N, int K, int Np, DevBuf& dw, DevBuf& db) -> int {
        pack_panel(W.data(), N, K, Np / 128, 8, (K + 15) / 16, f);
        int r = upload(ctx, dw, f);
        if (r) return r;
        return upload(ctx, db, padded(b.data(), (size_t)N, (size_t)Np));
    };
    // GRU tensors with every gate padded from H to Hp rows (and R's columns to Hp)
    auto gru = [&](const std::vector<float>& W, const std::vector<float>& R, const std::vector<float>& B, int K, DevBuf& dw, DevBuf& db,
                   DevBuf& dr, DevBuf& dbr) -> int {
        const int H = g.H, Hp = g.Hp;
        std::vector<float> Wp((size_t)3 * Hp * K, 0.0f), Rp((size_t)3 * Hp * Hp, 0.0f), wb((size_t)g.Gp, 0.0f), rb((size_t)3 * Hp, 0.0f);
        for (int gate = 0; gate < 3; ++gate)
            for (int u = 0; u < H; ++u) {
                std::copy(W.begin() + (size_t)(gate * H + u) * K, W.begin() + (size_t)(gate * H + u + 1) * K, Wp.begin() + (size_t)(gate * Hp + u) * K);
                std::copy(R.begin() + (size_t)(gate * H + u) * H, R.begin() + (size_t)(gate * H + u + 1) * H, Rp.begin() + (size_t)(gate * Hp + u) * Hp);
                wb[(size_t)gate * Hp + u] = B[(size_t)gate * H + u];
                rb[(size_t)gate * Hp + u] = B[(size_t)(3 + gate) * H + u];
            }
        pack_panel(Wp.data(), 3 * Hp, K, g.Gp / 128, 8, (K + 15) / 16, f);
        int r = upload(ctx, dw, f);
        if (r) return r;
        if ((r = upload(ctx, db, wb))) return r;
        pack_gru_r2(Rp.data(), Hp, f);
        if ((r = upload(ctx, dr, f))) return r;
        return upload(ctx, dbr, rb);
    };
    if ((rc = dense(w.fc1_w, w.fc1_b, g.F1, 161, g.F1p, m.g_fc1_w, m.g_fc1_b))) return rc;
    if ((rc = gru(w.gru1_w, w.gru1_r, w.gru1_b, g.F1, m.g_gi1_w, m.g_gi1_b, m.g_r1, m.g_br1))) return rc;
    if ((rc = gru(w.gru2_w, w.gru2_r, w.gru2_b, g.H, m.g_gi2_w, m.g_gi2_b, m.g_r2, m.g_br2))) return rc;
    if ((rc = dense(w.fc2_w, w.fc2_b, g.N2, g.H, g.N2p, m.g_fc2_w, m.g_fc2_b))) return rc;
    if ((rc = dense(w.fc3_w, w.fc3_b, g.N3, g.N2, g.N3p, m.g_fc3_w, m.g_fc3_b))) return rc;
    // fc4: 161 outputs = one block of 11 tiles (the gains rows are 176 floats wide)
    pack_panel(w.fc4_w.data(), 161, g.N3, 1, 11, (g.N3 + 15) / 16, f);
    if ((rc = upload(ctx, m.g_fc4_w, f))) return rc;
    if ((rc = upload(ctx, m.g_fc4_b, padded(w.fc4_b.data(), 161, 176)))) return rc;
    m.w_a1 = g.F1p; m.w_gi = g.Gp; m.w_h = g.Hp; m.w_f = std::max(g.N2p, g.N3p);
    m.generic = true;
    m.h3_ok = false;
    m.loaded = true;
    return FVAD_OK;
}

int upload_model(fvad_ctx* ctx)
{
    const HostWeights& w = ctx->hw;
    std::string err;
    if (!w.check_dims(err)) return set_err(ctx, FVAD_ERR_MODEL_FORMAT, err);
    DeviceModel& m = ctx->dm;
    m.loaded = false;
    if (!w.is_baseline()) return upload_model_generic(ctx);
    m.generic = false;
    m.w_a1 = 400; m.w_gi = 1200; m.w_h = 400; m.w_f = 640;
    const int H = 400;
    std::vector<float> f, gi1f_folded; // gi1f_folded: fc1 folded into GRU1's input projection, [1200][161]
    int rc;
    // fc1: 161 -> 400, K padded to 176 (11 super-steps), one block of 25 tiles
    pack_panel(w.fc1_w.data(), 400, 161, 1, 25, 11, f);
    if ((rc = upload(ctx, m.fc1_w, f))) return rc;
    if ((rc = upload(ctx, m.fc1_b, w.fc1_b))) return rc;
    // recurrent biases Rb and recurrent weights as one 75 KB slab per unit tile
    if ((rc = upload(ctx, m.br1, std::vector<float>(w.gru1_b.begin() + 3 * H, w.gru1_b.end())))) return rc;
    pack_gru_r2(w.gru1_r.data(), H, f);
    if ((rc = upload(ctx, m.r1v2, f))) return rc;
    if ((rc = upload(ctx, m.br2, std::vector<float>(w.gru2_b.begin() + 3 * H, w.gru2_b.end())))) return rc;
    pack_gru_r2(w.gru2_r.data(), H, f);
    if ((rc = upload(ctx, m.r2v2, f))) return rc;
    // large-batch layouts: 1200 = 5 column blocks of 15 tiles, output units in TILE-major order
    // (new row 48 J + 16 g + u = old row 400 g + 16 J + u): the GEMM then writes gi rows as
    // [25 J][3 gates][16 units], what gru_rec3_kernel reads 192 contiguous bytes at a time
    auto tile_major_rows = [&](const float* W, int K) {
        std::vector<float> out((size_t)3 * H * K);
        for (int J = 0; J < 25; ++J)
            for (int g = 0; g < 3; ++g)
                for (int u = 0; u < 16; ++u)
                    std::copy(W + (size_t)(g * H + 16 * J + u) * K, W + (size_t)(g * H + 16 * J + u + 1) * K,
                              out.begin() + (size_t)(48 * J + 16 * g + u) * K);
        return out;
    };
    pack_panel(tile_major_rows(w.gru1_w.data(), 400).data(), 1200, 400, 5, 15, 25, f);
    if ((rc = upload(ctx, m.gi1v2_w, f))) return rc;
    pack_panel(tile_major_rows(w.gru2_w.data(), 400).data(), 1200, 400, 5, 15, 25, f);
    if ((rc = upload(ctx, m.gi2v2_w, f))) return rc;
    if ((rc = upload(ctx, m.gi1_btm, tile_major_rows(w.gru1_b.data(), 1)))) return rc; // Wb only (unfolded fc1 path)
    {
        // fc1 has no activation (x = fc1(x); x, _ = rnn1(x)), so fc1 followed by GRU1's input
        // projection is one linear map 161 -> 1200: W' = W_ih W_fc1, b' = W_ih b_fc1 + Wb.  Folded
        // once on the host in double and rounded to f32: algebraically exact, differs from the
        // two-GEMM form only by round-off (~1e-7 rel), and removes 12 % of the network's FLOPs.
        std::vector<float>& wf = gi1f_folded;
        wf.assign((size_t)1200 * 161, 0.0f);
        std::vector<float> bf(1200);
        std::vector<double> row(161);
        for (int o = 0; o < 1200; ++o) {
            std::fill(row.begin(), row.end(), 0.0);
            double b = (double)w.gru1_b[o];
            const float* wi = w.gru1_w.data() + (size_t)o * 400;
            for (int j = 0; j < 400; ++j) {
                const double a = (double)wi[j];
                const float* f1 = w.fc1_w.data() + (size_t)j * 161;
                for (int k = 0; k < 161; ++k) row[k] += a * (double)f1[k];
                b += a * (double)w.fc1_b[j];
            }
            for (int k = 0; k < 161; ++k) wf[(size_t)o * 161 + k] = (float)row[k];
            bf[o] = (float)b;
        }
        pack_panel(tile_major_rows(wf.data(), 161).data(), 1200, 161, 5, 15, 11, f);
        if ((rc = upload(ctx, m.gi1f_w, f))) return rc;
        // small batches (panel_gemm_s_kernel): the same matrices cut into column blocks of 2 tiles (launches of up to
        // ~2000 rows) and of 4 tiles (larger ones): 75 unit tiles -> 38 / 19 blocks, the 76th tile never stored
        for (int fam = 0; fam < 2; ++fam) {
            const int nt = fam ? 4 : 2, nb = (75 + nt - 1) / nt;
            pack_panel(tile_major_rows(wf.data(), 161).data(), 1200, 161, nb, nt, 11, f);
            if ((rc = upload(ctx, m.s_gi1f_w[fam], f))) return rc;
            pack_panel(tile_major_rows(w.gru2_w.data(), 400).data(), 1200, 400, nb, nt, 25, f);
            if ((rc = upload(ctx, m.s_gi2_w[fam], f))) return rc;
        }
        // gru_ws2_kernel computes layer 2's input projection itself: W_ih2 as stationary fragments like R, Wb gate-major
        pack_gru_r2(w.gru2_w.data(), H, f);
        if ((rc = upload(ctx, m.s_w2frag, f))) return rc;
        // ... and gru_ws2k_kernel layer 1's as well: the folded W' (gate-major rows, K = 161) in the same fragment order
        pack_gru_frag(wf.data(), H, 161, f);
        if ((rc = upload(ctx, m.s_w1frag, f))) return rc;
        if ((rc = upload(ctx, m.s_bw2, std::vector<float>(w.gru2_b.begin(), w.gru2_b.begin() + 3 * H)))) return rc;
        if ((rc = upload(ctx, m.gi1f_b, tile_major_rows(bf.data(), 1)))) return rc;
        // f16x3 form of the same folded layer; its input, the log-power features, is bounded by log10 of the
        // largest f32 squared (NSNet2.zig:266-287)
        {
            const std::vector<float> wt = tile_major_rows(wf.data(), 161);
            m.h3_gi1f.sw = h3_weight_scale(wt.data(), wt.size());
            m.h3_gi1f.sx = h3_activation_scale(80.0);
            pack_panel_h3(wt.data(), 1200, 161, 5, 15, m.h3_gi1f.sw, f);
            if ((rc = upload(ctx, m.gi1f_h3, f))) return rc;
        }
        // gru_rec3_kernel adds only the n-gate recurrent bias itself: for z and r, Wb + Rb is one constant
        for (int o = 0; o < 2 * H; ++o) bf[o] += w.gru1_b[3 * H + o];
        if ((rc = upload(ctx, m.gi1f_bzr, tile_major_rows(bf.data(), 1)))) return rc;
        std::vector<float> b2(w.gru2_b.begin(), w.gru2_b.begin() + 3 * H);
        if ((rc = upload(ctx, m.gi2_btm, tile_major_rows(b2.data(), 1)))) return rc;
        for (int o = 0; o < 2 * H; ++o) b2[o] += w.gru2_b[3 * H + o];
        if ((rc = upload(ctx, m.gi2_bzr, tile_major_rows(b2.data(), 1)))) return rc;
    }
    // small batches: fc2 400 -> 600 and fc3 600 -> 600 as 19 column blocks of 2 tiles or 10 of 4 (rows of 640 floats,
    // K of fc3 padded to 608 = 38 super-steps), fc4 600 -> 161 as 6 blocks of 2 or 3 of 4, of which 11 tiles are stored
    for (int fam = 0; fam < 2; ++fam) {
        const int nt = fam ? 4 : 2;
        pack_panel(w.fc2_w.data(), 600, 400, (38 + nt - 1) / nt, nt, 25, f);
        if ((rc = upload(ctx, m.s_fc2_w[fam], f))) return rc;
        pack_panel(w.fc3_w.data(), 600, 600, (38 + nt - 1) / nt, nt, 38, f);
        if ((rc = upload(ctx, m.s_fc3_w[fam], f))) return rc;
        pack_panel(w.fc4_w.data(), 161, 600, (11 + nt - 1) / nt, nt, 38, f);
        if ((rc = upload(ctx, m.s_fc4_w[fam], f))) return rc;
    }
    if ((rc = upload(ctx, m.fc2_b, padded(w.fc2_b.data(), 600, 640)))) return rc;
    if ((rc = upload(ctx, m.fc3_b, padded(w.fc3_b.data(), 600, 640)))) return rc;
    if ((rc = upload(ctx, m.s_fc4_b, padded(w.fc4_b.data(), 161, 192)))) return rc;
    // the same two layers as 3 blocks of 13 tiles (39 tiles, the 39th is padding and never stored):
    // 104 accumulator + 52 fragment registers fit the persistent kernel, 19-tile blocks do not
    pack_panel(w.fc2_w.data(), 600, 400, 3, 13, 25, f);
    if ((rc = upload(ctx, m.fc2v3_w, f))) return rc;
    if ((rc = upload(ctx, m.fc2v3_b, padded(w.fc2_b.data(), 600, 624)))) return rc;
    pack_panel(w.fc3_w.data(), 600, 600, 3, 13, 38, f);
    if ((rc = upload(ctx, m.fc3v3_w, f))) return rc;
    if ((rc = upload(ctx, m.fc3v3_b, padded(w.fc3_b.data(), 600, 624)))) return rc;
    // fc4: 600 -> 161 (N padded to 176 = 11 tiles)
    pack_panel(w.fc4_w.data(), 161, 600, 1, 11, 38, f);
    if ((rc = upload(ctx, m.fc4_w, f))) return rc;
    if ((rc = upload(ctx, m.fc4_b, padded(w.fc4_b.data(), 161, 176)))) return rc;
    // ---- f16x3 layouts of the remaining layers.  Input scales from rigorous bounds: GRU states lie in (-1, 1)
    // (h = (1 - z) n + z h with |n| < 1, z in (0, 1), h_0 = 0); a dense layer's outputs are bounded by its
    // rows' l1 norms times the input bound plus the bias.
    {
        auto l1_bound = [](const std::vector<float>& W, const std::vector<float>& b, int N, int K, double in_bound) {
            double mx = 0.0;
            for (int n = 0; n < N; ++n) {
                double a = 0.0;
                for (int k = 0; k < K; ++k) a += fabs((double)W[(size_t)n * K + k]);
                mx = std::max(mx, a * in_bound + fabs((double)b[n]));
            }
            return mx;
        };
        const double b_fc2 = l1_bound(w.fc2_w, w.fc2_b, 600, 400, 1.0);
        const double b_fc3 = l1_bound(w.fc3_w, w.fc3_b, 600, 600, b_fc2);
        const std::vector<float> g2 = tile_major_rows(w.gru2_w.data(), 400);
        m.h3_gi2 = {h3_weight_scale(g2.data(), g2.size()), h3_activation_scale(1.0)};
        pack_panel_h3(g2.data(), 1200, 400, 5, 15, m.h3_gi2.sw, f);
        if ((rc = upload(ctx, m.gi2_h3, f))) return rc;
        m.h3_fc2 = {h3_weight_scale(w.fc2_w.data(), w.fc2_w.size()), h3_activation_scale(1.0)};
        pack_panel_h3(w.fc2_w.data(), 600, 400, 4, 10, m.h3_fc2.sw, f);
        if ((rc = upload(ctx, m.fc2_h3, f))) return rc;
        if ((rc = upload(ctx, m.fc2h3_b, padded(w.fc2_b.data(), 600, 640)))) return rc;
        m.h3_fc3 = {h3_weight_scale(w.fc3_w.data(), w.fc3_w.size()), h3_activation_scale(b_fc2)};
        pack_panel_h3(w.fc3_w.data(), 600, 600, 4, 10, m.h3_fc3.sw, f);
        if ((rc = upload(ctx, m.fc3_h3, f))) return rc;
        if ((rc = upload(ctx, m.fc3h3_b, padded(w.fc3_b.data(), 600, 640)))) return rc;
        m.h3_fc4 = {h3_weight_scale(w.fc4_w.data(), w.fc4_w.size()), h3_activation_scale(b_fc3)};
        pack_panel_h3(w.fc4_w.data(), 161, 600, 1, 12, m.h3_fc4.sw, f);
        if ((rc = upload(ctx, m.fc4_h3, f))) return rc;
        if ((rc = upload(ctx, m.fc4h3_b, padded(w.fc4_b.data(), 161, 192)))) return rc;
        m.h3_r1 = {h3_weight_scale(w.gru1_r.data(), w.gru1_r.size()), h3_activation_scale(1.0)};
        pack_gru_r_h3(w.gru1_r.data(), H, m.h3_r1.sw, f);
        if ((rc = upload(ctx, m.r1_h3, f))) return rc;
        m.h3_r2 = {h3_weight_scale(w.gru2_r.data(), w.gru2_r.size()), h3_activation_scale(1.0)};
        pack_gru_r_h3(w.gru2_r.data(), H, m.h3_r2.sw, f);
        if ((rc = upload(ctx, m.r2_h3, f))) return rc;
        // the scales are finite powers of two whenever the weights and the bounds are finite
        // ... and the split keeps its 22 bits only for values within ~18 binades below the bound: a model whose l1
        // bounds are absurdly loose (activations expected around 1 against a bound above 2^17) keeps the f32 kernels
        // bf16x3 layouts of the same five layers (kernels_b3.hip): three exact pieces per weight, no scales, no bounds
        pack_panel_b3(tile_major_rows(gi1f_folded.data(), 161).data(), 1200, 161, 5, 15, f);
        if ((rc = upload(ctx, m.gi1f_b3, f))) return rc;
        pack_panel_b3(g2.data(), 1200, 400, 5, 15, f);
        if ((rc = upload(ctx, m.gi2_b3, f))) return rc;
        pack_panel_b3(w.fc2_w.data(), 600, 400, 4, 10, f);
        if ((rc = upload(ctx, m.fc2_b3, f))) return rc;
        pack_panel_b3(w.fc3_w.data(), 600, 600, 4, 10, f);
        if ((rc = upload(ctx, m.fc3_b3, f))) return rc;
        pack_panel_b3(w.fc4_w.data(), 161, 600, 1, 12, f);
        if ((rc = upload(ctx, m.fc4_b3, f))) return rc;
        m.h3_ok = std::isfinite(b_fc3) && b_fc3 <= 131072.0 && b_fc2 <= 131072.0;
        for (const DeviceModel::H3Scale* sc : {&m.h3_gi1f, &m.h3_gi2, &m.h3_fc2, &m.h3_fc3, &m.h3_fc4, &m.h3_r1, &m.h3_r2})
            m.h3_ok = m.h3_ok && std::isfinite(sc->sw) && std::isfinite(sc->sx) && sc->sw > 0.0f && sc->sx > 0.0f &&
                      std::isfinite(sc->sw * sc->sx) && std::isfinite(1.0f / (sc->sw * sc->sx));
    }
    m.loaded = true;
    return FVAD_OK;
}

static void free_workspace_nn(Workspace& ws)
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

static long padded_batch(const fvad_ctx* ctx, long n, int T, int skip);

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

// Large batches: the LDS-DMA kernels with 192 / 128 / 64 sequences per workgroup; small batches keep
// one wavefront (16 sequences) per workgroup so that more CUs take part.
struct GruChoice {
    int version; // 3: gru_rec3 (expects the z/r recurrent biases folded into gi),
                 // 4: gru_lat (16 sequences per workgroup, tiles split over 8 waves),
                 // 5: gru_ws (weights stationary in registers across 25 x G workgroups, kernels_ws.hip)
    int waves;
};

constexpr size_t kWsSyncWords = 520 + 2000; // 2 x 256 flags + error word, padded to a multiple of 16 bytes; then gru_ws2k's step trace (ws2_variant 64)

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
    // gru_ws2k (one row tile per group: the hand-off chain alone) / gru_ws2m (row tiles streamed: ~5.5k clocks each, MFMA-paced)
    return RT == 1 ? 55.0 * 14e3 : 55.0 * (8e3 + 5.5e3 * RT);
}

static double gru_cost(long n_pad, int waves, int n_cu)
{
    const double per_step = waves == 12 ? 25 * 32.3e3 : waves == 8 ? 25 * 23.4e3 : waves == 4 ? 25 * 13.0e3 : 120e3;
    const long wgs = n_pad / (waves ? 16 * waves : 16);
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

static long padded_batch(const fvad_ctx* ctx, long n, int T, int skip)
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
    if (c.version == 4) return fvad_launch_gru_lat(gi, r_v2.p, bR, hout, n_pad, T, nullptr, tile_major, ctx->stream);
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

int run_nn(fvad_ctx* ctx, long n_pad, int T, int skip)
{
    if (!ctx->dm.loaded) return set_err(ctx, FVAD_ERR_NO_MODEL, "NSNet2 weights not loaded");
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
    const int nb_gi = (75 + snt - 1) / snt, nb_fc = (38 + snt - 1) / snt, nb_fc4 = (11 + snt - 1) / snt;
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
                                       ws.hx, ws.ws_sync, err, n_pad, T, ctx->n_cu, ws_spin_deadline(ctx, gru_ws2_cost_both_layers(n_pad, T, ctx->n_cu, tn.ws2_variant) * T / 55.0), tn.ws2_variant, st);
        });
        // one launch behind it: the whole fallback (layer 1, layer 2's input projection, layer 2 -- run only if the
        // error word was raised), the pass count, and the reset of the polled words for the next pass
        rc |= fvad_launch_gru_ws2_fallback(ws.gi, gi1_in_kernel ? ws.feat : nullptr, m.s_gi1f_w[0].p, m.gi1f_b.p, m.r1v2.p, m.br1.p, m.s_gi2_w[0].p,
                                           m.gi2_btm.p, m.r2v2.p, m.br2.p, ws.h1, ws.h2, n_pad, T, ws.ws_sync, ws.ws_fallbacks, st);
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
    time_begin(ctx, "fc2_gemm");
    rc |= fvad_launch_panel_gemm_s(ws.h2, 400, m.s_fc2_w[fam].p, m.fc2_b.p, ws.f2, 640, rows_out, snt, nb_fc, 25, FVAD_ACT_RELU, skip ? T : 0, skip, st);
    time_end(ctx);
    time_begin(ctx, "fc3_gemm");
    rc |= fvad_launch_panel_gemm_s(ws.f2, 640, m.s_fc3_w[fam].p, m.fc3_b.p, ws.f3, 640, rows_out, snt, nb_fc, 38, FVAD_ACT_RELU, 0, 0, st);
    time_end(ctx);
    time_begin(ctx, "fc4_gemm");
    rc |= fvad_launch_panel_gemm_s(ws.f3, 640, m.s_fc4_w[fam].p, m.s_fc4_b.p, ws.gains, kFeatStride, rows_out, snt, nb_fc4, 38, FVAD_ACT_SIGMOID, 0, 0, st, 11);
    time_end(ctx);
    if (rc) return set_err(ctx, FVAD_ERR_INVALID_ARGUMENT, "no kernel instance for this layer shape");
    FVAD_HIP(ctx, hipGetLastError());
    return FVAD_OK;
}

// Chunks per launch of a call over `total` chunks; max_chunks <= 0: the caller leaves it to the engine.
// Launch planning (only then, f32, default kernel selection): between the largest batch the pipelined recurrence takes
// (1536 chunks: 16 row tiles per group) and ~2700 chunks ONE launch would fall to the large-batch family's low-latency
// recurrence, which keeps 128-170 of the 256 CUs busy (2048 chunks: 7.4 ms = 13.8 M frames/s); two launches of half the size
// stay on the pipelined kernels (2 x 3.1 ms = 16.5 M frames/s).  Measured crossover (bench.py's batch curve): 4.86 ms + 1.26 us
// per chunk against 3.03 us per chunk.
static long planned_max_chunks(const fvad_ctx* ctx, long total, long max_chunks)
{
    if (max_chunks > 0) return max_chunks;
    max_chunks = ctx->tune.max_chunks;
    const Tuning& tn = ctx->tune;
    if (!ctx->dm.generic && nn_math_effective(ctx) == FVAD_NN_MATH_F32 && !tn.reproducible && tn.gru_kernel.empty() && tn.gemm_kernel.empty() &&
        total > 1536 && total <= 2700 && max_chunks >= total)
        max_chunks = (total / 2 + 15) / 16 * 16;
    return max_chunks;
}

// K1 -> NSNet2 -> K3 over every chunk of every job, in launches of <= max_chunks chunks.
int run_chunks(fvad_ctx* ctx, std::vector<LaneJob>& jobs, long max_chunks, ChunkDesc* capture_descs, ChunkDesc* capture_dev)
{
    // capture_descs != nullptr: the call is being captured into a hipGraph.  Every launch gets its own
    // region of the graph's private descriptor table (host copy capture_descs, device copy capture_dev,
    // uploaded once by the caller after the capture): the graph holds no copy node and does not depend on
    // the workspace's shared table, which direct calls overwrite.  No event is waited for or recorded.
    size_t capture_off = 0;
    long total = 0;
    for (auto& j : jobs) total += (long)j.n_chunks;
    if (total == 0) return FVAD_OK;
    max_chunks = planned_max_chunks(ctx, total, max_chunks);
    int rc = ensure_workspace(ctx, std::min(total, max_chunks), kRowsPerChunk, kWarmupRows, total % std::min(total, max_chunks));
    if (rc) return rc;
    Workspace& ws = ctx->ws;
    const long cap = std::min<long>(max_chunks, ws.cap_chunks);

    size_t job = 0, chunk_in_job = 0;
    while (job < jobs.size()) {
        // fill one launch, lane-contiguous
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
        rc = run_nn(ctx, n_pad, kRowsPerChunk, kWarmupRows);
        if (rc) return rc;
        time_begin(ctx, "istft320_ola_up3");
        fvad_launch_istft(dd, (int)n, ctx->tb, ws.spec, ws.gains, kFramesPerChunk, 0, ctx->stream, fft_parts);
        time_end(ctx);
        for (size_t t : touched) jobs[t].cur ^= 1;
        job = j;
        chunk_in_job = c;
    }
    FVAD_HIP(ctx, hipGetLastError());
    return FVAD_OK;
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
        if (!unset && (!to_long(c) || c < 0 || c >= (1 << 25))) return FVAD_ERR_INVALID_ARGUMENT;
#if !FVAD_DIAG
        // the timing-only bits (1, 2, 4, 32: WRONG results) and the step trace (64) exist in the diagnostics build only
        // (make diag -> libfvad_hip_diag.so); the shipping library has no way to ask for wrong results
        if (c & (1 | 2 | 4 | 32 | 64 | 256 | 512 | 4096 | 8192 | 16384)) return FVAD_ERR_INVALID_ARGUMENT;
#endif
        tn.ws2_variant = (int)c;
    } else if (name == "no_pipeline") { if (!to_bool(tn.no_pipeline)) return FVAD_ERR_INVALID_ARGUMENT; }
    else if (name == "trace_kernels") { if (!to_bool(tn.trace_kernels)) return FVAD_ERR_INVALID_ARGUMENT; }
    else if (name == "reproducible") { if (!to_bool(tn.reproducible)) return FVAD_ERR_INVALID_ARGUMENT; }
    else return FVAD_ERR_INVALID_ARGUMENT;
    ctx->ws.generation++; // a captured launch sequence holds the kernels of the old selection
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
                            "no_pipeline", "trace_kernels", "reproducible"}) {
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
    rc = run_nn(ctx, n_pad, (int)T, 0);
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

static int grow(fvad_ctx* ctx, float** p, size_t* cap, size_t need)
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

} // extern "C"

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

void parallel_memcpy(const std::vector<CopySeg>& blocks, size_t first, size_t n, char* slots, bool to_pinned, int n_threads)
{
    auto work = [&](size_t t) {
        for (size_t i = t; i < n; i += (size_t)n_threads) {
            const CopySeg& b = blocks[first + i];
            if (to_pinned) memcpy(slots + i * kPinSlotBytes, b.host, b.bytes);
            else memcpy(b.host, slots + i * kPinSlotBytes, b.bytes);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads && (size_t)t < n; ++t) th.emplace_back(work, (size_t)t);
    work(0);
    for (auto& x : th) x.join();
}

// host -> device (to_device) or device -> host, ordered on ctx->stream; returns after the last DMA has
// been enqueued (to_device) or after the data is in user memory (!to_device)
int staged_copy(fvad_ctx* ctx, const std::vector<CopySeg>& segs, bool to_device, hipStream_t st)
{
    Workspace::PinRing& ring = to_device ? ctx->ws.ring_in : ctx->ws.ring_out;
    size_t total = 0, total_padded = 0;
    for (const CopySeg& s : segs) { total += s.bytes; total_padded += (s.bytes + 63) & ~(size_t)63; }
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
        FVAD_HIP(ctx, hipStreamSynchronize(st));
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
    const size_t n_waves = (blocks.size() + kPinSlots - 1) / kPinSlots;
    auto wave_n = [&](size_t w) { return std::min((size_t)kPinSlots, blocks.size() - w * kPinSlots); };
    auto half = [&](size_t w) { return ring.base + (w & 1) * kPinSlots * kPinSlotBytes; };
    if (to_device) {
        for (size_t w = 0; w < n_waves; ++w) {
            if (w >= 2) FVAD_HIP(ctx, hipEventSynchronize(ring.ev[w & 1])); // this half's previous DMA is done
            parallel_memcpy(blocks, w * kPinSlots, wave_n(w), half(w), true, ctx->tune.copy_threads);
            for (size_t i = 0; i < wave_n(w); ++i) {
                const CopySeg& b = blocks[w * kPinSlots + i];
                FVAD_HIP(ctx, hipMemcpyAsync(b.dev, half(w) + i * kPinSlotBytes, b.bytes, hipMemcpyHostToDevice, st));
            }
            FVAD_HIP(ctx, hipEventRecord(ring.ev[w & 1], st));
        }
        // the ring may be reused by a later call: its last two halves must have left the host
        for (size_t w = (n_waves >= 2 ? n_waves - 2 : 0); w < n_waves; ++w) FVAD_HIP(ctx, hipEventSynchronize(ring.ev[w & 1]));
    } else {
        for (size_t w = 0; w <= n_waves; ++w) {
            if (w < n_waves) {
                for (size_t i = 0; i < wave_n(w); ++i) {
                    const CopySeg& b = blocks[w * kPinSlots + i];
                    FVAD_HIP(ctx, hipMemcpyAsync(half(w) + i * kPinSlotBytes, b.dev, b.bytes, hipMemcpyDeviceToHost, st));
                }
                FVAD_HIP(ctx, hipEventRecord(ring.ev[w & 1], st));
            }
            if (w >= 1) { // drain the previous wave while this one's DMA runs
                FVAD_HIP(ctx, hipEventSynchronize(ring.ev[(w - 1) & 1]));
                parallel_memcpy(blocks, (w - 1) * kPinSlots, wave_n(w - 1), half(w - 1), false, ctx->tune.copy_threads);
            }
        }
    }
    return FVAD_OK;
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
    if (!opts.on_device && n_lanes >= 8 && h2d_bytes >= (64u << 20) && !ctx->tune.no_pipeline) G = 4;
    if (G > 1) {
        if (!ws.copy_in) FVAD_HIP(ctx, hipStreamCreateWithFlags(&ws.copy_in, hipStreamNonBlocking));
        if (!ws.copy_out) FVAD_HIP(ctx, hipStreamCreateWithFlags(&ws.copy_out, hipStreamNonBlocking));
        for (int g = 0; g < G; ++g) {
            if (!ws.grp_in[g]) FVAD_HIP(ctx, hipEventCreateWithFlags(&ws.grp_in[g], hipEventDisableTiming));
            if (!ws.grp_k[g]) FVAD_HIP(ctx, hipEventCreateWithFlags(&ws.grp_k[g], hipEventDisableTiming));
        }
    }
    hipStream_t s_in = G > 1 ? ws.copy_in : st, s_out = G > 1 ? ws.copy_out : st;
    // group boundaries: contiguous lanes, about equal chunk counts
    std::vector<size_t> gb(G + 1, n_lanes);
    gb[0] = 0;
    {
        size_t acc = 0, g = 1;
        for (size_t l = 0; l < n_lanes && g < (size_t)G; ++l) {
            acc += lanes[l].n_chunks;
            if (acc * G >= chunks_total * g) gb[g++] = l + 1;
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
        for (size_t l = l0; l < l1; ++l) {
            fvad_lane& L = lanes[l];
            if (L.n_fft_frames) {
                d2h.push_back({L.band_sum, ws.band + band_off[l], L.n_fft_frames * sizeof(float)});
                if (L.fft_bins) d2h.push_back({L.fft_bins, ws.bins + band_off[l] * NB, L.n_fft_frames * NB * sizeof(float)});
            }
            if (L.n_chunks) {
                d2h.push_back({L.chunk_rms, d_rms + rms_off[l], L.n_chunks * sizeof(float)});
                if (L.denoised && !opts.on_device) d2h.push_back({L.denoised, jobs[l].d_den, L.n_chunks * kChunk48 * sizeof(float)});
                if (L.denoised_i16 && !opts.on_device) d2h.push_back({L.denoised_i16, jobs[l].d_den16, L.n_chunks * kChunk48 * sizeof(int16_t)});
            }
        }
        return staged_copy(ctx, d2h, false, s_out);
    };

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
                if ((rc_out = outputs_of(gb[g], gb[g + 1]))) return;
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
            fvad_launch_vadfft_jobs(ws.fft_jobs + l0, (int)(l1 - l0), mf, plan, opts.min_bin, opts.max_bin, st);
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
        if ((long)(n_lanes * n_chunks) > planned_max_chunks(ctx, (long)(n_lanes * n_chunks), opts.max_chunks_per_launch))
            ws.carries_clean = 0; // several launches: the even carries get written
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
        int r = run_chunks(ctx, jobs, opts.max_chunks_per_launch, capture_descs, capture_dev);
        if (r) return r;
        // one K4 launch for every lane's frames
        for (size_t l = 0; l < n_lanes; ++l) h_jobs[l] = {den + l * n_den, d_band_sum + l * n_frames, nullptr, (long)n_frames};
        if (!capture_descs && (ws.jobs_mirror.size() != n_lanes || memcmp(ws.jobs_mirror.data(), h_jobs, n_lanes * sizeof(VadFftJob)) != 0)) {
            FVAD_HIP(ctx, hipMemcpyAsync(d_jobs, h_jobs, n_lanes * sizeof(VadFftJob), hipMemcpyHostToDevice, st));
            ws.jobs_mirror.assign(h_jobs, h_jobs + n_lanes);
        }
        time_begin(ctx, "fft1024_bandsum");
        fvad_launch_vadfft_jobs(d_jobs, (int)n_lanes, (long)n_frames, plan, opts.min_bin, opts.max_bin, st);
        time_end(ctx);
        return FVAD_OK;
    };

    // Opt-in (fvad_engine_opts.use_graph): capture the sequence into a hipGraph once and replay it while the arguments
    // and the workspace stay the same -- BASELINE config 5's "hipGraph-captured steady-state loop".  A step
    // is ~12 launches per 100 ms of GPU work, so this saves well under 1 % (measured in bench.py's extras).
    if (opts.use_graph && !ctx->timing) {
        const long total = (long)(n_lanes * n_chunks);
        const long maxc = planned_max_chunks(ctx, total, opts.max_chunks_per_launch);
        if ((rc = ensure_workspace(ctx, std::min(total, maxc), kRowsPerChunk, kWarmupRows, total % std::min(total, maxc)))) return rc; // no allocation while capturing
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
