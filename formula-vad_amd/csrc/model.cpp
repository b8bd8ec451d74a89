// model.cpp -- NSNet2 weights on the device: every layer packed for the kernel families that read it (fragment-major f32 for
// the MFMA kernels, f16x3 / bf16x3 pieces for the emulations, the run-time-sized layouts of other model dimensions).
// Part of libfvad_hip.so; the entry point is fvad::upload_model (internal.h).
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

} // namespace fvad
