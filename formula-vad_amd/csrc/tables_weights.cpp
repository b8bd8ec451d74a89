// tables_weights.cpp -- host-side constants and NSNet2 weight handling (no device code).
//
// Windows restate src/audio_utils/window_fn.zig in f32 exactly (the reference builds them at init
// on the CPU too); twiddles follow kissfft's recipe: evaluate in double, round to float once.
// The ONNX reader replaces onnx.OnnxInstance.init (src/NSNet2.zig:53-61): a minimal protobuf
// walk that pulls the initializers of the NSNet2-baseline graph out of the file the reference
// loads (data/nsnet2-20ms-baseline.onnx).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

#include "internal.h"

namespace fvad {

static const float kTwoPiF = (float)(2.0 * 3.14159265358979323846264338327950288);

void hann_window_symmetric(float* result, size_t n)
{
    const float a0 = 0.5f, a1 = 0.5f;
    const float N = (float)n;
    const float step = kTwoPiF / (N - 1);
    for (size_t i = 0; i < n; ++i) result[i] = a0 - a1 * cosf((float)i * step);
}

void hann_window_periodic(float* result, size_t n)
{
    const float alphas[2] = {0.5f, 1.0f - 0.5f};
    const float N = (float)n;
    for (size_t i = 0; i < n; ++i) {
        const float x = (float)i;
        float acc = 0;
        for (int k = 0; k < 2; ++k) {
            const float sign = (k & 1) ? -1.0f : 1.0f;
            acc += sign * alphas[k] * cosf((kTwoPiF * (float)k * x) / N);
        }
        result[i] = acc;
    }
}

float window_norm_factor(const float* w, size_t n)
{
    float sum = 0;
    for (size_t i = 0; i < n; ++i) sum += w[i];
    return (float)n / sum;
}

void nsnet2_window(float* w320)
{
    hann_window_symmetric(w320, 320);
    for (int i = 0; i < 320; ++i) w320[i] = sqrtf(w320[i]);
}

static const double kPi = 3.14159265358979323846264338327;

void make_twiddles(int n, std::vector<float>& out)
{
    out.resize(2 * (size_t)n);
    for (int j = 0; j < n; ++j) {
        const double phase = -2.0 * kPi * (double)j / (double)n;
        out[2 * j] = (float)cos(phase);
        out[2 * j + 1] = (float)sin(phase);
    }
}

void make_super_twiddles(int ncfft, std::vector<float>& out)
{
    out.resize(2 * (size_t)(ncfft / 2));
    for (int i = 0; i < ncfft / 2; ++i) {
        const double phase = -kPi * ((double)(i + 1) / (double)ncfft + 0.5);
        out[2 * i] = (float)cos(phase);
        out[2 * i + 1] = (float)sin(phase);
    }
}

// ------------------------------------------------------------------ HostWeights

void HostWeights::view(fvad_nsnet2_weights* o) const
{
    o->n_bins = n_bins; o->n_fc1 = n_fc1; o->n_hidden = n_hidden; o->n_fc2 = n_fc2; o->n_fc3 = n_fc3;
    o->fc1_w = fc1_w.data(); o->fc1_b = fc1_b.data();
    o->gru1_w = gru1_w.data(); o->gru1_r = gru1_r.data(); o->gru1_b = gru1_b.data();
    o->gru2_w = gru2_w.data(); o->gru2_r = gru2_r.data(); o->gru2_b = gru2_b.data();
    o->fc2_w = fc2_w.data(); o->fc2_b = fc2_b.data();
    o->fc3_w = fc3_w.data(); o->fc3_b = fc3_b.data();
    o->fc4_w = fc4_w.data(); o->fc4_b = fc4_b.data();
}

bool HostWeights::check_dims(std::string& err) const
{
    // The spectral front end fixes the input and output width (n_fft = 320 -> 161 bins, NSNet2.zig:12,379-381); the
    // hidden widths come from the file.  NSNet2-baseline's 400/400/600/600 run on the specialised kernels, anything
    // else on the run-time-sized ones (engine.cpp run_nn_generic), within their limits.
    if (n_bins != 161 || n_fc1 < 1 || n_hidden < 1 || n_fc2 < 1 || n_fc3 < 1 || n_hidden > 1024 || n_fc1 > 8192 ||
        n_fc2 > 8192 || n_fc3 > 8192) {
        char b[200];
        snprintf(b, sizeof b, "unsupported NSNet2 dims bins=%d fc1=%d hidden=%d fc2=%d fc3=%d "
                 "(bins must be 161; hidden <= 1024; fc widths <= 8192)", n_bins, n_fc1, n_hidden, n_fc2, n_fc3);
        err = b;
        return false;
    }
    return true;
}

bool HostWeights::from_view(const fvad_nsnet2_weights* in, std::string& err)
{
    if (!in) { err = "null weights"; return false; }
    n_bins = in->n_bins; n_fc1 = in->n_fc1; n_hidden = in->n_hidden; n_fc2 = in->n_fc2; n_fc3 = in->n_fc3;
    if (n_bins <= 0 || n_fc1 <= 0 || n_hidden <= 0 || n_fc2 <= 0 || n_fc3 <= 0) { err = "bad dims"; return false; }
    const float* ptrs[] = {in->fc1_w, in->fc1_b, in->gru1_w, in->gru1_r, in->gru1_b, in->gru2_w, in->gru2_r,
                           in->gru2_b, in->fc2_w, in->fc2_b, in->fc3_w, in->fc3_b, in->fc4_w, in->fc4_b};
    for (const float* p : ptrs) if (!p) { err = "null weight tensor"; return false; }
    const size_t H = (size_t)n_hidden;
    auto cp = [](std::vector<float>& d, const float* s, size_t n) { d.assign(s, s + n); };
    cp(fc1_w, in->fc1_w, (size_t)n_fc1 * n_bins); cp(fc1_b, in->fc1_b, (size_t)n_fc1);
    cp(gru1_w, in->gru1_w, 3 * H * (size_t)n_fc1); cp(gru1_r, in->gru1_r, 3 * H * H); cp(gru1_b, in->gru1_b, 6 * H);
    cp(gru2_w, in->gru2_w, 3 * H * H); cp(gru2_r, in->gru2_r, 3 * H * H); cp(gru2_b, in->gru2_b, 6 * H);
    cp(fc2_w, in->fc2_w, (size_t)n_fc2 * H); cp(fc2_b, in->fc2_b, (size_t)n_fc2);
    cp(fc3_w, in->fc3_w, (size_t)n_fc3 * n_fc2); cp(fc3_b, in->fc3_b, (size_t)n_fc3);
    cp(fc4_w, in->fc4_w, (size_t)n_bins * n_fc3); cp(fc4_b, in->fc4_b, (size_t)n_bins);
    return true;
}

// splitmix64 -> xoshiro256**, uniform in [-a, a)
struct Rng {
    uint64_t s[4];
    explicit Rng(uint64_t seed)
    {
        uint64_t x = seed;
        for (int i = 0; i < 4; ++i) {
            x += 0x9E3779B97F4A7C15ull;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t r = rotl(s[1] * 5, 7) * 9;
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    float uniform(float a) // [-a, a)
    {
        const double u = (double)(next() >> 11) * (1.0 / 9007199254740992.0);
        return (float)((2.0 * u - 1.0) * (double)a);
    }
};

static void fill(std::vector<float>& v, size_t n, Rng& r, float a)
{
    v.resize(n);
    for (size_t i = 0; i < n; ++i) v[i] = r.uniform(a);
}

// Random-init weights of the NSNet2-baseline architecture.  Scales keep every layer in its
// responsive range (variance-preserving uniform init; the sigmoid head gets a wider spread so
// the gains are not all ~0.5), which is what makes the parity tests discriminating.
void synth_weights(uint64_t seed, HostWeights& w)
{
    w.n_bins = 161; w.n_fc1 = 400; w.n_hidden = 400; w.n_fc2 = 600; w.n_fc3 = 600;
    Rng r(seed);
    const size_t H = 400;
    auto bound = [](double gain, double fan_in) { return (float)(gain * std::sqrt(3.0 / fan_in)); };
    fill(w.fc1_w, 400 * 161, r, bound(0.35, 161)); fill(w.fc1_b, 400, r, 0.1f);
    fill(w.gru1_w, 3 * H * 400, r, bound(1.0, 400)); fill(w.gru1_r, 3 * H * H, r, bound(1.0, 400));
    fill(w.gru1_b, 6 * H, r, 0.1f);
    fill(w.gru2_w, 3 * H * H, r, bound(1.5, 400)); fill(w.gru2_r, 3 * H * H, r, bound(1.0, 400));
    fill(w.gru2_b, 6 * H, r, 0.1f);
    fill(w.fc2_w, 600 * H, r, bound(2.0, 400)); fill(w.fc2_b, 600, r, 0.1f);
    fill(w.fc3_w, 600 * 600, r, bound(1.4, 600)); fill(w.fc3_b, 600, r, 0.1f);
    fill(w.fc4_w, 161 * 600, r, bound(2.5, 600)); fill(w.fc4_b, 161, r, 0.5f);
}

// ------------------------------------------------------------------ fragment packing

void pack_panel(const float* W, int N, int K, int n_blocks, int NT, int S, std::vector<float>& out)
{
    out.assign((size_t)n_blocks * S * NT * 256, 0.0f);
    for (int b = 0; b < n_blocks; ++b)
        for (int s = 0; s < S; ++s)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int n = (b * NT + t) * 16 + (lane & 15);
                        const int k = 16 * s + 4 * (lane >> 4) + r;
                        if (n < N && k < K)
                            out[(((size_t)b * S + s) * NT + t) * 256 + lane * 4 + r] = W[(size_t)n * K + k];
                    }
}

// v2 layout: one contiguous 75 KB slab per unit tile: [J][g][S][64][4]
void pack_gru_r2(const float* R, int H, std::vector<float>& out)
{
    const int J = H / 16;
    out.assign((size_t)J * 3 * J * 256, 0.0f);
    for (int j = 0; j < J; ++j)
        for (int g = 0; g < 3; ++g)
            for (int s = 0; s < J; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int n = g * H + 16 * j + (lane & 15);
                        const int k = 16 * s + 4 * (lane >> 4) + r;
                        out[(((size_t)j * 3 + g) * J + s) * 256 + lane * 4 + r] = R[(size_t)n * H + k];
                    }
}

// the same fragment order for a GRU input matrix W [3 H][K]: [H/16 unit tiles][3 gates][ceil(K/16) super-steps][64 lanes][4],
// K zero-padded (kernels_ws.hip: layer 1's stationary input-projection fragments)
void pack_gru_frag(const float* W, int H, int K, std::vector<float>& out)
{
    const int J = H / 16, S = (K + 15) / 16;
    out.assign((size_t)J * 3 * S * 256, 0.0f);
    for (int j = 0; j < J; ++j)
        for (int g = 0; g < 3; ++g)
            for (int s = 0; s < S; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int n = g * H + 16 * j + (lane & 15);
                        const int k = 16 * s + 4 * (lane >> 4) + r;
                        if (k < K) out[(((size_t)j * 3 + g) * S + s) * 256 + lane * 4 + r] = W[(size_t)n * K + k];
                    }
}

// ---- f16x3 layouts (kernels_h3.hip): every weight as two f16 pieces of W * sw
// f32 -> f16 bits, round to nearest even (weights are finite; |v| < 65520 by the choice of sw)
static uint16_t f32_to_f16_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);           // >= 65536 (or NaN): infinity, never reached
    if (u < 0x38800000u) {                                             // below 2^-14: subnormal half
        if (u < 0x33000000u) return (uint16_t)sign;                    // below 2^-25: zero
        const int e = (int)(u >> 23);                                  // biased f32 exponent, 102..112
        const uint32_t mant = (u & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;                                     // 14..24: mant >> shift is the half mantissa
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((u - 0x38000000u) >> 13);
    const uint32_t rem = u & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;            // may carry into the exponent: still right
    return (uint16_t)(sign | h);
}
static float f16_bits_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3FFu;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400u), e - 25);
    uint32_t u;
    memcpy(&u, &v, 4);
    u |= sign;
    memcpy(&v, &u, 4);
    return v;
}

float h3_weight_scale(const float* W, size_t n)
{
    float mx = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(W[i])) return 0.0f; // not representable: the caller keeps the f32 kernels
        mx = std::max(mx, fabsf(W[i]));
    }
    if (!(mx > 0.0f)) return 1.0f;
    int e;
    frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 15 - e);    // mx * sw in [2^14, 2^15)
}

float h3_activation_scale(double bound)
{
    if (!(bound > 0.0) || !std::isfinite(bound)) return 1.0f;
    int e;
    const double f = frexp(bound, &e); // bound = f * 2^e
    if (f == 0.5) --e;                 // exact power of two
    return ldexpf(1.0f, 14 - e);       // |x| <= bound  ->  |x * sx| <= 2^14
}

// the two f16 fragment dwords of one value pair are written as bit patterns into float storage
static void put_half(std::vector<float>& out, size_t dword, int upper, uint16_t bits)
{
    uint32_t u;
    memcpy(&u, &out[dword], 4);
    u = upper ? ((u & 0x0000FFFFu) | ((uint32_t)bits << 16)) : ((u & 0xFFFF0000u) | bits);
    memcpy(&out[dword], &u, 4);
}
static inline int h3_slot_k(int s, int qa, int j) { return j < 4 ? 32 * s + 4 * qa + j : 32 * s + 16 + 4 * qa + (j - 4); }
static void h3_put(std::vector<float>& out, size_t block_dword, int lane, int j, float w_scaled)
{
    const uint16_t hi = f32_to_f16_bits(w_scaled);
    const uint16_t lo = f32_to_f16_bits(w_scaled - f16_bits_to_f32(hi));
    const size_t d = block_dword + (size_t)lane * 4 + (size_t)(j >> 1);
    put_half(out, d, j & 1, hi);
    put_half(out, d + 256, j & 1, lo);
}

// [n_blocks][S32][NT][hi, lo][64 lanes][8 halves], S32 = ceil(ceil(K / 16) / 2) K-steps of 32 slots
void pack_panel_h3(const float* W, int N, int K, int n_blocks, int NT, float sw, std::vector<float>& out)
{
    const int S = ((K + 15) / 16 + 1) / 2;
    out.assign((size_t)n_blocks * S * NT * 512, 0.0f);
    for (int b = 0; b < n_blocks; ++b)
        for (int s = 0; s < S; ++s)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = (b * NT + t) * 16 + (lane & 15);
                        const int k = h3_slot_k(s, lane >> 4, j);
                        if (n < N && k < K)
                            h3_put(out, (((size_t)b * S + s) * NT + t) * 512, lane, j, W[(size_t)n * K + k] * sw);
                    }
}

// one contiguous slab per unit tile: [J][S32 = 13][3 gates][hi, lo][64][8 halves]  (78 KB)
void pack_gru_r_h3(const float* R, int H, float sw, std::vector<float>& out)
{
    const int J = H / 16, S = (J + 1) / 2;
    out.assign((size_t)J * S * 3 * 512, 0.0f);
    for (int jt = 0; jt < J; ++jt)
        for (int s = 0; s < S; ++s)
            for (int g = 0; g < 3; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = g * H + 16 * jt + (lane & 15);
                        const int k = h3_slot_k(s, lane >> 4, j);
                        if (k < H) h3_put(out, (((size_t)jt * S + s) * 3 + g) * 512, lane, j, R[(size_t)n * H + k] * sw);
                    }
}

// ---- bf16x3 layout (kernels_b3.hip): every weight as three bf16 pieces, W = h + m + l exactly
// f32 -> bf16 bits, round to nearest even (what v_cvt_pk_bf16_f32 does on the device; NaN stays NaN)
static uint16_t f32_to_bf16_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u); // quiet NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_bits_to_f32(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}

// [n_blocks][S32][NT][h, m, l][64 lanes][8 bf16], S32 = ceil(ceil(K / 16) / 2) K-steps of 32 slots, kernels_h3's slot order
void pack_panel_b3(const float* W, int N, int K, int n_blocks, int NT, std::vector<float>& out)
{
    const int S = ((K + 15) / 16 + 1) / 2;
    out.assign((size_t)n_blocks * S * NT * 768, 0.0f);
    for (int b = 0; b < n_blocks; ++b)
        for (int s = 0; s < S; ++s)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = (b * NT + t) * 16 + (lane & 15);
                        const int k = h3_slot_k(s, lane >> 4, j);
                        if (n >= N || k >= K) continue;
                        const float w = W[(size_t)n * K + k];
                        const uint16_t h = f32_to_bf16_bits(w);
                        const float r1 = w - bf16_bits_to_f32(h);
                        const uint16_t m = f32_to_bf16_bits(r1);
                        const uint16_t l = f32_to_bf16_bits(r1 - bf16_bits_to_f32(m));
                        const size_t d = (((size_t)b * S + s) * NT + t) * 768 + (size_t)lane * 4 + (size_t)(j >> 1);
                        put_half(out, d, j & 1, h);
                        put_half(out, d + 256, j & 1, m);
                        put_half(out, d + 512, j & 1, l);
                    }
}

// ------------------------------------------------------------------ ONNX (protobuf) reader
// Wire format only: varint, 64-bit, length-delimited, 32-bit.  Message/field numbers from
// onnx.proto3: ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5; NodeProto.input = 1,
// .output = 2, .op_type = 4, .attribute = 5; TensorProto.dims = 1, .data_type = 2,
// .float_data = 4, .name = 8, .raw_data = 9; AttributeProto.name = 1, .i = 3.
namespace {
struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    uint64_t varint()
    {
        uint64_t v = 0;
        int shift = 0;
        while (p < end) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7F) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
            if (shift > 63) break;
        }
        ok = false;
        return 0;
    }
    bool next(uint32_t& field, uint32_t& wire, Cursor& sub, uint64_t& val)
    {
        if (p >= end || !ok) return false;
        const uint64_t key = varint();
        if (!ok) return false;
        field = (uint32_t)(key >> 3);
        wire = (uint32_t)(key & 7);
        switch (wire) {
        case 0: val = varint(); return ok;
        case 1: if (end - p < 8) { ok = false; return false; } memcpy(&val, p, 8); p += 8; return true;
        case 2: {
            const uint64_t n = varint();
            if (!ok || (uint64_t)(end - p) < n) { ok = false; return false; }
            sub.p = p; sub.end = p + n; sub.ok = true;
            p += n;
            return true;
        }
        case 5: { if (end - p < 4) { ok = false; return false; } uint32_t v32; memcpy(&v32, p, 4); val = v32; p += 4; return true; }
        default: ok = false; return false;
        }
    }
};

struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
};
struct Node {
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, int64_t> iattr;
    bool has_tensor = false; // AttributeProto.t (a Constant node's value)
    Tensor tensor;
};

bool parse_tensor(Cursor c, std::string& name, Tensor& t)
{
    uint32_t f, w; Cursor sub{nullptr, nullptr}; uint64_t v;
    int data_type = 0;
    std::vector<float> fdata;
    const uint8_t* raw = nullptr; size_t raw_n = 0;
    while (c.next(f, w, sub, v)) {
        if (f == 1) {
            if (w == 0) t.dims.push_back((int64_t)v);
            else if (w == 2) { Cursor d = sub; while (d.p < d.end && d.ok) t.dims.push_back((int64_t)d.varint()); }
        } else if (f == 2 && w == 0) data_type = (int)v;
        else if (f == 4) {
            if (w == 5) { float x; uint32_t u = (uint32_t)v; memcpy(&x, &u, 4); fdata.push_back(x); }
            else if (w == 2) { const size_t n = (size_t)(sub.end - sub.p) / 4; const size_t o = fdata.size(); fdata.resize(o + n); memcpy(fdata.data() + o, sub.p, n * 4); }
        } else if (f == 8 && w == 2) name.assign((const char*)sub.p, (size_t)(sub.end - sub.p));
        else if (f == 9 && w == 2) { raw = sub.p; raw_n = (size_t)(sub.end - sub.p); }
    }
    if (!c.ok) return false;
    if (data_type != 1) return true; // not FLOAT: keep dims only
    if (raw) { t.data.resize(raw_n / 4); memcpy(t.data.data(), raw, (raw_n / 4) * 4); }
    else t.data = std::move(fdata);
    return true;
}

bool parse_node(Cursor c, Node& n)
{
    uint32_t f, w; Cursor sub{nullptr, nullptr}; uint64_t v;
    while (c.next(f, w, sub, v)) {
        if (w != 2) continue;
        if (f == 1) n.in.emplace_back((const char*)sub.p, (size_t)(sub.end - sub.p));
        else if (f == 2) n.out.emplace_back((const char*)sub.p, (size_t)(sub.end - sub.p));
        else if (f == 4) n.op.assign((const char*)sub.p, (size_t)(sub.end - sub.p));
        else if (f == 5) {
            Cursor a = sub; uint32_t af, aw; Cursor as{nullptr, nullptr}; uint64_t av;
            std::string an; int64_t ai = 0; bool has_i = false;
            while (a.next(af, aw, as, av)) {
                if (af == 1 && aw == 2) an.assign((const char*)as.p, (size_t)(as.end - as.p));
                else if (af == 3 && aw == 0) { ai = (int64_t)av; has_i = true; }
                else if (af == 5 && aw == 2) { // AttributeProto.t
                    std::string tn; Tensor t;
                    if (!parse_tensor(as, tn, t)) return false;
                    n.tensor = std::move(t); n.has_tensor = true;
                }
            }
            if (has_i) n.iattr[an] = ai;
        }
    }
    return c.ok;
}
} // namespace

int read_onnx_nsnet2(const char* path, HostWeights& hw, std::string& err)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("cannot open ") + path; return FVAD_ERR_IO; }
    std::vector<uint8_t> buf;
    {
        fseek(fp, 0, SEEK_END);
        const long n = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        if (n <= 0) { fclose(fp); err = "empty model file"; return FVAD_ERR_MODEL_FORMAT; }
        buf.resize((size_t)n);
        if (fread(buf.data(), 1, (size_t)n, fp) != (size_t)n) { fclose(fp); err = "short read"; return FVAD_ERR_IO; }
        fclose(fp);
    }
    Cursor model{buf.data(), buf.data() + buf.size()};
    uint32_t f, w; Cursor sub{nullptr, nullptr}; uint64_t v;
    Cursor graph{nullptr, nullptr};
    bool have_graph = false;
    while (model.next(f, w, sub, v)) if (f == 7 && w == 2) { graph = sub; have_graph = true; }
    if (!model.ok || !have_graph) { err = "not an ONNX ModelProto (no graph)"; return FVAD_ERR_MODEL_FORMAT; }

    std::map<std::string, Tensor> inits;
    std::vector<Node> nodes;
    std::vector<std::string> graph_inputs; // GraphProto.input = 11 -> ValueInfoProto.name = 1
    while (graph.next(f, w, sub, v)) {
        if (w != 2) continue;
        if (f == 11) {
            Cursor vi = sub; uint32_t vf, vw; Cursor vs{nullptr, nullptr}; uint64_t vv;
            while (vi.next(vf, vw, vs, vv))
                if (vf == 1 && vw == 2) graph_inputs.emplace_back((const char*)vs.p, (size_t)(vs.end - vs.p));
        } else if (f == 5) { std::string name; Tensor t; if (!parse_tensor(sub, name, t)) { err = "bad TensorProto"; return FVAD_ERR_MODEL_FORMAT; } inits[name] = std::move(t); }
        else if (f == 1) { Node n; if (!parse_node(sub, n)) { err = "bad NodeProto"; return FVAD_ERR_MODEL_FORMAT; } nodes.push_back(std::move(n)); }
    }
    if (!graph.ok) { err = "truncated GraphProto"; return FVAD_ERR_MODEL_FORMAT; }
    // some exporters emit weights and biases as Constant nodes instead of initializers
    for (Node& n : nodes)
        if (n.op == "Constant" && n.has_tensor && !n.out.empty() && !inits.count(n.out[0])) inits[n.out[0]] = std::move(n.tensor);

    // walk the nodes in file (topological) order: 4 dense layers (MatMul or Gemm + Add) and 2 GRUs
    struct Dense { std::vector<float> w; std::vector<float> b; int in = 0, out = 0; bool have_b = false; };
    std::vector<Dense> dense;
    struct Gru { const Tensor *W, *R, *B; };
    std::vector<Gru> grus;
    std::map<std::string, size_t> produced_by_dense; // output name -> dense index awaiting its Add
    for (const Node& n : nodes) {
        auto init_of = [&](const std::string& s) -> const Tensor* { auto it = inits.find(s); return it == inits.end() ? nullptr : &it->second; };
        if (n.op == "MatMul" && n.in.size() == 2) {
            const Tensor* t = init_of(n.in[1]);
            if (!t || t->dims.size() != 2 || t->data.empty()) continue;
            Dense d; d.in = (int)t->dims[0]; d.out = (int)t->dims[1]; // B operand is [in][out]
            d.w.resize((size_t)d.in * d.out);
            for (int i = 0; i < d.in; ++i) for (int o = 0; o < d.out; ++o) d.w[(size_t)o * d.in + i] = t->data[(size_t)i * d.out + o];
            d.b.assign((size_t)d.out, 0.0f);
            dense.push_back(std::move(d));
            if (!n.out.empty()) produced_by_dense[n.out[0]] = dense.size() - 1;
        } else if (n.op == "Gemm" && n.in.size() >= 2) {
            const Tensor* t = init_of(n.in[1]);
            if (!t || t->dims.size() != 2 || t->data.empty()) continue;
            const bool transB = n.iattr.count("transB") && n.iattr.at("transB") != 0;
            Dense d;
            if (transB) { d.out = (int)t->dims[0]; d.in = (int)t->dims[1]; d.w = t->data; }
            else {
                d.in = (int)t->dims[0]; d.out = (int)t->dims[1]; d.w.resize((size_t)d.in * d.out);
                for (int i = 0; i < d.in; ++i) for (int o = 0; o < d.out; ++o) d.w[(size_t)o * d.in + i] = t->data[(size_t)i * d.out + o];
            }
            d.b.assign((size_t)d.out, 0.0f);
            if (n.in.size() >= 3) if (const Tensor* bt = init_of(n.in[2])) if (bt->data.size() == (size_t)d.out) { d.b = bt->data; d.have_b = true; }
            dense.push_back(std::move(d));
        } else if (n.op == "Add" && n.in.size() == 2) {
            for (int side = 0; side < 2; ++side) {
                auto it = produced_by_dense.find(n.in[side]);
                const Tensor* bt = init_of(n.in[1 - side]);
                if (it != produced_by_dense.end() && bt && bt->data.size() == (size_t)dense[it->second].out) {
                    dense[it->second].b = bt->data; dense[it->second].have_b = true;
                }
            }
        } else if (n.op == "GRU" && n.in.size() >= 3) {
            const Tensor *W = init_of(n.in[1]), *R = init_of(n.in[2]);
            const Tensor* B = n.in.size() >= 4 ? init_of(n.in[3]) : nullptr;
            if (!W || !R) { err = "GRU weights are not initializers"; return FVAD_ERR_MODEL_FORMAT; }
            if (!(n.iattr.count("linear_before_reset") && n.iattr.at("linear_before_reset") == 1)) {
                err = "GRU without linear_before_reset=1 is not the NSNet2-baseline export"; return FVAD_ERR_MODEL_FORMAT;
            }
            // initial_h: the reference feeds the session one tensor and no state (NSNet2.zig:57-58,71-112), so an
            // initial state can only be a constant of the graph.  PyTorch exports of nn.GRU without h0 wire a
            // zeros tensor built from shape ops into this input: accepted.  An all-zero initializer: accepted.
            // A non-zero initializer or a real graph input would be a stateful model: refused.
            if (n.in.size() >= 6 && !n.in[5].empty()) {
                const std::string& h0 = n.in[5];
                if (const Tensor* t0 = init_of(h0)) {
                    for (float x : t0->data) if (x != 0.0f) { err = "GRU with a non-zero initial_h is not supported"; return FVAD_ERR_MODEL_FORMAT; }
                } else {
                    for (const std::string& gi_name : graph_inputs)
                        if (gi_name == h0) { err = "GRU state is a graph input: stateful models are not supported"; return FVAD_ERR_MODEL_FORMAT; }
                }
            }
            grus.push_back({W, R, B});
        }
    }
    if (dense.size() != 4 || grus.size() != 2) {
        char b[128]; snprintf(b, sizeof b, "expected 4 dense + 2 GRU layers, found %zu + %zu", dense.size(), grus.size());
        err = b; return FVAD_ERR_MODEL_FORMAT;
    }
    auto gru_ok = [&](const Gru& g, int in, int H) {
        return g.W->dims.size() == 3 && g.W->dims[0] == 1 && g.W->dims[1] == 3 * H && g.W->dims[2] == in &&
               g.R->dims.size() == 3 && g.R->dims[1] == 3 * H && g.R->dims[2] == H &&
               (!g.B || g.B->data.size() == (size_t)6 * H);
    };
    hw.n_bins = dense[0].in; hw.n_fc1 = dense[0].out;
    hw.n_hidden = (int)(grus[0].R->dims.size() == 3 ? grus[0].R->dims[2] : 0);
    hw.n_fc2 = dense[1].out; hw.n_fc3 = dense[2].out;
    const int H = hw.n_hidden;
    if (H <= 0 || !gru_ok(grus[0], hw.n_fc1, H) || !gru_ok(grus[1], H, H) || dense[1].in != H ||
        dense[2].in != hw.n_fc2 || dense[3].in != hw.n_fc3 || dense[3].out != hw.n_bins) {
        err = "layer shapes do not chain like NSNet2"; return FVAD_ERR_MODEL_FORMAT;
    }
    hw.fc1_w = dense[0].w; hw.fc1_b = dense[0].b;
    hw.gru1_w = grus[0].W->data; hw.gru1_r = grus[0].R->data;
    hw.gru1_b = grus[0].B ? grus[0].B->data : std::vector<float>((size_t)6 * H, 0.0f);
    hw.gru2_w = grus[1].W->data; hw.gru2_r = grus[1].R->data;
    hw.gru2_b = grus[1].B ? grus[1].B->data : std::vector<float>((size_t)6 * H, 0.0f);
    hw.fc2_w = dense[1].w; hw.fc2_b = dense[1].b;
    hw.fc3_w = dense[2].w; hw.fc3_b = dense[2].b;
    hw.fc4_w = dense[3].w; hw.fc4_b = dense[3].b;
    return FVAD_OK;
}

} // namespace fvad

// ------------------------------------------------------------------ C ABI (host-only pieces)
extern "C" {

void fvad_hann_window_periodic(float* r, size_t n) { fvad::hann_window_periodic(r, n); }
void fvad_hann_window_symmetric(float* r, size_t n) { fvad::hann_window_symmetric(r, n); }
float fvad_window_norm_factor(const float* w, size_t n) { return fvad::window_norm_factor(w, n); }
void fvad_nsnet2_window(float* w) { fvad::nsnet2_window(w); }

int fvad_onnx_read_nsnet2(const char* path, fvad_nsnet2_weights* out, void** owner)
{
    if (!path || !out || !owner) return FVAD_ERR_INVALID_ARGUMENT;
    auto* hw = new fvad::HostWeights();
    std::string err;
    const int rc = fvad::read_onnx_nsnet2(path, *hw, err);
    if (rc != FVAD_OK) { delete hw; return rc; }
    hw->view(out);
    *owner = hw;
    return FVAD_OK;
}

int fvad_synth_nsnet2(uint64_t seed, fvad_nsnet2_weights* out, void** owner)
{
    if (!out || !owner) return FVAD_ERR_INVALID_ARGUMENT;
    auto* hw = new fvad::HostWeights();
    fvad::synth_weights(seed, *hw);
    hw->view(out);
    *owner = hw;
    return FVAD_OK;
}

void fvad_weights_free(void* owner) { delete static_cast<fvad::HostWeights*>(owner); }

} // extern "C"
