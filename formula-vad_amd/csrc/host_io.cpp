// host_io.cpp -- minimal RIFF/WAVE reader.
// The reference reads audio through libsndfile (src/audio_utils/AudioFileStream.zig:18-102,
// AudioBuffer.zig:26-59: sf_readf_float, then de-interleave into planar f32).  libsndfile is not
// part of this build; this reader covers what the simulator harness needs: PCM16 and IEEE float32
// WAV, any channel count, de-interleaved to channel-planar f32.  PCM16 is scaled by 1/32768 like
// libsndfile's normalised float read.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fvad.h"

namespace {
// unaligned little-endian loads as plain typed accesses (a chunk's body may start anywhere in the file); the loops below vectorise
typedef int16_t i16u __attribute__((aligned(1), may_alias));
typedef float f32u __attribute__((aligned(1), may_alias));
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
} // namespace

namespace {
struct WavInfo {
    // (malloc, not a vector: a vector would zero-fill the 100+ MB it is about to read over)
    struct Bytes { uint8_t* p = nullptr; size_t n = 0; ~Bytes() { free(p); } uint8_t* data() { return p; } size_t size() const { return n; }
                   bool resize(size_t k) { free(p); p = (uint8_t*)malloc(k ? k : 1); n = p ? k : 0; return p != nullptr; } } buf;
    int fmt_tag = 0, channels = 0, bits = 0; uint32_t rate = 0; const uint8_t* data = nullptr; size_t data_bytes = 0; };

int wav_parse(const char* path, WavInfo& w)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return FVAD_ERR_IO;
    WavInfo::Bytes& buf = w.buf;
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 12) { fclose(fp); return FVAD_ERR_MODEL_FORMAT; }
    if (!buf.resize((size_t)sz)) { fclose(fp); return FVAD_ERR_ALLOC_FAILED; }
    const size_t got = fread(buf.data(), 1, (size_t)sz, fp);
    fclose(fp);
    if (got != (size_t)sz) return FVAD_ERR_IO;
    if (memcmp(buf.data(), "RIFF", 4) != 0 || memcmp(buf.data() + 8, "WAVE", 4) != 0) return FVAD_ERR_MODEL_FORMAT;
    size_t pos = 12;
    while (pos + 8 <= buf.size()) {
        const uint8_t* ck = buf.data() + pos;
        const uint32_t len = rd32(ck + 4);
        const size_t body = pos + 8;
        if (memcmp(ck, "fmt ", 4) == 0 && len >= 16 && body + 16 <= buf.size()) {
            w.fmt_tag = rd16(buf.data() + body);
            w.channels = rd16(buf.data() + body + 2);
            w.rate = rd32(buf.data() + body + 4);
            w.bits = rd16(buf.data() + body + 14);
            if (w.fmt_tag == 0xFFFE && len >= 26 && body + 26 <= buf.size()) w.fmt_tag = rd16(buf.data() + body + 24); // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
        } else if (memcmp(ck, "data", 4) == 0) {
            w.data = buf.data() + body;
            w.data_bytes = (body + len <= buf.size()) ? len : buf.size() - body; // tolerate a truncated / streaming length
            break;
        }
        pos = body + len + (len & 1);
    }
    if (!w.data || w.channels <= 0 || w.rate == 0) return FVAD_ERR_MODEL_FORMAT;
    return FVAD_OK;
}
} // namespace

extern "C" {

int fvad_wav_read(const char* path, float*** channel_pcm, size_t* n_channels, size_t* n_frames, size_t* sample_rate)
{
    if (!path || !channel_pcm || !n_channels || !n_frames || !sample_rate) return FVAD_ERR_INVALID_ARGUMENT;
    WavInfo w;
    const int prc = wav_parse(path, w);
    if (prc) return prc;
    const int fmt_tag = w.fmt_tag, channels = w.channels, bits = w.bits;
    const uint32_t rate = w.rate;
    const uint8_t* data = w.data;
    const size_t data_bytes = w.data_bytes;
    const bool pcm16 = (fmt_tag == 1 && bits == 16);
    const bool f32 = (fmt_tag == 3 && bits == 32);
    if (!pcm16 && !f32) return FVAD_ERR_MODEL_FORMAT;
    const size_t frame_bytes = (size_t)channels * (bits / 8);
    const size_t frames = data_bytes / frame_bytes;

    float** out = (float**)calloc((size_t)channels, sizeof(float*));
    if (!out) return FVAD_ERR_ALLOC_FAILED;
    for (int c = 0; c < channels; ++c) {
        out[c] = (float*)malloc(sizeof(float) * (frames ? frames : 1));
        if (!out[c]) { fvad_wav_free(out, (size_t)channels); return FVAD_ERR_ALLOC_FAILED; }
    }
    // de-interleave (AudioFileStream.zig:88-95): per channel, unaligned typed loads (little-endian host like every target of this
    // library) in loops the compiler vectorises -- the byte-by-byte form read 1 GB/s
    for (int c = 0; c < channels; ++c) {
        float* o = out[c];
        if (pcm16) {
            const uint8_t* s = data + (size_t)c * 2;
            for (size_t i = 0; i < frames; ++i) o[i] = (float)*(const i16u*)(s + i * frame_bytes) * (1.0f / 32768.0f);
        } else if (channels == 1) {
            memcpy(o, data, frames * 4);
        } else {
            const uint8_t* s = data + (size_t)c * 4;
            for (size_t i = 0; i < frames; ++i) o[i] = *(const f32u*)(s + i * frame_bytes);
        }
    }
    *channel_pcm = out;
    *n_channels = (size_t)channels;
    *n_frames = frames;
    *sample_rate = rate;
    return FVAD_OK;
}

void fvad_wav_free(float** channel_pcm, size_t n_channels)
{
    if (!channel_pcm) return;
    for (size_t c = 0; c < n_channels; ++c) free(channel_pcm[c]);
    free(channel_pcm);
}

int fvad_wav_read_i16(const char* path, int16_t*** channel_pcm, size_t* n_channels, size_t* n_frames, size_t* sample_rate)
{
    if (!path || !channel_pcm || !n_channels || !n_frames || !sample_rate) return FVAD_ERR_INVALID_ARGUMENT;
    WavInfo w;
    const int prc = wav_parse(path, w);
    if (prc) return prc;
    if (!(w.fmt_tag == 1 && w.bits == 16)) return FVAD_ERR_MODEL_FORMAT;
    const size_t frame_bytes = (size_t)w.channels * 2;
    const size_t frames = w.data_bytes / frame_bytes;
    int16_t** out = (int16_t**)calloc((size_t)w.channels, sizeof(int16_t*));
    if (!out) return FVAD_ERR_ALLOC_FAILED;
    for (int c = 0; c < w.channels; ++c) {
        out[c] = (int16_t*)malloc(sizeof(int16_t) * (frames ? frames : 1));
        if (!out[c]) { fvad_wav_free_i16(out, (size_t)w.channels); return FVAD_ERR_ALLOC_FAILED; }
    }
    for (int c = 0; c < w.channels; ++c) { // de-interleave (AudioFileStream.zig:88-95), samples untouched
        if (w.channels == 1) { memcpy(out[c], w.data, frames * 2); continue; }
        const uint8_t* s = w.data + (size_t)c * 2;
        int16_t* o = out[c];
        for (size_t i = 0; i < frames; ++i) o[i] = *(const i16u*)(s + i * frame_bytes);
    }
    *channel_pcm = out;
    *n_channels = (size_t)w.channels;
    *n_frames = frames;
    *sample_rate = w.rate;
    return FVAD_OK;
}

void fvad_wav_free_i16(int16_t** channel_pcm, size_t n_channels)
{
    if (!channel_pcm) return;
    for (size_t c = 0; c < n_channels; ++c) free(channel_pcm[c]);
    free(channel_pcm);
}

// AudioBuffer.saveToFile (src/audio_utils/AudioBuffer.zig:61-118) for the WAV container: interleave the planar
// channels and write IEEE float32 (as_pcm16 == 0) or PCM16.  The reference writes through libsndfile
// (sf_writef_float: WAV, FLAC or Vorbis by Format); PCM16 here is libsndfile's default float -> short conversion
// with normalisation on, lrintf(x * 32767) after clipping to [-1, 1] [external: libsndfile's f2s_clip_array].
int fvad_wav_write(const char* path, const float* const* channel_pcm, size_t n_channels, size_t n_frames, size_t sample_rate,
                   int as_pcm16)
{
    if (!path || (n_frames && !channel_pcm) || n_channels == 0 || n_channels > 65535 || sample_rate == 0) return FVAD_ERR_INVALID_ARGUMENT;
    for (size_t c = 0; c < n_channels && n_frames; ++c) if (!channel_pcm[c]) return FVAD_ERR_CHANNEL_COUNT_MISMATCH;
    const size_t bytes_per = as_pcm16 ? 2 : 4;
    const uint64_t data_bytes = (uint64_t)n_frames * n_channels * bytes_per;
    if (data_bytes > 0xFFFFFF00ull) return FVAD_ERR_INVALID_ARGUMENT; // RIFF sizes are 32-bit
    std::vector<uint8_t> buf(44 + (size_t)data_bytes);
    auto wr32 = [&](size_t o, uint32_t v) { buf[o] = (uint8_t)v; buf[o + 1] = (uint8_t)(v >> 8); buf[o + 2] = (uint8_t)(v >> 16); buf[o + 3] = (uint8_t)(v >> 24); };
    auto wr16 = [&](size_t o, uint16_t v) { buf[o] = (uint8_t)v; buf[o + 1] = (uint8_t)(v >> 8); };
    memcpy(&buf[0], "RIFF", 4); wr32(4, (uint32_t)(36 + data_bytes)); memcpy(&buf[8], "WAVEfmt ", 8);
    wr32(16, 16); wr16(20, as_pcm16 ? 1 : 3); wr16(22, (uint16_t)n_channels); wr32(24, (uint32_t)sample_rate);
    wr32(28, (uint32_t)(sample_rate * n_channels * bytes_per)); wr16(32, (uint16_t)(n_channels * bytes_per)); wr16(34, (uint16_t)(bytes_per * 8));
    memcpy(&buf[36], "data", 4); wr32(40, (uint32_t)data_bytes);
    size_t o = 44;
    for (size_t i = 0; i < n_frames; ++i)
        for (size_t c = 0; c < n_channels; ++c) {
            const float x = channel_pcm[c][i];
            if (as_pcm16) {
                const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
                wr16(o, (uint16_t)(int16_t)lrintf(cl * 32767.0f));
                o += 2;
            } else {
                uint32_t u;
                memcpy(&u, &x, 4);
                wr32(o, u);
                o += 4;
            }
        }
    FILE* fp = fopen(path, "wb");
    if (!fp) return FVAD_ERR_IO;
    const size_t put = fwrite(buf.data(), 1, buf.size(), fp);
    const int cl = fclose(fp);
    return (put == buf.size() && cl == 0) ? FVAD_OK : FVAD_ERR_IO;
}

} // extern "C"
