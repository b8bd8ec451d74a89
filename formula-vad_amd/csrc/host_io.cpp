// host_io.cpp -- minimal RIFF/WAVE reader.
// The reference reads audio through libsndfile (src/audio_utils/AudioFileStream.zig:18-102,
// AudioBuffer.zig:26-59: sf_readf_float, then de-interleave into planar f32).  libsndfile is not
// part of this build; this reader covers what the simulator harness needs: PCM16 and IEEE float32
// WAV, any channel count, de-interleaved to channel-planar f32.  PCM16 is scaled by 1/32768 like
// libsndfile's normalised float read.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fvad.h"

namespace {
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
} // namespace

namespace {
struct WavInfo { std::vector<uint8_t> buf; int fmt_tag = 0, channels = 0, bits = 0; uint32_t rate = 0; const uint8_t* data = nullptr; size_t data_bytes = 0; };

int wav_parse(const char* path, WavInfo& w)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return FVAD_ERR_IO;
    std::vector<uint8_t>& buf = w.buf;
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 12) { fclose(fp); return FVAD_ERR_MODEL_FORMAT; }
    buf.resize((size_t)sz);
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
    // de-interleave (AudioFileStream.zig:88-95)
    for (size_t i = 0; i < frames; ++i)
        for (int c = 0; c < channels; ++c) {
            const uint8_t* s = data + i * frame_bytes + (size_t)c * (bits / 8);
            if (pcm16) out[c][i] = (float)(int16_t)rd16(s) * (1.0f / 32768.0f);
            else { uint32_t u = rd32(s); float f; memcpy(&f, &u, 4); out[c][i] = f; }
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
    for (size_t i = 0; i < frames; ++i) // de-interleave (AudioFileStream.zig:88-95), samples untouched
        for (int c = 0; c < w.channels; ++c) out[c][i] = (int16_t)rd16(w.data + i * frame_bytes + (size_t)c * 2);
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

} // extern "C"
