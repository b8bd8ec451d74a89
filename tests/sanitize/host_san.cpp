// Sanitizer driver for the host-only part of libfvad_hip (no GPU): built by tests/test_sanitizers.py with
// -fsanitize=address,undefined from host_vad.cpp, host_stats.cpp, host_io.cpp and tables_weights.cpp.
//   host_san onnx <file>...      parse each file as an NSNet2 ONNX model (errors are fine, crashes are not)
//   host_san wav <file>...       same for the WAV reader
//   host_san audacity <file>...  same for the Audacity label parser
//   host_san vad <seed>          VAD state machines on a random script: run() vs run_many(), stats
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fvad.h"

static std::vector<char> slurp(const char* path)
{
    std::vector<char> v;
    FILE* f = fopen(path, "rb");
    if (!f) return v;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    int ok = 0, err = 0;
    if (mode == "onnx") {
        for (int i = 2; i < argc; ++i) {
            fvad_nsnet2_weights w;
            void* owner = nullptr;
            const int rc = fvad_onnx_read_nsnet2(argv[i], &w, &owner);
            if (rc == FVAD_OK) { ++ok; volatile float s = w.fc1_w[0] + w.fc4_b[160]; (void)s; fvad_weights_free(owner); }
            else ++err;
        }
    } else if (mode == "wav") {
        for (int i = 2; i < argc; ++i) {
            float** pcm = nullptr;
            size_t nc = 0, nf = 0, sr = 0;
            const int rc = fvad_wav_read(argv[i], &pcm, &nc, &nf, &sr);
            if (rc == FVAD_OK) {
                ++ok;
                double s = 0;
                for (size_t c = 0; c < nc; ++c) for (size_t k = 0; k < nf; ++k) s += pcm[c][k];
                volatile double sink = s; (void)sink;
                fvad_wav_free(pcm, nc);
            } else ++err;
        }
    } else if (mode == "audacity") {
        for (int i = 2; i < argc; ++i) {
            const std::vector<char> t = slurp(argv[i]);
            std::vector<fvad_segment_sec> out(t.size() / 2 + 4);
            size_t n = 0;
            const int rc = fvad_parse_audacity(t.data(), t.size(), out.data(), out.size(), &n);
            // a too-small buffer must be reported, not overrun
            size_t n2 = 0;
            fvad_segment_sec one;
            const int rc2 = fvad_parse_audacity(t.data(), t.size(), &one, 1, &n2);
            if (rc == FVAD_OK) ++ok; else ++err;
            (void)rc2;
        }
    } else if (mode == "vad") {
        unsigned x = (unsigned)atoi(argv[2]) * 2654435761u + 12345u;
        auto rnd = [&] { x = x * 1664525u + 1013904223u; return (x >> 8) * (1.0f / (1 << 24)); };
        const size_t n_streams = 5, n_frames = 30000, n_ch = 2;
        fvad_vad_config cfg;
        fvad_vad_config_default(&cfg);
        std::vector<std::vector<float>> band(n_streams), ratio(n_streams);
        for (size_t s = 0; s < n_streams; ++s) {
            band[s].resize(n_frames * n_ch);
            ratio[s].resize(n_frames);
            float level = 0.002f;
            for (size_t k = 0; k < n_frames; ++k) {
                level *= 1.0f + 0.02f * (rnd() - 0.5f);
                const bool burst = ((k / 97 + s) % 11) == 0;
                for (size_t c = 0; c < n_ch; ++c) band[s][k * n_ch + c] = level * (0.5f + rnd()) * (burst ? 40.0f : 1.0f);
                ratio[s][k] = (k % 501 == 7) ? NAN : 0.3f + 0.7f * rnd();
            }
        }
        std::vector<fvad_vad*> a(n_streams), b(n_streams);
        for (size_t s = 0; s < n_streams; ++s) {
            if (fvad_vad_create(&cfg, 48000, n_ch, 1024, &a[s]) || fvad_vad_create(&cfg, 48000, n_ch, 1024, &b[s])) return 3;
            for (size_t k = 0; k < n_frames; ++k) {
                fvad_vad_result r;
                const bool has = !std::isnan(ratio[s][k]);
                if (fvad_vad_run(a[s], 1024 * k, &band[s][k * n_ch], has, has ? ratio[s][k] : 0, &r)) return 4;
            }
        }
        std::vector<const float*> bp(n_streams), rp(n_streams);
        std::vector<size_t> nf(n_streams, n_frames);
        std::vector<uint64_t> fi(n_streams, 0);
        for (size_t s = 0; s < n_streams; ++s) { bp[s] = band[s].data(); rp[s] = ratio[s].data(); }
        if (fvad_vad_run_many(b.data(), n_streams, bp.data(), rp.data(), nf.data(), n_ch, fi.data(), 1024, 3)) return 5;
        for (size_t s = 0; s < n_streams; ++s) {
            const size_t na = fvad_vad_segment_count(a[s]), nb = fvad_vad_segment_count(b[s]);
            if (na != nb) { fprintf(stderr, "segment count mismatch\n"); return 6; }
            std::vector<fvad_speech_segment> sa(na + 1), sb(nb + 1);
            size_t n1 = 0, n2 = 0;
            fvad_vad_segments(a[s], sa.data(), sa.size(), &n1);
            fvad_vad_segments(b[s], sb.data(), sb.size(), &n2);
            if (n1 != n2 || memcmp(sa.data(), sb.data(), n1 * sizeof(fvad_speech_segment))) { fprintf(stderr, "segments differ\n"); return 7; }
            // Evaluator statistics on the segments against themselves shifted
            std::vector<fvad_segment_sec> vs(n1), ref(n1);
            for (size_t i = 0; i < n1; ++i) {
                vs[i] = fvad_segment_to_sec(&sa[i], 48000);
                ref[i] = vs[i];
                ref[i].from_sec += 0.25f;
            }
            fvad_stat_config sc = {0.7f, 5.0f, 10.0f, 5.0f};
            fvad_single_stats st;
            if (fvad_stats_from_segments(vs.data(), n1, ref.data(), n1, &sc, &st)) return 8;
            fvad_vad_destroy(a[s]);
            fvad_vad_destroy(b[s]);
            ++ok;
        }
    } else {
        return 2;
    }
    printf("%s: ok=%d err=%d\n", mode.c_str(), ok, err);
    return 0;
}
