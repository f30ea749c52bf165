// Shared declarations for the MI355X (gfx950) Whisper path.  Product code: never includes anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ss {

typedef __bf16 bf16;
typedef _Float16 f16;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define SS_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw ss::Error(-4, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

// after every kernel launch: a bad launch configuration must fail loudly, not leave stale outputs behind
#define SS_LAUNCH_CHECK()                                                                                    \
    do {                                                                                                     \
        hipError_t _e = hipGetLastError();                                                                   \
        if (_e != hipSuccess) throw ss::Error(-4, std::string("kernel launch failed: ") + hipGetErrorString(_e) + " @" + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

// Run `fn` once per HIP device (kernel attributes such as the dynamic-LDS limit are per device; one process may hold one
// engine per GPU).  `mask` is a function-local static owned by the caller.
template <typename F>
inline void once_per_device(std::atomic<uint64_t>& mask, F fn) {
    int dev = 0;
    SS_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (mask.load(std::memory_order_acquire) & bit) return;
    fn();
    mask.fetch_or(bit, std::memory_order_release);
}
inline int device_cu_count() {   // cached per device: hipGetDeviceProperties is far too slow for a launch path
    static std::atomic<int> cu[64];
    int dev = 0;
    SS_HIP(hipGetDevice(&dev));
    int v = cu[dev & 63].load(std::memory_order_relaxed);
    if (v == 0) {
        hipDeviceProp_t p;
        SS_HIP(hipGetDeviceProperties(&p, dev));
        v = p.multiProcessorCount;
        cu[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

// OCP e4m3 (1-4-3, bias 7, no infinities, 0x7f = NaN, largest finite 448) of a float, round to nearest even, saturating: the conversion
// v_cvt_pk_fp8_f32 does on gfx950, used here for the weights of the fp8 engine (quantised once on the host at load)
inline uint8_t f32_to_e4m3(float x) {
    uint32_t b; memcpy(&b, &x, 4);
    const uint8_t sign = (uint8_t)((b >> 24) & 0x80);
    b &= 0x7fffffffu;
    float a; memcpy(&a, &b, 4);
    if (!(a == a)) return (uint8_t)(sign | 0x7f);
    if (a >= 464.0f) return (uint8_t)(sign | 0x7e);            // beyond the last rounding boundary: saturate to 448
    if (a < 0.0009765625f) return sign;                         // below half the smallest subnormal (2^-10): zero (ties at exactly 2^-10 go to even = 0)
    int e = (int)(b >> 23) - 127;                               // unbiased exponent of a
    uint32_t q;                                                 // magnitude in units of the target step
    if (e < -6) {                                               // subnormal target: step 2^-9
        const float t = a * 512.0f;                             // exact
        q = (uint32_t)t;
        const float r = t - (float)q;
        if (r > 0.5f || (r == 0.5f && (q & 1))) q++;
        return (uint8_t)(sign | q);                             // q == 8 is the smallest normal: the encoding carries over
    }
    const uint32_t mant = b & 0x7fffffu;
    q = mant >> 20;                                             // 3 mantissa bits
    const uint32_t rest = mant & 0xfffffu;
    if (rest > 0x80000u || (rest == 0x80000u && (q & 1))) q++;
    if (q == 8) { q = 0; e++; }
    return (uint8_t)(sign | ((uint32_t)(e + 7) << 3) | q);
}

// ggml legacy header; field order is the file order
struct HParams {
    int32_t n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
    int32_t n_text_ctx, n_text_state, n_text_head, n_text_layer, n_mels, ftype;
};

struct Vocab {
    int n_vocab = 51864;
    std::vector<std::string> id_to_token;
    std::map<std::string, int> token_to_id;
    int token_eot = 50256, token_sot = 50257, token_translate = 50357, token_transcribe = 50358;
    int token_solm = 50359, token_prev = 50360, token_nosp = 50361, token_not = 50362, token_beg = 50363;
    bool is_multilingual() const { return n_vocab >= 51865; }
    int num_languages() const { return n_vocab - 51765 - (is_multilingual() ? 1 : 0); }
};
int lang_id(const char* code);  // -1 if unknown
const char* lang_code(int id);   // nullptr if out of range
// whisper.cpp's non_speech_tokens (whisper_full_params.suppress_non_speech_tokens): ids of the vocabulary entries equal to one of the listed symbols or
// to ' ' + the symbol, plus " -" and " '"; ascending, without duplicates
std::vector<int> non_speech_token_ids(const Vocab& vocab);
std::vector<int> tokenize(const Vocab& vocab, const std::string& text);   // whisper.cpp's tokenize(): GPT-2 pre-split + greedy longest match

// Host-side tensor as read from the file: f32 copy + the raw f16 payload when the file stored f16
struct HostTensor {
    std::vector<int> ne;            // ggml order (ne[0] fastest)
    int ttype = 0;                  // 0 f32, 1 f16
    std::vector<float> f32;         // always filled
    size_t n() const { return f32.size(); }
};

struct HostModel {
    HParams hp{};
    int filt_n_mel = 0, filt_n_fft = 0;
    std::vector<float> filters;  // [n_mel][n_fft]
    Vocab vocab;
    std::map<std::string, HostTensor> t;
    int n_quantised = 0;         // tensors that came as ggml block-quantised data (q4_0 .. q8_0) and were de-quantised at load
    const HostTensor& get(const std::string& name) const;
};
void load_ggml_model(const char* path, HostModel& m, bool vocab_only = false);  // throws ss::Error(-2,...)

// audio constants (whisper.cpp: WHISPER_SAMPLE_RATE / N_FFT / HOP_LENGTH / CHUNK_SIZE)
constexpr int kSampleRate = 16000, kNFft = 400, kHop = 160, kChunkSec = 30, kNBins = 201;
inline int mel_n_len(int n_samples) { return (n_samples + kSampleRate * kChunkSec + 2 * (kNFft / 2) - kNFft) / kHop; }
inline int mel_n_len_org(int n_samples) { return 1 + (n_samples + kNFft / 2 - kNFft) / kHop; }

}  // namespace ss
