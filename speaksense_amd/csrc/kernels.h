// Launch wrappers for the gfx950 kernels (definitions in kernels_*.hip).  T = ss::bf16 or ss::f16.
#pragma once
#include "common.h"

namespace ss {

// ---------------------------------------------------------------------------------------------
// log-mel (kernels_mel.hip)
// ---------------------------------------------------------------------------------------------
struct MelTables {       // device pointers, built once per engine
    const float* sin_t;  // [400]
    const float* cos_t;  // [400]
    const float* hann;   // [400]
    const float* filt;   // [n_mel][201]
    int n_mel;
};
// pcm: device f32 [n_samples]; mel_out: device f32 [n_mel][n_len]; scratch: device f32 [>= 1 + 2048]
void launch_log_mel(const MelTables& mt, const float* pcm, int n_samples, float* mel_out, int n_len, float* scratch, hipStream_t st);
void launch_signal_energy(const float* pcm, int n_samples, float* energy, hipStream_t st);   // whisper.cpp get_signal_energy(.., 32): one value per sample
// mel [n_mel][n_len] f32 -> time-major window x0[T2+2][n_mel] (rows 0 and T2+1 zero) in T, frames [seek, seek+T2)
template <typename T>
void launch_mel_window(const float* mel, int n_mel, int n_len, int seek, int T2, T* x0, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// MFMA GEMM (kernels_gemm.hip):  D[m][n] = sum_k A[m][k] * W[n][k]   (both operands K-contiguous)
// ---------------------------------------------------------------------------------------------
enum EpiKind {
    EPI_STORE_T = 0,   // out T[m][n] = (acc + bias[n]) * scale
    EPI_GELU_T,        // out T[m][n] = gelu(acc + bias[n])
    EPI_RES_F32,       // out f32[m][n] = res[m][n] + acc + bias[n]           (in place on the residual stream)
    EPI_GELU_POS_F32,  // out f32[m][n] = gelu(acc + bias[n]) + pos[(m % rows_per_batch)][n]   (conv2 + positional embedding)
    EPI_VT,            // out T: V^T layout [b][h][64][Tpad], b = m / rows_per_batch, t = m % rows_per_batch (acc + bias)
    EPI_CROSS_KV,      // out T: cross cache [l][b][kv][h][t][64]; n = l*2d + kv*d + h*64 + j; K part scaled by `scale`
    EPI_STORE_F32,     // out f32[m][n] = acc + bias[n]
};
struct GemmDesc {
    // A operand: row m at A + m * lda, or with a_rows_per_batch > 0 (conv stem) at A + (m / a_rows_per_batch) * a_batch_stride + (m % a_rows_per_batch) * lda   (elements)
    const void* A; long lda; int a_rows_per_batch; long a_batch_stride;
    const void* W;            // [N][K], K contiguous
    int M, N, K;
    int kind;
    const float* bias;        // [N] or null
    void* out; long ldo;      // row m at out + m * ldo, or with o_rows_per_batch > 0 as for A
    int o_rows_per_batch; long o_batch_stride;
    const float* res;         // EPI_RES_F32 (same addressing as out)
    const float* pos;         // EPI_GELU_POS_F32: [rows_per_batch][N]
    float scale;              // EPI_STORE_T / EPI_CROSS_KV(K part)
    int rows_per_batch;       // T (1500) for EPI_VT / EPI_CROSS_KV / EPI_GELU_POS_F32
    int d, Tpad, n_batch;     // EPI_VT / EPI_CROSS_KV geometry
    int gelu_f16_in;          // f16 engines: gelu(f16(x)) like ggml's table (no-op for bf16)
    int use_batch_map;        // EPI_CROSS_KV: window b of this launch writes cache slot batch_map[b] instead of b
    unsigned char batch_map[128];
    int cache_rows;           // EPI_CROSS_KV: key rows per (slot, head) of the CACHE (n_audio_ctx); 0 = rows_per_batch.  Differs when whisper_full_params.audio_ctx shortens the pass
    int nt_out;               // set by launch_gemm: 16-bit outputs of more than kNtOutBytes leave as non-temporal stores (kernels_gemm.hip st16_out)
    long long* trace;         // dev tool (tools/gemm_bench.cpp): per workgroup and tile {start, loop start, loop end, stores issued} s_memtime stamps; null in the product
};
template <typename T> void launch_gemm(const GemmDesc& g, hipStream_t st);
// self-test: tiled kernel vs a one-thread-per-output reference on seeded operands; kind in {EPI_STORE_T, EPI_GELU_T, EPI_RES_F32, EPI_STORE_F32}
// reps > 0: afterwards the same launch is repeated `reps` times between two events on `st` and the average duration returned in *avg_ms
template <typename T> void gemm_selftest(int M, int N, int K, int kind, float* max_err, float* max_ref, hipStream_t st, int reps = 0, float* avg_ms = nullptr);

// ---------------------------------------------------------------------------------------------
// fp8 GEMM (kernels_gemm_fp8.hip):  OCP e4m3 operands on the MX-scaled 32x32x64 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4)
//   D[m][n] = w_scale[n] * sum_blk 2^(a_scale[m][blk] - 127) * sum_{k in blk} A8[m][k] * W8[n][k]      (blocks of 64 k)
// Activations carry one E8M0 exponent byte per (row, 64-column block), the hardware applies it inside the MFMA; weights carry one f32
// scale per output channel, applied in the epilogue.  The exponent bytes live in a tile-aware layout so that one 256-byte LDS-DMA
// per k-step brings a 256-row tile's bytes and a lane's four row groups sit in one dword: f8_scale_index().
// ---------------------------------------------------------------------------------------------
enum Fp8Epi {
    F8_STORE_T = 0,   // out T[m][n] = (acc * ws[n] + bias[n]) * scale
    F8_GELU_F8,       // out e4m3[m][n] = quantised gelu(acc * ws[n] + bias[n]) with its own exponent bytes (out_scale)
    F8_RES_F32,       // out f32[m][n] = res[m][n] + acc * ws[n] + bias[n]
    F8_VT,            // as EPI_VT
    F8_CROSS_KV,      // as EPI_CROSS_KV
    F8_STORE_F32,     // out f32[m][n] = acc * ws[n] + bias[n]
    F8_CROSS_KV8,     // the cross cache itself in e4m3: codes [l][b][kv][h][t][64] + one exponent byte per (l, b, kv, h, t) in out_scale (K pre-scaled)
};
__host__ __device__ inline long f8_scale_index(long m, long blk, long ldsc) {
    return blk * ldsc + (m & ~(long)127) + (m & 31) * 4 + ((m >> 5) & 3);
}
struct GemmF8Desc {
    const unsigned char* A; long lda;           // e4m3 [M][lda]
    const unsigned char* a_scale; long ldsc;    // E8M0 bytes, f8_scale_index(m, k / 64, ldsc); ldsc = rows padded to a multiple of 256
    const unsigned char* W;                     // e4m3 [N][K]
    const float* w_scale;                       // [N]
    int M, N, K, kind;
    const float* bias;
    void* out; long ldo;
    unsigned char* out_scale; long ld_osc;      // F8_GELU_F8
    const float* res;
    float scale;
    int rows_per_batch, d, Tpad, n_batch, gelu_f16_in;
    int use_batch_map; unsigned char batch_map[128];
    int cache_rows;           // as in GemmDesc
};
template <typename T> void launch_gemm_f8(const GemmF8Desc& g, hipStream_t st);
// LayerNorm with quantised output (one wave per row), and plain quantisation of a T matrix (attention output): e4m3 + exponent bytes
void launch_layernorm_f8(const float* x, const float* w, const float* b, unsigned char* y8, unsigned char* y_scale, long ldsc, int rows, int d, hipStream_t st);
template <typename T> void launch_quantize_f8(const T* x, long ldx, unsigned char* y8, unsigned char* y_scale, long ldsc, int rows, int d, hipStream_t st);
// self-test: the fp8 kernel against a one-thread-per-output reference on seeded e4m3 operands and exponent bytes; kind in Fp8Epi except VT / CROSS_KV
template <typename T> void gemm_f8_selftest(int M, int N, int K, int kind, float* max_err, float* max_ref, hipStream_t st, int reps = 0, float* avg_ms = nullptr);
// host + device: exponent byte for a block whose largest magnitude is amax (amax / 2^(e-127) <= 448), and the OCP e4m3 code of a float
__host__ __device__ inline int e8m0_for_amax(float amax) {
    unsigned bits; __builtin_memcpy(&bits, &amax, 4);
    int e = (int)((bits >> 23) & 0xff) - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
    return e < 1 ? 1 : (e > 253 ? 253 : e);
}

// device helpers shared by the producers of e4m3 activations (kernels_gemm_fp8.hip, the fp8 epilogue of enc_attn_lds_kernel)
__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d) {
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (unsigned)p;
}
__device__ __forceinline__ float pow2_neg_of_e8m0(int e) {   // 2^-(e - 127), e in [1, 253]
    return __builtin_bit_cast(float, (unsigned)(254 - e) << 23);
}

struct RowCtl {       // one per decode row; lives in pinned host memory mapped into the device
    int32_t token;    // input token id
    int32_t pos;      // position (= n_past)
    int32_t slot;     // self-KV slot
    int32_t cross;    // cross-KV index (window in the current batch)
    int32_t n_hist;   // tokens sampled so far in this window (0 => is_initial)
    int32_t last_ts, penult_ts;   // rule state (whisper_process_logits)
    int32_t has_ts, ts_min;       // decoder.has_ts, seek_delta/2   (RuleConsts.openai_ts: any id >= beg sampled so far, index of the last such id)
    float temperature;
    int32_t want_probs;           // t > 0: also write the full probability row
    int32_t n_keys;               // cross-attention keys of this row's window (whisper_full_params.audio_ctx); 0 = the kernel's Tn (n_audio_ctx)
};

// Fused decode-step GEMV for M <= 16 rows (kernels_decode.hip): prologue + 16-row weight tiles x split-K + epilogue
constexpr int kPartRows = 128;  // most token rows ONE decoder pass carries (row stride of the split-K partial buffers [S][kPartRows][N]; the control blocks of a
                                // pass are ctl[0, kPartRows) = its rows, ctl[kPartRows, 2 kPartRows) = its sampling rows).  Round 4: 64 -> 128 (CT = 8 column tiles)
enum DecPro { PRO_LN = 0, PRO_T = 1 };   // PRO_LN: the descriptor of a dec_reduce_ln launch; PRO_T: a GEMV whose activations are T rows (Xt)
enum DecEpi { DEPI_PART = 0, DEPI_QKV = 1, DEPI_GELU_T = 2, DEPI_LOGITS = 3 };
// Layout of every weight matrix the decode-step GEMVs read ([N][K] in the file, N a multiple of 16, K of 32): FRAGMENT-MAJOR.  The 16 x 32 block
// (rows 16 t .. 16 t + 15, columns 32 b .. 32 b + 31) -- one MFMA 16x16x32 A operand -- is one contiguous kilobyte in lane order: lane
// l = (n % 16) + 16 ((k % 32) / 8) holds its 8 consecutive k.  A wave's weight-load instruction then reads 8 full 128-byte lines instead of 64 bytes
// in each of 16 lines that lie a row pitch apart: ~2.7 x less time on the CU's address path per instruction (tools/diag/dma_issue_bench.cpp
// measures the same effect for LDS-DMA) and sequential HBM bursts; the stand-in chain of tools/diag/hetero_tick_bench.cpp runs 7.5 % faster
// for it (profiles/r04_ak_packed_weights_bench.txt).  The weights are constants: the engine packs them once at load.
// The ACTIVATION rows a decode-step GEMV multiplies them with (the B operand: LayerNorm output, attention output, GELU output; <= 128 token rows)
// use the same layout with n = token row: their producers (dec_reduce_ln, dec_self_attn, dec_cross_attn / combine, the GELU epilogue) store
// through dec_wpack_off, and the GEMV's three times more numerous activation loads become contiguous kilobytes too (stand-in chain: 59.5 ->
// 41.5 us per layer, same file).  Buffers hold a multiple of 16 rows; rows of a tile beyond M are stale but finite and feed MFMA columns
// that are never stored.
__host__ __device__ inline long dec_wpack_off(long n, int k, int K) {
    return ((((n >> 4) * (K >> 5) + (k >> 5)) * 64) + (n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7);
}

struct DecGemvDesc {
    int pro, epi;
    // PRO_LN: x = x_in (or tok/pos embedding when ctl != null) + bias_prev + sum_p parts[p]; optional write-back; LayerNorm
    const float* x_in; float* x_out;
    const float* parts; int n_parts;        // [n_parts][kPartRows][K] f32 split-K partials of the previous projection
    const float* bias_prev;                 // [K] or null
    const float* ln_w; const float* ln_b;
    const RowCtl* ctl; const void* tok_emb; const float* pos_emb;   // embedding prologue (layer 0); tok_emb in the decoder-weight layout below
    const int* row_idx;                     // optional row gather (final LayerNorm of the sampling rows)
    const void* Xt; long ldx;               // PRO_T: T [M][ldx]
    const void* W; int M, N, K, S;          // W: T [N][K]
    const float* bias; void* out; long ldo; float scale; int n_valid;
    float* part_out;                        // DEPI_PART: [S][kPartRows][N]
    const RowCtl* ctl_rows; void* kcache; void* vcache; long slot_stride; int d;   // DEPI_QKV
    int gelu_f16_in;
};
void dec_gemv_plan(int N, int K, int* S_out, int* NW_out, bool whole_heads = false);
template <typename T> void launch_dec_gemv(const DecGemvDesc& g, int NW, hipStream_t st);
// stand-alone prologue: out T [M][K] = LayerNorm(x_in + bias_prev + sum parts) with optional row gather / x_out write-back
template <typename T> void launch_dec_reduce_ln(const DecGemvDesc& g, T* out, hipStream_t st);
// Few rows (M <= kLnFuseRows: the latency configuration, one chunk at a time): the residual update + LayerNorm of launch_dec_reduce_ln as the
// PROLOGUE of the GEMV that consumes it -- one launch instead of two.  `g` carries both halves (x_in / parts / bias_prev / ln_w / ln_b / x_out /
// ctl / row_idx AND W / N / K / epilogue fields); S == 1.  Every workgroup normalises the M rows itself (M x K f32 from L2: nothing at M <= 4;
// at 32 rows x 320 workgroups it was 157 MB per launch, which is why the same fusion lost at the benchmark's row counts).
constexpr int kLnFuseRows = 4;
template <typename T> void launch_dec_gemv_ln(const DecGemvDesc& g, int NW, hipStream_t st);
// cross-attention whose q comes as split-K partials: q = round_T((sum_p qpart[p] + qbias) * qscale); writes (m,l,o[64]) partials
template <typename T>
void launch_dec_cross_attention_q(const float* qpart, int n_qpart, const float* qbias, float qscale, const T* kc, const T* vc, long b_stride, int d, int H,
                                  int Tn, const RowCtl* ctl, int M, float* scratch, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// attention (kernels_attn.hip)
// ---------------------------------------------------------------------------------------------
// Encoder self-attention, non-causal.  q,k: T [B*Tn][ld] (head h at column h*64); vT: T [B][H][64][Tpad]; out T [B*Tn][ldo]
template <typename T>
void launch_enc_attention(const T* q, const T* k, long ld, const T* vT, int Tpad, T* out, long ldo, int B, int H, int Tn, hipStream_t st);
// V rows [B*Tn][ld] (head h at column h*64) -> V^T [B][H][64][Tpad] for the kernel above (zeros in the key columns >= Tn of the last 64-key tile)
template <typename T>
void launch_v_transpose(const T* v, long ld, T* vT, int Tpad, int B, int H, int Tn, hipStream_t st);
// fp8 engine: the same attention with its output rounded to T and then quantised in the epilogue: e4m3 [B*Tn][ldo] + one exponent byte per (row, head)
template <typename T>
void launch_enc_attention_f8(const T* q, const T* k, long ld, const T* vT, int Tpad, unsigned char* out8, long ldo, unsigned char* out_scale, long ldsc, int B, int H, int Tn, hipStream_t st);
// Decoder self-attention for M rows (one new token each): q T [M][d] (pre-scaled), caches [slot][n_ctx][d]; n_kv = pos+1
template <typename T>
void launch_dec_self_attention(const T* q, const T* kcache, const T* vcache, long slot_stride, int d, int H, const RowCtl* ctl, int M, T* out, hipStream_t st);

// one-workgroup variant (NR = 4): one workgroup per (row, head) walks the SAME four key ranges, merges them with the combine kernel's expression and writes
// the normalised output T [M][d] directly (no partials, no combine launch): bit-identical to launch_dec_cross_attention_q + launch_dec_cross_combine
template <typename T>
void launch_dec_cross_attention_direct(const float* qpart, int n_qpart, const float* qbias, float qscale, const T* kc, const T* vc, long b_stride, int d,
                                       int H, int Tn, const RowCtl* ctl, int M, T* out, hipStream_t st);
// flash-decoding combine of the cross-attention partials: out T [M][d]
template <typename T> void launch_dec_cross_combine(const float* scratch, int d, int H, int M, T* out, hipStream_t st);
// fp8 engine: the same two cross-attention forms over an e4m3 cross cache (codes [kv][h][t][64] per window, exponent bytes [kv][h][t]); b_stride /
// sc_stride = bytes per window of the two buffers; scratch != null: 4 key splits into scratch (then launch_dec_cross_combine), else direct into out
template <typename T>
void launch_dec_cross_attention_f8(const float* qpart, int n_qpart, const float* qbias, float qscale, const unsigned char* kc, const unsigned char* ksc,
                                   long b_stride, long sc_stride, int d, int H, int Tn, const RowCtl* ctl, int M, float* scratch, T* out, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// misc (kernels_misc.hip)
// ---------------------------------------------------------------------------------------------
// LayerNorm over rows of f32 -> T   (eps 1e-5, double-free two-pass in registers)
// row_idx (optional): output row r normalises input row row_idx[r] (gather)
template <typename T> void launch_layernorm(const float* x, const float* w, const float* b, T* y, int rows, int d, const int* row_idx, hipStream_t st);
template <typename T> void launch_layernorm_f32out(const float* x, const float* w, const float* b, float* y, int rows, int d, hipStream_t st);
// conversions
template <typename T> void launch_f32_to_T(const float* in, T* out, size_t n, hipStream_t st);
template <typename T> void launch_T_to_f32(const T* in, float* out, size_t n, hipStream_t st);

// Fused whisper_process_logits + log-softmax + whisper_sample_token(best) (kernels_misc.hip)
struct RuleConsts {
    int32_t n_vocab, eot, sot, translate, transcribe, solm, prev, nosp, not_, beg, blank /* id of " " or -1 */, n_lang;
    int32_t suppress_blank, no_timestamps, tdrz_enable, max_initial_tid /* -1 = off */, suppress_eot /* Mode F */;
    int32_t openai_ts;   // SS_COMPAT_OPENAI_TS_RULES: RowCtl.has_ts / ts_min then mean "a timestamp (id >= beg) was sampled" / "the last one's index"
    const uint32_t* ns_mask;   // ss_params.suppress_non_speech_tokens: device bitmask over the vocabulary (bit i of word i / 32), null = off
};
static_assert(sizeof(RuleConsts) == 18 * 4 + 8, "RuleConsts is compared with memcmp: no padding");
struct SampleOut { int32_t id, tid; float p, plog, pt, ptsum; int32_t pad[2]; };
// scratch: >= M * 64 * 8 floats
// ctl_upd (optional): base of the step's device control blocks; the pick kernel then rewrites rows [64 + m] and [row_of[m]] into the
// control block of the next greedy step (token = the pick, pos + 1, rule state advanced)
void launch_logits_rules(const float* logits, long ld, const RowCtl* ctl, int M, const RuleConsts& rc, SampleOut* out, float* probs /* [M][ld] or null */,
                         float* scratch, hipStream_t st, RowCtl* ctl_upd = nullptr, const int* row_of = nullptr);

// stage hook: after launch_logits_rules, the processed log-softmax rows (-inf where a rule masks the id), [M][ld]
void launch_logits_logprob_rows(const float* logits, long ld, const RowCtl* ctl, int M, const RuleConsts& rc, const SampleOut* picked, float* logprobs, hipStream_t st);

// rows with ctl[m].want_probs: out[m].id / out[m].p <- the index std::discrete_distribution would return for the uniform draw u[m]
void launch_sample_draw(const float* probs, long ld, int n_vocab, const RowCtl* ctl, int M, const double* u, SampleOut* out, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// STFT denoiser, frame size 2048 (kernels_denoise.hip) -- SURVEY.md §8f next #1
// ---------------------------------------------------------------------------------------------
void launch_dn_chunk_power(const float* x, int n_chunks, const float2* tw, float* power, hipStream_t st);
void launch_dn_spectra(const float* power, int n_chunks, float* noise, float* signal, float* var_out, hipStream_t st);
void launch_dn_frames(int mode, const float* x, int n_frames, int step, const float2* tw, const float* noise, const float* signal, float strength,
                      float* frames_out, hipStream_t st);
void launch_dn_overlap_add(const float* frames, int n_frames, int step, int n, float* out, hipStream_t st);
void launch_dn_noise_gate(const float* in, float* out, int n, float gate, hipStream_t st);
// sinc resampler (kernels_resample.hip) -- SURVEY.md §8f next #4: y[k] at instant idx_rel[k] of read chunk chunk_of[k]
void launch_resample(const float* x, const double* idx_rel, const int* chunk_of, long n_out, const float* sincs, float* y, hipStream_t st);
// stream pre-processor (StreamAudioProcessor, src/audio/mod.rs:67-155) over a whole mono 16 kHz stream; frames of 2048, last one zero padded
void launch_pp_stream(const float* x, long n, const long* chunk_off, int n_chunks, int chunk_len, int n_frames, const float2* tw, float strength, float gate,
                      int denoise, float* y, float* energy, float* sub_energy, float* gain, float* out, hipStream_t st);

}  // namespace ss
