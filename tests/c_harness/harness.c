/* C11 consumer of the drop-in boundary: includes both public headers as a C (not C++) translation unit, links against
 * libspeaksense_hip.so and transcribes one synthetic chunk through (1) the native API and (2) the whisper.h-compatible subset,
 * printing the segments.  tests/test_host_cpu.py compiles and links it (no GPU needed); tests/test_gpu_variants.py runs it. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "speaksense.h"
#include "whisper_compat.h"

static float* make_audio(int n) {
    float* x = (float*)malloc(sizeof(float) * (size_t)n);
    unsigned s = 12345u;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        const float noise = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.02f;
        const float t = (float)i / 16000.0f;
        x[i] = 0.3f * sinf(6.2831853f * 180.0f * t) * (0.6f + 0.4f * sinf(6.2831853f * 4.0f * t)) + 0.15f * sinf(6.2831853f * 360.0f * t) + noise;
    }
    return x;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: harness <ggml model> [seconds]\n"); return 2; }
    const int n = 16000 * (argc > 2 ? atoi(argv[2]) : 8);
    float* pcm = make_audio(n);

    /* (1) native API */
    ss_engine* eng = NULL;
    if (ss_engine_create(argv[1], NULL, &eng) != SS_OK) { fprintf(stderr, "ss_engine_create: %s\n", ss_last_error()); return 1; }
    ss_session* ses = ss_session_create(eng);
    ss_params p;
    ss_default_params(&p);
    p.temperature_inc = 0.0f;
    strcpy(p.language, "en");
    if (ss_transcribe(ses, pcm, n, &p) != SS_OK) { fprintf(stderr, "ss_transcribe: %s\n", ss_last_error()); return 1; }
    const int ns = ss_result_n_segments(ses);
    printf("native %d\n", ns);
    for (int i = 0; i < ns; i++)
        printf("N %lld %lld %s\n", (long long)ss_result_segment_t0(ses, i), (long long)ss_result_segment_t1(ses, i), ss_result_segment_text(ses, i));
    ss_session_free(ses);
    ss_engine_free(eng);

    /* (2) whisper.h subset, as whisper-rs-sys binds it */
    struct whisper_context_params cp = whisper_context_default_params();
    struct whisper_context* ctx = whisper_init_from_file_with_params_no_state(argv[1], cp);
    if (!ctx) { fprintf(stderr, "whisper_init_from_file_with_params_no_state failed\n"); return 1; }
    struct whisper_state* st = whisper_init_state(ctx);
    struct whisper_full_params wp = whisper_full_default_params(WHISPER_SAMPLING_GREEDY);
    wp.language = "en";
    wp.temperature_inc = 0.0f;
    wp.no_context = true;
    if (whisper_full_with_state(ctx, st, wp, pcm, n) != 0) { fprintf(stderr, "whisper_full_with_state failed\n"); return 1; }
    const int nw = whisper_full_n_segments_from_state(st);
    printf("compat %d\n", nw);
    for (int i = 0; i < nw; i++)
        printf("W %lld %lld %s\n", (long long)whisper_full_get_segment_t0_from_state(st, i), (long long)whisper_full_get_segment_t1_from_state(st, i),
               whisper_full_get_segment_text_from_state(st, i));
    whisper_free_state(st);
    whisper_free(ctx);
    free(pcm);
    return 0;
}
