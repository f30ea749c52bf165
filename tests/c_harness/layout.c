/* Prints "struct field offset size" for every field of the two by-value parameter structs of the C ABI, as a C11 compiler lays them out from
 * the public headers.  tests/test_host_cpu.py compares the output with the committed tables (tests/golden/abi_layout.txt: whisper.h v1.5.4; with
 * -DSS_WHISPER_H_POST_1_5_4 abi_layout_post_1_5_4.txt: v1.5.5; both derived by hand for LP64) and with the ctypes mirrors the Python binding uses. */
#include <stddef.h>
#include <stdio.h>
#include "speaksense.h"
#include "whisper_compat.h"

#define F(S, f) printf(#S " " #f " %zu %zu\n", offsetof(struct S, f), sizeof(((struct S*)0)->f))

int main(void) {
    F(whisper_full_params, strategy); F(whisper_full_params, n_threads); F(whisper_full_params, n_max_text_ctx);
    F(whisper_full_params, offset_ms); F(whisper_full_params, duration_ms); F(whisper_full_params, translate);
    F(whisper_full_params, no_context); F(whisper_full_params, no_timestamps); F(whisper_full_params, single_segment);
    F(whisper_full_params, print_special); F(whisper_full_params, print_progress); F(whisper_full_params, print_realtime);
    F(whisper_full_params, print_timestamps); F(whisper_full_params, token_timestamps); F(whisper_full_params, thold_pt);
    F(whisper_full_params, thold_ptsum); F(whisper_full_params, max_len); F(whisper_full_params, split_on_word);
    F(whisper_full_params, max_tokens); F(whisper_full_params, speed_up); F(whisper_full_params, debug_mode);
    F(whisper_full_params, audio_ctx); F(whisper_full_params, tdrz_enable); F(whisper_full_params, initial_prompt);
    F(whisper_full_params, prompt_tokens); F(whisper_full_params, prompt_n_tokens); F(whisper_full_params, language);
    F(whisper_full_params, detect_language); F(whisper_full_params, suppress_blank); F(whisper_full_params, suppress_non_speech_tokens);
    F(whisper_full_params, temperature); F(whisper_full_params, max_initial_ts); F(whisper_full_params, length_penalty);
    F(whisper_full_params, temperature_inc); F(whisper_full_params, entropy_thold); F(whisper_full_params, logprob_thold);
    F(whisper_full_params, no_speech_thold); F(whisper_full_params, greedy); F(whisper_full_params, beam_search);
    F(whisper_full_params, new_segment_callback); F(whisper_full_params, new_segment_callback_user_data);
    F(whisper_full_params, progress_callback); F(whisper_full_params, progress_callback_user_data);
    F(whisper_full_params, encoder_begin_callback); F(whisper_full_params, encoder_begin_callback_user_data);
    F(whisper_full_params, abort_callback); F(whisper_full_params, abort_callback_user_data);
    F(whisper_full_params, logits_filter_callback); F(whisper_full_params, logits_filter_callback_user_data);
    F(whisper_full_params, grammar_rules); F(whisper_full_params, n_grammar_rules); F(whisper_full_params, i_start_rule);
    F(whisper_full_params, grammar_penalty);
    printf("whisper_full_params sizeof %zu 0\n", sizeof(struct whisper_full_params));
    F(whisper_context_params, use_gpu);
#ifdef SS_WHISPER_H_POST_1_5_4
    F(whisper_context_params, gpu_device); F(whisper_context_params, dtw_token_timestamps); F(whisper_context_params, dtw_aheads_preset);
    F(whisper_context_params, dtw_n_top); F(whisper_context_params, dtw_aheads); F(whisper_context_params, dtw_mem_size);
#endif
    printf("whisper_context_params sizeof %zu 0\n", sizeof(struct whisper_context_params));
    F(whisper_token_data, id); F(whisper_token_data, tid); F(whisper_token_data, p); F(whisper_token_data, plog); F(whisper_token_data, pt);
    F(whisper_token_data, ptsum); F(whisper_token_data, t0); F(whisper_token_data, t1);
#ifdef SS_WHISPER_H_POST_1_5_4
    F(whisper_token_data, t_dtw);
#endif
    F(whisper_token_data, vlen);
    printf("whisper_token_data sizeof %zu 0\n", sizeof(struct whisper_token_data));
    F(ss_params, best_of); F(ss_params, temperature); F(ss_params, temperature_inc); F(ss_params, entropy_thold); F(ss_params, logprob_thold);
    F(ss_params, max_initial_ts); F(ss_params, length_penalty); F(ss_params, no_context); F(ss_params, single_segment); F(ss_params, no_timestamps);
    F(ss_params, suppress_blank); F(ss_params, tdrz_enable); F(ss_params, print_special); F(ss_params, max_tokens); F(ss_params, audio_ctx);
    F(ss_params, translate); F(ss_params, fixed_steps); F(ss_params, language); F(ss_params, n_max_text_ctx); F(ss_params, offset_ms);
    F(ss_params, duration_ms); F(ss_params, detect_language); F(ss_params, prompt_tokens); F(ss_params, prompt_n_tokens); F(ss_params, token_timestamps);
    F(ss_params, initial_prompt); F(ss_params, thold_pt); F(ss_params, thold_ptsum);
    F(ss_params, suppress_non_speech_tokens); F(ss_params, max_len); F(ss_params, split_on_word);
    printf("ss_params sizeof %zu 0\n", sizeof(struct ss_params));
    F(ss_engine_opts, device); F(ss_engine_opts, dtype); F(ss_engine_opts, max_batch); F(ss_engine_opts, max_decoders);
    F(ss_engine_opts, batch_wait_us); F(ss_engine_opts, n_lanes); F(ss_engine_opts, compat); F(ss_engine_opts, reserved);
    printf("ss_engine_opts sizeof %zu 0\n", sizeof(struct ss_engine_opts));
    return 0;
}
