# Host-side memory checking of the C ABI's lifetime paths (dev tool; gpurun -- 'bash tools/diag/heap_check_run.sh').
# AddressSanitizer itself does not run here: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and needs an ASan build of ROCr + xnack
# ("AddressSanitizer: out of memory: allocator is trying to allocate 0x400000 bytes" at the first hipMalloc, r05_q).  What does run: glibc's heap
# consistency checks on every malloc / free (MALLOC_CHECK_=3 aborts on a corrupted chunk, a double free or an invalid pointer) with freed and fresh memory
# filled with a pattern (MALLOC_PERTURB_), over the ownership tests, the soak and the whisper.h surface -- the tests that free sessions / engines / tickets
# in every order.
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
OUT=gpurun_out/${TAG:-r05_q}_malloc_check.txt
( MALLOC_CHECK_=3 MALLOC_PERTURB_=165 SS_CRASH_BACKTRACE=1 SS_SOAK_SECONDS=45 timeout 1500 python -m pytest tests/test_gpu_lifetime.py tests/test_gpu_variants.py tests/test_gpu_features.py tests/test_gpu_multi.py tests/test_gpu_audio_ctx.py tests/test_gpu_wrap.py -q -m gpu -x -p no:cacheprovider 2>&1 | tail -12 ) > $OUT 2>&1
cat $OUT | cut -c1-250
