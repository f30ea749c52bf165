// Excerpt of speaksense_amd/csrc/kernels_gemm.hip at the time of the experiment (uses its Mfma<T>, glds16, wait_vmcnt, tile_of_block, row_off, epilogue256;
// GemmDesc had an extra `int ko` knock-out field).  See README.md.

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, eight phases per two k-tiles ("8-phase" schedule, cdna_hip_programming.md section 5 / T3+T4+T5): the same wave tiling
// and epilogue as gemm256_kernel (8 waves = 2 m x 4 n, 128 x 64 per wave), a different main loop.
//
// gemm256_kernel interleaves {8 MFMA, 1 LDS-DMA, 3 ds_read} in every wave, so both waves of a SIMD issue memory instructions at the same
// time and the matrix pipe waits (knock-out analysis, DESIGN.md section 8.2: the loop is bound by memory-instruction issue).  Here the two
// waves of a SIMD (wave w and w + 4 = the two m halves) run one barrier apart: while one is in a pure 16-MFMA cluster (one 64 x 32 quadrant
// of its tile x 64 k, s_setprio 1) its partner issues the ds_reads and LDS-DMAs of its next quadrant, then they swap.
//
// LDS: 2 k-tile buffers x 4 half-tiles [W cols 0-127 | W cols 128-255 | X rows 0-127 | X rows 128-255] x 16 KB (128 rows x 64 k), rows of
// 128 B, 16-B chunk c of row r at position c ^ ((r >> 1) & 7) (applied on the DMA source address and on the read: conflict-free ds_read_b128).
// Staging is a continuous stream of half-tiles, one per phase (2 DMA instructions per wave), 6 phases ahead of its first read: a half-tile's
// region is rewritten one phase after its last read (every phase waits for its own ds_reads BEFORE its first barrier, so that is safe for
// both wave groups), and is waited for (counted vmcnt, 2 half-tiles stay in flight) two phases after its last piece was issued.  The stream
// runs across tile boundaries of the persistent loop: the first 6 half-tiles of tile i + 1 are in flight during the epilogue of tile i.
// ---------------------------------------------------------------------------------------------
constexpr int kP8Half = 128 * 64 * 2;          // 16 KB
constexpr int kP8Lds = 2 * 4 * kP8Half;        // 128 KB

template <typename T, int KIND>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(GemmDesc g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mfma<T>::V8 V8;
    constexpr bool SWAP = (KIND == EPI_VT);
    constexpr bool ST16 = KIND == EPI_STORE_T || KIND == EPI_GELU_T || KIND == EPI_CROSS_KV;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    const int nbn = g.N / 256, nbm = (g.M + 255) / 256, n_tiles = nbn * nbm;
    const int nk = g.K / 64;                                   // k-tiles per output tile
    const char* __restrict__ Ab = (const char*)g.A;
    const char* __restrict__ Wb = (const char*)g.W;

    // ---- staging stream -------------------------------------------------------------------------------------------
    // half-tile j of a k-tile: 0, 1 = W columns [128 j, +128); 2, 3 = X rows [128 (j - 2), +128).  This wave stages rows [16 wave, +16) of every
    // half-tile with two DMA instructions (8 rows x 128 B each); lane l writes LDS row l >> 3, chunk position l & 7, i.e. it fetches chunk
    // (l & 7) ^ ((row >> 1) & 7) of that row.  The stream order is k-tile major, j minor, 6 half-tiles ahead of the phase that issues it:
    // phase p of k-tile kt stages half-tile (p + 2) & 3 of k-tile kt + 1 (p < 2) or kt + 2 -- past the end of the tile that is the next tile's
    // k-tile 0 / 1 (offsets `sn`), so when a tile starts its k-tile 0 and half of its k-tile 1 are already on their way.
    unsigned sc[4][2], sn[4][2];     // [half-tile][piece]: byte offset of this lane's 16 B at k = 0, current / next tile of this workgroup
    auto tile_offsets = [&](unsigned (&so)[4][2], int vb) {
        int mb, nb;
        tile_of_block(vb, nbm, nbn, &mb, &nb);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int lr = wave * 16 + i * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((lr >> 1) & 7);
                if (j < 2) so[j][i] = (unsigned)(((long)(nb * 256 + j * 128 + lr) * g.K + c * 8) * (long)sizeof(T));
                else {
                    long m = (long)mb * 256 + (j - 2) * 128 + lr;
                    if (m > g.M - 1) m = g.M - 1;
                    so[j][i] = (unsigned)((row_off(m, g.a_rows_per_batch, g.a_batch_stride, g.lda) + c * 8) * (long)sizeof(T));
                }
            }
    };
    bool has_next = false;           // this workgroup has a tile after the current one
    int gkt = 0;                     // running k-tile count of this workgroup: k-tile G lives in LDS buffer G & 1
#define P8_STAGE(J, KTS)                                                                                              \
    {                                                                                                                \
        const int kts_ = (KTS);                                                                                      \
        const bool nxt_ = kts_ >= nk;                                                                                \
        if ((!nxt_ || has_next) && !(g.ko & 1)) {                                                                    \
            const int kk_ = nxt_ ? kts_ - nk : kts_;                                                                 \
            char* dst_ = smem + ((((gkt + kts_ - kt_cur) & 1) * 4 + (J)) * kP8Half) + wave * (16 * 128);             \
            const char* src_ = ((J) < 2 ? Wb : Ab) + (long)kk_ * (64 * sizeof(T));                                   \
            glds16<T>((const T*)(src_ + (nxt_ ? sn[J][0] : sc[J][0])), dst_);                                        \
            glds16<T>((const T*)(src_ + (nxt_ ? sn[J][1] : sc[J][1])), dst_ + 8 * 128);                              \
        }                                                                                                            \
    }

    // ---- fragment addressing ----------------------------------------------------------------------------------------
    // row r of a half-tile, k-chunk (4 kk + fg): byte r * 128 + ((4 kk + fg) ^ ((r >> 1) & 7)) * 16; r = 16 x + frow, so the swizzle is frow >> 1
    const int fo0 = frow * 128 + (((0 + fg) ^ (frow >> 1)) * 16), fo1 = frow * 128 + (((4 + fg) ^ (frow >> 1)) * 16);
    const int xbase = (2 + wm) * kP8Half;                             // this wave's X half-tile
    const int wbase = (wn >> 1) * kP8Half + (wn & 1) * (64 * 128);    // its 64 W rows inside W half-tile wn >> 1

    // prologue of the stream for this workgroup's first tile: k-tile 0 and the W half of k-tile 1; k-tile 0 visible to every wave
    if ((int)blockIdx.x < n_tiles) {
        tile_offsets(sn, blockIdx.x);
        has_next = true;             // "the next tile" is the first one until the loop below adopts it
        const int kt_cur = nk;       // so that (gkt + kts - kt_cur) = 0, 1: the first tile's k-tiles 0 and 1
        P8_STAGE(0, nk) P8_STAGE(1, nk) P8_STAGE(2, nk) P8_STAGE(3, nk) P8_STAGE(0, nk + 1) P8_STAGE(1, nk + 1)
        wait_vmcnt<4>();
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();      // the m = 1 waves run one barrier behind their SIMD partners from here on

    V8 wl[2][2], wr[2][2];          // W fragments of the two column halves; which one holds the left half alternates per k-tile
    for (int vb = blockIdx.x; vb < n_tiles; vb += gridDim.x) {
        int mb, nb;
        tile_of_block(vb, nbm, nbn, &mb, &nb);
        const int m0 = mb * 256, n0 = nb * 256;
#pragma unroll
        for (int j = 0; j < 4; j++) { sc[j][0] = sn[j][0]; sc[j][1] = sn[j][1]; }
        has_next = vb + (int)gridDim.x < n_tiles;
        if (has_next) tile_offsets(sn, vb + gridDim.x);
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        V8 xa[4][2];
#define P8_MMA(WF, NI0, MI0)                                                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 2; kk++)          /* k outermost: 8 independent accumulators between two uses of one */ \
        _Pragma("unroll") for (int ni = 0; ni < 2; ni++)                                                             \
        _Pragma("unroll") for (int mi = 0; mi < 4; mi++) {                                                           \
            if (SWAP) acc[NI0 + ni][MI0 + mi] = Mfma<T>::mma(xa[mi][kk], WF[ni][kk], acc[NI0 + ni][MI0 + mi]);       \
            else acc[NI0 + ni][MI0 + mi] = Mfma<T>::mma(WF[ni][kk], xa[mi][kk], acc[NI0 + ni][MI0 + mi]);            \
        }
#define P8_SYNC_MMA(WF, NI0, MI0)                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (!(g.ko & 16)) __builtin_amdgcn_s_setprio(1);                                                             \
        if (!(g.ko & 4)) { P8_MMA(WF, NI0, MI0) }                                                                    \
        if (!(g.ko & 16)) __builtin_amdgcn_s_setprio(0);                                                             \
        __builtin_amdgcn_sched_barrier(0);
        // One k-tile = 4 phases; each phase: ds_reads of the fragments the NEXT cluster needs that are not in registers yet + one half-tile of
        // the staging stream, wait for the reads, barrier, 16-MFMA cluster, barrier.  Reads per phase: 8 / 4 / 8 / 4 -- the W fragments of the
        // left column half of k-tile kt + 1 are fetched in phase 3 of k-tile kt into the registers the right half has just vacated, so the two
        // W register sets swap roles every k-tile (the loop is unrolled by two; nk is even, checked at launch).
#define P8_WAIT(N_FULL, N_END)                                                                                      \
        { if (more_) wait_vmcnt<N_FULL>(); else wait_vmcnt<N_END>(); }
#define P8_KTILE(WL, WR, KT)                                                                                        \
        {                                                                                                            \
            const int kt_cur = (KT);                                                                                 \
            const char* xb = smem + (gkt & 1) * (4 * kP8Half) + xbase;                                               \
            const char* wb = smem + (gkt & 1) * (4 * kP8Half) + wbase;                                               \
            const char* wbn = smem + ((gkt + 1) & 1) * (4 * kP8Half) + wbase;                                        \
            const bool need_ = kt_cur + 1 < nk || has_next, more_ = kt_cur + 2 < nk || has_next;                     \
            /* phase 0: (rows 0-63, cols 0-31) */                                                                    \
            if (!(g.ko & 2)) _Pragma("unroll") for (int mi = 0; mi < 4; mi++) { xa[mi][0] = *(const V8*)(xb + mi * 16 * 128 + fo0); xa[mi][1] = *(const V8*)(xb + mi * 16 * 128 + fo1); } \
            P8_STAGE(2, kt_cur + 1)                                                                                  \
            P8_SYNC_MMA(WL, 0, 0)                                                                                    \
            if (!(g.ko & 8)) __builtin_amdgcn_s_barrier();                                                            \
            /* phase 1: (rows 0-63, cols 32-63) */                                                                   \
            if (!(g.ko & 2)) _Pragma("unroll") for (int ni = 0; ni < 2; ni++) { WR[ni][0] = *(const V8*)(wb + (2 + ni) * 16 * 128 + fo0); WR[ni][1] = *(const V8*)(wb + (2 + ni) * 16 * 128 + fo1); } \
            P8_STAGE(3, kt_cur + 1)                                                                                  \
            P8_SYNC_MMA(WR, 2, 0)                                                                                    \
            if (!(g.ko & 8)) __builtin_amdgcn_s_barrier();                                                            \
            /* phase 2: (rows 64-127, cols 32-63); the W half-tiles of the next k-tile must be visible after its closing barrier */ \
            if (!(g.ko & 2)) _Pragma("unroll") for (int mi = 0; mi < 4; mi++) { xa[mi][0] = *(const V8*)(xb + (4 + mi) * 16 * 128 + fo0); xa[mi][1] = *(const V8*)(xb + (4 + mi) * 16 * 128 + fo1); } \
            P8_STAGE(0, kt_cur + 2)                                                                                  \
            if (wm == 1 && need_) P8_WAIT(6, 4)                                                                      \
            P8_SYNC_MMA(WR, 2, 4)                                                                                    \
            if (wm == 0 && need_) P8_WAIT(6, 4)                                                                      \
            if (!(g.ko & 8)) __builtin_amdgcn_s_barrier();                                                            \
            /* phase 3: (rows 64-127, cols 0-31); W left half of the next k-tile into the registers of the right half; its X half-tiles visible */ \
            if (need_ && !(g.ko & 2)) { _Pragma("unroll") for (int ni = 0; ni < 2; ni++) { WR[ni][0] = *(const V8*)(wbn + ni * 16 * 128 + fo0); WR[ni][1] = *(const V8*)(wbn + ni * 16 * 128 + fo1); } } \
            P8_STAGE(1, kt_cur + 2)                                                                                  \
            if (wm == 1 && need_) P8_WAIT(4, 0)                                                                      \
            P8_SYNC_MMA(WL, 0, 4)                                                                                    \
            if (wm == 0 && need_) P8_WAIT(4, 0)                                                                      \
            __builtin_amdgcn_s_barrier();                                                                            \
            gkt++;                                                                                                   \
        }
        if (vb == (int)blockIdx.x) {   // first tile: the left W half of k-tile 0 (later tiles got it in phase 3 of the previous tile's last k-tile)
            const char* wb0 = smem + (gkt & 1) * (4 * kP8Half) + wbase;
#pragma unroll
            for (int ni = 0; ni < 2; ni++) { wl[ni][0] = *(const V8*)(wb0 + ni * 16 * 128 + fo0); wl[ni][1] = *(const V8*)(wb0 + ni * 16 * 128 + fo1); }
        }
        for (int kt = 0; kt < nk; kt += 2) {
            P8_KTILE(wl, wr, kt)
            P8_KTILE(wr, wl, kt + 1)
        }
#undef P8_KTILE
#undef P8_WAIT
#undef P8_SYNC_MMA
#undef P8_MMA
        f32x4 bias_v[4];
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            bias_v[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g.bias) {
                if constexpr (SWAP) bias_v[ni][0] = g.bias[n0 + wn * 64 + ni * 16 + frow];
                else bias_v[ni] = *(const f32x4*)(g.bias + n0 + wn * 64 + ni * 16 + fg * 4);
            }
            asm volatile("" : "+v"(bias_v[ni][0]), "+v"(bias_v[ni][1]), "+v"(bias_v[ni][2]), "+v"(bias_v[ni][3]));
        }
        epilogue256<T, KIND, ST16>(g, acc, bias_v, m0, n0, wm, wn, frow, fg);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();      // matches the extra barrier of the m = 1 waves
#undef P8_STAGE
}

