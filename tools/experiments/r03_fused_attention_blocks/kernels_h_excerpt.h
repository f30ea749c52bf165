// 16..64-row passes (kernels_decode_fused.hip): the attention blocks with their row-local projections folded in, one workgroup per (row, head).
// Per-head partials live in [H][kPartRows][d] f32 buffers and are summed in head order by the consumer (n_parts = H).
template <typename T>
void launch_dec_self_attention_wo(const T* q, const T* kcache, const T* vcache, long slot_stride, int d, int H, const RowCtl* ctl, int M, const T* Wo, float* part,
                                  hipStream_t st);
struct DecCrossFusedDesc {
    const float* x_in; float* x_out;           // residual stream rows [M][d]; x_out = x_in + bias_prev + sum_p parts[p]
    const float* parts; int n_parts;           // [n_parts][kPartRows][d]: per-head partials of the self-attention out-projection
    const float* bias_prev;                    // [d]
    const float* ln_w; const float* ln_b;      // cross_attn_ln
    const void* Wq; const float* bq; float qscale;   // cross query projection T [d][d], bias, dh^-1/4
    const void* kc; const void* vc; long b_stride;   // this layer's cross K / V: T [window][h][Tn][64] (b_stride between windows)
    const RowCtl* ctl;
    const void* Wo;                            // cross out-projection T [d][d]
    float* part_out;                           // [H][kPartRows][d]
    int M, H, d, Tn;
};
template <typename T> void launch_dec_cross_fused(const DecCrossFusedDesc& g, hipStream_t st);
