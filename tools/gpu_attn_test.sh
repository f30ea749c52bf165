mkdir -p gpurun_out
./tools/attn_test.bin /tmp/a_new.bin; SS_ATTN_LDS=0 ./tools/attn_test.bin /tmp/a_old.bin
python - <<'PY'
import numpy as np
a=np.fromfile('/tmp/a_new.bin',np.float16).astype(np.float32); b=np.fromfile('/tmp/a_old.bin',np.float16).astype(np.float32)
print("new vs old kernel: n", a.size, "max abs diff", np.abs(a-b).max(), "mean abs", np.abs(a-b).mean(), "nan", np.isnan(a).sum(), np.isnan(b).sum())
cfgs=[(1,2,1500),(3,2,1500),(2,20,1500),(1,2,100),(3,6,1471)]
off=0
for B,H,T in cfgs:
    n=B*T*H*64; d=np.abs(a[off:off+n]-b[off:off+n]).reshape(B,T,H*64); print((B,H,T), "max", d.max(), "argmax", np.unravel_index(d.argmax(), d.shape)); off+=n
PY
