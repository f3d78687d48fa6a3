# SC arena variant: first arena stage s0 against throughput (diagnostic build).  usage: bash tools/sc_s0_try.sh
export PYPOLAR_AMD_LIB=$PWD/polar-codes_amd/lib/libpolar_amd_diag.so
cat > /tmp/sc_n.py << 'PY'
import os, sys
R = os.getcwd(); sys.path[:0] = [R, os.path.join(R, 'polar-codes_amd')]
import numpy as np, torch, bench, pypolar_amd
from pypolar_amd.construction import frozen_bits
dev = torch.device("cuda", 0)
for N in (2048, 4096, 8192):
    K = N // 2
    cfg = dict(N=N, K=K, L=1, dsnr=0.0, crc=0, crc_type="crc", ebn0=2.0)
    fz = frozen_bits(N, K, 0.0, "BB")
    B = 1 << (19 if N <= 2048 else 17)
    llr, info = bench.make_batch(cfg, fz, B, dev, seed=5)
    d = pypolar_amd.PolarDecoder(N, 1, fz, "float"); d.setErrorDetection(0, "crc")
    out, ok = d.decode_batch(llr)
    match = float((out == info).all(dim=1).float().mean())
    d.set_timing(True)
    for _ in range(4): d.decode_batch(llr)
    torch.cuda.synchronize()
    ms = float(np.mean(d.kernel_times_ms()[-4:]))
    print(f"N={N}: {B / ms / 1e3:.1f} M cw/s  match {match:.4f}", flush=True)
PY
for s in default 7 8 9 10; do
  echo "s0=$s"
  if [ $s = default ]; then python3 /tmp/sc_n.py 2>/dev/null | grep "N="; else PCD_SC_S0=$s python3 /tmp/sc_n.py 2>/dev/null | grep "N="; fi
done
