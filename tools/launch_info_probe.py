import sys, os
sys.path[:0] = ['/root/repo/polar-codes_amd', '/root/repo/tests', '/root/repo']
import numpy as np, pypolar_amd
from pypolar_amd.construction import frozen_bits
fz = np.asarray(frozen_bits(4096, 2048, 0.0, "BB"), dtype=np.uint32)
d = pypolar_amd.PolarDecoder(4096, 32, fz, "float"); d.setErrorDetection(32, "crc")
llr = np.random.default_rng(1).normal(1, 1, (64, 4096)).astype(np.float32)
d.decode_batch(llr)
print(os.environ.get("PYPOLAR_AMD_LIB"), d.launch_info(True), d.program_stats(True))
