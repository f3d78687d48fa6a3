"""One-off check of 64-bit addressing: encoder and puncturer on buffers larger than 4 GiB (device resident), rows repeating
a 2^20-row pattern so that the last block must equal the first.   usage: python tools/large_buffer_check.py"""
import os
import sys

import torch

_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, "polar-codes_amd")]
import pypolar_amd
from pypolar_amd.construction import frozen_bits

dev = torch.device("cuda", 0)
F = frozen_bits(1024, 512, 0.0, "BB")
g = torch.Generator(device=dev)
g.manual_seed(1)
P = 1 << 20
# encoder: 2^25 + 2^20 rows -> code buffer 4.1 GiB
reps = 33
pat = torch.randint(0, 256, (P, 64), dtype=torch.uint8, device=dev, generator=g)
info = pat.repeat(reps, 1).contiguous()
enc = pypolar_amd.PolarEncoder(1024, F)
enc.setErrorDetection(24, "crcnr")
code = enc.encode_batch(info)
torch.cuda.synchronize()
assert code.numel() > (1 << 32)
assert torch.equal(code[:P], code[-P:]) and torch.equal(info[:P], info[-P:])
assert torch.equal(code[:P], code[16 * P:17 * P])
print("encoder ok:", code.shape, "bytes", code.numel())
del code, info
# puncturer: 2^21 + 2^19 rows of 1024 floats -> 10 GiB
punc = pypolar_amd.Puncturer(864, F)
Q = 1 << 19
patf = torch.randn((Q, 864), dtype=torch.float32, device=dev, generator=g)
x = patf.repeat(5, 1).contiguous()
y = punc.depuncture_batch(x)
torch.cuda.synchronize()
assert y.numel() * 4 > (1 << 32)
assert torch.equal(y[:Q], y[-Q:])
z = punc.puncture_batch(y)
assert torch.equal(z, x)
print("puncturer ok:", y.shape)
