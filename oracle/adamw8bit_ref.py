"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of AdamW with block-wise 8-bit moments, the optimiser the reference selects with
optim="adamw_bnb_8bit" (/root/reference/mm/forget.py:156).  The implementation lives in the third-party package
bitsandbytes==0.43.3 (req.txt), which is NOT in /root/reference and not installed here: this follows the published
algorithm (Dettmers et al., "8-bit Optimizers via Block-wise Quantization", ICLR 2022; bitsandbytes
`create_dynamic_map`, block size 2048, signed map for the first moment, unsigned for the second, nearest-code
quantisation, one fp32 absmax per block).  PARITY UNPINNED against bitsandbytes itself: the reference holds no test,
fixture or golden output for its optimiser.  tests/ use it to check the HIP kernel against the same published rules.
"""
from __future__ import annotations

import math

import torch

BLOCK = 2048


def dynamic_map(signed: bool, max_exponent_bits: int = 7, total_bits: int = 8) -> torch.Tensor:
    non_sign_bits = total_bits - 1
    data = []
    for i in range(max_exponent_bits):
        items = (2 ** (i + non_sign_bits - max_exponent_bits) + 1) if signed else (2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1)
        b = torch.linspace(0.1, 1, items)
        means = (b[:-1] + b[1:]) / 2.0
        vals = (10.0 ** (-(max_exponent_bits - 1) + i)) * means
        data += vals.tolist()
        if signed:
            data += (-vals).tolist()
    data += [0.0, 1.0]
    assert len(data) == 2 ** total_bits
    return torch.tensor(sorted(data), dtype=torch.float32)


def quantize(map_: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """index of the nearest code (ties to the upper code)"""
    mids = 0.5 * (map_[:-1] + map_[1:])
    return torch.bucketize(x, mids, right=True).to(torch.uint8)


def adamw8bit_step(p, g, c1, c2, am1, am2, map1, map2, lr, beta1, beta2, eps, wd, step, clip_scale=1.0):
    """One update in fp32 on CPU tensors: p (bf16), g (bf16), codes (uint8), absmax (fp32 per block).  Returns the new
    (p, c1, c2, am1, am2)."""
    n = p.numel()
    nb = (n + BLOCK - 1) // BLOCK
    pad = nb * BLOCK - n
    gf = torch.nn.functional.pad(g.float() * clip_scale, (0, pad)).view(nb, BLOCK)
    m = map1[torch.nn.functional.pad(c1.long(), (0, pad), value=int((map1 == 0).nonzero()[0]))].view(nb, BLOCK) * am1[:, None]
    v = map2[torch.nn.functional.pad(c2.long(), (0, pad), value=int((map2 == 0).nonzero()[0]))].view(nb, BLOCK) * am2[:, None]
    m = beta1 * m + (1 - beta1) * gf
    v = beta2 * v + (1 - beta2) * gf * gf
    nm1, nm2 = m.abs().amax(dim=1), v.amax(dim=1)
    bc1 = 1 - beta1 ** step
    bc2s = math.sqrt(1 - beta2 ** step)
    pf = torch.nn.functional.pad(p.float(), (0, pad)).view(nb, BLOCK) * (1 - lr * wd)
    pf = pf - (lr / bc1) * m / (v.sqrt() / bc2s + eps)
    r1 = torch.where(nm1 > 0, 1 / nm1, torch.zeros_like(nm1))
    r2 = torch.where(nm2 > 0, 1 / nm2, torch.zeros_like(nm2))
    nc1 = quantize(map1, (m * r1[:, None]).reshape(-1))[:n]
    nc2 = quantize(map2, (v * r2[:, None]).reshape(-1))[:n]
    return pf.reshape(-1)[:n].to(torch.bfloat16), nc1, nc2, nm1, nm2
