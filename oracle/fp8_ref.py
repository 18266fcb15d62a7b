"""Oracle for the fp8 weight path of BASELINE.json configs[4] (test infrastructure only).

THE REFERENCE HAS NO FP8 PATH ("no reference counterpart for fp8"), so there is nothing
under /root/reference to pin this file against: it restates the engine's own definition of the path in plain torch-CPU
ops, pinned only against torch's `float8_e4m3fn` (OCP e4m3, the gfx950 format) conversion semantics.  Model-level
parity of the fp8 path is therefore reported as agreement with the bf16 path (which IS pinned), see tests/.

Definition (include/pcy.h, pcy_quant_rows_fp8 / pcy_gemm_fp8):
  * per-row symmetric quantisation with a POWER-OF-TWO scale: scale = the smallest 2^e (e >= -126) with amax|row| / 2^e <= 448
    (1 for an all-zero row), q = e4m3_rne(x / scale) -- the division is exact (a multiply on the GPU), and the code of an
    element depends on the rest of its row only through the binade of the row maximum;
  * linear: y = bf16( ((q_x . q_w^T) * s_x[:,None]) * s_w[None,:] ), exact products, wide accumulation;
    weights quantised once per output channel, activations per token at every call.

Sensitivity (tools/diag_fp8.py): an activation that differs by ONE bf16 ulp between two implementations lands in a
neighbouring e4m3 bucket with probability ~1/16 and then moves by 6-12 % of its value, so end-to-end agreement of two
exact-class fp8 pipelines is far looser than for bf16: one small Llama layer, GPU vs this oracle: median row 3e-3, worst
rows 3e-2 (a dominant SwiGLU activation re-bucketed), 1.3e-2 overall against an fp8-vs-bf16 distance of 6e-2.  The
kernel-level tests (identical inputs) are strict; model-level bars are stated relative to the quantisation distance.
"""
from __future__ import annotations

import torch

E4M3_MAX = 448.0


def quant_rows(x: torch.Tensor):
    """x [rows,K] (any float dtype) -> (q float8_e4m3fn [rows,K], scale fp32 [rows])."""
    xf = x.to(torch.float32)
    amax = xf.abs().amax(dim=-1)
    scale = pow2_scale(amax)
    q = (xf / scale[:, None]).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q, scale


def pow2_scale(amax: torch.Tensor) -> torch.Tensor:
    """smallest power of two s >= 2^-126 with amax / s <= 448 (fp32, exact); 1 where amax == 0."""
    m, e = torch.frexp(amax)                       # amax = m * 2^e, m in [0.5, 1)
    e0 = (e - 9).clamp_min(-126)                   # 448 = 1.75 * 2^8: amax / 2^(e-9) = 2m * 2^8 <= 448 iff 2m <= 1.75
    s0 = torch.ldexp(torch.ones_like(amax), e0)
    s = torch.where(amax <= E4M3_MAX * s0, s0, 2 * s0)
    return torch.where(amax > 0, s, torch.ones_like(amax))


def dequant(q: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q.to(torch.float32) * scale[:, None]


def linear_q(qx, sx, qw, sw, out_dtype=torch.bfloat16):
    """((qx . qw^T) * sx) * sw, accumulated in fp64 (products of two e4m3 values and their sums are exact there)."""
    acc = (qx.to(torch.float64) @ qw.to(torch.float64).T).to(torch.float32)
    return ((acc * sx[:, None]) * sw[None, :]).to(out_dtype)


def linear_fp8(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Drop-in for F.linear(x, w) on the fp8 path: x [..., K] bf16, w [N, K] bf16 -> [..., N] in x.dtype."""
    lead = x.shape[:-1]
    qx, sx = quant_rows(x.reshape(-1, x.shape[-1]))
    qw, sw = quant_rows(w)
    return linear_q(qx, sx, qw, sw, x.dtype).reshape(*lead, w.shape[0])
