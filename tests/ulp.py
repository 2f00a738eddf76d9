"""bf16 ulp arithmetic for the flat-bound parity tests.

bf16 keeps 8 significant bits, so for 2^e <= |x| < 2^(e+1) one unit in the last place is 2^(e-7).  A correctly rounded
bf16 result is within 0.5 ulp of the exact value; two correct bf16 evaluations that sum in different orders differ by at
most 1 ulp.  The bounds below are FLAT: they do not scale with the noise band of a deep random network, so a kernel that
reads a wrong key, skips a k-step or rounds at the wrong place cannot pass them.
"""
import torch


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """ulp of bf16 at |x| (fp32 / fp64 tensor in, same dtype out)."""
    ax = x.abs().to(torch.float64).clamp_min(2.0 ** -126)
    e = torch.floor(torch.log2(ax))
    return torch.pow(torch.tensor(2.0, dtype=torch.float64, device=x.device), e - 7).to(x.dtype if x.dtype.is_floating_point else torch.float32)


def ulp_error(got: torch.Tensor, want: torch.Tensor, floor: float = 0.0, slack: torch.Tensor = None) -> torch.Tensor:
    """|got - want| in bf16 ulps of max(|want|, floor), after removing `slack` (an absolute allowance, e.g. the fp32
    accumulation error bound 1e-5 * sum|x||w| of a dot product).  fp64 throughout."""
    g, w = got.to(torch.float64), want.to(torch.float64)
    d = (g - w).abs()
    if slack is not None:
        d = (d - slack.to(torch.float64)).clamp_min(0.0)
    ref = w.abs()
    if floor > 0:
        ref = ref.clamp_min(floor)
    return d / bf16_ulp(ref)


def rms(x: torch.Tensor) -> float:
    return float(x.to(torch.float64).pow(2).mean().sqrt())


def report(name: str, err: torch.Tensor, got: torch.Tensor = None, want: torch.Tensor = None) -> str:
    s = f"{name}: max {err.max().item():.3f} ulp, p99.9 {err.flatten().to(torch.float32).kthvalue(max(1, int(err.numel() * 0.999))).values.item():.3f}"
    if got is not None and want is not None and got.dtype == want.dtype:
        s += f", bit-equal {(got == want).float().mean().item() * 100:.2f}%"
    return s
