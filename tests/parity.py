"""Shared helpers of the GPU parity tests: reduced-precision reference inputs and per-tensor error measures."""
import torch

LOW = {"bf16": torch.bfloat16, "fp16": torch.float16}


def round_to(t: torch.Tensor, dtype: str) -> torch.Tensor:
    if dtype == "fp32":
        return t
    if t.is_complex():
        return t
    return t.to(LOW[dtype]).to(torch.float32)


def operands_as_device_sees_them(P: dict, dtype: str) -> dict:
    """The 16-bit paths keep fp32 master parameters and round the CONV WEIGHTS (3-d `.weight` tensors) to the compute
    dtype when packing MFMA operands; biases, BatchNorm parameters and `z` stay fp32.  Feeding the oracle the same
    rounded weights removes the operand rounding from the comparison: what is left is the storage rounding of the
    activations/gradients between kernels."""
    if dtype == "fp32":
        return P
    return {k: (round_to(v, dtype) if (k.endswith(".weight") and v.dim() == 3) else v) for k, v in P.items()}


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = torch.view_as_real(got) if got.is_complex() else got
    ref = torch.view_as_real(ref) if ref.is_complex() else ref
    got, ref = got.double().reshape(-1).cpu(), ref.double().reshape(-1).cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-300))


def temp_grad_terms_norm(logits: torch.Tensor, reduction: str = "mean") -> float:
    """d loss / d temp = sum_ij g_ij * logits_ij with g = d loss / d logits (the logits are exp(temp) * cosine, so
    d logits / d temp = logits): a sum whose diagonal (g_ii < 0) and off-diagonal (g_ij > 0) parts largely cancel.  Returns the
    Frobenius norm of its terms — the scale an error of that sum is to be judged against."""
    lg = logits.double()
    B = lg.shape[0]
    g = torch.softmax(lg, dim=1) + torch.softmax(lg, dim=0) - 2.0 * torch.eye(B, dtype=torch.float64)
    g = g / (2.0 * B) if reduction == "mean" else g / 2.0
    return float((g * lg).norm())
