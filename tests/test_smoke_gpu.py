"""The driver's round-end smoke call must keep working: run it as a GPU test."""
import pytest

pytestmark = pytest.mark.gpu


def test_graft_entry_smoke():
    import __graft_entry__ as g
    g.smoke()


def test_lse_and_dmask_are_not_differentiable():
    import torch
    import flash_attn
    q = torch.randn(1, 64, 2, 64, device="cuda", dtype=torch.float16, requires_grad=True)
    out, lse, dmask = flash_attn.flash_attn_func(q, q, q, causal=True, return_attn_probs=True)
    assert out.requires_grad and not lse.requires_grad and not dmask.requires_grad
    lse.cpu().numpy()          # usable without detach(), like the reference's plain tensors
