"""One tiny pass of the hot path on cuda:0, checked against the oracle handed in by the caller
(__graft_entry__.smoke / tests).  This module never imports the oracle itself."""
import torch


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


def run(oracle, verbose=True):
    from losses.approxNDCG import approxNDCGLoss
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    torch.manual_seed(2020)
    B, S = 8, 128
    s = torch.randn(B, S)
    y = torch.randint(0, 5, (B, S)).float()
    sd = s.to(dev).requires_grad_(True)
    loss = approxNDCGLoss(sd, y.to(dev))
    loss.backward()
    ref_loss, ref_grad, _ = oracle.approx_ndcg_closed_form(s.double(), y.double())
    e1, e2 = _rel(loss.cpu(), ref_loss), _rel(sd.grad.cpu(), ref_grad)
    if verbose:
        print(f"[smoke] approxNDCG S={S}: loss rel err {e1:.2e}, grad rel err {e2:.2e}")
    assert e1 < 1e-5 and e2 < 1e-5, (e1, e2)
    try:
        from ltr_mi355x import fused
    except ImportError:
        fused = None
    if fused is not None and hasattr(fused, "smoke"):
        fused.smoke(oracle, verbose)
    return True
