"""One tiny pass of the hot path on cuda:0, checked against the oracle handed in by the caller
(__graft_entry__.smoke / tests).  This module never imports the oracle itself."""
import torch


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


def run(oracle, verbose=True):
    from architeture.doubleLayer import DoubleLayerNet
    from losses.approxNDCG import approxNDCGLoss
    from ltr_mi355x.scorer import FusedRanker
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    torch.manual_seed(2020)
    B, S = 8, 128
    # 1) standalone loss kernel
    s = torch.randn(B, S)
    y = torch.randint(0, 5, (B, S)).float()
    sd = s.to(dev).requires_grad_(True)
    loss = approxNDCGLoss(sd, y.to(dev))
    loss.backward()
    ref_loss, ref_grad, _ = oracle.approx_ndcg_closed_form(s.double(), y.double())
    e1, e2 = _rel(loss.detach().cpu(), ref_loss), _rel(sd.grad.cpu(), ref_grad)
    if verbose:
        print(f"[smoke] approxNDCG loss kernel  S={S}: loss rel err {e1:.2e}, grad rel err {e2:.2e}")
    assert e1 < 1e-5 and e2 < 1e-5, (e1, e2)
    # 2) the fused slate pipeline: DoubleLayerNet forward + approxNDCG + backward in one launch
    net = DoubleLayerNet(136)
    ref_p = {k: v.detach().double().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.to(dev).eval()
    x = torch.randn(B, S, 136)
    ranker = FusedRanker(net, loss="approxNDCG")
    lf = ranker.step(x.to(dev), y.to(dev))
    lr = oracle.approx_ndcg(oracle.double_layer_forward(x.double(), ref_p).squeeze(-1), y.double())
    lr.backward()
    e3 = _rel(lf.detach().cpu(), lr.detach())
    top = max(float(v.grad.abs().max()) for v in ref_p.values())
    e4 = max(float((p.grad.cpu().double() - ref_p[k].grad).abs().max()) / top for k, p in net.named_parameters())
    if verbose:
        print(f"[smoke] fused DoubleLayerNet+approxNDCG: loss rel err {e3:.2e}, param-grad max-norm rel err {e4:.2e}")
    assert e3 < 1e-5 and e4 < 1e-5, (e3, e4)
    # 3) TripleLayerNet: its fused step runs l2 . l1 folded into one layer (tripleLayer.py:14-16 has no activation between them)
    from architeture.tripleLayer import TripleLayerNet
    net3 = TripleLayerNet(136)
    ref3 = {k: v.detach().double().clone().requires_grad_(True) for k, v in net3.state_dict().items()}
    net3 = net3.to(dev)
    r3 = FusedRanker(net3, loss="approxNDCG")
    l3 = r3.step(x.to(dev), y.to(dev))
    lr3 = oracle.approx_ndcg(oracle.triple_layer_forward(x.double(), ref3).squeeze(-1), y.double())
    lr3.backward()
    e5 = _rel(l3.detach().cpu(), lr3.detach())
    top3 = max(float(v.grad.abs().max()) for v in ref3.values())
    e6 = max(float((p.grad.cpu().double() - ref3[k].grad).abs().max()) / top3 for k, p in net3.named_parameters())
    if verbose:
        print(f"[smoke] fused TripleLayerNet (folded={r3.fold is not None})+approxNDCG: loss rel err {e5:.2e}, param-grad max-norm rel err {e6:.2e}")
    assert e5 < 1e-5 and e6 < 1e-5, (e5, e6)
    return True
