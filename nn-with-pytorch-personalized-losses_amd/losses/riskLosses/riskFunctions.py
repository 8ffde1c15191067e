"""MI355X drop-in for losses/riskLosses/riskFunctions.py of the reference (zRisk :4-22, geoRisk :25-33).

`mat` is a [queries, systems] effectiveness matrix on the device; column `i` is the system under test
(`i=-1`: the last column).  Each call is ONE HIP launch (csrc/ltr_risk.hip: fp64 accumulation, fixed-order
reductions) that also produces the analytic gradient w.r.t. every matrix entry, so both functions are
differentiable whenever `mat` requires grad; `requires_grad` is accepted for signature compatibility (the
reference only uses it for its CPU `alpha` tensor)."""
from ltr_mi355x import risk as _risk


def zRisk(mat, alpha, requires_grad=False, i=0):
    """sum_q d_q (1 + alpha [d_q < 0]),  d_q = (mat[q,i] - e_q) / sqrt(e_q),  e_q = S_i T_q / N.  0-dim tensor."""
    return _risk.z_risk(mat, alpha, i).reshape(())


def geoRisk(mat, alpha, requires_grad=False, i=0):
    """sqrt(mean_q mat[q,i] * Phi(zRisk(mat, alpha, i) / Q)).  Shape [1], like the reference's broadcast against
    its Normal(tensor([0.]), tensor([1.]))."""
    return _risk.geo_risk(mat, alpha, i)
