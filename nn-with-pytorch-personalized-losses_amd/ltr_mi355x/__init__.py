"""ltr_mi355x -- host runtime of the MI355X-native listwise learning-to-rank path.

Python above a C ABI (include/ltr_mi355x.h -> libltr_mi355x.so, hand-written HIP for gfx950).  PyTorch is
used for device memory, streams, autograd plumbing and torch.distributed (RCCL) only.  There is NO CPU
fallback: every op raises `LtrDeviceError` on non-device tensors and `LtrBuildError` if the HIP library
is missing.  The reference's import surface lives next to this package (`losses/`, `architeture/`).
"""
from ._lib import LtrBuildError, LtrDeviceError, LtrError, lib, library_path  # noqa: F401
