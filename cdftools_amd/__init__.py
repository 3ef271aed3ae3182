"""cdftools_amd -- MI355X-native hot path of CDFTOOLS (cdfmoc / cdfmhst / cdfvT / cdfheatc /
cdfpsi / cdfcurl / sigma0-sigmai) behind a C ABI (include/cdftools_hip.h).

Only what the path needs lives here: csrc/ (hand-written HIP for gfx950 + the C ABI), lib/ (the
built shared object), fortran/ (the drop-in host programs and the cdfio / modcdfnames surface) and
hip.py (the same calls for Python tests and bench.py).
"""
from ._lib import CdftError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
