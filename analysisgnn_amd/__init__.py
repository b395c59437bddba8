"""analysisgnn_amd — MI355X-native (gfx950) hot path of manoskary/analysisgnn.

Only the heterogeneous message-passing encoder path is here: hand-written HIP
kernels behind a C-ABI (`include/agnn.h`, `analysisgnn_amd/csrc/`), and the Python host side
that mirrors the reference's encoder / operator interface.  There is no CPU fallback: every
compute entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
