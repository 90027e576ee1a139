"""openairinterface5g_amd -- MI355X (gfx950) implementation of OAI's NR LDPC coding hot path.

Product = csrc/ (HIP kernels + C ABI, built into lib/libldpc_hip.so) with this thin Python host mirror
of the reference's plugin interface.  See DESIGN.md / INTEGRATION.md.
"""
from . import ldpc  # noqa: F401
from .ldpc import (LDPCinit, LDPCshutdown, LDPCdecoder, LDPCencoder, decode_batch_host, decode_batch_device,  # noqa: F401
                   encode_batch_host, encode_batch_device, load_library, make_dec_params)

__all__ = ["ldpc", "LDPCinit", "LDPCshutdown", "LDPCdecoder", "LDPCencoder", "decode_batch_host",
           "decode_batch_device", "encode_batch_host", "encode_batch_device", "load_library", "make_dec_params"]
