"""ddmp-hip: MI355X-native dual-GCN mesh-denoising training step.

Host-side mirror of the reference's operator interface for the hot path
(``util/networks.py``, ``util/loss.py``, ``main.py:88-110`` of astaka-pe/Dual-DMP)
over the C-ABI HIP library ``csrc/libddmp_hip.so`` (see ``include/ddmp_hip.h``).
"""
__version__ = "0.1.0"
