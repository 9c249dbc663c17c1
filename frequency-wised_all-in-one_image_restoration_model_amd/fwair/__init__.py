"""fwair: host side of the MI355X-native AirNet training hot path (PyTorch-ROCm tensors + ctypes calls
into libfwair_hip.so).  No CPU fallback exists: every op raises if the HIP library is missing."""
