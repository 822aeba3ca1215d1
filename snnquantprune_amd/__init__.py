"""snnquantprune_amd -- MI355X-native quantized / pruned SNN forward pass.

Host-side mirror of the reference's module surface (QuantDense, QuantConv,
quantisers, neurons, SpikingBlock, eval_step) on torch-ROCm tensors; all
arithmetic runs in libsnnqp.so (hand-written gfx950 HIP, include/snnqp.h).
"""

__version__ = "0.1.0"
