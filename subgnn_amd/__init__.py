"""subgnn_amd -- MI355X-native (gfx950) implementation of SubGNN's anchor-patch sampling +
three-channel subgraph message-passing hot path, behind the reference's own Python interface
(SubGNN / SG_MPN / anchor_patch_samplers names).  Compute = hand-written HIP kernels in
libsubgnn_hip.so (see include/subgnn_hip.h); there is no CPU fallback."""
__version__ = '0.1.0'
