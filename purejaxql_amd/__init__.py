"""purejaxql_amd -- MI355X-native PQN hot path (rollout + Q(lambda) + minibatch update).

Drop-in for the make_train / gymnax-env surface of mttga/purejaxql's
pqn_minatar.py / pqn_gymnax.py.  Compute runs in hand-written HIP kernels
(csrc/, C ABI in include/pqn_hotpath.h); PyTorch-ROCm is device memory, streams
and torch.distributed.  There is no CPU fallback.
"""
__version__ = "0.1.0"
