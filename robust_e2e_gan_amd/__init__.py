"""MI355X-native joint adversarial-enhancement + attention-ASR training path.

Mirror of the reference's module surface (model/*, data/* collate, options/*, utils/*) for the
``joint_train.py`` hot path; all arithmetic runs in hand-written HIP kernels (csrc/) behind the
C ABI of include/re2e.h.  There is no CPU fallback: ops raise if libre2e_hip.so is missing.
"""
__version__ = '0.1.0'
