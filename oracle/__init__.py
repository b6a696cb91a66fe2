"""CPU oracle for the joint_train hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A PyTorch-CPU (fp32) restatement of the reference's algorithm for the path that
BASELINE.json names (bliunlpr/Robust_e2e_gan joint_train.py step).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only
as the checker / timed baseline -- never from ``robust_e2e_gan_amd`` (the product fails loudly
without its HIP library instead of falling back to this).

Pinned: ``tests/test_oracle_golden.py`` checks every function here against the vectors in
``tests/golden/*.npz`` which were produced by importing the reference's own modules
(``tests/golden/make_fixtures.py``).  Third-party arithmetic that is absent from the reference
tree -- warp-ctc (``model/e2e_ctc.py:11,30,63``; version unpinned upstream) -- is restated as
the textbook CTC negative log-likelihood (blank 0, sum over batch / B); that boundary has no
reference test, so CTC parity is pinned only through the same restatement ("parity unpinned"
for warp-ctc's behaviour on infeasible alignments).
"""
