"""CPU oracle for the CoR2 / ODA hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product path.  It may be imported only
by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- always as the checker / the reported CPU baseline, never as the
thing shipped or measured as the GPU number.  The product package
(``vqa_playground_pytorch_amd``) never imports this package and raises when its
HIP library is missing instead of falling back to anything here.

Parity status: PINNED.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself: ``tests/golden/make_golden.py`` imports ``/root/reference/config/CoR2.py``
and ``config/ODA.py`` in the authoring container, runs them on seeded inputs and
commits the outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function here against those vectors.

Contents
  seeded.py              version-stable parameter / input generator (numpy RandomState)
  reference_faithful.py  torch-CPU restatement that follows the reference's op sequence
                         (per-sample python loops, materialised [B,N,N,D] tensor)
  kernels_np.py          float64 numpy closed forms (forward AND hand-derived backward)
                         of the four kernels the HIP library implements
"""
