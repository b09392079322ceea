"""CPU oracle for the three-stage style-transfer training path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32) restatement of the
arithmetic of the reference's hot path (src/model/*.py and the loss composition in
src/main_{pretrain,warmup,optimize}.py).  It is the *checker* for the HIP product path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it.  The product package ``consistent__style_transfer_amd`` never does.

Pinning: the oracle is pinned against golden vectors generated in the build container by
importing the reference's own ``src/model/*.py`` (``tests/golden/make_golden.py``; fixtures
in ``tests/golden/*.npz``).  The reference ships no tests or fixtures of its own
(SURVEY.md section 4), so those generated vectors are the pin.
"""
