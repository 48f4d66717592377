"""CPU oracle for the Segmentation_Factory hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-fp32 / numpy restatement of the reference's
forward/backward path (models/backbones + models/heads + engine.criterion +
util/metrics), written as pure functions over a ``state_dict``.  It exists so
that the HIP product path in ``segmentation_factory_amd`` can be checked
against something that runs on any CPU.

Rules (see DESIGN.md "Oracle"):
  * Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here.  The product package
    never does; it fails loudly when its HIP library is missing.
  * Parity pin: every function here is checked against the *imported
    reference* (``oracle/ref_shim.py``, this container only) by
    ``oracle/make_goldens.py``; the reference's outputs are committed as
    ``tests/golden/*.npz`` and re-checked by ``tests/test_oracle_golden.py``
    on every box.  The reference itself ships no tests / golden vectors for
    this path, so those captured outputs are the pin.
  * Unpinned: timm 0.9.2 optimizer / AGC arithmetic (timm is not installed
    here and is outside the fwd+bwd metric) -- see ``oracle/optim.py``.
"""
