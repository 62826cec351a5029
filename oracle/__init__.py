"""CPU oracle for the dilated-CNN patch path of keillernogueira/dynamic-rs-segmentation.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it, and there only as the checker / the timed CPU baseline.  The product
path (``dynamic-rs-segmentation_amd``) never imports this package and fails loudly
when its HIP library is missing.

Pinning status
--------------
* Host half (sampling, crop/augment, normalise, sliding-window tiling, confusion
  matrix, patch-size scoring): pinned.  ``oracle/host_ref.py`` is checked against
  fixtures in ``tests/golden/`` that were produced by importing the reference's own
  numpy helpers in the build container (``tests/golden/make_goldens.py``).
* Graph half (dilated conv, batch-norm, (leaky) ReLU, max-pool, classifier,
  softmax-CE + L2, momentum): **parity unpinned**.  The arithmetic lives in
  TensorFlow 1.x, which the reference neither vendors nor pins and which is not
  installable here; the reference holds no tests or golden vectors for it.
  ``oracle/tf_ops.py`` restates the published TF 1.x op semantics at the reference's
  call sites (cited per function); ``oracle/torch_ref.py`` is an independent second
  implementation on PyTorch-CPU autograd, and the two are required to agree
  (``tests/test_oracle_graph.py``).
"""
