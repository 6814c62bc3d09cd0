"""CPU oracle for the MoCo-v2 ResNet-50 hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain torch-CPU fp32 (+ numpy fp64 for the loss head)
restatement of the reference's algorithm, written to be read side by side with
the reference files it cites.  It is NOT part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  Nothing under ``passl_amd/`` imports it.

Parity status: **pinned at the PASSL-Python level, unpinned at the Paddle
kernel boundary**.  ``import paddle`` is impossible in the build container, so
the oracle is pinned by executing the reference's own ``moco.py`` /
``contrastive_head.py`` / ``base_neck.py`` sources under a torch-backed
``paddle`` shim (``oracle/ref_runner.py``) and comparing outputs
(``tests/test_oracle_vs_reference.py``; golden vectors in ``tests/golden``).
What remains assumed about Paddle itself is listed in ``oracle/README.md``.
"""
