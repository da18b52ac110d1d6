"""CPU oracle for the multilevel-GNN hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch on the CPU, the arithmetic the
reference executes on its forward/backward hot path.
It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``multilevel-gnn_amd/`` (the product) imports it, and the product has no CPU
fallback: it raises when the HIP extension is missing.

Pinning status (also recorded in DESIGN.md):

* Formulas authored *inside* the reference tree (message, aggregator
  composition, MsgNorm, MLP layout, residual wiring, SAGE message/update,
  projection pooling, conv head, feature loss, DiffPool wiring) are pinned by
  golden vectors generated from the reference's own classes
  (``tests/golden/make_golden.py``).
* The third-party primitives those classes call (torch_geometric 2.2.0 /
  torch_scatter 2.1.0: ``scatter``, ``scatter_softmax``, ``degree``,
  ``add/remove_self_loops``, ``global_*_pool``, ``DenseSAGEConv``,
  ``dense_diff_pool``, ``MessagePassing.propagate``) are absent from the
  container and from the reference tree; the reference holds no test that pins
  them.  For those primitives: **parity unpinned** -- they follow the
  published algorithm of the pinned versions, checked against hand-computed
  cases in ``tests/test_oracle_primitives.py``.
"""
