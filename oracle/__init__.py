"""CPU oracle for the graph-VAE message-passing train step.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (``hgvae_amd``) may
import this directory; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED.  The arithmetic of the reference's hot path lives in
PyTorch Geometric (``torch_geometric``, version not pinned by the reference:
no requirements file, no lock file).  That package is not installed here and
the reference ships no tests, golden vectors or recorded outputs, so this
oracle is a restatement of the published PyG 2.3-era operator algorithms
(``SAGEConv``, ``GATConv``, ``GraphConv``, ``GCNConv``, ``SAGPooling``,
``topk``, ``filter_adj``, ``Batch.from_data_list``) anchored on the
reference's own call sites.  It is pinned by hand-computed known answers,
by an independent dense-adjacency formulation and by float64 gradient
checks (``tests/test_oracle_*.py``), not by an executed PyG.
"""
