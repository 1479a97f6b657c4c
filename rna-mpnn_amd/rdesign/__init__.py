"""MI355X-native mirror of the reference's sibling package ``rdesign``: the ``RNAModel`` forward
(k-NN graph, RBF / orientation / dihedral features, ``MPNNLayer`` stack, read-out) behind the reference's module names, computed by
``librnampnn_hip.so`` (C ABI: ``include/rdesign_hip.h``).  PARITY UNPINNED - see ``oracle/rdesign_oracle.py``."""
