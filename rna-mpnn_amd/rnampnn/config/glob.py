"""Module constants of the hot path.

Same names and values as the reference's ``rnampnn/config/glob.py:11-21`` (they are
inputs of the arithmetic: atom count, class count, hidden width, the two epsilons and
the nucleotide vocabulary).  Paths of the reference's competition I/O are not part of
the hot path and are not mirrored.
"""

NUM_MAIN_SEQ_ATOMS = 7
NUM_RES_TYPES = 4
DEFAULT_HIDDEN_DIM = 128

LEPS = 1e6
SEPS = 1e-6
DEFAULT_SEED = 42

VOCAB = {'A': 0, 'U': 1, 'C': 2, 'G': 3}
REVERSE_VOCAB = {0: 'A', 1: 'U', 2: 'C', 3: 'G'}

# Feature widths that follow from 7 / 6 / 6 atoms (reference feature.py:181-182).
RAW_NODE_DIM = 28   # 21 intra-residue distances + 4 bond-angle cosines + 3 dihedral cosines
RAW_EDGE_DIM = 90   # 49 cross distances + 25 bond-vector cosines + 16 plane-normal cosines
