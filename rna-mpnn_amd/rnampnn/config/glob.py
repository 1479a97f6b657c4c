"""Constants the hot path consumes, under the names the reference exports from
``rnampnn/config/glob.py:11-21`` (atom count, class count, hidden width, the two epsilons, the
nucleotide vocabulary).  The reference's data / competition paths are not part of the path."""

_NUCLEOTIDES = "AUCG"                      # class id = position in this string

VOCAB = {nt: i for i, nt in enumerate(_NUCLEOTIDES)}
REVERSE_VOCAB = dict(enumerate(_NUCLEOTIDES))
NUM_RES_TYPES = len(_NUCLEOTIDES)

NUM_MAIN_SEQ_ATOMS = 7                     # backbone atoms kept per residue; coords are (L, 7, 3)
DEFAULT_HIDDEN_DIM = 128                   # node / edge embedding width the kernels are built for
DEFAULT_SEED = 42

SEPS = 1e-6                                # added under every square root / to variances
LEPS = 1e6                                 # "infinite" distance of padded / invalid pairs

RAW_NODE_DIM = 28    # 21 intra-residue distances + 4 bond-angle cosines + 3 dihedral cosines (feature.py:181)
RAW_EDGE_DIM = 90    # 49 cross distances + 25 bond-vector cosines + 16 plane-normal cosines (feature.py:182)
