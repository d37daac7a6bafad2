"""Matrix bijections on the hot path, by the reference's spelling (bijections/finite/matrix/__init__.py:4): the two
permutations every preset is built from.  (Identity / LU / QR / Householder / triangular matrices are out of scope.)"""
from torchflows_amd.bijections.finite.matrix.permutation import (  # noqa: F401
    ReversePermutationMatrix, RandomPermutationMatrix)
