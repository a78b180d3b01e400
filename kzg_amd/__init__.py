"""kzg_amd -- MI355X-native (gfx950) engine for the KZG commit/open hot path of proxima-one/kzg.

The product is the C-ABI shared library libkzg_mi355x.so (include/kzg_mi355x.h) built from the
hand-written HIP sources in kzg_amd/csrc/.  This package is the thin host-side mirror of the
reference's prover surface used by tests/ and bench.py.  There is no CPU fallback.
"""
from ._lib import (FR_CANONICAL, FR_MONT, G1_AFFINE_MONT, G1_JACOBIAN_MONT, G1_ZCASH_COMPRESSED,
                   G1_ZCASH_UNCOMPRESSED, G2_AFFINE_MONT, G2_COMPRESSED, G2_JACOBIAN_MONT, G2_UNCOMPRESSED, IN_DEVICE,
                   OUT_DEVICE, SO_PATH, load)
from .api import (DeviceBuffer, DeviceGroup, Engine, EngineError, EvaluationDomain, KZGBatchWitness, KZGError, KZGParams,
                  KZGProver, KZGProverEvalForm, KZGVerifier, KZGVerifierEvalForm, PointNotOnPolynomial, Polynomial,
                  PolynomialDegreeTooLarge, ReferencePanic, ShardedSrs, Srs, SrsG2, compute_lagrange_basis, compute_lagrange_basis_g2, compute_omega,
                  pack_scalars, setup, setup_g2, setup_lagrange, setup_lagrange_g2, setup_shard, splitmix_scalar,
                  unpack_scalars)
