// witness.hip -- (temporary stubs; replaced by the batched-witness / coset / group-iNTT implementation)
#include "common.h"
using namespace kzg;
extern "C" int kzg_witness_coeff_batched(kzg_ctx *ctx, const kzg_srs *, const void *, size_t, const void *, const void *,
                                         size_t, int, int, void *, int, void *, size_t *) {
    return ctx ? fail(ctx, KZG_ERR_INTERNAL, "kzg_witness_coeff_batched: not implemented yet") : KZG_ERR_SHAPE;
}
extern "C" int kzg_coset_ntt_fr(kzg_ctx *ctx, void *, uint32_t, int, int, int) {
    return ctx ? fail(ctx, KZG_ERR_INTERNAL, "kzg_coset_ntt_fr: not implemented yet") : KZG_ERR_SHAPE;
}
extern "C" int kzg_srs_lagrange_from_monomial_g1(kzg_ctx *ctx, const kzg_srs *, kzg_srs **) {
    return ctx ? fail(ctx, KZG_ERR_INTERNAL, "kzg_srs_lagrange_from_monomial_g1: not implemented yet") : KZG_ERR_SHAPE;
}
