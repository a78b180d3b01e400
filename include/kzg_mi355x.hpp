// kzg_mi355x.hpp -- header-only C++ host mirror of the reference's prover surface over the C ABI
// (include/kzg_mi355x.h).  Same names, argument meaning and error behaviour as proxima-one/kzg:
//   setup / KZGParams            src/lib.rs:14-55           Polynomial          src/polynomial.rs:24-165
//   EvaluationDomain             src/ft.rs:17-140           KZGProver           src/coeff_form.rs:37-112
//   KZGProverEvalForm            src/eval_form.rs:39-147    KZGVerifier         src/coeff_form.rs:114-183
// Result<_, KZGError> becomes a thrown kzg::KZGError; a reference panic becomes kzg::ReferencePanic.
// Scalars are 32-byte canonical little-endian (kzg::Scalar), points 96-byte affine Montgomery (kzg::G1Affine).
#pragma once
#include <array>
#include <cstring>
#include <stdexcept>
#include <cstdlib>
#include <string>
#include <vector>

#include "kzg_mi355x.h"

namespace kzg {

struct Scalar {
    std::array<uint8_t, 32> le{};  // canonical, < r
    static Scalar from_u64(uint64_t v) {
        Scalar s;
        for (int i = 0; i < 8; i++) s.le[i] = (uint8_t)(v >> (8 * i));
        return s;
    }
    bool operator==(const Scalar &o) const { return le == o.le; }
    bool is_zero() const {
        for (auto b : le) if (b) return false;
        return true;
    }
};
struct G1Affine {
    std::array<uint8_t, 96> bytes{};  // blst_p1_affine; identity = all zero
    bool operator==(const G1Affine &o) const { return bytes == o.bytes; }
};
using KZGCommitment = G1Affine;  // src/lib.rs:22
using KZGWitness = G1Affine;     // src/lib.rs:24

struct KZGError : std::runtime_error {  // src/lib.rs:26-36
    enum Kind { PointNotOnPolynomial = 1, PolynomialDegreeTooLarge = 2 } kind;
    KZGError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
struct ReferencePanic : std::runtime_error { using std::runtime_error::runtime_error; };
struct EngineError : std::runtime_error { using std::runtime_error::runtime_error; };

class Engine {
  public:
    explicit Engine(int device = 0) {
        if (int rc = kzg_ctx_create(device, &ctx_)) throw EngineError("kzg_ctx_create failed (no CPU fallback): " + std::to_string(rc));
    }
    ~Engine() { kzg_ctx_destroy(ctx_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    kzg_ctx *ctx() const { return ctx_; }
    // the batched pipeline's plan and whether the process' hardware-queue pool narrowed it (kzg_ctx_info)
    std::string info() const {
        char buf[512];
        check(kzg_ctx_info(ctx_, buf, sizeof buf));
        return buf;
    }
    void check(int rc) const {
        if (rc == KZG_OK) return;
        std::string msg = kzg_last_error(ctx_);
        if (rc == KZG_ERR_POINT_NOT_ON_POLY) throw KZGError(KZGError::PointNotOnPolynomial, "point not on polynomial!");
        if (rc == KZG_ERR_DEGREE_TOO_LARGE) throw KZGError(KZGError::PolynomialDegreeTooLarge, "polynomial degree too large");
        if (rc == KZG_ERR_SHAPE) throw ReferencePanic(msg);
        throw EngineError("kzg_mi355x error " + std::to_string(rc) + ": " + msg);
    }

  private:
    kzg_ctx *ctx_ = nullptr;
};

// KZGParams: `gs` (G1 powers) and `hs` (G2 powers, read by the verifier only) live on the GPU.
struct KZGParams {
    const Engine *engine = nullptr;
    kzg_srs *gs = nullptr;
    kzg_srs_g2 *hs = nullptr;
    size_t len() const { return kzg_srs_len(gs); }
    KZGParams() = default;
    KZGParams(const KZGParams &) = delete;
    KZGParams &operator=(const KZGParams &) = delete;
    KZGParams(KZGParams &&o) noexcept : engine(o.engine), gs(o.gs), hs(o.hs) { o.gs = nullptr; o.hs = nullptr; }
    ~KZGParams() {
        if (gs) kzg_srs_free(engine->ctx(), gs);
        if (hs) kzg_srs_g2_free(engine->ctx(), hs);
    }
};

// src/lib.rs:38-55.  The reference builds num_coeffs G2 powers too; only the verifier reads them (hs[0], hs[1] and
// hs[..k+1] for a k-point batched opening), so `g2_len` caps that half (SIZE_MAX = min(num_coeffs, 257)).
inline KZGParams setup(const Engine &e, const Scalar &s, size_t num_coeffs, size_t g2_len = (size_t)-1) {
    KZGParams p;
    p.engine = &e;
    e.check(kzg_srs_setup_g1(e.ctx(), s.le.data(), KZG_FR_CANONICAL_LE_32, num_coeffs, &p.gs));
    if (g2_len == (size_t)-1) g2_len = num_coeffs < 257 ? num_coeffs : 257;
    if (g2_len) e.check(kzg_srs_setup_g2(e.ctx(), s.le.data(), KZG_FR_CANONICAL_LE_32, g2_len, &p.hs));
    return p;
}

struct Polynomial {  // src/polynomial.rs:24-27
    size_t degree = 0;
    std::vector<Scalar> coeffs;
    static size_t compute_degree(const std::vector<Scalar> &c, size_t upper) {  // :94-105
        size_t i = upper;
        while (i > 0 && c[i].is_zero()) i--;
        return i;
    }
    static Polynomial make(std::vector<Scalar> c) {  // Polynomial::new, :83-87
        Polynomial p;
        p.degree = compute_degree(c, c.size() - 1);
        p.coeffs = std::move(c);
        return p;
    }
    static Polynomial new_from_coeffs(std::vector<Scalar> c, size_t degree) {  // :89-92
        Polynomial p;
        p.degree = degree;
        p.coeffs = std::move(c);
        return p;
    }
    size_t num_coeffs() const { return degree + 1; }  // :135-137
    Scalar eval(const Engine &e, const Scalar &x) const {  // :156-165
        Scalar y;
        e.check(kzg_poly_eval(e.ctx(), coeffs.data(), num_coeffs(), x.le.data(), KZG_FR_CANONICAL_LE_32, 0, y.le.data()));
        return y;
    }
};

struct EvaluationDomain {  // src/ft.rs:17-25
    std::vector<Scalar> coeffs;
    size_t d = 0;
    uint32_t exp = 0;
    Scalar omega;
    static EvaluationDomain from_coeffs(std::vector<Scalar> c) {  // :94-109
        EvaluationDomain e;
        int rc = kzg_compute_omega(c.size(), &e.d, &e.exp, e.omega.le.data(), KZG_FR_CANONICAL_LE_32);
        if (rc == KZG_ERR_DEGREE_TOO_LARGE) throw KZGError(KZGError::PolynomialDegreeTooLarge, "polynomial degree too large");
        c.resize(e.d);
        e.coeffs = std::move(c);
        return e;
    }
    size_t len() const { return coeffs.size(); }
    void fft(const Engine &e) { e.check(kzg_ntt_fr(e.ctx(), coeffs.data(), exp, 0, 0)); }   // :111-113
    void ifft(const Engine &e) { e.check(kzg_ntt_fr(e.ctx(), coeffs.data(), exp, 1, 0)); }  // :115-140
    void coset_fft(const Engine &e) { e.check(kzg_coset_ntt_fr(e.ctx(), coeffs.data(), exp, 0, KZG_FR_CANONICAL_LE_32, 0)); }
    void icoset_fft(const Engine &e) { e.check(kzg_coset_ntt_fr(e.ctx(), coeffs.data(), exp, 1, KZG_FR_CANONICAL_LE_32, 0)); }
    Scalar z(const Scalar &tau) const {  // :182-187
        Scalar out;
        if (kzg_domain_z(coeffs.size(), tau.le.data(), KZG_FR_CANONICAL_LE_32, out.le.data())) throw ReferencePanic("z");
        return out;
    }
    void divide_by_z_on_coset(const Engine &e) { e.check(kzg_divide_by_z_on_coset(e.ctx(), coeffs.data(), exp, KZG_FR_CANONICAL_LE_32, 0)); }
    void mul_assign(const Engine &e, const EvaluationDomain &o) {  // :220-244
        if (o.coeffs.size() != coeffs.size()) throw ReferencePanic("assert_eq!(self.coeffs.len(), other.coeffs.len())");
        e.check(kzg_fr_vec_mul(e.ctx(), coeffs.data(), o.coeffs.data(), coeffs.size(), KZG_FR_CANONICAL_LE_32, 0));
    }
    void sub_assign(const Engine &e, const EvaluationDomain &o) {  // :247-271
        if (o.coeffs.size() != coeffs.size()) throw ReferencePanic("assert_eq!(self.coeffs.len(), other.coeffs.len())");
        e.check(kzg_fr_vec_sub(e.ctx(), coeffs.data(), o.coeffs.data(), coeffs.size(), KZG_FR_CANONICAL_LE_32, 0));
    }
};

struct KZGBatchWitness {  // src/coeff_form.rs:12-35
    Polynomial r;
    G1Affine w;
    const G1Affine &elem() const { return w; }
    const Polynomial &polynomial() const { return r; }
};

class KZGProver {  // src/coeff_form.rs:37-112
  public:
    explicit KZGProver(const KZGParams &params) : params_(params), e_(*params.engine) {}
    const KZGParams &parameters() const { return params_; }
    KZGCommitment commit(const Polynomial &p) const {  // :59-64
        G1Affine out;
        e_.check(kzg_commit_coeff(e_.ctx(), params_.gs, p.coeffs.data(), p.num_coeffs(), KZG_FR_CANONICAL_LE_32, 0,
                                  out.bytes.data(), KZG_G1_AFFINE_MONT_96));
        return out;
    }
    KZGWitness create_witness(const Polynomial &p, const Scalar &x, const Scalar &y) const {  // :66-81
        G1Affine out;
        e_.check(kzg_witness_coeff(e_.ctx(), params_.gs, p.coeffs.data(), p.num_coeffs(), x.le.data(), y.le.data(),
                                   KZG_FR_CANONICAL_LE_32, 0, out.bytes.data(), KZG_G1_AFFINE_MONT_96));
        return out;
    }
    // Not a reference method: `xs.size()` calls of create_witness on one polynomial, pipelined on the GPU.  ok[j] == false
    // where the reference would return Err(PointNotOnPolynomial) for opening j.
    std::vector<KZGWitness> create_witness_many(const Polynomial &p, const std::vector<Scalar> &xs, const std::vector<Scalar> &ys,
                                                std::vector<bool> *ok = nullptr) const {
        if (xs.size() != ys.size()) throw ReferencePanic("assert_eq!(xs.len(), ys.len())");
        std::vector<KZGWitness> out(xs.size());
        std::vector<int> status(xs.size() ? xs.size() : 1, 0);
        static_assert(sizeof(KZGWitness) == 96, "KZGWitness is the 96-byte affine encoding");
        e_.check(kzg_witness_coeff_many(e_.ctx(), params_.gs, p.coeffs.data(), p.num_coeffs(), xs.data(), ys.data(), xs.size(),
                                        KZG_FR_CANONICAL_LE_32, 0, out.data(), KZG_G1_AFFINE_MONT_96, status.data()));
        if (ok) {
            ok->assign(xs.size(), true);
            for (size_t j = 0; j < xs.size(); j++) (*ok)[j] = status[j] == 0;
        }
        return out;
    }
    KZGBatchWitness create_witness_batched(const Polynomial &p, const std::vector<Scalar> &xs,
                                           const std::vector<Scalar> &ys) const {  // :83-111
        if (xs.size() != ys.size()) throw ReferencePanic("assert_eq!(xs.len(), ys.len())");
        KZGBatchWitness wit;
        std::vector<Scalar> r(xs.size() < 2 ? 2 : xs.size());
        size_t rlen = 0;
        e_.check(kzg_witness_coeff_batched(e_.ctx(), params_.gs, p.coeffs.data(), p.num_coeffs(), xs.data(), ys.data(),
                                           xs.size(), KZG_FR_CANONICAL_LE_32, 0, wit.w.bytes.data(), KZG_G1_AFFINE_MONT_96,
                                           r.data(), &rlen));
        r.resize(rlen);
        wit.r = Polynomial::new_from_coeffs(std::move(r), rlen - 1);
        return wit;
    }
    bool verify_poly(const KZGCommitment &c, const Polynomial &p) const {  // KZGVerifier::verify_poly, :119-124
        int ok = 0;
        e_.check(kzg_verify_poly_coeff(e_.ctx(), params_.gs, c.bytes.data(), KZG_G1_AFFINE_MONT_96, p.coeffs.data(),
                                       p.num_coeffs(), KZG_FR_CANONICAL_LE_32, 0, &ok));
        return ok != 0;
    }

  private:
    const KZGParams &params_;
    const Engine &e_;
};

class KZGVerifier {  // src/coeff_form.rs:114-183; the pairing checks run on the GPU
  public:
    explicit KZGVerifier(const KZGParams &params) : params_(params), e_(*params.engine) {}
    bool verify_poly(const KZGCommitment &c, const Polynomial &p) const {  // :119-124
        int ok = 0;
        e_.check(kzg_verify_poly_coeff(e_.ctx(), params_.gs, c.bytes.data(), KZG_G1_AFFINE_MONT_96, p.coeffs.data(),
                                       p.num_coeffs(), KZG_FR_CANONICAL_LE_32, 0, &ok));
        return ok != 0;
    }
    bool verify_eval(const Scalar &x, const Scalar &y, const KZGCommitment &c, const KZGWitness &w) const {  // :126-142
        if (!params_.hs) throw ReferencePanic("KZGParams.hs is empty (index out of bounds)");
        uint8_t ok = 0;
        e_.check(kzg_verify_eval(e_.ctx(), params_.gs, params_.hs, x.le.data(), y.le.data(), KZG_FR_CANONICAL_LE_32,
                                 c.bytes.data(), w.bytes.data(), KZG_G1_AFFINE_MONT_96, 1, &ok));
        return ok != 0;
    }
    bool verify_eval_batched(const std::vector<Scalar> &xs, const KZGCommitment &c, const KZGBatchWitness &w) const {  // :144-182
        if (!params_.hs) throw ReferencePanic("KZGParams.hs is empty (index out of bounds)");
        int ok = 0;
        e_.check(kzg_verify_eval_batched(e_.ctx(), params_.gs, params_.hs, xs.data(), xs.size(), w.r.coeffs.data(),
                                         w.r.num_coeffs(), KZG_FR_CANONICAL_LE_32, c.bytes.data(), w.w.bytes.data(),
                                         KZG_G1_AFFINE_MONT_96, &ok));
        return ok != 0;
    }

  private:
    const KZGParams &params_;
    const Engine &e_;
};

// ---- multi-GPU: KZGParams.gs sharded over a group of GPUs, KZGProver::commit / create_witness over the group ----
// (the seam is the multi_exp call, src/coeff_form.rs:61,78; partial points are combined over RCCL inside the library)
// What a ONE-NODE host exports before the first RCCL call of the process (values already exported are kept): RCCL bootstraps every
// communicator over TCP on the first non-loopback interface it finds, which stalls formation for minutes on a host whose interface
// swallows packets; the library bounds formation (KZG_COMM_TIMEOUT_MS / "comm_timeout_ms") but the environment is the host's.
inline void single_node_rccl_env() {
    setenv("NCCL_SOCKET_IFNAME", "lo", 0);
    setenv("NCCL_RAS_ENABLE", "0", 0);
    setenv("NCCL_IB_DISABLE", "1", 0);
    setenv("NCCL_NET_PLUGIN", "none", 0);
}

class DeviceGroup {
  public:
    explicit DeviceGroup(const std::vector<int> &devices) {  // one process drives all GPUs
        single_node_rccl_env();
        if (int rc = kzg_mctx_create(devices.data(), (int)devices.size(), &m_))
            throw EngineError("kzg_mctx_create failed: " + std::to_string(rc) + ": " + kzg_mctx_create_error());
    }
    DeviceGroup(int device, int rank, int world, const void *unique_id) {  // one process per GPU
        single_node_rccl_env();
        if (int rc = kzg_mctx_create_rank(device, rank, world, unique_id, &m_))
            throw EngineError("kzg_mctx_create_rank failed: " + std::to_string(rc) + ": " + kzg_mctx_create_error());
    }
    ~DeviceGroup() { kzg_mctx_destroy(m_); }
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;
    kzg_mctx *handle() const { return m_; }
    int world() const { return kzg_mctx_world(m_); }
    // which RCCL the group runs on, what forming its communicator cost per phase, each local context's pipeline plan (kzg_mctx_info)
    std::string info() const {
        char buf[2048];
        check(kzg_mctx_info(m_, buf, sizeof buf));
        return buf;
    }
    void check(int rc) const {
        if (rc == KZG_OK) return;
        std::string msg = kzg_mctx_last_error(m_);
        if (rc == KZG_ERR_POINT_NOT_ON_POLY) throw KZGError(KZGError::PointNotOnPolynomial, "point not on polynomial!");
        if (rc == KZG_ERR_SHAPE) throw ReferencePanic(msg);
        throw EngineError("kzg_mi355x error " + std::to_string(rc) + ": " + msg);
    }

  private:
    kzg_mctx *m_ = nullptr;
};

struct ShardedParams {  // KZGParams.gs held as one contiguous shard per GPU
    const DeviceGroup *group = nullptr;
    kzg_msrs *gs = nullptr;
    ShardedParams() = default;
    ShardedParams(const ShardedParams &) = delete;
    ShardedParams &operator=(const ShardedParams &) = delete;
    ShardedParams(ShardedParams &&o) noexcept : group(o.group), gs(o.gs) { o.gs = nullptr; }
    ~ShardedParams() { if (gs) kzg_msrs_free(group->handle(), gs); }
    size_t len() const { return kzg_msrs_len(gs); }
};

inline ShardedParams setup_sharded(const DeviceGroup &g, const Scalar &s, size_t num_coeffs) {  // src/lib.rs:38-47
    ShardedParams p;
    p.group = &g;
    g.check(kzg_srs_setup_g1_sharded(g.handle(), s.le.data(), KZG_FR_CANONICAL_LE_32, num_coeffs, &p.gs));
    return p;
}

class ShardedKZGProver {  // KZGProver (src/coeff_form.rs:37-81) with the SRS spread over the group
  public:
    explicit ShardedKZGProver(const ShardedParams &params) : params_(params), g_(*params.group) {}
    KZGCommitment commit(const Polynomial &p) const {  // :59-64
        G1Affine out;
        g_.check(kzg_commit_coeff_sharded(g_.handle(), params_.gs, p.coeffs.data(), p.num_coeffs(), KZG_FR_CANONICAL_LE_32, 0,
                                          out.bytes.data(), KZG_G1_AFFINE_MONT_96));
        return out;
    }
    KZGWitness create_witness(const Polynomial &p, const Scalar &x, const Scalar &y) const {  // :66-81
        G1Affine out;
        g_.check(kzg_witness_coeff_sharded(g_.handle(), params_.gs, p.coeffs.data(), p.num_coeffs(), x.le.data(), y.le.data(),
                                           KZG_FR_CANONICAL_LE_32, 0, out.bytes.data(), KZG_G1_AFFINE_MONT_96));
        return out;
    }
    // :83-111; returns the witness, `r` receives the interpolant (as the reference's (KZGWitness, Polynomial) pair)
    KZGWitness create_witness_batched(const Polynomial &p, const std::vector<Scalar> &xs, const std::vector<Scalar> &ys,
                                      Polynomial *r) const {
        G1Affine out;
        const size_t k = xs.size();
        std::vector<uint8_t> xb(32 * k), yb(32 * k), rb(32 * (k < 2 ? 2 : k));
        for (size_t i = 0; i < k; i++) {
            std::memcpy(&xb[32 * i], xs[i].le.data(), 32);
            std::memcpy(&yb[32 * i], ys[i].le.data(), 32);
        }
        size_t rl = 0;
        g_.check(kzg_witness_coeff_batched_sharded(g_.handle(), params_.gs, p.coeffs.data(), p.num_coeffs(), xb.data(), yb.data(), k,
                                                   KZG_FR_CANONICAL_LE_32, 0, out.bytes.data(), KZG_G1_AFFINE_MONT_96, rb.data(), &rl));
        if (r) {
            std::vector<Scalar> c(rl);
            std::memcpy(c.data(), rb.data(), 32 * rl);
            *r = Polynomial::make(std::move(c));
        }
        return out;
    }

    // KZGProverEvalForm::create_witness over the group (src/eval_form.rs:124-140); `lagrange` = the Lagrange-basis SRS sharded
    // over the same group (kzg_srs_upload_g1_sharded), evals = the d evaluations
    KZGWitness create_witness_eval(const kzg_msrs *lagrange, const std::vector<Scalar> &evals, size_t index) const {
        G1Affine out;
        g_.check(kzg_witness_eval_sharded(g_.handle(), lagrange, evals.data(), evals.size(), index, KZG_FR_CANONICAL_LE_32, 0,
                                          out.bytes.data(), KZG_G1_AFFINE_MONT_96));
        return out;
    }

  private:
    const ShardedParams &params_;
    const DeviceGroup &g_;
};

class KZGProverEvalForm {  // src/eval_form.rs:39-147
  public:
    KZGProverEvalForm(const KZGParams &params, const kzg_srs *lagrange_basis_g)  // :88-100
        : params_(params), lag_(lagrange_basis_g), e_(*params.engine) {
        int rc = kzg_compute_omega(params.len(), &d_, &exp_, omega_.le.data(), KZG_FR_CANONICAL_LE_32);
        if (rc) throw ReferencePanic("compute_omega(...).unwrap()");
    }
    size_t degree() const { return d_; }
    Scalar omega() const { return omega_; }
    KZGCommitment commit(const EvaluationDomain &ev) const {  // :114-122
        if (d_ != ev.d) throw ReferencePanic("assert!(self.d == evals.d)");
        G1Affine out;
        e_.check(kzg_commit_eval(e_.ctx(), lag_, ev.coeffs.data(), ev.len(), KZG_FR_CANONICAL_LE_32, 0, out.bytes.data(),
                                 KZG_G1_AFFINE_MONT_96));
        return out;
    }
    KZGWitness create_witness(const EvaluationDomain &ev, size_t i) const {  // :124-140
        G1Affine out;
        e_.check(kzg_witness_eval(e_.ctx(), lag_, ev.coeffs.data(), ev.len(), i, KZG_FR_CANONICAL_LE_32, 0,
                                  out.bytes.data(), KZG_G1_AFFINE_MONT_96));
        return out;
    }
    KZGWitness create_witness_all() const { return G1Affine{}; }  // :142-146: the identity

  private:
    const KZGParams &params_;
    const kzg_srs *lag_;
    const Engine &e_;
    size_t d_ = 0;
    uint32_t exp_ = 0;
    Scalar omega_;
};

}  // namespace kzg
