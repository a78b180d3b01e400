// Exercises include/kzg_mi355x.hpp (the C++ host mirror) on the GPU: the reference's test_eval_basic
// degree-1 edge case (src/coeff_form.rs:332-341) incl. verify_eval, commit/verify_poly (test_basic, :271-285) and
// a 3-point batched opening with verify_eval_batched (test_eval_batched, :344-376).
// `cpp_mirror_test group` runs the device-group block as well (it forms an RCCL communicator: tests/test_gpu_mgpu.py, behind the
// suite's RCCL probe); without the argument only the single-GPU surface runs (tests/test_gpu_kzg.py).
#include <cstdio>
#include <cstring>
#include "../include/kzg_mi355x.hpp"
using namespace kzg;
int main(int argc, char **argv) {
    const bool with_group = argc > 1 && !std::strcmp(argv[1], "group");
    Engine e(0);
    KZGParams params = setup(e, Scalar::from_u64(0x1234567), 13);
    KZGProver prover(params);
    std::vector<Scalar> c(13);
    c[0] = Scalar::from_u64(3);
    c[1] = Scalar::from_u64(1);
    Polynomial p = Polynomial::make(c);
    if (p.num_coeffs() != 2) return 1;
    if (!(p.eval(e, Scalar::from_u64(1)) == Scalar::from_u64(4))) return 2;
    KZGCommitment cm = prover.commit(p);
    if (!prover.verify_poly(cm, p)) return 3;
    KZGWitness w = prover.create_witness(p, Scalar::from_u64(1), Scalar::from_u64(4));  // quotient = 1 -> gs[0]
    uint8_t g[96];
    if (kzg_srs_download_g1(e.ctx(), params.gs, 0, 1, g, KZG_G1_AFFINE_MONT_96)) return 4;
    if (std::memcmp(w.bytes.data(), g, 96) != 0) return 5;
    try {
        prover.create_witness(p, Scalar::from_u64(1), Scalar::from_u64(5));
        return 6;
    } catch (const KZGError &err) {
        if (err.kind != KZGError::PointNotOnPolynomial) return 7;
    }
    EvaluationDomain ev = EvaluationDomain::from_coeffs({Scalar::from_u64(1), Scalar::from_u64(2), Scalar::from_u64(3)});
    if (ev.d != 4 || ev.exp != 2) return 8;
    std::vector<Scalar> orig = ev.coeffs;
    ev.fft(e);
    ev.ifft(e);
    if (!(ev.coeffs == orig)) return 9;
    KZGVerifier verifier(params);
    if (!verifier.verify_eval(Scalar::from_u64(1), Scalar::from_u64(4), cm, w)) return 10;
    if (verifier.verify_eval(Scalar::from_u64(1), Scalar::from_u64(5), cm, w)) return 11;
    std::vector<Scalar> c2(13);
    for (int i = 0; i < 7; i++) c2[i] = Scalar::from_u64(1000 + 17 * i);
    Polynomial p2 = Polynomial::make(c2);
    KZGCommitment cm2 = prover.commit(p2);
    std::vector<Scalar> xs = {Scalar::from_u64(5), Scalar::from_u64(6), Scalar::from_u64(7)}, ys;
    for (auto &x : xs) ys.push_back(p2.eval(e, x));
    KZGBatchWitness bw = prover.create_witness_batched(p2, xs, ys);
    if (!verifier.verify_eval_batched(xs, cm2, bw)) return 12;
    xs[1] = Scalar::from_u64(9);
    if (verifier.verify_eval_batched(xs, cm2, bw)) return 13;
    {   // create_witness_many == create_witness per opening; a wrong y is flagged, not thrown
        std::vector<Scalar> mx = {Scalar::from_u64(11), Scalar::from_u64(12), Scalar::from_u64(13)}, my;
        for (auto &x : mx) my.push_back(p2.eval(e, x));
        my[2] = Scalar::from_u64(1);
        std::vector<bool> ok;
        std::vector<KZGWitness> ws = prover.create_witness_many(p2, mx, my, &ok);
        if (ws.size() != 3 || !ok[0] || !ok[1] || ok[2]) return 14;
        if (!(ws[0] == prover.create_witness(p2, mx[0], my[0]))) return 15;
        if (!verifier.verify_eval(mx[1], my[1], cm2, ws[1])) return 16;
    }
    if (with_group) {   // the multi-GPU prover over every visible GPU (a group of one on the test box): same commitment / witness as one GPU
        int ndev = kzg_device_count();
        if (ndev < 1) return 20;
        std::vector<int> devs;
        for (int i = 0; i < ndev && i < 8; i++) devs.push_back(i);
        DeviceGroup group(devs);
        if (kzg_mctx_set_option(group.handle(), "always_gather", 1)) return 21;  // exercise the RCCL exchange even alone
        ShardedParams sp = setup_sharded(group, Scalar::from_u64(0x1234567), 13);
        ShardedKZGProver sprover(sp);
        if (!(sprover.commit(p2) == cm2)) return 22;
        if (group.info().find("formation_ms=") == std::string::npos) return 26;
        Scalar x5 = Scalar::from_u64(5), y5 = p2.eval(e, x5);
        if (!(sprover.create_witness(p2, x5, y5) == prover.create_witness(p2, x5, y5))) return 23;
        try {
            sprover.create_witness(p2, x5, Scalar::from_u64(1));
            return 24;
        } catch (const KZGError &err) {
            if (err.kind != KZGError::PointNotOnPolynomial) return 25;
        }
    }
    if (e.info().find("device=0") == std::string::npos) return 30;
    std::printf(with_group ? "cpp mirror ok (with the device group)\n" : "cpp mirror ok\n");
    return 0;
}
