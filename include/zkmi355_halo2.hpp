// C++ host-side mirror of the halo2_proofs items on the hot path, over the C ABI (zkmi355.h).
//
// The reference is a Rust workspace and Rust is not available in this build image, so this header
// is the compiled-language counterpart of the shim crate described in INTEGRATION.md: same names,
// argument meaning and error behaviour as the halo2 items it mirrors, so call sites translate
// one-to-one.  Header-only; link with -lzkmi355.
//
//   halo2_proofs::arithmetic::best_fft            -> zk::halo2::best_fft
//   halo2_proofs::arithmetic::best_multiexp       -> zk::halo2::best_multiexp
//   halo2_proofs::arithmetic::eval_polynomial     -> zk::halo2::eval_polynomial
//   halo2_proofs::arithmetic::kate_division       -> zk::halo2::kate_division
//   halo2_proofs::poly::EvaluationDomain          -> zk::halo2::EvaluationDomain
//   halo2_proofs::poly::kzg::commitment::ParamsKZG-> zk::halo2::ParamsKZG
//   halo2_proofs::plonk::{keygen_pk, create_proof}-> zk::halo2::{ProvingKey, create_proof}
//   halo2_proofs::dev::MockProver::{run, verify_par, verify_at_rows_par} -> zk::halo2::mock_verify
//   (reference call sites: circuit-benchmarks/src/super_circuit.rs:104-132, zkevm-circuits/src/test_util.rs:272,
//    prover/src/common/prover/utils.rs:31,55, prover/src/utils.rs:77)
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "zkmi355.h"

namespace zk {
namespace halo2 {

using Fr = std::array<uint64_t, 4>;        // Montgomery limbs, as halo2curves holds them
using G1Affine = std::array<uint64_t, 8>;  // {x, y}; identity = zeros

// halo2's plonk::Error / io errors surface as one exception type carrying the library message
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

class Context {
   public:
    explicit Context(int device = 0) {
        int rc = zk_ctx_create(device, &ctx_);
        if (rc) throw Error(rc, rc == ZK_ERR_NO_DEVICE ? "no gfx950 device (there is no CPU fallback)" : "zk_ctx_create failed");
    }
    ~Context() { zk_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    zk_ctx* raw() const { return ctx_; }
    void check(int rc) const { if (rc) throw Error(rc, zk_last_error(ctx_)); }

    // RAII device column
    class Buffer {
       public:
        Buffer(const Context& c, size_t bytes) : c_(c), bytes_(bytes) { c_.check(zk_buf_alloc(c_.raw(), bytes, &p_)); }
        ~Buffer() { zk_buf_free(c_.raw(), p_); }
        Buffer(const Buffer&) = delete;
        Buffer& operator=(const Buffer&) = delete;
        void* ptr() const { return p_; }
        size_t bytes() const { return bytes_; }
        void upload(const void* h, size_t n) { c_.check(zk_h2d(c_.raw(), p_, h, n)); }
        void download(void* h, size_t n) const { c_.check(zk_d2h(c_.raw(), h, p_, n)); }
       private:
        const Context& c_;
        void* p_ = nullptr;
        size_t bytes_;
    };

   private:
    zk_ctx* ctx_ = nullptr;
};

// arithmetic::best_fft(a, omega, log_n): in place, natural order in and out
inline void best_fft(const Context& c, std::vector<Fr>& a, const Fr& omega, uint32_t log_n) {
    if (a.size() != (size_t(1) << log_n)) throw Error(ZK_ERR_INVALID_ARG, "best_fft: a.len() != 1 << log_n");
    Context::Buffer d(c, a.size() * sizeof(Fr));
    d.upload(a.data(), a.size() * sizeof(Fr));
    c.check(zk_ntt_omega(c.raw(), d.ptr(), log_n, omega.data()));
    d.download(a.data(), a.size() * sizeof(Fr));
}

// arithmetic::best_multiexp(coeffs, bases)
inline G1Affine best_multiexp(const Context& c, const std::vector<Fr>& coeffs, const std::vector<G1Affine>& bases) {
    if (coeffs.size() != bases.size()) throw Error(ZK_ERR_INVALID_ARG, "best_multiexp: coeffs.len() != bases.len()");
    G1Affine out{};
    c.check(zk_msm_g1_host(c.raw(), coeffs.data(), bases.data(), coeffs.size(), out.data()));
    return out;
}

// arithmetic::eval_polynomial(poly, point)
inline Fr eval_polynomial(const Context& c, const std::vector<Fr>& poly, const Fr& point) {
    Context::Buffer d(c, poly.size() * sizeof(Fr) + 32);
    d.upload(poly.data(), poly.size() * sizeof(Fr));
    Fr out{};
    c.check(zk_poly_eval(c.raw(), d.ptr(), poly.size(), point.data(), out.data()));
    return out;
}

// arithmetic::kate_division(a, b): quotient of (a(X) - a(b)) / (X - b)
inline std::vector<Fr> kate_division(const Context& c, const std::vector<Fr>& a, const Fr& b) {
    if (a.size() < 2) return {};
    Context::Buffer d(c, a.size() * sizeof(Fr)), q(c, (a.size() - 1) * sizeof(Fr));
    d.upload(a.data(), a.size() * sizeof(Fr));
    c.check(zk_kate_division(c.raw(), d.ptr(), a.size(), b.data(), q.ptr()));
    std::vector<Fr> out(a.size() - 1);
    q.download(out.data(), out.size() * sizeof(Fr));
    return out;
}

// poly::EvaluationDomain::new(j, k)
class EvaluationDomain {
   public:
    EvaluationDomain(const Context& c, uint32_t j, uint32_t k) : c_(c), k_(k) {
        extended_k_ = k;
        while ((size_t(1) << extended_k_) < (size_t(1) << k) * (j - 1)) ++extended_k_;
    }
    uint32_t k() const { return k_; }
    uint32_t extended_k() const { return extended_k_; }
    void lagrange_to_coeff(Context::Buffer& a) const { c_.check(zk_ntt(c_.raw(), a.ptr(), k_, 1)); }
    void coeff_to_lagrange(Context::Buffer& a) const { c_.check(zk_ntt(c_.raw(), a.ptr(), k_, 0)); }
    void coeff_to_extended(const Context::Buffer& coeffs, Context::Buffer& out) const { c_.check(zk_coeff_to_extended(c_.raw(), coeffs.ptr(), k_, extended_k_, out.ptr())); }
    void extended_to_coeff(Context::Buffer& a) const { c_.check(zk_extended_to_coeff(c_.raw(), a.ptr(), extended_k_)); }
   private:
    const Context& c_;
    uint32_t k_, extended_k_;
};

// poly::kzg::commitment::ParamsKZG<Bn256>
class ParamsKZG {
   public:
    // ParamsKZG::unsafe_setup_with_s(k, s)
    static ParamsKZG unsafe_setup_with_s(const Context& c, uint32_t k, const Fr& s) {
        zk_srs* p = nullptr;
        c.check(zk_srs_setup_with_s(c.raw(), k, s.data(), &p));
        ParamsKZG out(c, p);
        out.g2_.resize(128);
        out.s_g2_.resize(128);
        c.check(zk_g2_setup(s.data(), out.g2_.data(), out.s_g2_.data()));
        return out;
    }
    // after ParamsKZG::read_custom on the host: g and g_lagrange in RawBytes layout
    static ParamsKZG from_points(const Context& c, uint32_t k, const std::vector<G1Affine>& g, const std::vector<G1Affine>& g_lagrange) {
        zk_srs* p = nullptr;
        c.check(zk_srs_create(c.raw(), k, g.data(), g_lagrange.empty() ? nullptr : g_lagrange.data(), &p));
        return ParamsKZG(c, p);
    }
    // ParamsKZG::read_custom(reader, format): `file` is the whole params{k} file; the G2 encodings are
    // kept as they came (the verifier side needs them, the prover does not)
    enum class SerdeFormat : int { Processed = ZK_SERDE_PROCESSED, RawBytes = ZK_SERDE_RAW, RawBytesUnchecked = ZK_SERDE_RAW_UNCHECKED };
    static ParamsKZG read_custom(const Context& c, const std::vector<uint8_t>& file, SerdeFormat format = SerdeFormat::RawBytesUnchecked) {
        zk_srs* p = nullptr;
        std::vector<uint8_t> g2(128), s_g2(128);
        c.check(zk_params_read(c.raw(), file.data(), file.size(), (int)format, &p, g2.data(), s_g2.data()));
        ParamsKZG out(c, p);
        const size_t gl = format == SerdeFormat::Processed ? 64 : 128;
        g2.resize(gl);
        s_g2.resize(gl);
        out.g2_ = std::move(g2);
        out.s_g2_ = std::move(s_g2);
        return out;
    }
    // ParamsKZG::write_custom(writer, format)
    std::vector<uint8_t> write_custom(SerdeFormat format = SerdeFormat::RawBytesUnchecked) const {
        size_t len = 0;
        c_.check(zk_params_write(c_.raw(), srs_, g2_.data(), s_g2_.data(), (int)format, nullptr, 0, &len));
        std::vector<uint8_t> out(len);
        c_.check(zk_params_write(c_.raw(), srs_, g2_.data(), s_g2_.data(), (int)format, out.data(), out.size(), &len));
        return out;
    }
    const std::vector<uint8_t>& g2() const { return g2_; }
    const std::vector<uint8_t>& s_g2() const { return s_g2_; }
    // ParamsKZG::downsize(k): g truncated, Lagrange basis of the smaller domain recomputed on the device
    ParamsKZG downsize(uint32_t new_k) const {
        zk_srs* p = nullptr;
        c_.check(zk_srs_downsize(c_.raw(), srs_, new_k, &p));
        ParamsKZG out(c_, p);
        out.g2_ = g2_;
        out.s_g2_ = s_g2_;
        return out;
    }
    ParamsKZG(ParamsKZG&& o) noexcept : c_(o.c_), srs_(o.srs_), g2_(std::move(o.g2_)), s_g2_(std::move(o.s_g2_)) { o.srs_ = nullptr; }
    ~ParamsKZG() { if (srs_) zk_srs_destroy(c_.raw(), srs_); }
    uint32_t k() const { return zk_srs_k(srs_); }
    uint64_t n() const { return uint64_t(1) << k(); }
    // ParamsKZG::commit (coefficient basis) / commit_lagrange
    G1Affine commit(const Context::Buffer& poly, size_t n) const { G1Affine o{}; c_.check(zk_commit(c_.raw(), srs_, 0, poly.ptr(), n, o.data())); return o; }
    G1Affine commit_lagrange(const Context::Buffer& poly, size_t n) const { G1Affine o{}; c_.check(zk_commit(c_.raw(), srs_, 1, poly.ptr(), n, o.data())); return o; }
    // every column of a phase in one pipelined batch
    std::vector<G1Affine> commit_lagrange_batch(const std::vector<const void*>& cols, size_t n) const {
        std::vector<G1Affine> o(cols.size());
        c_.check(zk_commit_batch(c_.raw(), srs_, 1, cols.data(), cols.size(), n, o.data()));
        return o;
    }
    const zk_srs* raw() const { return srs_; }
   private:
    ParamsKZG(const Context& c, zk_srs* p) : c_(c), srs_(p) {}
    const Context& c_;
    zk_srs* srs_;
    std::vector<uint8_t> g2_, s_g2_;   // G2 generator and s * generator, file encoding
};

// plonk::keygen_pk result
class ProvingKey {
   public:
    ProvingKey(const Context& c, const ParamsKZG& params, const std::vector<uint8_t>& circuit_blob) : c_(c) {
        c_.check(zk_pk_create(c_.raw(), params.raw(), circuit_blob.data(), circuit_blob.size(), &pk_));
    }
    ~ProvingKey() { zk_pk_destroy(c_.raw(), pk_); }
    ProvingKey(const ProvingKey&) = delete;
    ProvingKey& operator=(const ProvingKey&) = delete;
    const zk_pk* raw() const { return pk_; }
    // install upstream's `vk.transcript_repr()` (the first scalar every proof absorbs)
    void set_transcript_repr(const Fr& repr) { c_.check(zk_pk_set_transcript_repr(c_.raw(), pk_, repr.data())); }
   private:
    const Context& c_;
    zk_pk* pk_ = nullptr;
};

// plonk::create_proof(params, pk, circuits, instances, rng, transcript): the transcript bytes come back
inline std::vector<uint8_t> create_proof(const Context& c, const ProvingKey& pk, const std::vector<const void*>& advice_columns,
                                         const std::vector<const void*>& instance_columns, const std::array<uint8_t, 16>& rng_seed) {
    std::vector<uint8_t> proof(size_t(1) << 20);
    size_t len = 0;
    c.check(zk_create_proof(c.raw(), pk.raw(), advice_columns.data(), instance_columns.data(), rng_seed.data(), proof.data(), proof.size(), &len));
    proof.resize(len);
    return proof;
}

// The phase-by-phase session behind plonk::create_proof for circuits whose later phases depend on challenges (the SuperCircuit has
// three phases [REF zkevm-circuits/src/util.rs:120-133]): begin -> advice_phase per phase (the host synthesises that phase's columns
// with the challenges returned so far) -> finish.  What the Rust shim drives through ffi.rs.
class ProofSession {
   public:
    // instances as halo2 takes them: exactly these values are absorbed, the columns are zero-padded on the device
    ProofSession(const Context& c, const ProvingKey& pk, const std::vector<std::vector<Fr>>& instances, const std::array<uint8_t, 16>& rng_seed, bool shplonk = true) : c_(c) {
        std::vector<const void*> ptrs;
        std::vector<uint32_t> lens;
        for (const auto& col : instances) { ptrs.push_back(col.data()); lens.push_back((uint32_t)col.size()); }
        c_.check(zk_proof_begin_instances(c_.raw(), pk.raw(), ptrs.data(), lens.data(), rng_seed.data(), &s_));
        uint32_t shape[16] = {0};
        int rc = zk_pk_shape(c_.raw(), pk.raw(), shape);
        if (rc == ZK_OK) rc = zk_proof_set_multiopen(c_.raw(), s_, shplonk ? ZK_MULTIOPEN_SHPLONK : ZK_MULTIOPEN_GWC);
        if (rc != ZK_OK) { zk_proof_abort(c_.raw(), s_); s_ = nullptr; c_.check(rc); }
        num_challenges_ = shape[10];
    }
    ~ProofSession() { if (s_) zk_proof_abort(c_.raw(), s_); }
    ProofSession(const ProofSession&) = delete;
    ProofSession& operator=(const ProofSession&) = delete;
    // Poseidon (gen_snark_shplonk) or Keccak / EVM (gen_evm_proof_shplonk) instead of Blake2b: right after construction
    void set_transcript_kind(int kind) { c_.check(zk_proof_set_transcript_kind(c_.raw(), s_, kind)); }
    // commits the columns of the current phase (column_index[j] -> columns[j], n x 32 B each); returns the challenges that become usable after it
    std::vector<Fr> advice_phase(const std::vector<uint32_t>& column_index, const std::vector<const void*>& columns) {
        std::vector<Fr> ch(num_challenges_ ? num_challenges_ : 1);
        uint32_t cnt = (uint32_t)ch.size();
        c_.check(zk_proof_advice_phase(c_.raw(), s_, column_index.data(), columns.data(), (uint32_t)column_index.size(), ch.data(), &cnt));
        ch.resize(cnt);
        return ch;
    }
    // the same for columns resident on the device (device pointers); in_place: the session works in those buffers (it overwrites their
    // blinding rows) until finish() or the destructor returns
    std::vector<Fr> advice_phase_dev(const std::vector<uint32_t>& column_index, const std::vector<const void*>& device_columns, bool in_place = false) {
        std::vector<Fr> ch(num_challenges_ ? num_challenges_ : 1);
        uint32_t cnt = (uint32_t)ch.size();
        c_.check(zk_proof_advice_phase_dev(c_.raw(), s_, column_index.data(), device_columns.data(), (uint32_t)column_index.size(), in_place ? ZK_ADVICE_DEV_IN_PLACE : 0u, ch.data(), &cnt));
        ch.resize(cnt);
        return ch;
    }
    // the vanishing argument's "random" polynomial: ZK_VANISHING_ONE (the reference's own proofs; default) or ZK_VANISHING_UNIFORM (upstream halo2)
    void set_vanishing_random(int kind) { c_.check(zk_proof_set_vanishing_random(c_.raw(), s_, kind)); }
    // MockProver's row checks over the columns the session holds, under its own challenges (after the last phase)
    std::vector<zk_mock_failure> mock_verify(size_t max_records = 4096) {
        std::vector<zk_mock_failure> out(max_records ? max_records : 1);
        size_t count = 0;
        c_.check(zk_proof_mock_verify(c_.raw(), s_, nullptr, 0, nullptr, 0, out.data(), max_records, &count));
        out.resize(count < max_records ? count : max_records);
        return out;
    }
    // the proof bytes; the session is gone afterwards
    std::vector<uint8_t> finish() {
        std::vector<uint8_t> proof(size_t(1) << 20);
        size_t len = 0;
        zk_proof* s = s_;
        s_ = nullptr;                       // zk_proof_finish consumes the session on success and on failure
        c_.check(zk_proof_finish(c_.raw(), s, proof.data(), proof.size(), &len));
        proof.resize(len);
        return proof;
    }
   private:
    const Context& c_;
    zk_proof* s_ = nullptr;
    uint32_t num_challenges_ = 0;
};

// dev::MockProver::run(k, &circuit, instances) + verify_par() / verify_at_rows_par(gate_rows, lookup_rows): the failures, sorted
// (empty = assert_satisfied_par passes).  challenges empty = MockProver's own chain (zk_host_mock_challenges).
inline std::vector<zk_mock_failure> mock_verify(const Context& c, const ProvingKey& pk, const std::vector<const void*>& advice_columns,
                                                const std::vector<const void*>& instance_columns, const std::vector<Fr>& challenges = {},
                                                const std::vector<uint32_t>* gate_rows = nullptr, const std::vector<uint32_t>* lookup_rows = nullptr,
                                                size_t max_records = 4096) {
    std::vector<zk_mock_failure> out(max_records ? max_records : 1);
    size_t count = 0;
    c.check(zk_mock_verify(c.raw(), pk.raw(), advice_columns.data(), instance_columns.data(), challenges.empty() ? nullptr : challenges.data(),
                           gate_rows ? gate_rows->data() : nullptr, gate_rows ? gate_rows->size() : 0, lookup_rows ? lookup_rows->data() : nullptr,
                           lookup_rows ? lookup_rows->size() : 0, out.data(), max_records, &count));
    out.resize(count < max_records ? count : max_records);
    return out;
}

// The same with `instances: &[&[Fr]]` as halo2 takes them -- exactly these values are absorbed into
// the transcript, the columns are zero-padded on the device -- for single-phase circuits (with
// several phases the host re-synthesises between zk_proof_advice_phase calls).
inline std::vector<uint8_t> create_proof(const Context& c, const ProvingKey& pk, const std::vector<const void*>& advice_columns,
                                         const std::vector<std::vector<Fr>>& instances, const std::array<uint8_t, 16>& rng_seed, bool shplonk = true) {
    std::vector<const void*> ptrs;
    std::vector<uint32_t> lens, index;
    for (const auto& col : instances) { ptrs.push_back(col.data()); lens.push_back((uint32_t)col.size()); }
    for (uint32_t i = 0; i < advice_columns.size(); ++i) index.push_back(i);
    zk_proof* sess = nullptr;
    c.check(zk_proof_begin_instances(c.raw(), pk.raw(), ptrs.data(), lens.data(), rng_seed.data(), &sess));
    std::vector<uint8_t> proof(size_t(1) << 20);
    size_t len = 0;
    uint32_t shape[16] = {0};
    int rc = zk_pk_shape(c.raw(), pk.raw(), shape);
    std::vector<Fr> challenges(shape[10] ? shape[10] : 1);              // sized from the key: a phase writes every challenge it yields
    uint32_t num_challenges = (uint32_t)challenges.size();
    if (rc == ZK_OK) rc = zk_proof_set_multiopen(c.raw(), sess, shplonk ? 1 : 0);
    if (rc == ZK_OK) rc = zk_proof_advice_phase(c.raw(), sess, index.data(), advice_columns.data(), (uint32_t)index.size(), challenges.data(), &num_challenges);
    if (rc != ZK_OK) { zk_proof_abort(c.raw(), sess); c.check(rc); }
    c.check(zk_proof_finish(c.raw(), sess, proof.data(), proof.size(), &len));
    proof.resize(len);
    return proof;
}

}  // namespace halo2
}  // namespace zk
