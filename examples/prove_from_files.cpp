// A prover process in C++ over the C ABI alone (include/zkmi355_halo2.hpp -> libzkmi355.so): what a compiled-language host does with
// the library when the circuit and its witness arrive as files -- the params{k} file of the reference's prover
// [REF prover/src/utils.rs:39-84], the key blob of INTEGRATION.md, the witness columns in their in-memory form.
//
//   prove_from_files <dir> [shplonk|gwc]
//     <dir>/params.bin     ParamsKZG::write_custom, RawBytesUnchecked
//     <dir>/blob.bin       key blob v3 (constraint system, fixed and sigma columns)
//     <dir>/advice.bin     A columns of 2^k x 32 B (Montgomery limbs), one after the other
//     <dir>/instance.bin   I columns of 2^k x 32 B
//     <dir>/repr.bin       vk.transcript_repr, 32 B (optional: without it the key keeps the library's stand-in)
//     <dir>/seed.bin       16 B XorShift seed of the blinding RNG (optional: zeros)
//   writes <dir>/proof_cpp.bin and prints what it did.  Exit status 0 only if MockProver's row checks pass and a proof came out.
//
// Build: g++ -std=c++17 -I include examples/prove_from_files.cpp -o prove_from_files -L zkevm-circuits_amd/lib -lzkmi355 -Wl,-rpath,$PWD/zkevm-circuits_amd/lib
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>

#include "zkmi355_halo2.hpp"

using namespace zk::halo2;

static std::vector<uint8_t> slurp(const std::string& path, bool required = true) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        if (required) throw std::runtime_error("cannot read " + path);
        return {};
    }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <dir> [shplonk|gwc]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const bool shplonk = argc < 3 || std::string(argv[2]) != "gwc";
    try {
        Context ctx(0);
        ParamsKZG params = ParamsKZG::read_custom(ctx, slurp(dir + "/params.bin"), ParamsKZG::SerdeFormat::RawBytesUnchecked);
        const std::vector<uint8_t> blob = slurp(dir + "/blob.bin");
        ProvingKey pk(ctx, params, blob);
        const std::vector<uint8_t> repr = slurp(dir + "/repr.bin", false);
        if (repr.size() == 32) {
            Fr r;
            std::memcpy(r.data(), repr.data(), 32);
            pk.set_transcript_repr(r);
        }
        uint32_t shape[16] = {0};
        ctx.check(zk_pk_shape(ctx.raw(), pk.raw(), shape));
        const size_t n = size_t(1) << shape[0], A = shape[4], I = shape[5], usable = n - shape[11] - 1;
        const std::vector<uint8_t> advice = slurp(dir + "/advice.bin"), instance = slurp(dir + "/instance.bin", I != 0);
        if (advice.size() != A * n * 32 || instance.size() != I * n * 32) throw std::runtime_error("advice.bin / instance.bin do not hold A / I columns of 2^k field elements");
        std::vector<const void*> adv_ptrs, inst_ptrs;
        for (size_t c = 0; c < A; ++c) adv_ptrs.push_back(advice.data() + c * n * 32);
        for (size_t c = 0; c < I; ++c) inst_ptrs.push_back(instance.data() + c * n * 32);
        // MockProver::run(..).assert_satisfied_par() first, as the reference's tests do before they prove
        const std::vector<zk_mock_failure> failures = mock_verify(ctx, pk, adv_ptrs, inst_ptrs);
        for (const zk_mock_failure& f : failures) std::printf("mock: kind %u index %u sub %u row %u\n", f.kind, f.index, f.sub, f.row);
        if (!failures.empty()) return 1;
        std::array<uint8_t, 16> seed{};
        const std::vector<uint8_t> seed_file = slurp(dir + "/seed.bin", false);
        if (seed_file.size() == 16) std::memcpy(seed.data(), seed_file.data(), 16);
        // instances as halo2 takes them: the values of every usable row of each instance column
        std::vector<std::vector<Fr>> inst(I, std::vector<Fr>(usable));
        for (size_t c = 0; c < I; ++c) std::memcpy(inst[c].data(), instance.data() + c * n * 32, usable * 32);
        // the session a multi-phase circuit would drive phase by phase; these files hold a single-phase witness
        ProofSession session(ctx, pk, inst, seed, shplonk);
        std::vector<uint32_t> all(A);
        for (uint32_t c = 0; c < A; ++c) all[c] = c;
        (void)session.advice_phase(all, adv_ptrs);
        if (!session.mock_verify().empty()) throw std::runtime_error("the session's own row checks disagree with the stand-alone ones");
        const std::vector<uint8_t> proof = session.finish();
        std::ofstream(dir + "/proof_cpp.bin", std::ios::binary).write((const char*)proof.data(), (std::streamsize)proof.size());
        std::printf("k = %u, %zu advice / %zu instance columns, mock checks passed, %s proof of %zu bytes written\n", shape[0], A, I, shplonk ? "SHPLONK" : "GWC", proof.size());
        return 0;
    } catch (const Error& e) {
        std::printf("zkmi355 error %d: %s\n", e.status, e.what());
        return 3;
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 4;
    }
}
