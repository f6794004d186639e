// AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST half of the library (loaders, transpiler, key codec,
// transcript, verifier with its pairing): gcc, no HIP, no GPU.  The four translation units are compiled in directly; the
// only symbols they take from the device half (error text storage, witness un-registration) are provided here.
// Inputs: the golden files named on the command line, then thousands of random mutations of each (byte flips, truncations,
// size-field overwrites): every call must return a status — no out-of-bounds access, no overflow, no exception escaping the
// extern "C" boundary, no abort.  Built and run by tests/test_sanitizers.py (SURVEY.md §5 asked for an ASan / UBSan pass).
#include "../../plonkit_amd/csrc/circuit.cpp"
#include "../../plonkit_amd/csrc/hostapi.cpp"
#include "../../plonkit_amd/csrc/pairing.cpp"
#include "../../plonkit_amd/csrc/verify.cpp"
#include <cstdio>
#include <fstream>

namespace plk {
static thread_local std::string g_err;
void set_error(const std::string &m) { g_err = m; }
}
extern "C" const char *plk_last_error(void) { return plk::g_err.c_str(); }
void plk_circuit_unregister(plk_circuit *) {}

static std::vector<uint8_t> slurp(const char *p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
static uint64_t rs = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

static void mutate(std::vector<uint8_t> &b) {
    if (b.empty()) return;
    switch (rnd() % 5) {
        case 0: b[rnd() % b.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 1: b.resize(rnd() % (b.size() + 1)); break;
        case 2: { size_t o = rnd() % b.size(); for (size_t i = 0; i < 4 && o + i < b.size(); i++) b[o + i] = 0xff; break; }
        case 3: { size_t o = rnd() % b.size(); uint32_t v = (uint32_t)rnd(); for (size_t i = 0; i < 4 && o + i < b.size(); i++) b[o + i] = (uint8_t)(v >> (8 * i)); break; }
        default: for (int k = 0; k < 8; k++) b[rnd() % b.size()] = (uint8_t)rnd();
    }
}

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: sanitize_host r1cs.bin circuit.json witness.json vk.bin proof.bin key.bin iterations\n"); return 2; }
    const std::vector<uint8_t> r1cs = slurp(argv[1]), cj = slurp(argv[2]), wj = slurp(argv[3]), vk = slurp(argv[4]), proof = slurp(argv[5]), key = slurp(argv[6]);
    const int iters = atoi(argv[7]);
    int accepted = 0, rejected = 0;
    auto circuit = [&](const std::vector<uint8_t> &r, int rj, const std::vector<uint8_t> *w, int wjson) {
        plk_circuit *c = nullptr;
        int32_t rc = plk_circuit_load(r.data(), r.size(), rj, w ? w->data() : nullptr, w ? w->size() : 0, wjson, &c);
        if (rc == PLK_OK) {
            std::vector<char> out(1 << 20);
            (void)plk_circuit_analyse(c, out.data(), out.size());
            uint64_t len = 0;
            (void)plk_circuit_export(c, 0, nullptr, 0, &len);
            std::vector<uint8_t> e(len);
            if (plk_circuit_export(c, 0, e.data(), len, &len) == PLK_OK) { plk_circuit *c2 = nullptr; if (plk_circuit_load(e.data(), len, 0, nullptr, 0, 0, &c2) == PLK_OK) plk_circuit_free(c2); }
            plk_circuit_free(c);
            accepted++;
        } else rejected++;
    };
    auto verify = [&](const std::vector<uint8_t> &v, const std::vector<uint8_t> &p) {
        int32_t ok = 0;
        int32_t rc = plk_verify(v.data(), v.size(), p.data(), p.size(), &ok);
        if (rc == PLK_OK && ok) accepted++; else rejected++;
        return rc == PLK_OK && ok;
    };
    auto keyfile = [&](const std::vector<uint8_t> &k) {
        uint64_t n = 0; uint8_t g2[256];
        if (plk_key_parse(k.data(), k.size(), nullptr, 0, &n, g2) != PLK_OK) { rejected++; return; }
        std::vector<plk_g1_affine> pts(n);
        if (plk_key_parse(k.data(), k.size(), pts.data(), n, &n, g2) == PLK_OK) accepted++; else rejected++;
    };
    // the untouched files first: all must be accepted, and the golden proof must verify
    circuit(r1cs, 0, nullptr, 0); circuit(cj, 1, &wj, 1);
    if (!verify(vk, proof)) { fprintf(stderr, "golden proof rejected\n"); return 1; }
    keyfile(key);
    if (rejected) { fprintf(stderr, "a golden file was rejected\n"); return 1; }
    int forged = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> a = r1cs, b = cj, w = wj, v = vk, p = proof;
        mutate(a); circuit(a, 0, nullptr, 0);
        mutate(b); circuit(b, 1, &wj, 1);
        mutate(w); circuit(cj, 1, &w, 1);
        if (it & 1) mutate(v); else mutate(p);
        if (verify(v, p) && (v != vk || p != proof)) forged++;      // a mutated vk/proof pair must never verify
        if (it % 16 == 0) { std::vector<uint8_t> k = key; mutate(k); keyfile(k); }
    }
    // transcript / keccak on odd lengths
    for (size_t n = 0; n < 300; n++) { std::vector<uint8_t> m(n, (uint8_t)n); uint8_t h[32]; plk_keccak256(m.data(), n, h); }
    printf("sanitize_host: %d accepted, %d rejected, %d forged\n", accepted, rejected, forged);
    return forged ? 1 : 0;
}
