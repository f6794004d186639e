// Host-side circuit pipeline: circom R1CS / witness loaders, CircomCircuit::synthesize into the
// R1CS -> width-4 PLONK gate transpiler, and the gate list the setup / prover consume.
// Mirrors (reference file:line):
//   R1CS, CircomCircuit            src/circom_circuit.rs:33-47
//   synthesize                     src/circom_circuit.rs:74-133
//   load_r1cs / json / bin         src/reader.rs:178-241 ; src/r1cs_file.rs:44-154
//   load_witness_*                 src/reader.rs:92-175
//   transpile (TranspilerWrapper)  src/transpile.rs:18-139 -> bellman_ce adaptor::Transpiler (absent;
//                                  gate shapes per SURVEY.md A.3; long-LC chains from recollection)
//   analyse                        src/plonk.rs:72-93
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <mutex>
#include "hostmath.h"

#include <new>
#include <exception>
#include <functional>

#include <cstdlib>
#include <sys/mman.h>

namespace plk {

// Allocator of the few very large host vectors of the loaders, the transpiler and the setup (hundreds of MB written once by
// many threads): 2 MiB-aligned blocks marked MADV_HUGEPAGE.  With 4 KiB pages the first touch of 650 MB at the 2^20 domain is
// 160 K page faults serialised on the address-space lock — a large part of what `plonkit prove` spends outside the GPU.
template <class T> struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        void *p = nullptr;
        if (bytes >= ((size_t)4 << 20)) {
            const size_t huge = (size_t)2 << 20, rounded = (bytes + huge - 1) & ~(huge - 1);
            if (posix_memalign(&p, huge, rounded) != 0) throw std::bad_alloc();
            (void)madvise(p, rounded, MADV_HUGEPAGE);
        } else {
            p = malloc(bytes ? bytes : 1);
            if (!p) throw std::bad_alloc();
        }
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) { free(p); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using big_vector = std::vector<T, HugeAlloc<T>>;

void set_error(const std::string &msg);
// nothing may unwind through the extern "C" boundary: a std::bad_alloc (a header that announces 2^32 constraints) or any
// other exception becomes a status code.  The Rust reference panics (aborts the call) at the same places.
template <class Fn> static inline int32_t guarded(const char *who, int32_t on_alloc, Fn fn) {
    try { return fn(); }
    catch (const std::bad_alloc &) { set_error(std::string(who) + ": out of host memory (malformed size field?)"); return on_alloc; }
    catch (const std::exception &e) { set_error(std::string(who) + ": " + e.what()); return on_alloc; }
    catch (...) { set_error(std::string(who) + ": unexpected exception"); return on_alloc; }
}

using host::HFr;

struct LcTerm { uint32_t wire; HFr coeff; };
typedef std::vector<LcTerm> Lc;
struct LcView {                             // one linear combination inside R1cs::terms
    const LcTerm *p; size_t n;
    const LcTerm *begin() const { return p; }
    const LcTerm *end() const { return p + n; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    const LcTerm &operator[](size_t i) const { return p[i]; }
};

// R1CS of src/circom_circuit.rs:33-39, stored flat: all linear combinations back to back (A_0, B_0, C_0, A_1, ..) with one
// offset table — a 2^20-constraint circuit is three allocations, not three million, and the loaders fill it from many threads
struct R1cs {
    uint64_t num_inputs = 0, num_aux = 0, num_variables = 0;
    big_vector<LcTerm> terms;
    std::vector<uint64_t> off{0};           // 3 * num_constraints + 1 offsets into terms
    size_t num_constraints() const { return (off.size() - 1) / 3; }
    LcView lc(size_t constraint, int which) const { const uint64_t b = off[3 * constraint + which]; return {terms.data() + b, (size_t)(off[3 * constraint + which + 1] - b)}; }
    void clear() { terms.clear(); off.assign(1, 0); }
    void push_lc(const LcTerm *t, size_t n) { terms.insert(terms.end(), t, t + n); off.push_back(terms.size()); }
};

// runs fn(lo, hi) over [0, n) on up to `max_threads` host threads (serially when the range is small)
void parallel_for(size_t n, size_t min_per_thread, const std::function<void(size_t, size_t)> &fn, unsigned max_threads = 32);

// variable ids: 0 = dummy (value 0); w in [1, num_variables) = circom wire w (Input(w) for
// w < num_inputs, Aux(w - num_inputs + AUX_OFFSET) otherwise, src/circom_circuit.rs:107-113);
// ids >= num_variables are temporaries allocated by the transpiler.
struct Gate {
    uint32_t v[4];
    HFr q[7];           // q_a q_b q_c q_d q_m q_const q_d_next
};

struct ConstraintStat { std::string name; uint64_t num_gates; };

// temporary variable id (num_variables + index) = constant + sum coeff * value[var]
struct WitnessTerm { uint32_t var; HFr coeff; };
struct WitnessOp { uint32_t first, count; HFr constant; };

struct Transpiled {
    // The gates (without the public-input gates) stay in the pieces the host threads produced, in constraint order: piece k
    // holds gates [gate0[k], gate0[k + 1]); inside a piece the transpiler's temporaries are numbered from first_tmp as if the
    // piece were alone, so a consumer adds tmp_shift[k] to every variable id >= first_tmp (gate_at() does).  The setup reads
    // them once, piece-parallel, straight into its columns — a 2^20-gate circuit never holds a second 250 MB copy.
    std::vector<big_vector<Gate>> pieces;
    std::vector<uint64_t> gate0;        // pieces.size() + 1 entries
    std::vector<uint32_t> tmp_shift;
    uint64_t num_gates = 0, first_tmp = 0;
    Gate gate_at(size_t piece, size_t i) const {
        Gate g = pieces[piece][i];
        for (int j = 0; j < 4; j++) if (g.v[j] >= first_tmp) g.v[j] += tmp_shift[piece];
        return g;
    }
    std::vector<HFr> values;            // per variable id; empty when there is no witness
    std::vector<ConstraintStat> stats;  // per-constraint gate counts (plonk::analyse); skipped when collect_stats is false
    bool collect_stats = true;
    uint64_t num_hints = 0;
    uint64_t num_vars = 0;              // including temporaries
    std::vector<WitnessOp> ops;         // one per temporary, in allocation order
    std::vector<WitnessTerm> op_terms;
};

}  // namespace plk

struct plk_circuit {
    plk::R1cs r1cs;
    plk::big_vector<plk::HFr> witness;
    bool has_witness = false;
    mutable bool witness_registered = false;   // page-locked for fast upload (done by plk_prove from the second proof on)
    mutable uint32_t proofs_started = 0;
    mutable std::mutex reg_mu;                 // the two fields above: one circuit object may be proved from two contexts / host threads at once
};

namespace plk {

// parsers: return false and set_error() on malformed input
bool parse_r1cs_bin(const uint8_t *data, size_t len, R1cs *out);
bool parse_r1cs_json(const uint8_t *data, size_t len, R1cs *out);
bool parse_wtns_bin(const uint8_t *data, size_t len, big_vector<HFr> *out);
bool parse_witness_json(const uint8_t *data, size_t len, big_vector<HFr> *out);
bool fr_from_decimal(const std::string &s, HFr *out);

// transpile; `witness` may be null.  Returns false (set_error) on an unsatisfiable constant constraint.
bool transpile(const R1cs &r, const big_vector<HFr> *witness, Transpiled *out);
std::string analyse_json(const R1cs &r, const Transpiled &t);

}  // namespace plk
