// see circuit.h for the reference mapping
#include "circuit.h"
#include "../../include/plonkit_amd.h"
#include <algorithm>
#include <cstring>
#include <atomic>
#include <map>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace plk {

// ------------------------------------------------------------------------------- scalars
bool fr_from_decimal(const std::string &s, HFr *out) {
    if (s.empty()) return false;
    HFr acc = HFr::zero(), ten = HFr::from_u64(10);
    for (char ch : s) {
        if (ch < '0' || ch > '9') return false;
        acc = acc * ten + HFr::from_u64((uint64_t)(ch - '0'));
    }
    *out = acc;
    return true;
}

static bool fr_from_le32(const uint8_t *p, HFr *out) {
    uint64_t c[4];
    memcpy(c, p, 32);
    if (HFr::geq_p(c)) return false;          // read_field: rejected when >= r (src/r1cs_file.rs:37-42)
    *out = HFr::from_canonical(c);
    return true;
}

static const uint8_t BN254_R_LE[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                       0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

struct Rd {
    const uint8_t *p; size_t len, off = 0; bool ok = true;
    Rd(const uint8_t *p_, size_t l) : p(p_), len(l) {}
    bool need(size_t n) { if (off + n > len || off + n < off) { ok = false; return false; } return true; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v; memcpy(&v, p + off, 4); off += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; memcpy(&v, p + off, 8); off += 8; return v; }
};

// Nothing may unwind out of a worker thread (std::terminate would abort the host process under the extern "C" boundary):
// every worker catches, the first exception is kept, all threads are always joined, and it is rethrown on the calling
// thread where guarded() maps it to a status code.  If a thread cannot be created the remaining ranges run on the caller.
void parallel_for(size_t n, size_t min_per_thread, const std::function<void(size_t, size_t)> &fn, unsigned max_threads) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > max_threads) nt = max_threads;
    if (min_per_thread && n / min_per_thread < nt) nt = (unsigned)(n / min_per_thread);
    if (nt <= 1) { if (n) fn(0, n); return; }
    std::exception_ptr first;
    std::mutex mu;
    auto run = [&](size_t lo, size_t hi) noexcept {
        try { fn(lo, hi); }
        catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    };
    std::vector<std::thread> th;
    try { th.reserve(nt); } catch (...) { fn(0, n); return; }
    const size_t per = (n + nt - 1) / nt;
    size_t serial_from = n;                                   // ranges [serial_from, n) run here if thread creation fails
    for (unsigned t = 0; t < nt; t++) {
        const size_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        try { th.emplace_back(run, lo, hi); }
        catch (...) { serial_from = lo; break; }              // std::system_error: out of threads — never leave joinable threads behind
    }
    if (serial_from < n) run(serial_from, n);
    for (auto &x : th) x.join();
    if (first) std::rethrow_exception(first);
}

// ------------------------------------------------------------------------- .r1cs (binary)
// src/r1cs_file.rs:100-154.  Two passes over the constraint section: a serial walk that only reads the length words and
// lays out the offset table, then the terms are converted (range checks + canonical -> Montgomery, the expensive part)
// by all host threads — 0.16 s single-threaded for the 115 MB of a 2^20-constraint circuit.
bool parse_r1cs_bin(const uint8_t *data, size_t len, R1cs *out) {
    Rd r(data, len);
    if (len < 12 || memcmp(data, "r1cs", 4) != 0) { set_error("Invalid magic number"); return false; }
    r.off = 4;
    if (r.u32() != 1) { set_error("Unsupported version"); return false; }
    uint32_t nsec = r.u32();
    std::map<uint32_t, std::pair<size_t, uint64_t>> secs;
    for (uint32_t i = 0; i < nsec; i++) {
        uint32_t t = r.u32(); uint64_t sz = r.u64();
        if (!r.ok || !r.need(sz)) { set_error("InvalidData: truncated section table"); return false; }
        secs[t] = {r.off, sz};
        r.off += sz;
    }
    if (!secs.count(1) || !secs.count(2) || !secs.count(3)) { set_error("InvalidData: missing section"); return false; }
    r.off = secs[1].first;
    uint32_t field_size = r.u32();
    if (!r.ok || !r.need(field_size)) { set_error("InvalidData: truncated header"); return false; }
    const uint8_t *prime = data + r.off;
    r.off += field_size;
    if (secs[1].second != 32 + (uint64_t)field_size) { set_error("Invalid header section size"); return false; }
    uint32_t n_wires = r.u32(), n_pub_out = r.u32(), n_pub_in = r.u32(), n_prv_in = r.u32();
    uint64_t n_labels = r.u64(); uint32_t n_constraints = r.u32();
    (void)n_prv_in; (void)n_labels;
    if (!r.ok) { set_error("InvalidData: truncated header"); return false; }
    if (field_size != 32) { set_error("This parser only supports 32-byte fields"); return false; }
    if (memcmp(prime, BN254_R_LE, 32) != 0) { set_error("This parser only supports bn256"); return false; }
    // a constraint takes at least its three length words: the header cannot announce more than the section holds
    if ((uint64_t)n_constraints * 12 > secs[2].second) { set_error("InvalidData: constraint count exceeds the constraint section"); return false; }
    const size_t sec2_end = secs[2].first + secs[2].second;
    out->clear();
    out->off.resize((size_t)3 * n_constraints + 1);
    std::vector<size_t> src((size_t)3 * n_constraints);          // file offset of the first term of every linear combination
    r.off = secs[2].first;
    uint64_t total = 0;
    for (size_t k = 0; k < (size_t)3 * n_constraints; k++) {
        uint32_t nv = r.u32();
        if (!r.ok || !r.need((size_t)nv * 36) || r.off + (size_t)nv * 36 > sec2_end) { set_error("InvalidData: truncated constraint"); return false; }
        out->off[k] = total; src[k] = r.off;
        total += nv; r.off += (size_t)nv * 36;
    }
    out->off[(size_t)3 * n_constraints] = total;
    out->terms.resize(total);
    std::atomic<int> bad(0);                                       // 1: coefficient not in the field, 2: wire id out of range
    parallel_for((size_t)3 * n_constraints, 4096, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi && !bad.load(std::memory_order_relaxed); k++) {
            const uint8_t *p = data + src[k];
            LcTerm *dst = out->terms.data() + out->off[k];
            for (uint64_t j = 0, nv = out->off[k + 1] - out->off[k]; j < nv; j++, p += 36) {
                memcpy(&dst[j].wire, p, 4);
                // the reference indexes its variable table with the wire id and panics when it is out of range
                // (src/circom_circuit.rs:107-113); here ids >= n_wires would alias the transpiler's temporaries
                if (dst[j].wire >= n_wires) { bad = 2; return; }
                if (!fr_from_le32(p + 4, &dst[j].coeff)) { bad = 1; return; }
            }
        }
    });
    if (bad == 1) { set_error("InvalidData: coefficient not in field"); return false; }
    if (bad == 2) { set_error("InvalidData: wire index out of range"); return false; }
    if (secs[3].second != (uint64_t)n_wires * 8) { set_error("Invalid map section size"); return false; }
    r.off = secs[3].first;
    if (n_wires) { uint64_t first = r.u64(); if (!r.ok || first != 0) { set_error("Wire 0 should always be mapped to 0"); return false; } }
    out->num_inputs = 1 + (uint64_t)n_pub_in + n_pub_out;        // src/reader.rs:229
    out->num_variables = n_wires;
    if (out->num_variables < out->num_inputs) { set_error("InvalidData: fewer wires than inputs"); return false; }
    out->num_aux = out->num_variables - out->num_inputs;
    return true;
}

// ---------------------------------------------------------------------------- tiny JSON
struct JVal {
    enum { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    std::string s;                                   // NUM / STR text
    std::vector<JVal> a;
    std::vector<std::pair<std::string, JVal>> o;
    const JVal *get(const char *k) const { for (auto &kv : o) if (kv.first == k) return &kv.second; return nullptr; }
};

struct JParser {
    const char *p, *e; bool ok = true;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    bool str(std::string *out) {
        if (p >= e || *p != '"') return ok = false;
        p++; out->clear();
        while (p < e && *p != '"') { if (*p == '\\' && p + 1 < e) { p++; } out->push_back(*p++); }
        if (p >= e) return ok = false;
        p++; return true;
    }
    bool val(JVal *v, int depth = 0) {
        if (depth > 64) return ok = false;
        ws();
        if (p >= e) return ok = false;
        if (*p == '{') {
            v->t = JVal::OBJ; p++; ws();
            if (p < e && *p == '}') { p++; return true; }
            while (ok) {
                ws(); std::string k; if (!str(&k)) return false;
                ws(); if (p >= e || *p != ':') return ok = false; p++;
                v->o.emplace_back(k, JVal());
                if (!val(&v->o.back().second, depth + 1)) return false;
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; return true; }
                return ok = false;
            }
        } else if (*p == '[') {
            v->t = JVal::ARR; p++; ws();
            if (p < e && *p == ']') { p++; return true; }
            while (ok) {
                v->a.emplace_back();
                if (!val(&v->a.back(), depth + 1)) return false;
                ws(); if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; return true; }
                return ok = false;
            }
        } else if (*p == '"') { v->t = JVal::STR; return str(&v->s);
        } else if (*p == 't' && e - p >= 4) { v->t = JVal::BOOL; v->s = "1"; p += 4; return true;
        } else if (*p == 'f' && e - p >= 5) { v->t = JVal::BOOL; v->s = "0"; p += 5; return true;
        } else if (*p == 'n' && e - p >= 4) { v->t = JVal::NUL; p += 4; return true;
        } else {
            v->t = JVal::NUM; const char *s0 = p;
            while (p < e && (isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E')) p++;
            if (p == s0) return ok = false;
            v->s.assign(s0, p); return true;
        }
        return ok;
    }
};

static bool json_parse(const uint8_t *data, size_t len, JVal *out) {
    JParser P{(const char *)data, (const char *)data + len};
    if (!P.val(out)) return false;
    P.ws();
    return P.p == P.e;
}

static bool json_u64(const JVal *v, uint64_t *out) {
    if (!v || (v->t != JVal::NUM && v->t != JVal::STR) || v->s.empty()) return false;
    uint64_t x = 0;
    for (char c : v->s) {
        if (c < '0' || c > '9') return false;
        if (x > (UINT64_MAX - 9) / 10) return false;                 // does not fit 64 bits
        x = x * 10 + (uint64_t)(c - '0');
    }
    *out = x; return true;
}

// src/reader.rs:194-218: constraints are BTreeMap<String,String> => terms ordered by the string key
bool parse_r1cs_json(const uint8_t *data, size_t len, R1cs *out) {
    JVal root;
    if (!json_parse(data, len, &root) || root.t != JVal::OBJ) { set_error("unable to read: malformed circuit json"); return false; }
    uint64_t n_pub, n_out, n_vars;
    if (!json_u64(root.get("nPubInputs"), &n_pub) || !json_u64(root.get("nOutputs"), &n_out) || !json_u64(root.get("nVars"), &n_vars)) {
        set_error("unable to read: nPubInputs/nOutputs/nVars missing"); return false; }
    const JVal *cons = root.get("constraints");
    if (!cons || cons->t != JVal::ARR) { set_error("unable to read: constraints missing"); return false; }
    if (n_vars >= (1ull << 32) || n_pub >= (1ull << 32) || n_out >= (1ull << 32)) { set_error("unable to read: nVars / nPubInputs / nOutputs out of range"); return false; }
    out->num_inputs = n_pub + n_out + 1;
    if (n_vars < out->num_inputs) { set_error("unable to read: nVars < number of inputs"); return false; }
    out->num_aux = n_vars - out->num_inputs;
    out->num_variables = n_vars;
    out->clear();
    Lc one_lc;
    for (size_t i = 0; i < cons->a.size(); i++) {
        const JVal &c = cons->a[i];
        if (c.t != JVal::ARR || c.a.size() < 3) { set_error("unable to read: constraint is not [A,B,C]"); return false; }
        for (int k = 0; k < 3; k++) {
            one_lc.clear();
            if (c.a[k].t != JVal::OBJ) { set_error("unable to read: linear combination is not an object"); return false; }
            std::vector<std::pair<std::string, std::string>> terms;
            for (auto &kv : c.a[k].o) terms.emplace_back(kv.first, kv.second.s);
            std::sort(terms.begin(), terms.end(), [](const std::pair<std::string, std::string> &x, const std::pair<std::string, std::string> &y) { return x.first < y.first; });
            for (auto &t : terms) {
                uint64_t w = 0; HFr cf;
                JVal kv; kv.t = JVal::STR; kv.s = t.first;
                if (!json_u64(&kv, &w) || !fr_from_decimal(t.second, &cf)) { set_error("unable to read: bad term in linear combination"); return false; }
                if (w >= n_vars) { set_error("unable to read: wire index out of range"); return false; }      // checked before the 32-bit cast
                one_lc.push_back({(uint32_t)w, cf});
            }
            out->push_lc(one_lc.data(), one_lc.size());
        }
    }
    return true;
}

// ------------------------------------------------------------------------------ witness
bool parse_wtns_bin(const uint8_t *data, size_t len, big_vector<HFr> *out) {
    Rd r(data, len);
    if (len < 4 || memcmp(data, "wtns", 4) != 0) { set_error("invalid file header"); return false; }
    r.off = 4;
    uint32_t version = r.u32();
    if (!r.ok || version > 2) { set_error("unsupported file version"); return false; }
    if (r.u32() != 2 || !r.ok) { set_error("invalid num sections"); return false; }
    if (r.u32() != 1 || !r.ok) { set_error("invalid section type"); return false; }
    if (r.u64() != 4 + 32 + 4 || !r.ok) { set_error("invalid section len"); return false; }
    if (r.u32() != 32 || !r.ok) { set_error("invalid field byte size"); return false; }
    if (!r.need(32) || memcmp(data + r.off, BN254_R_LE, 32) != 0) { set_error("invalid curve prime"); return false; }
    r.off += 32;
    uint32_t wl = r.u32();
    if (r.u32() != 2 || !r.ok) { set_error("invalid section type"); return false; }
    uint64_t sz = r.u64();
    if (!r.ok || sz != (uint64_t)wl * 32) { set_error("invalid witness section size"); return false; }
    if (!r.need(sz)) { set_error("read witness failed: truncated"); return false; }
    out->resize(wl);
    std::atomic<int> bad(0);
    const uint8_t *src = data + r.off;
    parallel_for(wl, 16384, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) if (!fr_from_le32(src + 32 * i, &(*out)[i])) { bad = 1; return; }
    });
    if (bad) { set_error("read witness failed: not in field"); return false; }
    return true;
}

bool parse_witness_json(const uint8_t *data, size_t len, big_vector<HFr> *out) {
    JVal root;
    if (!json_parse(data, len, &root) || root.t != JVal::ARR) { set_error("unable to read: witness json is not an array"); return false; }
    out->resize(root.a.size());
    for (size_t i = 0; i < root.a.size(); i++)
        if (!fr_from_decimal(root.a[i].s, &(*out)[i])) { set_error("unable to read: bad witness entry"); return false; }
    return true;
}

// --------------------------------------------------------------------------- transpiler
namespace {

struct Term { uint32_t var; HFr coeff; };

// One Builder transpiles a contiguous range of constraints into its own piece: gates, temporaries (numbered from
// r.num_variables upwards inside the piece) and their defining linear forms.  transpile() runs one Builder per chunk on
// the host threads and stitches the pieces in constraint order, shifting every temporary id by the number of temporaries
// of the pieces before it — the result is what a single serial pass produces (gate order, ids, statistics).
struct Piece {
    big_vector<Gate> gates;
    std::vector<HFr> tmp_values;            // per temporary; only when a witness is given
    std::vector<WitnessOp> ops;
    std::vector<WitnessTerm> op_terms;
    std::vector<ConstraintStat> stats;
    uint64_t num_hints = 0;
    bool failed = false;
};

struct Builder {
    Piece *t;
    const big_vector<HFr> *witness;         // circom wires (index 0 = ONE in the file, the dummy variable here), or null
    uint64_t first_tmp;                     // r.num_variables
    const HFr zero = HFr::zero(), one = HFr::one(), minus_one = -HFr::one();
    bool have_values() const { return witness != nullptr; }

    // a temporary is always a linear form over earlier variables: record it, so that a later proof
    // can recompute the temporaries from a fresh witness without re-running the transpiler
    uint32_t alloc(const HFr &v, const Term *terms, size_t n_terms, const HFr &constant) {
        uint32_t id = (uint32_t)(first_tmp + t->ops.size());
        if (have_values()) t->tmp_values.push_back(v);
        WitnessOp op; op.first = (uint32_t)t->op_terms.size(); op.count = (uint32_t)n_terms; op.constant = constant;
        for (size_t i = 0; i < n_terms; i++) t->op_terms.push_back({terms[i].var, terms[i].coeff});
        t->ops.push_back(op);
        return id;
    }
    HFr val(uint32_t v) const {
        if (!have_values() || v == 0) return HFr::zero();              // id 0 is the dummy variable, not circom's ONE
        return v < first_tmp ? (*witness)[v] : t->tmp_values[v - first_tmp];
    }
    void gate(const uint32_t v[4], const HFr q[7]) {
        Gate g; memcpy(g.v, v, sizeof g.v); for (int i = 0; i < 7; i++) g.q[i] = q[i];
        t->gates.push_back(g);
    }

    // stable de-duplication; wire 0 (the constant ONE) is folded into the constant term
    static void split(const LcTerm *lc, size_t n, HFr *constant, std::vector<Term> *terms) {
        *constant = HFr::zero(); terms->clear();
        for (size_t i = 0; i < n; i++) {
            const LcTerm &x = lc[i];
            if (x.wire == 0) { *constant = *constant + x.coeff; continue; }
            bool found = false;
            for (Term &y : *terms) if (y.var == x.wire) { y.coeff = y.coeff + x.coeff; found = true; break; }
            if (!found) terms->push_back({x.wire, x.coeff});
        }
        terms->erase(std::remove_if(terms->begin(), terms->end(), [](const Term &y) { return y.coeff.is_zero(); }), terms->end());
    }

    HFr eval(const std::vector<Term> &lc, const HFr &free) const {
        HFr s = free;
        if (have_values()) for (const Term &x : lc) s = s + x.coeff * val(x.var);
        return s;
    }

    // bellman adaptor::enforce_lc_as_gates [recollection]; single gate pinned by SURVEY.md A.3 row 2
    void lc_as_gates(std::vector<Term> lc, HFr free, bool collapse, uint32_t *var_out, HFr *coeff_out) {
        if (lc.size() == 1 && free.is_zero() && collapse) { *var_out = lc[0].var; *coeff_out = lc[0].coeff; return; }
        uint32_t fin = 0;
        if (collapse) { fin = alloc(eval(lc, free), lc.data(), lc.size(), free); lc.push_back({fin, minus_one}); }
        if (lc.size() <= 4) {
            uint32_t v[4] = {0, 0, 0, 0}; HFr q[7];
            for (int i = 0; i < 7; i++) q[i] = zero;
            for (size_t i = 0; i < lc.size(); i++) { v[i] = lc[i].var; q[i] = lc[i].coeff; }
            q[5] = free;
            gate(v, q);
        } else {                                             // UNPINNED: chain through d / d_next
            size_t pos = 0;
            uint32_t v[4]; HFr q[7];
            for (int i = 0; i < 7; i++) q[i] = zero;
            HFr s = free;
            for (int i = 0; i < 4; i++, pos++) { v[i] = lc[pos].var; q[i] = lc[pos].coeff; if (have_values()) s = s + q[i] * val(v[i]); }
            q[5] = free; q[6] = minus_one;
            uint32_t nxt = alloc(s, lc.data(), 4, free);
            gate(v, q);
            while (lc.size() - pos > 3) {
                for (int i = 0; i < 7; i++) q[i] = zero;
                s = val(nxt);
                for (int i = 0; i < 3; i++, pos++) { v[i] = lc[pos].var; q[i] = lc[pos].coeff; if (have_values()) s = s + q[i] * val(v[i]); }
                v[3] = nxt; q[3] = one; q[6] = minus_one;
                Term chain[4] = {lc[pos - 3], lc[pos - 2], lc[pos - 1], {nxt, one}};
                uint32_t nn = alloc(s, chain, 4, zero);
                gate(v, q);
                nxt = nn;
            }
            for (int i = 0; i < 7; i++) q[i] = zero;
            v[0] = v[1] = v[2] = 0;
            for (int i = 0; pos < lc.size(); i++, pos++) { v[i] = lc[pos].var; q[i] = lc[pos].coeff; }
            v[3] = nxt; q[3] = one;
            gate(v, q);
        }
        *var_out = fin; *coeff_out = one;
    }

    // CircomCircuit::synthesize (src/circom_circuit.rs:114-131) for the constraints [lo, hi) fed to the gate adaptor
    void run(const R1cs &r, size_t lo, size_t hi, bool collect_stats) {
        std::vector<Term> al, bl, cl;
        HFr ac, bc, cc;
        t->gates.reserve((hi - lo) * 2);
        for (size_t idx = lo; idx < hi; idx++) {
            const LcView ka = r.lc(idx, 0), kb = r.lc(idx, 1), kc = r.lc(idx, 2);
            if ((ka.empty() || kb.empty()) && kc.empty()) continue;              // src/circom_circuit.rs:121-122
            size_t g0 = t->gates.size();
            split(ka.p, ka.n, &ac, &al); split(kb.p, kb.n, &bc, &bl); split(kc.p, kc.n, &cc, &cl);
            bool a_k = al.empty(), b_k = bl.empty(), c_k = cl.empty();
            uint32_t dv; HFr dc;
            if (a_k && b_k) {
                HFr free = cc - ac * bc;
                if (c_k) { if (!free.is_zero()) { t->failed = true; return; } }
                else lc_as_gates(cl, free, false, &dv, &dc);
            } else if (a_k || b_k) {                                              // UNPINNED: constant * LC = LC
                const HFr &kk = a_k ? ac : bc; const std::vector<Term> &lin = a_k ? bl : al; const HFr &lin_c = a_k ? bc : ac;
                Lc merged;
                for (const Term &x : lin) merged.push_back({x.var, x.coeff * kk});
                for (const Term &x : cl) merged.push_back({x.var, -x.coeff});
                HFr free = kk * lin_c - cc, dummy;
                std::vector<Term> m2;
                split(merged.data(), merged.size(), &dummy, &m2);
                if (!m2.empty()) lc_as_gates(m2, free, false, &dv, &dc);
                else if (!free.is_zero()) { t->failed = true; return; }
            } else {
                bool same = al.size() == 1 && bl.size() == 1 && al[0].var == bl[0].var && (c_k || (cl.size() == 1 && cl[0].var == al[0].var));
                if (same) {                                                       // UNPINNED: quadratic gate
                    HFr a1 = al[0].coeff, b1 = bl[0].coeff, c1 = c_k ? HFr::zero() : cl[0].coeff;
                    uint32_t v[4] = {al[0].var, al[0].var, 0, 0}; HFr q[7];
                    for (int i = 0; i < 7; i++) q[i] = HFr::zero();
                    q[0] = ac * b1 + a1 * bc - c1; q[4] = a1 * b1; q[5] = ac * bc - cc;
                    gate(v, q);
                } else {
                    uint32_t av, bv, cv; HFr acoef, bcoef, ccoef;
                    lc_as_gates(al, ac, true, &av, &acoef);
                    lc_as_gates(bl, bc, true, &bv, &bcoef);
                    HFr q[7];
                    for (int i = 0; i < 7; i++) q[i] = HFr::zero();
                    q[4] = acoef * bcoef;
                    if (c_k) { uint32_t v[4] = {av, bv, 0, 0}; q[5] = -cc; gate(v, q); }
                    else { lc_as_gates(cl, cc, true, &cv, &ccoef); uint32_t v[4] = {av, bv, cv, 0}; q[2] = -ccoef; gate(v, q); }
                }
            }
            if (collect_stats) t->stats.push_back({std::to_string(idx), (uint64_t)(t->gates.size() - g0)});
            t->num_hints++;
        }
    }
};

}  // namespace

bool transpile(const R1cs &r, const big_vector<HFr> *witness, Transpiled *out) {
    out->pieces.clear(); out->gate0.clear(); out->tmp_shift.clear(); out->num_gates = 0; out->first_tmp = r.num_variables;
    out->values.clear(); out->stats.clear(); out->num_hints = 0;
    out->ops.clear(); out->op_terms.clear();
    out->num_vars = r.num_variables;
    if (witness && witness->size() < r.num_variables) { set_error("witness shorter than the number of variables"); return false; }
    const size_t nc = r.num_constraints();
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    size_t chunks = nc / 8192;                                   // pieces of >= 8192 constraints, a few per thread
    if (chunks > 4 * (size_t)nt) chunks = 4 * (size_t)nt;
    if (chunks < 1) chunks = 1;
    std::vector<Piece> pieces(chunks);
    const size_t per = (nc + chunks - 1) / chunks;
    const bool stats = out->collect_stats;
    parallel_for(chunks, 1, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) {
            Builder B{&pieces[k], witness, r.num_variables};
            B.run(r, std::min(nc, k * per), std::min(nc, (k + 1) * per), stats);
        }
    }, nt);
    for (const Piece &p : pieces) if (p.failed) { set_error("unsatisfiable constant constraint"); return false; }
    // stitch: prefix sums of gates / temporaries / terms, then every piece copies itself into place with its ids shifted
    std::vector<size_t> g0(chunks + 1, 0), t0(chunks + 1, 0), o0(chunks + 1, 0), s0(chunks + 1, 0);
    for (size_t k = 0; k < chunks; k++) {
        g0[k + 1] = g0[k] + pieces[k].gates.size(); t0[k + 1] = t0[k] + pieces[k].ops.size();
        o0[k + 1] = o0[k] + pieces[k].op_terms.size(); s0[k + 1] = s0[k] + pieces[k].stats.size();
        out->num_hints += pieces[k].num_hints;
    }
    if (r.num_variables + t0[chunks] >= (1ull << 32) || o0[chunks] >= (1ull << 32)) { set_error("circuit too large: variable ids do not fit 32 bits"); return false; }
    out->num_gates = g0[chunks];
    out->gate0.assign(g0.begin(), g0.end());
    out->tmp_shift.resize(chunks);
    out->pieces.resize(chunks);
    out->ops.resize(t0[chunks]); out->op_terms.resize(o0[chunks]);
    if (stats) out->stats.resize(s0[chunks]);
    out->num_vars = r.num_variables + t0[chunks];
    if (witness) {
        out->values.resize(out->num_vars);
        std::copy(witness->begin(), witness->begin() + r.num_variables, out->values.begin());
        out->values[0] = HFr::zero();                        // id 0 is the dummy variable, not circom's ONE
    }
    const uint32_t nv = (uint32_t)r.num_variables;
    parallel_for(chunks, 1, [&](size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) {
            Piece &p = pieces[k];
            const uint32_t shift = (uint32_t)t0[k];
            out->tmp_shift[k] = shift;
            out->pieces[k].swap(p.gates);                      // the gates stay where they were produced
            for (size_t i = 0; i < p.ops.size(); i++) { WitnessOp op = p.ops[i]; op.first += (uint32_t)o0[k]; out->ops[t0[k] + i] = op; }
            for (size_t i = 0; i < p.op_terms.size(); i++) { WitnessTerm wt = p.op_terms[i]; if (wt.var >= nv) wt.var += shift; out->op_terms[o0[k] + i] = wt; }
            if (stats) for (size_t i = 0; i < p.stats.size(); i++) out->stats[s0[k] + i] = std::move(p.stats[i]);
            if (witness) std::copy(p.tmp_values.begin(), p.tmp_values.end(), out->values.begin() + r.num_variables + t0[k]);
        }
    }, nt);
    return true;
}

std::string analyse_json(const R1cs &r, const Transpiled &t) {
    std::string s = "{\"num_inputs\":" + std::to_string(r.num_inputs) + ",\"num_aux\":" + std::to_string(r.num_aux) +
                    ",\"num_variables\":" + std::to_string(r.num_variables) + ",\"num_constraints\":" + std::to_string(r.num_constraints()) +
                    ",\"num_nontrivial_constraints\":" + std::to_string(t.stats.size()) + ",\"num_gates\":" + std::to_string(t.num_gates) +
                    ",\"num_hints\":" + std::to_string(t.num_hints);
    if (!t.stats.empty()) {
        s += ",\"constraint_stats\":[";
        for (size_t i = 0; i < t.stats.size(); i++) {
            if (i) s += ",";
            s += "{\"name\":\"" + t.stats[i].name + "\",\"num_gates\":" + std::to_string(t.stats[i].num_gates) + "}";
        }
        s += "]";
    }
    return s + "}";
}

}  // namespace plk

// ------------------------------------------------------------------------------ C ABI
using namespace plk;

extern "C" int32_t plk_circuit_load(const uint8_t *r1cs, uint64_t r1cs_len, int32_t r1cs_is_json,
                                    const uint8_t *witness, uint64_t witness_len, int32_t witness_is_json, plk_circuit **out) {
    if (!r1cs || !out) { set_error("plk_circuit_load: bad argument"); return PLK_ERR_ARG; }
    *out = nullptr;
    return guarded("plk_circuit_load", PLK_ERR_FORMAT, [&]() -> int32_t {
        std::unique_ptr<plk_circuit> c(new plk_circuit());
        bool ok = r1cs_is_json ? parse_r1cs_json(r1cs, r1cs_len, &c->r1cs) : parse_r1cs_bin(r1cs, r1cs_len, &c->r1cs);
        if (ok && witness) {
            ok = witness_is_json ? parse_witness_json(witness, witness_len, &c->witness) : parse_wtns_bin(witness, witness_len, &c->witness);
            c->has_witness = ok;
            if (ok && c->witness.size() < c->r1cs.num_variables) { set_error("witness shorter than the number of variables"); ok = false; }
        }
        if (!ok) return PLK_ERR_FORMAT;
        *out = c.release();
        return PLK_OK;
    });
}

void plk_circuit_unregister(plk_circuit *c);
extern "C" void plk_circuit_free(plk_circuit *c) { if (c) { plk_circuit_unregister(c); delete c; } }

extern "C" int32_t plk_circuit_analyse(const plk_circuit *c, char *out_json, uint64_t cap) {
    if (!c || !out_json) { set_error("plk_circuit_analyse: bad argument"); return PLK_ERR_ARG; }
    return guarded("plk_circuit_analyse", PLK_ERR_FORMAT, [&]() -> int32_t {
        Transpiled t;
        if (!transpile(c->r1cs, nullptr, &t)) return PLK_ERR_UNSAT;
        std::string s = analyse_json(c->r1cs, t);
        if (s.size() + 1 > cap) { set_error("plk_circuit_analyse: buffer too small"); return PLK_ERR_ARG; }
        memcpy(out_json, s.c_str(), s.size() + 1);
        return PLK_OK;
    });
}

// ------------------------------------------------------------------ synthetic R1CS (bench input)
// a seeded multiplication chain over Fr, witness included, using only
// constraint shapes whose transpilation is pinned by the golden circuit:
//   type 1 (1 gate):  (ca*u) * (cb*v) = cc*w                      -> new wire w
//   type 2 (2 gates): (ca*u) * (cb*v) = k + c1*v + c2*q           -> new wire q
// plus a final  1*pub = last  linear constraint (1 gate) tying the single public input to the chain.
// `target_gates` transpiled gates are produced exactly, so that 1 (public input) + target_gates + 1
// is the domain size: 2^20 - 2 gates give the 2^20 domain of BASELINE.json configs[1].
namespace plk {
namespace {
struct Xoshiro256ss {
    uint64_t s[4];
    explicit Xoshiro256ss(uint64_t seed) {
        for (int i = 0; i < 4; i++) {                                   // splitmix64 seeding
            seed += 0x9E3779B97F4A7C15ULL;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            s[i] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    HFr fr() {                                                          // uniform by rejection from 254 bits
        for (;;) {
            uint64_t c[4] = {next(), next(), next(), next()};
            c[3] &= (1ULL << 62) - 1;
            if (!HFr::geq_p(c)) return HFr::from_canonical(c);
        }
    }
    HFr fr_nonzero() { HFr v = fr(); return v.is_zero() ? HFr::one() : v; }
};
}  // namespace
}  // namespace plk

static int32_t circuit_synthetic_impl(uint64_t target_gates, uint64_t seed, uint64_t witness_seed, uint32_t lc_terms, plk_circuit **out);
extern "C" int32_t plk_circuit_synthetic(uint64_t target_gates, uint64_t seed, plk_circuit **out) {
    if (!out || target_gates < 4 || target_gates >= (1ull << 28)) { set_error("plk_circuit_synthetic: bad argument"); return PLK_ERR_ARG; }
    *out = nullptr;
    return guarded("plk_circuit_synthetic", PLK_ERR_ARG, [&] { return circuit_synthetic_impl(target_gates, seed, 0, 0, out); });
}
// The same generator with two more knobs:
//   witness_seed != 0: the same R1CS (it depends on `seed` only) with ANOTHER satisfying witness — the free starting wires are
//       drawn from `witness_seed` and the chain is re-walked: what a prover that serves many requests for one circuit sees.
//   lc_terms >= 5 ("dense"): the body is made of constraints whose A side is a linear combination of `lc_terms` earlier wires
//       plus a constant, (sum_j a_j w_{u-j} + k) * (cb * w_v) = cc * w_new — the shape of a circom Poseidon round (S-box input =
//       MDS row of the previous state + round constant).  The transpiler folds such a combination through the d column
//       (q_d_next = -1 chains, src/circom_circuit.rs:114-131), so the d wire, q_d_next and the fourth quotient chunk are all
//       live: the prover does 11 non-trivial commitments instead of the 9 of the pinned-subset circuit.  PARITY UNPINNED: the
//       chaining rule is a recollection of bellman's adaptor, equal in product and oracle, pinned by neither.
//       1 + ceil((lc_terms - 3) / 3) + 1 gates per such constraint; the remainder to `target_gates` is filled with the
//       one-gate pinned shape.  lc_terms = 0: the pinned-subset circuit of plk_circuit_synthetic.
extern "C" int32_t plk_circuit_synthetic_ex(uint64_t target_gates, uint64_t seed, uint64_t witness_seed, uint32_t lc_terms, plk_circuit **out) {
    if (!out || target_gates < 4 || target_gates >= (1ull << 28) || (lc_terms != 0 && (lc_terms < 5 || lc_terms > 64))) {
        set_error("plk_circuit_synthetic_ex: bad argument (target_gates in [4, 2^28), lc_terms 0 or 5..64)"); return PLK_ERR_ARG; }
    *out = nullptr;
    return guarded("plk_circuit_synthetic_ex", PLK_ERR_ARG, [&] { return circuit_synthetic_impl(target_gates, seed, witness_seed, lc_terms, out); });
}
static int32_t circuit_synthetic_impl(uint64_t target_gates, uint64_t seed, uint64_t witness_seed, uint32_t lc_terms, plk_circuit **out) {
    std::unique_ptr<plk_circuit> holder(new plk_circuit());
    plk_circuit *c = holder.get();
    Xoshiro256ss rng(seed);
    big_vector<HFr> &w = c->witness;
    w.reserve(target_gates + 8 + lc_terms);
    w.push_back(HFr::one()); w.push_back(HFr::zero()); w.push_back(rng.fr()); w.push_back(rng.fr());
    const uint32_t n_free = lc_terms ? lc_terms + 1 : 2;                // wires nothing constrains: private inputs of the chain
    for (uint32_t i = 2; i < n_free; i++) w.push_back(rng.fr());
    if (witness_seed) { Xoshiro256ss wr(witness_seed ^ 0x7769746e65737321ULL); for (uint32_t i = 0; i < n_free; i++) w[2 + i] = wr.fr(); }
    R1cs &R = c->r1cs;
    R.clear();
    R.terms.reserve(target_gates * 4); R.off.reserve(target_gates * 3 + 4);
    // Pass 1 draws every coefficient in the generator's order (the draws do not depend on the witness), pass 2 walks
    // the chain.  The divisions w = (...) / c are by those coefficients, so all of them are inverted together with
    // Montgomery's trick — one field inversion instead of one per constraint (4 s at 2^20 gates).
    struct Draw { HFr ca, cb, kk, c1, cdiv; uint32_t kind; size_t lc0; };      // kind 0: one gate, 1: two gates, 2: dense
    std::vector<Draw> draws;
    std::vector<HFr> lc_coeffs;                                           // lc_terms - 1 further A-side coefficients per dense constraint
    draws.reserve(target_gates);
    uint64_t gates = 0;
    const uint64_t body = target_gates - 1;                             // the last gate is the public-input tie
    const uint64_t dense_gates = lc_terms ? 1 + (lc_terms + 1 - 4 + 2) / 3 + 1 : 0;
    while (gates < body) {
        Draw d;
        d.ca = rng.fr_nonzero(); d.cb = rng.fr_nonzero();
        if (lc_terms && gates + dense_gates <= body) {
            d.kind = 2; d.lc0 = lc_coeffs.size();
            for (uint32_t j = 1; j < lc_terms; j++) lc_coeffs.push_back(rng.fr_nonzero());
            d.kk = rng.fr(); d.cdiv = rng.fr_nonzero(); gates += dense_gates;
        } else if (!lc_terms && (draws.size() & 1) && (gates + 2 <= body)) {
            d.kind = 1; d.kk = rng.fr(); d.c1 = rng.fr_nonzero(); d.cdiv = rng.fr_nonzero(); gates += 2;
        } else { d.kind = 0; d.cdiv = rng.fr_nonzero(); gates += 1; }
        draws.push_back(d);
    }
    std::vector<HFr> inv_c(draws.size());
    {
        HFr acc = HFr::one();
        for (size_t i = 0; i < draws.size(); i++) { inv_c[i] = acc; acc = acc * draws[i].cdiv; }      // prefix products
        HFr run = acc.inv();
        for (size_t i = draws.size(); i-- > 0;) { HFr t = run * inv_c[i]; run = run * draws[i].cdiv; inv_c[i] = t; }
    }
    std::vector<LcTerm> ka_long(lc_terms + 1);
    for (size_t i = 0; i < draws.size(); i++) {
        const Draw &d = draws[i];
        uint32_t u = (uint32_t)w.size() - 1, v = (uint32_t)w.size() - 2;
        const LcTerm kb{v, d.cb};
        if (d.kind == 2) {
            // A = k + ca * w_u + sum_{j >= 1} a_j * w_{u - j}: distinct wires, the constant first (wire 0 is folded by the transpiler)
            HFr a_val = d.kk + d.ca * w[u];
            ka_long[0] = {0, d.kk}; ka_long[1] = {u, d.ca};
            for (uint32_t j = 1; j < lc_terms; j++) { const HFr &a = lc_coeffs[d.lc0 + j - 1]; ka_long[1 + j] = {u - j, a}; a_val = a_val + a * w[u - j]; }
            R.push_lc(ka_long.data(), lc_terms + 1); R.push_lc(&kb, 1);
            w.push_back(a_val * d.cb * w[v] * inv_c[i]);
            const LcTerm kc{(uint32_t)w.size() - 1, d.cdiv};
            R.push_lc(&kc, 1);
            continue;
        }
        HFr prod = d.ca * w[u] * d.cb * w[v];
        const LcTerm ka{u, d.ca};
        R.push_lc(&ka, 1); R.push_lc(&kb, 1);
        if (d.kind == 0) {
            w.push_back(prod * inv_c[i]);
            const LcTerm kc{(uint32_t)w.size() - 1, d.cdiv};
            R.push_lc(&kc, 1);
        } else {
            w.push_back((prod - d.kk - d.c1 * w[v]) * inv_c[i]);
            const LcTerm kc[3] = {{0, d.kk}, {v, d.c1}, {(uint32_t)w.size() - 1, d.cdiv}};
            R.push_lc(kc, 3);
        }
    }
    w[1] = w.back();
    {
        const LcTerm ta{1, HFr::one()}, tb{0, HFr::one()}, tc{(uint32_t)w.size() - 1, HFr::one()};
        R.push_lc(&ta, 1); R.push_lc(&tb, 1); R.push_lc(&tc, 1);
    }
    c->r1cs.num_inputs = 2;
    c->r1cs.num_variables = w.size();
    c->r1cs.num_aux = w.size() - 2;
    c->has_witness = true;
    *out = holder.release();
    return PLK_OK;
}

// exports the circuit in the reference's own file formats (iden3 .r1cs v1 / .wtns v2), so that the
// same synthetic input can be fed to a real `plonkit` binary (PLONKIT_REF_BIN)
static int32_t circuit_export_impl(const plk_circuit *c, int32_t what, uint8_t *out, uint64_t cap, uint64_t *len);
extern "C" int32_t plk_circuit_export(const plk_circuit *c, int32_t what, uint8_t *out, uint64_t cap, uint64_t *len) {
    if (!c || !len) { set_error("plk_circuit_export: bad argument"); return PLK_ERR_ARG; }
    return guarded("plk_circuit_export", PLK_ERR_ARG, [&] { return circuit_export_impl(c, what, out, cap, len); });
}
static int32_t circuit_export_impl(const plk_circuit *c, int32_t what, uint8_t *out, uint64_t cap, uint64_t *len) {
    std::vector<uint8_t> b;
    auto u32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); };
    auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); };
    auto fr = [&](const HFr &v) { uint64_t cc[4]; v.to_canonical(cc); for (int i = 0; i < 4; i++) u64(cc[i]); };
    if (what == 0) {                                                    // .r1cs
        b.insert(b.end(), {'r', '1', 'c', 's'}); u32(1); u32(3);
        u32(1); u64(64); u32(32); b.insert(b.end(), BN254_R_LE, BN254_R_LE + 32);
        u32((uint32_t)c->r1cs.num_variables); u32(0); u32((uint32_t)c->r1cs.num_inputs - 1); u32((uint32_t)c->r1cs.num_aux);
        const R1cs &R = c->r1cs;
        u64(R.num_variables); u32((uint32_t)R.num_constraints());
        u32(2); u64(12 * (uint64_t)R.num_constraints() + 36 * (uint64_t)R.terms.size());
        b.reserve(b.size() + 12 * R.num_constraints() + 36 * R.terms.size() + 8 * R.num_variables + 64);
        for (size_t i = 0; i < R.num_constraints(); i++)
            for (int which = 0; which < 3; which++) { const LcView lc = R.lc(i, which); u32((uint32_t)lc.size()); for (const LcTerm &t : lc) { u32(t.wire); fr(t.coeff); } }
        u32(3); u64(8 * c->r1cs.num_variables);
        for (uint64_t i = 0; i < c->r1cs.num_variables; i++) u64(i);
    } else {                                                            // .wtns
        if (!c->has_witness) { set_error("plk_circuit_export: no witness"); return PLK_ERR_ARG; }
        b.insert(b.end(), {'w', 't', 'n', 's'}); u32(2); u32(2);
        u32(1); u64(40); u32(32); b.insert(b.end(), BN254_R_LE, BN254_R_LE + 32); u32((uint32_t)c->witness.size());
        u32(2); u64(32 * c->witness.size());
        for (const HFr &v : c->witness) fr(v);
    }
    *len = b.size();
    if (out) { if (b.size() > cap) { set_error("plk_circuit_export: buffer too small"); return PLK_ERR_ARG; } memcpy(out, b.data(), b.size()); }
    return PLK_OK;
}
