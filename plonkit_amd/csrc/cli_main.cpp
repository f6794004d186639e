// `plonkit` command line over the C ABI — the five prover commands of the reference's CLI
// (src/bin/main.rs:27-53): setup, dump-lagrange, prove, export-verification-key, analyse, verify.
// Same option names, short flags and defaults (src/bin/main.rs:55-136,176-190), same refusal to overwrite
// (src/bin/main.rs:336-339,374-377,403-406) and the circuit-file default rule (src/bin/main.rs:346-357).
// Everything arithmetic goes through include/plonkit_amd.h.
//
// Multi-GPU (an extension; the reference is one process): start one `plonkit prove` / `export-verification-key` per GPU
// with PLONKIT_WORLD=<ranks> PLONKIT_RANK=<r> and PLONKIT_COMM=rccl:<id file> (rank 0 writes the RCCL unique id there, the
// others wait for it; PLONKIT_RUN_ID=<nonce of this run> on all ranks is mandatory with it) or PLONKIT_COMM=tcp:<port> (ranks sharing one device).  Rank r uses device PLONKIT_DEVICE or
// r mod #devices, keeps only its 1/world slice of the key resident and computes its share of every commitment
// (plk_comm_init, include/plonkit_amd.h); every rank derives the same bytes and rank 0 writes the files.
#include "../../include/plonkit_amd.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <ctime>
#include <sys/stat.h>
#include <unistd.h>

// A fatal error ends the process with the Rust panic exit code — but only ever from the MAIN thread: exit() on a helper
// thread would run static destructors and atexit handlers (HIP, RCCL, libstdc++) under the feet of the main thread.  On the
// start-up GPU thread fatal() throws instead; the thread records it and main reports it after join() (worker_guard below).
struct Fatal { int code; std::string msg; };
static thread_local bool t_in_worker = false;
static std::thread *g_helper = nullptr;                       // the start-up GPU thread while it may be running
[[noreturn]] static void fatal(int code, const std::string &msg) {
    if (t_in_worker) throw Fatal{code, msg};
    fprintf(stderr, "%s\n", msg.c_str());
    if (g_helper && g_helper->joinable()) g_helper->join();  // (exit() does not unwind: wait for the helper to leave HIP first)
    exit(code);
}
static bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static bool ends_with(const std::string &s, const char *suf) { size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
static std::vector<uint8_t> slurp(const std::string &p, const char *what) {           // one read() of the whole file
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) fatal(101, std::string(what) + ": cannot open " + p);
    std::vector<uint8_t> data;
    struct stat st;
    if (fstat(fileno(f), &st) == 0 && st.st_size > 0) {
        data.resize((size_t)st.st_size);
        size_t got = fread(data.data(), 1, data.size(), f);
        data.resize(got);
    } else {                                                                            // not a regular file: stream it
        uint8_t buf[1 << 16];
        for (size_t got; (got = fread(buf, 1, sizeof buf, f)) > 0;) data.insert(data.end(), buf, buf + got);
    }
    fclose(f);
    return data;
}
static void spit(const std::string &p, const uint8_t *d, size_t n) { std::ofstream f(p, std::ios::binary); f.write((const char *)d, (std::streamsize)n); }
static void die(const char *what, int32_t rc) { fatal(101, std::string(what) + ": " + plk_last_error() + " (status " + std::to_string(rc) + ")"); }   // Rust panic exit code
// runs `body` on a helper thread's behalf: a Fatal raised inside is parked in *err (anything else too) instead of ending the process there
template <class F> static void worker_guard(Fatal *err, F &&body) {
    t_in_worker = true;
    try { body(); }
    catch (const Fatal &f) { *err = f; }
    catch (const std::exception &e) { *err = Fatal{101, std::string("start-up thread: ") + e.what()}; }
    catch (...) { *err = Fatal{101, "start-up thread: unknown failure"}; }
    t_in_worker = false;
}
static void rethrow_on_main(const Fatal &err) { if (err.code) fatal(err.code, err.msg); }
struct JoinOnExit { std::thread &t; ~JoinOnExit() { if (t.joinable()) t.join(); } };      // a thread object must not die joinable, whatever unwinds past it
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double g_t0 = 0;
// PLK_CLI_TIMING=1: phase times on stderr (whole-CLI measurement of DESIGN.md §4)
static void phase(const char *what) {
    if (!getenv("PLK_CLI_TIMING")) return;
    double t = now_s();
    if (g_t0 == 0) g_t0 = t;
    static double last = 0; if (last == 0) last = t;
    fprintf(stderr, "[timing] %-28s +%.3f s (%.3f)\n", what, t - last, t - g_t0);
    last = t;
}
#define CK(what, expr) do { int32_t _rc = (expr); if (_rc != PLK_OK) die(what, _rc); } while (0)

struct Args {
    std::map<std::string, std::string> kv; bool overwrite = false;
    std::string get(const char *k, const char *def = nullptr) const {
        auto it = kv.find(k);
        if (it != kv.end()) return it->second;
        if (def) return def;
        fprintf(stderr, "error: The following required argument was not provided: --%s\n", k); exit(2);
    }
    bool has(const char *k) const { return kv.count(k) != 0; }
};
static Args parse(int argc, char **argv, const std::map<std::string, std::string> &shorts) {
    Args a;
    for (int i = 2; i < argc; i++) {
        std::string s = argv[i];
        if (s == "--overwrite") { a.overwrite = true; continue; }
        std::string key;
        // an entry "canonical|alias" accepts both long names and stores the value under the canonical one
        auto canonical = [](const std::string &v) { size_t bar = v.find('|'); return bar == std::string::npos ? v : v.substr(0, bar); };
        if (s.rfind("--", 0) == 0) {
            const std::string name = s.substr(2);
            bool known = false;
            for (auto &kv : shorts) {
                size_t bar = kv.second.find('|');
                if (kv.second == name || (bar != std::string::npos && (kv.second.substr(0, bar) == name || kv.second.substr(bar + 1) == name))) { known = true; key = canonical(kv.second); }
            }
            if (!known) { fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", s.c_str()); exit(2); }
        } else if (s.size() == 2 && s[0] == '-' && shorts.count(s.substr(1))) key = canonical(shorts.at(s.substr(1)));
        else { fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", s.c_str()); exit(2); }
        if (i + 1 >= argc) { fprintf(stderr, "error: The argument '%s' requires a value\n", s.c_str()); exit(2); }
        a.kv[key] = argv[++i];
    }
    return a;
}
static std::string resolve_circuit(const Args &a) {            // src/bin/main.rs:346-357
    if (a.has("circuit")) return a.get("circuit");
    return (exists("circuit.r1cs") || !exists("circuit.json")) ? "circuit.r1cs" : "circuit.json";
}
static void refuse_duplicate(const Args &a, const std::string &path, const char *what) {
    if (!a.overwrite && exists(path)) { fprintf(stderr, "duplicate %s file: %s\n", what, path.c_str()); exit(101); }
}
static plk_circuit *load_circuit(const std::string &cf, const std::string *wf) {
    fprintf(stderr, "Loading circuit from %s...\n", cf.c_str());
    std::vector<uint8_t> r = slurp(cf, "unable to open."), w;
    if (wf) w = slurp(*wf, "unable to open.");
    plk_circuit *c = nullptr;
    CK("load circuit", plk_circuit_load(r.data(), r.size(), ends_with(cf, "json"), wf ? w.data() : nullptr, w.size(), wf && ends_with(*wf, "json"), &c));
    return c;
}
// proof.json / public.json of `prove` (src/bin/main.rs:410-424): bellman_vk_codegen::serialize_proof gives the public
// inputs and the 33 words of the Solidity verifier's deserialize_proof (contrib/template.sol:864-951: the proof.bin
// fields in the same order, without the length words, G1 as (x, y), infinity as (0, 0)), printed by
// serde_json::to_string_pretty.  UNPINNED: the crate is not in the reference tree and no fixture holds these files; the
// number format written here is web3's U256 one ("0x" + hex without leading zeros), which ethers accepts.
static std::string hex_word(const uint8_t *be32) {
    static const char *d = "0123456789abcdef";
    std::string s = "0x";
    bool started = false;
    for (int i = 0; i < 32; i++) {
        const int hi = be32[i] >> 4, lo = be32[i] & 15;
        if (started || hi) { s += d[hi]; started = true; }
        if (started || lo) { s += d[lo]; started = true; }
    }
    if (!started) s += '0';
    return s;
}
static std::string json_array(const std::vector<std::string> &items) {
    if (items.empty()) return "[]";
    std::string s = "[\n";
    for (size_t i = 0; i < items.size(); i++) s += "  \"" + items[i] + "\"" + (i + 1 < items.size() ? ",\n" : "\n");
    return s + "]";
}
static bool proof_words(const uint8_t *p, size_t len, std::vector<std::string> *inputs, std::vector<std::string> *words) {
    size_t off = 0;
    auto u64 = [&](uint64_t *v) { if (off + 8 > len) return false; *v = 0; for (int i = 0; i < 8; i++) *v = (*v << 8) | p[off + i]; off += 8; return true; };
    auto fr = [&](std::vector<std::string> *dst) { if (off + 32 > len) return false; dst->push_back(hex_word(p + off)); off += 32; return true; };
    auto g1 = [&]() {
        if (off + 64 > len) return false;
        static const uint8_t zero[32] = {0};
        const bool inf = (p[off] & 0x40) != 0;
        words->push_back(hex_word(inf ? zero : p + off)); words->push_back(hex_word(inf ? zero : p + off + 32));
        off += 64; return true;
    };
    uint64_t n, k;
    if (!u64(&n) || !u64(&k)) return false;
    for (uint64_t i = 0; i < k; i++) if (!fr(inputs)) return false;
    if (!u64(&k) || k != 4) return false;
    for (int i = 0; i < 4; i++) if (!g1()) return false;
    if (!g1()) return false;
    if (!u64(&k) || k != 4) return false;
    for (int i = 0; i < 4; i++) if (!g1()) return false;
    if (!u64(&k) || k != 4) return false;
    for (int i = 0; i < 4; i++) if (!fr(words)) return false;
    if (!u64(&k) || k != 1) return false;
    for (int i = 0; i < 4; i++) if (!fr(words)) return false;      // d(z*omega), z(z*omega), t(z), r(z)
    if (!u64(&k) || k != 3) return false;
    for (int i = 0; i < 3; i++) if (!fr(words)) return false;
    return g1() && g1() && off == len;
}

struct Ranks { int rank = 0, world = 1; std::string comm; };
static Ranks ranks_from_env() {
    Ranks r;
    if (const char *w = getenv("PLONKIT_WORLD")) r.world = atoi(w);
    if (const char *k = getenv("PLONKIT_RANK")) r.rank = atoi(k);
    if (const char *c = getenv("PLONKIT_COMM")) r.comm = c;
    if (r.world < 1 || r.rank < 0 || r.rank >= r.world) { fprintf(stderr, "PLONKIT_RANK / PLONKIT_WORLD out of range\n"); exit(2); }
    // (a rank other than 0 can only tell this run's id file from a crashed run's leftover by the nonce inside it)
    if (r.world > 1 && r.comm.rfind("rccl:", 0) == 0 && !(getenv("PLONKIT_RUN_ID") && *getenv("PLONKIT_RUN_ID"))) {
        fprintf(stderr, "PLONKIT_WORLD > 1 with PLONKIT_COMM=rccl:<id file> needs PLONKIT_RUN_ID=<nonce of this run> on every rank\n"); exit(2); }
    if (r.world > 1 && r.comm.rfind("rccl:", 0) != 0 && r.comm.rfind("tcp:", 0) != 0) { fprintf(stderr, "PLONKIT_WORLD > 1 needs PLONKIT_COMM=rccl:<id file> or tcp:<port>\n"); exit(2); }
    return r;
}
static plk_ctx *open_ctx(const Ranks &rk = Ranks()) {
    int dev = 0;
    if (const char *d = getenv("PLONKIT_DEVICE")) dev = atoi(d);
    else if (rk.world > 1) { int n = plk_device_count(); dev = n > 0 ? rk.rank % n : 0; }
    plk_ctx *ctx = nullptr; CK("plk_create", plk_create(dev, &ctx)); return ctx;
}
// joins the ranks: commitments over the N-point domain are split into world contiguous slices of N / world SRS points
static void join_ranks(plk_ctx *ctx, const Ranks &rk, uint64_t N) {
    if (rk.world == 1 && rk.comm.empty()) return;                  // (a communicator of one rank is allowed: it exercises the transport)
    if (N % (uint64_t)rk.world) { fprintf(stderr, "PLONKIT_WORLD must divide the domain size\n"); exit(101); }
    const uint64_t first = (uint64_t)rk.rank * (N / rk.world);
    if (rk.comm.rfind("tcp:", 0) == 0) { CK("plk_comm_init_tcp", plk_comm_init_tcp(ctx, rk.rank, rk.world, (uint16_t)atoi(rk.comm.c_str() + 4), first)); return; }
    // The id file is 128 bytes of ncclUniqueId followed by the run id (PLONKIT_RUN_ID, may be empty).  A file left behind
    // by an earlier run must never be taken for this run's: rank 0 unlinks the path before it does anything else and
    // removes the file again once the communicator exists (every rank has read it by then: ncclCommInitRank is
    // collective); the other ranks accept only a file that carries THEIR run id — mandatory for more than one rank (a rank
    // cannot tell a crashed run's leftover, written seconds ago, from this run's file by its age alone).
    const std::string path = rk.comm.substr(5);
    const char *rid_env = getenv("PLONKIT_RUN_ID");
    const std::string run_id = rid_env ? rid_env : "";
    plk_comm_id id;
    if (rk.rank == 0) {
        (void)unlink(path.c_str());
        CK("plk_comm_unique_id", plk_comm_unique_id(&id));
        std::vector<uint8_t> blob(id.bytes, id.bytes + sizeof id.bytes);
        blob.insert(blob.end(), run_id.begin(), run_id.end());
        spit(path + ".tmp", blob.data(), blob.size());
        if (rename((path + ".tmp").c_str(), path.c_str()) != 0) fatal(101, "cannot write " + path);
    } else {
        bool got = false;
        for (int i = 0; i < 1200 && !got; i++) {                                    // up to two minutes for rank 0
            struct stat st;
            if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == sizeof id.bytes + run_id.size()) {
                std::vector<uint8_t> raw = slurp(path, "RCCL unique id");
                if (raw.size() == sizeof id.bytes + run_id.size() && memcmp(raw.data() + sizeof id.bytes, run_id.data(), run_id.size()) == 0) {
                    memcpy(id.bytes, raw.data(), sizeof id.bytes);
                    got = true;
                    break;
                }
            }
            usleep(100000);
        }
        if (!got) fatal(101, "no RCCL id file of this run at " + path + " (rank 0 writes it; PLONKIT_RUN_ID must agree on all ranks)");
    }
    CK("plk_comm_init", plk_comm_init(ctx, rk.rank, rk.world, &id, first));
    if (rk.rank == 0) (void)unlink(path.c_str());
}
// PLK_SHARD_MODE=scatter with several ranks (owner computes, include/plonkit_amd.h): rank 0 runs the command, the others serve its
// commitments from their slice of the key and never touch the witness
static bool scatter_mode(const Ranks &rk) { const char *e = getenv("PLK_SHARD_MODE"); return rk.world > 1 && e && !strcmp(e, "scatter"); }
static int serve_owner(plk_ctx *ctx) {
    // one command, one job: an owner that exits on an error never sends the stop batch, and its workers must not outlive it for ever
    // (the reference's process simply ends when a step panics).  Ten minutes between two batches is far beyond any step of a 2^26 proof.
    setenv("PLK_COMM_IDLE_TIMEOUT_MS", "600000", 0);
    uint64_t batches = 0;
    CK("serve (owner-computes mode)", plk_comm_serve(ctx, &batches));
    fprintf(stderr, "served %llu batches of commitments\n", (unsigned long long)batches);
    return 0;
}
// a key file read and checked on the host (Crs::read: every point on the curve): no GPU needed, so the single-process commands do it on a
// thread of its own WHILE the HIP runtime initialises (0.06 s of the 0.24 s the GPU side of `prove` used to take at the 2^20 domain)
struct ParsedKey { std::vector<plk_g1_affine> pts; uint64_t n = 0; uint8_t g2[256]; };
static void parse_key(const std::string &path, bool lagrange, ParsedKey *k) {
    const char *what = lagrange ? "read key_lagrange_form err" : "read key_monomial_form err";
    std::vector<uint8_t> raw = slurp(path, what);
    CK(what, plk_key_parse(raw.data(), raw.size(), nullptr, 0, &k->n, k->g2));
    k->pts.resize(k->n);
    CK(what, plk_key_parse(raw.data(), raw.size(), k->pts.data(), k->n, &k->n, k->g2));
}
// uploads the key; with several ranks only this rank's slice [rank * N/world, (rank + 1) * N/world) of its first N points
static void upload_key(plk_ctx *ctx, const ParsedKey &k, bool lagrange, const Ranks &rk = Ranks(), uint64_t N = 0) {
    const char *what = lagrange ? "read key_lagrange_form err" : "read key_monomial_form err";
    const plk_g1_affine *first = k.pts.data();
    uint64_t n = k.n;
    if (rk.world > 1) {
        if (n < N) fatal(101, std::string(what) + ": key has " + std::to_string(n) + " points, the domain needs " + std::to_string(N));
        first += (uint64_t)rk.rank * (N / rk.world);
        n = N / rk.world;
    }
    if (lagrange) CK("srs upload", plk_srs_lagrange_upload(ctx, first, n));
    else CK("srs upload", plk_srs_upload(ctx, first, n));
}
static void load_key(plk_ctx *ctx, const std::string &path, uint8_t g2[256], bool lagrange = false, const Ranks &rk = Ranks(), uint64_t N = 0) {
    ParsedKey k;
    parse_key(path, lagrange, &k);
    memcpy(g2, k.g2, 256);
    upload_key(ctx, k, lagrange, rk, N);
}

static int run(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "plonkit (MI355X) — subcommands: analyse setup dump-lagrange prove export-verification-key verify\n"); return 2; }
    std::string cmd = argv[1];
    if (cmd == "analyse") {
        Args a = parse(argc, argv, {{"c", "circuit"}, {"o", "output"}});
        plk_circuit *c = load_circuit(resolve_circuit(a), nullptr);
        std::vector<char> buf(1 << 26);
        CK("analyse failed", plk_circuit_analyse(c, buf.data(), buf.size()));
        std::string out = a.get("output", "analyse.json");
        spit(out, (const uint8_t *)buf.data(), strlen(buf.data()));
        fprintf(stderr, "output to %s\n", out.c_str());
    } else if (cmd == "setup") {                                     // src/bin/main.rs:334-343, src/plonk.rs:30-48
        Args a = parse(argc, argv, {{"p", "power"}, {"m", "srs_monomial_form"}});
        int power = atoi(a.get("power").c_str());
        if (power < 10 || power > 26) { fprintf(stderr, "setup power of two is not in the correct range\n"); return 101; }
        std::string out = a.get("srs_monomial_form");
        phase("start");
        plk_ctx *ctx = open_ctx();
        phase("plk_create (HIP init)");
        uint64_t n = 1ull << power;
        CK("crs_42", plk_srs_generate(ctx, n, 0, 42));
        std::vector<plk_g1_affine> pts(n);
        CK("srs download", plk_srs_download(ctx, 0, n, pts.data()));
        phase("crs_42 + download");
        uint8_t g2[256]; plk_crs42_g2_bytes(g2);
        uint64_t len = 0;
        CK("serialize", plk_key_serialize(pts.data(), n, g2, nullptr, 0, &len));
        std::vector<uint8_t> bytes(len);
        CK("serialize", plk_key_serialize(pts.data(), n, g2, bytes.data(), len, &len));
        phase("serialize");
        refuse_duplicate(a, out, "srs_monomial_form");
        spit(out, bytes.data(), len);
        phase("file written");
        fprintf(stderr, "srs_monomial_form saved to %s\n", out.c_str());
    } else if (cmd == "dump-lagrange") {                             // src/bin/main.rs:360-381
        Args a = parse(argc, argv, {{"m", "srs_monomial_form"}, {"l", "srs_lagrange_form"}, {"c", "circuit"}});
        phase("start");
        // as in `prove`: HIP initialisation and the key (read + parse + upload) on a second thread while this one parses the circuit; of the setup
        // only the domain size is needed (plk_circuit_domain_size: transpile, no columns, no device work) — round 6, 0.55 -> 0.43 s at the 2^20 domain
        const std::string key_path = a.get("srs_monomial_form");
        uint8_t g2[256];
        Fatal gpu_err{0, ""};
        plk_ctx *ctx = nullptr;
        std::thread gpu([&gpu_err, &ctx, &g2, key_path] {
            worker_guard(&gpu_err, [&] {
                ParsedKey key;
                Fatal key_err{0, ""};
                std::thread reader([&] { worker_guard(&key_err, [&] { parse_key(key_path, false, &key); }); });
                JoinOnExit reader_guard{reader};
                ctx = open_ctx();
                reader.join();
                if (key_err.code) fatal(key_err.code, key_err.msg);
                memcpy(g2, key.g2, 256);
                upload_key(ctx, key, false);
            });
        });
        g_helper = &gpu;
        plk_circuit *c = load_circuit(resolve_circuit(a), nullptr);
        phase("load circuit");
        uint64_t N = 0;
        CK("prepare err", plk_circuit_domain_size(c, &N));
        phase("domain size (transpile)");
        gpu.join();
        g_helper = nullptr;
        rethrow_on_main(gpu_err);
        phase("HIP init + key (other thread)");
        uint32_t log_n = 0; while ((1ull << log_n) < N) log_n++;
        if (plk_srs_size(ctx) < N) { fprintf(stderr, "SRS too small for the circuit domain\n"); return 101; }
        std::vector<plk_g1_affine> mono(N), lag(N);
        CK("srs download", plk_srs_download(ctx, 0, N, mono.data()));
        CK("from_powers", plk_g1_intt(ctx, mono.data(), log_n, lag.data()));
        phase("download + G1 iNTT");
        uint64_t len = 0;
        CK("serialize", plk_key_serialize(lag.data(), N, g2, nullptr, 0, &len));
        std::vector<uint8_t> bytes(len);
        CK("serialize", plk_key_serialize(lag.data(), N, g2, bytes.data(), len, &len));
        phase("serialize");
        std::string out = a.get("srs_lagrange_form");
        refuse_duplicate(a, out, "srs_lagrange_form");
        spit(out, bytes.data(), len);
        phase("file written");
        fprintf(stderr, "srs_lagrange_form saved to %s\n", out.c_str());
    } else if (cmd == "export-verification-key") {                   // src/bin/main.rs:484-504
        Args a = parse(argc, argv, {{"m", "srs_monomial_form"}, {"c", "circuit"}, {"v", "vk"}});
        const Ranks rk = ranks_from_env();
        uint8_t g2[256];
        plk_ctx *ctx = nullptr;
        plk_setup *s = nullptr;
        plk_circuit *c = nullptr;
        if (rk.world == 1 && rk.comm.empty()) {                      // as in `prove`: the GPU comes up while the circuit is parsed and transpiled
            const std::string key_path = a.get("srs_monomial_form");   // (a missing option exits here, on the main thread)
            Fatal gpu_err{0, ""};
            plk_ctx *gpu_ctx = nullptr;
            std::thread gpu([&gpu_err, &gpu_ctx, &g2, rk, key_path] {
                worker_guard(&gpu_err, [&] {
                    ParsedKey key;                                      // read + checked beside the HIP initialisation
                    Fatal key_err{0, ""};
                    std::thread reader([&] { worker_guard(&key_err, [&] { parse_key(key_path, false, &key); }); });
                    JoinOnExit reader_guard{reader};
                    gpu_ctx = open_ctx(rk);
                    reader.join();
                    if (key_err.code) fatal(key_err.code, key_err.msg);
                    memcpy(g2, key.g2, 256);
                    upload_key(gpu_ctx, key, false);
                    CK("srs precompute", plk_srs_precompute(gpu_ctx));
                });
            });
            g_helper = &gpu;                                            // fatal() on this thread waits for it before exit()
            c = load_circuit(resolve_circuit(a), nullptr);
            CK("prepare err", plk_setup_prepare_host(c, &s));
            gpu.join();
            g_helper = nullptr;
            rethrow_on_main(gpu_err);
            ctx = gpu_ctx;
            CK("prepare err", plk_setup_upload(ctx, s));
        } else {
            c = load_circuit(resolve_circuit(a), nullptr);
            ctx = open_ctx(rk);
            if (scatter_mode(rk) && rk.rank != 0) CK("prepare err", plk_setup_prepare_host(c, &s));       // a worker only needs the domain size
            else CK("prepare err", plk_setup_prepare(ctx, c, &s));
            join_ranks(ctx, rk, plk_setup_domain_size(s));
            load_key(ctx, a.get("srs_monomial_form"), g2, false, rk, plk_setup_domain_size(s));
            if (scatter_mode(rk) && rk.rank != 0) return serve_owner(ctx);
        }
        std::vector<uint8_t> buf(4096); uint64_t len = 0;
        CK("make_verification_key", plk_setup_write_vk(ctx, s, g2, buf.data(), buf.size(), &len));
        if (scatter_mode(rk)) CK("stop workers", plk_comm_stop_workers(ctx));
        std::string out = a.get("vk", "vk.bin");
        if (rk.rank == 0) {
            refuse_duplicate(a, out, "vk");
            spit(out, buf.data(), len);
            fprintf(stderr, "Verification key saved to %s\n", out.c_str());
        }
    } else if (cmd == "prove") {                                     // src/bin/main.rs:384-424
        Args a = parse(argc, argv, {{"m", "srs_monomial_form"}, {"l", "srs_lagrange_form"}, {"c", "circuit"}, {"w", "witness"},
                                    {"p", "proof"}, {"j", "proofjson"}, {"i", "publicjson"}, {"t", "transcript"}});
        if (a.get("transcript", "keccak") != "keccak") { fprintf(stderr, "not implemented: transcript '%s' (only keccak; rescue needs franklin-crypto)\n", a.get("transcript").c_str()); return 101; }
        phase("start");
        const Ranks rk = ranks_from_env();
        std::string wf = a.get("witness", "witness.wtns");
        const bool single = rk.world == 1 && rk.comm.empty();
        uint8_t g2[256];
        plk_ctx *ctx = nullptr;
        plk_setup *s = nullptr;
        plk_circuit *c = nullptr;
        // a Lagrange-form key (-l) changes how the witness commitments are computed (commit_using_values), never the
        // proof bytes (src/plonk.rs:138-146); an empty or missing option means "monomial only" as in the reference
        const std::string lag = a.get("srs_lagrange_form", "");
        if (single) {
            // the GPU side of the start-up (HIP initialisation, key parse + upload, fixed-base table of the MSM) runs on a
            // second thread while this one parses the circuit and the witness: neither needs the other until the setup
            const std::string key_path = a.get("srs_monomial_form");   // (a missing option exits here, on the main thread)
            Fatal gpu_err{0, ""};
            plk_ctx *gpu_ctx = nullptr;
            std::thread gpu([&gpu_err, &gpu_ctx, &g2, rk, key_path, lag] {
                worker_guard(&gpu_err, [&] {
                    const bool tm = getenv("PLK_CLI_TIMING") != nullptr;
                    double t0 = now_s(), t1;
                    auto lap = [&](const char *what) { if (tm) { t1 = now_s(); fprintf(stderr, "[timing]   gpu thread: %-24s +%.3f s\n", what, t1 - t0); t0 = t1; } };
                    ParsedKey key, lkey;                                // read + checked beside the HIP initialisation
                    Fatal key_err{0, ""};
                    std::thread reader([&] { worker_guard(&key_err, [&] { parse_key(key_path, false, &key); if (!lag.empty()) parse_key(lag, true, &lkey); }); });
                    JoinOnExit reader_guard{reader};
                    gpu_ctx = open_ctx(rk);
                    lap("plk_create (HIP init)");
                    reader.join();
                    if (key_err.code) fatal(key_err.code, key_err.msg);
                    memcpy(g2, key.g2, 256);
                    upload_key(gpu_ctx, key, false);
                    if (!lag.empty()) upload_key(gpu_ctx, lkey, true);
                    lap("key upload (read + parse ran beside HIP init)");
                    CK("srs precompute", plk_srs_precompute(gpu_ctx));
                    lap("MSM table");
                });
            });
            g_helper = &gpu;                                            // a bad circuit file: fatal() on this thread waits for the helper before exit()
            c = load_circuit(resolve_circuit(a), &wf);
            phase("load circuit + witness");
            CK("prepare err", plk_setup_prepare_host(c, &s));             // pure CPU: does not wait for the GPU either
            phase("setup: host phase");
            gpu.join();
            g_helper = nullptr;
            rethrow_on_main(gpu_err);
            ctx = gpu_ctx;
            phase("HIP init + key + table (other thread)");
            CK("prepare err", plk_setup_upload(ctx, s));
            phase("setup: device phase");
        } else {
            c = load_circuit(resolve_circuit(a), &wf);
            phase("load circuit + witness");
            ctx = open_ctx(rk);
            phase("plk_create (HIP init)");
            const bool worker = scatter_mode(rk) && rk.rank != 0;
            if (worker) CK("prepare err", plk_setup_prepare_host(c, &s));  // a worker only needs the domain size
            else CK("prepare err", plk_setup_prepare(ctx, c, &s));         // the slice of the key depends on the domain size
            phase("setup_prepare");
            join_ranks(ctx, rk, plk_setup_domain_size(s));
            const uint64_t N = plk_setup_domain_size(s);
            load_key(ctx, a.get("srs_monomial_form"), g2, false, rk, N);
            if (!lag.empty()) { uint8_t g2l[256]; load_key(ctx, lag, g2l, true, rk, N); }
            phase("load key (parse + upload)");
            if (worker) return serve_owner(ctx);
        }
        if (!s) {
            CK("prepare err", plk_setup_prepare(ctx, c, &s));
            phase("setup_prepare");
        }
        fprintf(stderr, "Proving...\n");
        std::vector<uint8_t> buf(1 << 16); uint64_t len = 0;
        int32_t rc = plk_prove(ctx, s, c, buf.data(), buf.size(), &len);
        if (rc == PLK_ERR_ARG && len > buf.size()) {                 // many public inputs: the call reports the size it needs
            buf.resize(len);
            rc = plk_prove(ctx, s, c, buf.data(), buf.size(), &len);
        }
        if (scatter_mode(rk)) { const int32_t rs = plk_comm_stop_workers(ctx); if (rc == PLK_OK && rs != PLK_OK) die("stop workers", rs); }
        if (rc == PLK_ERR_UNSAT) { fprintf(stderr, "must satisfy: %s\n", plk_last_error()); return 101; }
        if (rc != PLK_OK) die("prove", rc);
        phase("prove");
        if (getenv("PLK_CLI_TIMING")) { CK("sync", plk_synchronize(ctx)); phase("device idle"); }
        if (rk.rank != 0) return 0;                                  // every rank holds the same bytes; rank 0 writes them
        std::string out = a.get("proof", "proof.bin");
        refuse_duplicate(a, out, "proof");
        spit(out, buf.data(), len);
        fprintf(stderr, "Proof saved to %s\n", out.c_str());
        std::vector<std::string> inputs, words;
        if (!proof_words(buf.data(), len, &inputs, &words)) { fprintf(stderr, "serialize_proof: malformed proof\n"); return 101; }
        std::string pj = a.get("proofjson", "proof.json"), ij = a.get("publicjson", "public.json");
        refuse_duplicate(a, pj, "proof json");
        refuse_duplicate(a, ij, "input json");
        std::string ps = json_array(words), is = json_array(inputs);
        spit(pj, reinterpret_cast<const uint8_t *>(ps.data()), ps.size());
        fprintf(stderr, "Proof json saved to %s\n", pj.c_str());
        spit(ij, reinterpret_cast<const uint8_t *>(is.data()), is.size());
        fprintf(stderr, "Public input json saved to %s\n", ij.c_str());
        phase("files written");
        if (getenv("PLK_CLI_FREE")) {                                // (measurement knob: what the exit would otherwise pay for)
            plk_circuit_free(c); phase("free: circuit (unregister)");
            plk_setup_free(s); phase("free: setup");
            plk_destroy(ctx); phase("free: context");
        }
    } else if (cmd == "verify") {                                    // src/bin/main.rs:425-437 (no GPU involved)
        // VerifyOpts (src/bin/main.rs:125-137): the key is `-v` / `--verification_key` here, while export-verification-key
        // names its output `--vk` (src/bin/main.rs:186-187); `--vk` is kept as an alias on verify
        Args a = parse(argc, argv, {{"p", "proof"}, {"v", "verification_key|vk"}, {"t", "transcript"}});
        if (a.get("transcript", "keccak") != "keccak") { fprintf(stderr, "not implemented: transcript '%s' (only keccak; rescue needs franklin-crypto)\n", a.get("transcript").c_str()); return 101; }
        std::vector<uint8_t> vk = slurp(a.get("verification_key", "vk.bin"), "read vk file err"), pr = slurp(a.get("proof", "proof.bin"), "read proof file err");
        int32_t valid = 0;
        CK("fail to verify proof", plk_verify(vk.data(), vk.size(), pr.data(), pr.size(), &valid));
        if (valid) fprintf(stderr, "Proof is valid.\n");
        else { fprintf(stderr, "Proof is invalid!\n"); return 400 & 0xff; }   // std::process::exit(400): the shell sees 144
    } else {
        fprintf(stderr, "error: unrecognized subcommand '%s'\n", cmd.c_str());
        return 2;
    }
    return 0;
}

// Every output file is written and closed inside run(); the process then leaves without tearing down the HIP
// runtime and freeing gigabytes of device memory one buffer at a time (0.25 s at the 2^20 domain) — the
// driver reclaims everything at exit.
int main(int argc, char **argv) {
    // one process per GPU joined over RCCL: on this driver device-memory IPC between processes only works in dmabuf mode, and the
    // switch is read when the HSA runtime initialises — before the first HIP call, whoever launched the ranks
    if (const char *w = getenv("PLONKIT_WORLD")) if (atoi(w) > 1) setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    int rc = run(argc, argv);
    fflush(stdout); fflush(stderr);
    _exit(rc);
}
