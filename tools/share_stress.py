"""lifetime / concurrency stress of plk_ctx_share_srs: random sequences of create / share / prove (threads) / re-key / destroy on
one device, every proof checked against the bytes of the same witness proved alone; python tools/share_stress.py [iterations=40] [seed=1]"""
import os, random, sys, threading
sys.path.insert(0, os.path.abspath(os.environ.get("PLK_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plonkit_amd as pa
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
owner = pa.Context(0)
proofs = errors = 0
for it in range(iters):
    log_n = rnd.choice([10, 12, 14, 16])
    lc = rnd.choice([0, 0, 6, 11])
    owner.srs_generate(1 << log_n, 0, 42)                          # (no borrower exists here: allowed)
    k = rnd.choice([1, 2, 3, 4])
    circs = [pa.Circuit.synthetic_ex((1 << log_n) - 2, seed=it + 1, witness_seed=(0 if j == 0 else 50 + j), lc_terms=lc) for j in range(k)]
    setup = pa.SetupForProver(owner, circs[0])
    want = [setup.prove(c) for c in circs] if rnd.random() < 0.7 else None      # sometimes the FIRST proofs of the setup race
    ctxs = [owner] + [pa.Context(0) for _ in range(k - 1)]
    for c in ctxs[1:]:
        c.share_srs_from(owner)
    try:
        owner.srs_generate(1 << log_n, 0, 42)
        if k > 1:
            errors += 1; print("iteration %d: the lender replaced a key on loan" % it)
    except pa.PlkError:
        pass
    got = [[] for _ in range(k)]
    reps = rnd.choice([1, 2, 5])

    def worker(j):
        for _ in range(reps):
            got[j].append(setup.prove(circs[j], ctx=ctxs[j]))
    th = [threading.Thread(target=worker, args=(j,)) for j in range(k)]
    for t in th: t.start()
    for t in th: t.join()
    if want is None:
        want = [setup.prove(c) for c in circs]
    for j in range(k):
        proofs += len(got[j])
        if len(got[j]) != reps or any(p != want[j] for p in got[j]):
            errors += 1; print("iteration %d: prover %d differs" % (it, j))
    order = list(range(1, k)); rnd.shuffle(order)
    if order and rnd.random() < 0.3:                              # a borrower gets a key of its own, proves, and is destroyed later
        b = ctxs[order[0]]
        b.srs_generate(1 << log_n, 0, 42)
        if setup.prove(circs[0], ctx=b) != want[0]:
            errors += 1; print("iteration %d: re-keyed borrower differs" % it)
    for j in order:
        ctxs[j].close()
    setup.close()
    for c in circs: c.close()
print("share_stress: %d iterations, %d concurrent proofs, %d errors" % (iters, proofs, errors))
sys.exit(1 if errors else 0)
