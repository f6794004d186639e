#!/bin/bash
# One parametrised runner for every gpurun call of the project (it replaces the 31 one-shot gpu_r2_*.sh of round 2,
# which stay in the history at commit 9b657a5).  Usage, from the repo root on the GPU box:
#   tools/gpu_run.sh <tag> <stage> [<stage> ...]        results land under gpurun_out/<tag>/
# stages:
#   tests[:<pytest args>]   pytest -m gpu (default: the whole suite; e.g. tests:"tests/test_gpu_prove.py -k zero")
#   smoke                   __graft_entry__.smoke()
#   bench[:<args>]          python bench.py <args> (default: the driver's --gpus 1 --steps 20 --warmup 5)
#   prof-solo | prof-pipe   rocprofv3 --kernel-trace --stats of `bench.py --msm-only` with one / three commitments in flight
#   prof-prove[:<log_n>]    rocprofv3 --kernel-trace --stats of tools/prove_probe.py <log_n> 6
#   prof-py:<name>=<script and args>   rocprofv3 --kernel-trace --stats of `python <script and args>`; summary <name>_kernel_stats.csv
#   pmc:<COUNTER>           one rocprofv3 --pmc pass (kernel trace only) of `bench.py --msm-only --pipeline-depth 1`
#   fuzz | soak             tools/msm_fuzz.py + tools/prove_fuzz.py | tools/soak.py
#   py:<script and args>    python <script and args>  (probes under tools/)
#   sh:<command>            anything else
cd "$(dirname "$0")/.."
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
show='import json,sys
ls=[l for l in sys.stdin.read().splitlines() if l.startswith("{")]
if not ls: sys.exit("no JSON line")
d=json.loads(ls[-1]); r=d["roofline"]
print("value %.1f M/s  ms_per_step %.4f  kernel_ms %.4f  pipelined %.4f  one_in_flight %.4f  sustained %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_pipelined"], r["ms_per_step_one_in_flight"], d.get("sustained", {}).get("value")))
p=d.get("prove")
if p and "rounds_ms" in p: print("prove wall %.4f s rounds %s" % (p["wall_s"], p["rounds_ms"]))
if "kernels" in d: print({k: v["ms"] for k, v in d["kernels"].items()})'
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_$name" -o "$name" -- "$@" > "$O/prof_$name.log" 2>&1
  python tools/rocpd_stats.py "$O/prof_$name/${name}_results.db" "$O/${name}_kernel_stats.csv" && head -12 "$O/${name}_kernel_stats.csv" | cut -c1-60,100-220
}
for st in "$@"; do
  kind=${st%%:*}; arg=""; [[ "$st" == *:* ]] && arg=${st#*:}
  echo "=== $st"
  case $kind in
    tests) ( time eval "timeout 1500 python -m pytest ${arg:-tests} -m gpu -x -q --durations=8" ) > "$O/pytest.log" 2>&1; tail -16 "$O/pytest.log" ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ;;
    bench) ( time timeout 900 python bench.py ${arg:---gpus 1 --steps 20 --warmup 5} ) > "$O/bench.log" 2> "$O/bench.err"; python -c "$show" < "$O/bench.log"; tail -3 "$O/bench.err" ;;
    prof-solo) prof solo python bench.py --msm-only --pipeline-depth 1 --steps 20 --warmup 5 ;;
    prof-pipe) prof pipe python bench.py --msm-only --steps 20 --warmup 5 ;;
    prof-prove) prof prove python tools/prove_probe.py ${arg:-20} 6 ;;
    prof-py) prof "${arg%%=*}" python ${arg#*=} ;;
    pmc) timeout 600 rocprofv3 --kernel-trace --pmc $arg -d "$O/pmc_$arg" -o pmc --output-format csv -- python bench.py --msm-only --pipeline-depth 1 --steps 10 --warmup 2 > "$O/pmc_$arg.log" 2>&1; ls "$O/pmc_$arg" | head ;;
    fuzz) timeout 600 python tools/msm_fuzz.py 120 5 2>&1 | tail -2 | tee "$O/fuzz.txt"; timeout 600 python tools/prove_fuzz.py 40 9 2>&1 | tail -2 | tee -a "$O/fuzz.txt" ;;
    soak) timeout 600 python tools/soak.py 2>&1 | tail -3 | tee "$O/soak.txt" ;;
    py) timeout 900 python $arg 2>&1 | tee "$O/py_$(echo "$arg" | tr -c 'A-Za-z0-9' _ | cut -c1-40).txt" | tail -40 ;;
    sh) timeout 900 bash -c "$arg" 2>&1 | tail -40 ;;
    *) echo "unknown stage $st" ;;
  esac
done
