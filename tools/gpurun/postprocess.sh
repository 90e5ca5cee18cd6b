#!/bin/bash
# after tools/gpurun/r04_final.sh <tag> has been merged back: copy what is judged from gpurun_out/<tag> into profiles/<tag>_*
T=${1:?tag}
O=gpurun_out/$T
grep '^{' $O/bench_line.json | tail -1 > profiles/${T}_bench_line.json
cp $O/trace/run_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
python3 tools/trace_by_operator.py $O/trace/run_kernel_trace.csv profiles/${T}_kernel_by_operator.csv
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write profiles/$T > /dev/null
cp $O/series_b.jsonl profiles/${T}_series_b.jsonl
[ -f $O/series_b_rank_blocks.jsonl ] && cp $O/series_b_rank_blocks.jsonl profiles/${T}_series_b_rank_blocks.jsonl
cp $O/series_b_kernel_by_operator.csv profiles/${T}_series_b_kernel_by_operator.csv
cp $O/trace_b/run_kernel_stats.csv profiles/${T}_series_b_kernel_stats.csv
tail -3 gpurun_out/${T}_suite/t_all.log | head -1
python3 - $T <<'PYEOF'
import json, sys
T = sys.argv[1]
d = json.load(open(f"profiles/{T}_bench_line.json"))
t = json.load(open("profiles/traffic.json"))
print("ms_per_step", round(d["ms_per_step"], 2), "value %.3e" % d["value"], "iters", d["iters"], "iters_match", d.get("iters_match"), "setup", round(d["setup_ms"], 1), "/", round(d["setup_cold_ms"], 1), "seam", round(d["seam"]["ms_per_step"], 2))
rf = d["roofline"]
print("roofline avg_ms", round(rf["avg_ms"], 4), "frac", round(rf["frac"], 4), "traffic GB", round((rf["traffic"] or 0) / 1e9, 3), "| separate passes", round(t.get("k_spmv_stream_jacobi_level1_bytes_per_launch", 0) / 1e9, 3))
print("solve_phase_hbm_frac", round(d["solve_phase_hbm_frac"], 4), "csr_equiv", round(d["solve_phase_csr_equiv_frac"], 4), "traffic_gb", round(d.get("solve_phase_traffic_gb") or 0, 1), "over_format", round(d.get("traffic_over_format") or 0, 4))
print("plain", round(d["plain_csr"]["ms_per_step"], 2), "uncoded", round(d["uncoded"]["ms_per_step"], 2), "agg", round(d["aggressive_1"]["ms_per_step"], 2), "cpu_baseline solve_s", round(d["cpu_baseline"]["solve_s"], 2))
cd = d.get("cpu_defaults", {})
print("cpu_defaults", {k: cd.get(k) for k in ("V", "iters", "iters_match")}, round(cd.get("ms_per_step", 0), 1), round(cd.get("setup_ms", 0), 0), "extras_skipped", d.get("extras_skipped"))
for l in open(f"profiles/{T}_series_b.jsonl"):
    q = json.loads(l)
    print("series B", q["grid"], "V", q["V"], "iters", q["iters"], round(q["ms_per_step"], 2), "ms, setup", round(q["setup_ms"], 0))
import os
if os.path.exists(f"profiles/{T}_series_b_rank_blocks.jsonl"):
    for l in open(f"profiles/{T}_series_b_rank_blocks.jsonl"):
        q = json.loads(l)
        print("series B rank blocks", q["grid"], "V", q["V"], "iters", q["iters"], round(q["ms_per_step"], 2), "ms, setup", round(q["setup_ms"], 0))
PYEOF
grep 'k_spmv_win<2, false, false, false, false>",0' profiles/${T}_kernel_by_operator.csv
