#!/bin/bash
# Copies the summaries of a finished GPU session (tools/gpu_final.sh <tag>: gpurun_out/<tag>/, gpurun_out/<tag>_final/) into profiles/
# under their per-round names and regenerates the floor table from them.
#   usage: bash tools/collect_round.sh <tag>
set -e
cd "$(dirname "$0")/.."
T=${1:-r05}; O=gpurun_out/$T; P=profiles
cp $O/bench.json $P/${T}_bench_c3_final.json
cp $O/bench_detail.json $P/${T}_bench_c3_final_detail.json
cp $O/bench_graph.json $P/${T}_bench_graph.json
cp $O/bench_c2.json $P/${T}_bench_c2.json
cp $O/bench_c5.json $P/${T}_bench_c5.json
cp $O/bench_c3_1k.json $P/${T}_bench_c3_1k.json
cp $O/bench_c3_1k_graph.json $P/${T}_bench_c3_1k_graph.json
cp $O/bench_dist1.json $P/${T}_bench_dist1_forced_rank.json
cp $O/bench_dist1_bf16.json $P/${T}_bench_dist1_forced_rank_bf16.json
cp $O/kernel_stats.txt $P/${T}_kernel_stats_final.txt
for c in rd wr l2 mfma sq inst; do cp $O/pmc_$c.txt $P/${T}_pmc_$c.txt; done
cp $O/hbm_traffic_$T.json $P/hbm_traffic_$T.json
cp $O/${T}_mfma_pmc.json $P/${T}_mfma_pmc.json
cp $O/${T}_counter_calibration.txt $P/${T}_counter_calibration.txt
{ echo "library sha256[:16] $(cat gpurun_out/${T}_final/lib_sha.txt); python -m pytest tests -m gpu -q -rfE --tb=short (tools/gpu_final.sh $T)"; tail -n 6 gpurun_out/${T}_final/pytest.log; } > $P/${T}_pytest_gpu_final.txt
python tools/floor_table.py $P/${T}_bench_c3_final_detail.json $O/pmc_inst.json $P/hbm_traffic_$T.json --stats $P/${T}_kernel_stats_final.txt --md > $P/${T}_floor_table.md
tail -n 2 $P/${T}_pytest_gpu_final.txt
