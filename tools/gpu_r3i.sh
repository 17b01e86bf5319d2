#!/bin/bash
# round 3, run I: SQ counters per kernel (LDS bank conflicts, VALU / LDS / VMEM busy, waits)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r3i}
mkdir -p $O
S="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --variant-steps 0 --profile-steps 0"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/sq$i -o p -- $S > $GRAFT_REPO_ROOT/$O/sq$i.log 2>&1 )
  python tools/rocpd_pmc.py $(find $O/sq$i -name "*.db" | head -1) --json $O/sq$i.json > $O/sq$i.txt 2>&1
  echo "pass $i: $(wc -l < $O/sq$i.txt) lines"; tail -2 $O/sq$i.log | cut -c1-200
  rm -rf $O/sq$i
done
python - $O <<'PY'
import json, sys
o = sys.argv[1]
a, b = json.load(open(o + "/sq1.json")), json.load(open(o + "/sq2.json"))
def pl(d, k, c):
    v = d.get(k, {}).get(c)
    return v["sum"] / v["launches"] if v and v["launches"] else float("nan")
rows = []
for k in a:
    wc = pl(a, k, "SQ_WAVE_CYCLES")
    if not wc == wc or wc < 1e7:
        continue
    rows.append((wc, k))
print("%-58s %9s %6s %6s %6s %6s %7s %7s %7s" % ("kernel (per launch)", "wavecyc", "wait%", "stall%", "valu%", "lds%", "bankcf%", "vmem%", "IPCv"))
for wc, k in sorted(rows, reverse=True)[:26]:
    f = lambda c: 100.0 * pl(a, k, c) / wc
    bc, ia = pl(a, k, "SQ_LDS_BANK_CONFLICT"), pl(a, k, "SQ_LDS_IDX_ACTIVE")
    print("%-58s %9.3g %6.1f %6.1f %6.1f %6.1f %7.1f %7.1f %7.2f" % (k[:58], wc, f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_ANY"), f("SQ_ACTIVE_INST_VALU"), f("SQ_ACTIVE_INST_LDS"),
          100.0 * bc / ia if ia else float("nan"), 100.0 * pl(b, k, "SQ_ACTIVE_INST_VMEM") / wc, pl(b, k, "SQ_INSTS_VALU") / max(pl(a, k, "SQ_BUSY_CYCLES"), 1)))
PY
ls $O
