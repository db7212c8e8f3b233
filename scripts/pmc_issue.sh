#!/bin/bash
# Instruction-issue counters of the grid search kernel for bench.py's `roofline.issue` (profiles/pmc_issue.json):
# one rocprofv3 --pmc pass (with --kernel-trace only) over scripts/one_align.py <size> grid, mean per launch of
# nn_quad_kernel<false,true,...> (the fused kernel of an alignment's ten sweeps), plus a counting run for the number of
# target points evaluated per launch.   usage: pmc_issue.sh <tag> [sizes...]
TAG=${1:-pmc_issue}; shift
SIZES=${@:-200000x200000 50000x50000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for S in $SIZES; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a_$S -- python $R/scripts/one_align.py $S grid > $O/a_$S.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b_$S -- python $R/scripts/one_align.py $S grid > $O/b_$S.log 2>&1
  python $R/scripts/count_candidates.py $S > $O/cand_$S.txt 2>&1
done
python - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, "$R/scripts")
from kernel_hash import search_kernel_source_sha256
out = {"kernel_source_sha256": search_kernel_source_sha256(),
       "_how": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... --kernel-trace (own pass) over scripts/one_align.py <size> grid: mean per launch of "
               "nn_quad_kernel<false,true,*> over the ten sweeps of each alignment; candidates = target points evaluated per launch "
               "(icpgpu_count_candidates, scripts/count_candidates.py); scripts/pmc_issue.sh"}
for S in "$SIZES".split():
    key = {"200000x200000": "200kx200k", "50000x50000": "50kx50k", "200000x1000000": "200kx1M", "5000x5000": "5kx5k"}.get(S, S)
    acc = collections.defaultdict(list)
    for sub in ("a", "b"):
        for path in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (sub, S), recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(path)):
                if "nn_quad_kernel<false, true" in r["Kernel_Name"]:
                    per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            for (d, c), v in per.items():
                acc[c].append(v)
    m = {c: sum(v) / len(v) for c, v in acc.items()}
    cand = None
    try:
        cand = float(open("$O/cand_%s.txt" % S).read().split("candidates_per_launch")[1].split()[0])
    except Exception:
        pass
    if not m:
        continue
    n_s = int(S.split("x")[0])
    out[key] = {"valu_insts_per_launch": m.get("SQ_INSTS_VALU"), "salu_insts_per_launch": m.get("SQ_INSTS_SALU"),
                "vmem_rd_insts_per_launch": m.get("SQ_INSTS_VMEM_RD"), "lds_insts_per_launch": m.get("SQ_INSTS_LDS"),
                "waves_per_launch": m.get("SQ_WAVES"), "valu_active_cycles_x4": m.get("SQ_ACTIVE_INST_VALU"),
                "wave_cycles": m.get("SQ_WAVE_CYCLES"), "wait_any_cycles": m.get("SQ_WAIT_ANY"),
                "grbm_gui_active_sum_over_xcds": m.get("GRBM_GUI_ACTIVE"), "launches": len(acc.get("SQ_INSTS_VALU", [])),
                "valu_insts_per_source_point": (m.get("SQ_INSTS_VALU", 0.0) / n_s), "candidates_per_launch": cand}
    print(key, json.dumps(out[key]))
json.dump(out, open("$O/pmc_issue.json", "w"), indent=1)
PY
