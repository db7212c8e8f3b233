"""Dev tool: reduce rocprofv3 --pmc counter_collection CSVs (one counter per pass) to per-launch HBM bytes of the NN kernels.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE
reports half of a wide coalesced read (x2); WRITE_SIZE is left uncorrected."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_hash import search_kernel_source_sha256

root, size = sys.argv[1], sys.argv[2]
KEY = "x".join(f"{int(v) // 1000}k" for v in size.split("x"))  # bench.py workload name, e.g. 200kx200k
KERNELS = {"grid": "nn_quad_kernel<false, true, false", "brute": "nn_brute_bf16_kernel"}  # 200k: the quad kernel; brute: the matrix-core kernel (bf16 bound)


def per_launch(mode, ctr):
    vals = []
    for path in glob.glob(os.path.join(root, f"{mode}_{ctr}", "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            rows = []
            for r in rd:
                if KERNELS[mode] in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    rows.append((r["Kernel_Name"], ctr, float(r["Counter_Value"]), r.get("Start_Timestamp", ""),
                                 r.get("End_Timestamp", ""), r.get("Grid_Size", "")))
            # a dispatch's counter may be split over several rows (one per XCD / dimension): sum per dispatch
        by_dispatch = {}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if KERNELS[mode] in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    by_dispatch[r["Dispatch_Id"]] = by_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        vals += list(by_dispatch.values())
        with open(os.path.join(root, f"{mode}_{ctr}.csv"), "w") as out:
            out.write("Kernel,Counter,Value,Start,End,Grid_Size\n")
            for k, c, v, s, e, g in rows:
                out.write(f'"{KERNELS[mode]}",{c},{v:.6f},{s},{e},{g}\n')
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (with --kernel-trace only) over "
               f"scripts/one_align.py {size} {{grid,brute}}; mean per launch of the named kernel. Units: the counters are in "
               "KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); "
               "WRITE_SIZE is uncorrected (uncalibrated per the guide).",
       "kernel_source_sha256": search_kernel_source_sha256(), KEY: {}}
blk = res[KEY]
for mode in ("grid", "brute"):
    f, nf = per_launch(mode, "FETCH_SIZE")
    w, nw = per_launch(mode, "WRITE_SIZE")
    if f is None or w is None:
        blk[f"nn_{mode}"] = None
        continue
    d = {"fetch_size_kb": f, "write_size_kb": w, "fetch_bytes_raw": f * 1024, "fetch_bytes_corrected": 2 * f * 1024,
         "write_bytes": w * 1024, "total": 2 * f * 1024 + w * 1024, "launches": nf}
    blk[f"nn_{mode}"] = d
    blk[f"nn_{mode}_hbm_bytes_per_launch"] = d["total"]
print(json.dumps(res, indent=1))
