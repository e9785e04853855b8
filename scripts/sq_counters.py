"""rocprofv3 --pmc passes -> profiles/r04_sq_counters*.json: per-kernel, per-launch averages of every counter collected, after
VALIDATING each pass against the library's calibration kernel (mgs_calibration_kernel: per wave and iteration exactly
64 v_fma_f32 + 8 v_mfma_f32_32x32x2_f32 + 4 ds_read_b32; 256 workgroups x 4 waves x 1000 iterations).

  python scripts/sq_counters.py <out.json> <mgs_build_id() of the library that ran> <pass dir> [<pass dir> ...]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE tallies 128-B requests at 64 B).  A pass whose calibration counts are off by more than 3 % (SQ_INSTS_VALU counts the 8 matrix instructions too: 72 per iteration) is dropped and
listed under "rejected_passes"."""
import collections
import csv
import glob
import json
import sys

OURS = ("preprocess", "bin_", "coop_fwd", "gm_bwd", "calibration")
WAVES = 256 * 4
ITERS = 1000
# (SQ_INSTS_VALU counts the matrix instructions as well: 64 v_fma_f32 + 8 v_mfma per iteration)
EXPECT = {"SQ_INSTS_VALU": (64 + 8) * ITERS * WAVES, "SQ_INSTS_MFMA": 8 * ITERS * WAVES, "SQ_INSTS_VALU_MFMA_MOPS_F32": None,
          "SQ_INSTS_LDS": 4 * ITERS * WAVES, "SQ_WAVES": WAVES}


def load(d):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return None
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"].replace("void ", "").replace("mgs::", "").split("(")[0]
        if any(s in k for s in OURS):
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return agg


def main():
    out_path, so_hash, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    kernels = collections.defaultdict(dict)
    checks, rejected = {}, []
    for d in dirs:
        agg = load(d)
        if agg is None:
            rejected.append({"pass": d, "why": "no counter_collection.csv (rocprofv3 refused a counter name?)"})
            continue
        cal = next((v for k, v in agg.items() if "calibration" in k), None)
        ok = True
        if cal:
            for name, vals in cal.items():
                want = EXPECT.get(name)
                if want:
                    got = sum(vals) / len(vals)
                    lo, hi = want * 0.97, want * 1.03  # every calibration count within 3 % of the exact number
                    checks[f"{d.split('/')[-1]}:{name}"] = {"expected": want, "measured": got, "tolerance": 0.03,
                                                            "ok": lo <= got <= hi}
                    ok = ok and lo <= got <= hi
        if not ok:
            rejected.append({"pass": d, "why": "calibration kernel counts off", "checks": {k: v for k, v in checks.items() if d.split('/')[-1] in k}})
            continue
        for k, c in agg.items():
            for name, vals in c.items():
                kernels[k][name] = sum(vals) / len(vals)
                kernels[k].setdefault("launches_seen", len(vals))
    for k, c in kernels.items():
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            c["hbm_bytes_per_launch"] = int((2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024)
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
            # quad-cycles of VALU issue summed over waves / (SIMDs x busy cycles): how full the VALU issue slots were
            c["note"] = "SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)"
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <group> -- python bench.py --mode eager-st --only-mode --calibrate "
                         "--steps 10 --warmup 5 --no-cpu-baseline (one counter group per pass; scripts/gpu_round3.sh)",
               "lib_build_id": so_hash, "calibration_checks": checks, "rejected_passes": rejected,
               "hbm_correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request)",
               "kernels": kernels}, open(out_path, "w"), indent=1)
    print(json.dumps({k: {n: round(v) for n, v in c.items() if isinstance(v, float)} for k, c in kernels.items()
                      if "gm_bwd" in k or "coop_fwd" in k or "calibration" in k}, indent=1)[:3000])
    print("rejected:", rejected)


if __name__ == "__main__":
    main()
