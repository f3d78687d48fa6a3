# kernel-trace rows (grid, LDS, registers, duration) of the dominant kernel for two libraries:  bash tools/kt_compare.sh cfg5 libA.so libB.so
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=$1; shift
for lib in "$@"; do
  rm -rf $R/gpurun_out/ktc; (cd /tmp && PYPOLAR_AMD_LIB=$R/polar-codes_amd/lib/$lib rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktc -o k -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-pmc-traffic > /dev/null 2>&1)
  python3 - "$lib" $R/gpurun_out/ktc <<'PY'
import csv, glob, sys
lib, d = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "decode_kernel" in r["Kernel_Name"]]
    r = rows[-1]
    keys = [k for k in r if any(t in k for t in ("Grid", "Workgroup", "LDS", "Scratch", "VGPR", "SGPR"))]
    dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in rows]
    print(lib, {k: r[k] for k in keys}, "ms", [round(x, 2) for x in dur])
PY
done
