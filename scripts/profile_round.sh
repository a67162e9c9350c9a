#!/bin/bash
# Profiles bench.py on the GPU box and leaves per-kernel summaries under gpurun_out/<tag>/ :
#   bash scripts/profile_round.sh r02a [passes]
# passes (default "trace fetch write l2 mfma"): trace = rocprofv3 --kernel-trace --stats; the others are separate --pmc
# passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Copy what is to be judged to profiles/.
set -e
TAG=${1:?tag}
PASSES=${2:-"trace fetch write l2 mfma"}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH_SHORT="$REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --in-flight 1 --no-bf16"
for p in $PASSES; do
  case $p in
    trace)
      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-bf16 \
        --no-cpu-baseline --in-flight 1 > "$OUT/trace.log" 2>&1
      cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
      python3 $REPO/scripts/prof_summary.py "$OUT/trace" 27 40 > "$OUT/kernel_stats.txt"
      ;;
    fetch)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $BENCH_SHORT > "$OUT/fetch.log" 2>&1
      python3 $REPO/scripts/pmc_summary.py "$OUT/fetch" FETCH_SIZE "$OUT/pmc_fetch.csv"
      ;;
    write)
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $BENCH_SHORT > "$OUT/write.log" 2>&1
      python3 $REPO/scripts/pmc_summary.py "$OUT/write" WRITE_SIZE "$OUT/pmc_write.csv"
      ;;
    l2)
      rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/l2" -- python3 $BENCH_SHORT > "$OUT/l2.log" 2>&1
      python3 $REPO/scripts/pmc_multi.py "$OUT/l2" "$OUT/pmc_l2.csv" TCC_HIT_sum TCC_MISS_sum
      ;;
    mfma)
      rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/mfma" \
        -- python3 $BENCH_SHORT > "$OUT/mfma.log" 2>&1
      python3 $REPO/scripts/pmc_multi.py "$OUT/mfma" "$OUT/pmc_mfma.csv" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
      ;;
    bf16trace)   # the bf16_bs4 region alone (BASELINE configs[4]: 4 x 1 M points, bf16 storage)
      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bf16trace" -- python3 $REPO/bench.py --bf16-only --bf16-steps 5 \
        > "$OUT/bf16trace.log" 2>&1
      cp "$(find "$OUT/bf16trace" -name '*kernel_stats.csv' | head -1)" "$OUT/bf16_kernel_stats.csv"
      python3 $REPO/scripts/prof_summary.py "$OUT/bf16trace" 7 30 > "$OUT/bf16_kernel_stats.txt"
      ;;
    bf16fetch)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/bf16fetch" -- python3 $REPO/bench.py --bf16-only --bf16-steps 2 \
        > "$OUT/bf16fetch.log" 2>&1
      python3 $REPO/scripts/pmc_summary.py "$OUT/bf16fetch" FETCH_SIZE "$OUT/bf16_pmc_fetch.csv"
      ;;
    bf16write)
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/bf16write" -- python3 $REPO/bench.py --bf16-only --bf16-steps 2 \
        > "$OUT/bf16write.log" 2>&1
      python3 $REPO/scripts/pmc_summary.py "$OUT/bf16write" WRITE_SIZE "$OUT/bf16_pmc_write.csv"
      ;;
    inst)
      rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/inst" \
        -- python3 $BENCH_SHORT > "$OUT/inst.log" 2>&1
      python3 $REPO/scripts/pmc_multi.py "$OUT/inst" "$OUT/pmc_inst.csv" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
      ;;
  esac
  echo "pass $p done"
  rm -rf "$OUT/$p"            # the raw per-dispatch CSVs are large; the summaries stay
done
python3 - "$OUT" "$REPO" "$TAG" <<'EOF'
import hashlib, json, os, sys, time
out, repo, tag = sys.argv[1:4]
sha = hashlib.sha256(open(os.path.join(repo, "detection_3d_amd/csrc/conv.hip"), "rb").read()).hexdigest()
meta = {"conv_hip_sha256": sha, "fetch_csv": f"{tag}_pmc_fetch.csv", "write_csv": f"{tag}_pmc_write.csv",
        "taken_at": time.strftime("%Y-%m-%d %H:%M:%S")}
if os.path.exists(os.path.join(out, "bf16_pmc_fetch.csv")) and os.path.exists(os.path.join(out, "bf16_pmc_write.csv")):
    sha16 = hashlib.sha256(open(os.path.join(repo, "detection_3d_amd/csrc/conv_bf16.hip"), "rb").read()).hexdigest()
    meta["bf16"] = {"sha256": sha16, "fetch_csv": f"{tag}_bf16_pmc_fetch.csv", "write_csv": f"{tag}_bf16_pmc_write.csv",
                    "taken_at": meta["taken_at"]}
json.dump(meta, open(os.path.join(out, "pmc_current.json"), "w"), indent=1)
EOF
ls -la "$OUT"
