#!/bin/bash
# HBM traffic of the K1 kernels + MFMA activity of K4: separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), each with --kernel-trace only.  CONFIG=2|3, TAG=rNN
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
CONFIG=${CONFIG:-2}; TAG=${TAG:-r01}
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --config $CONFIG --steps 30 --warmup 5 --profile-mode > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$n.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ_VALU_MFMA_BUSY_CYCLES > gpurun_out/${TAG}_pmc_summary_c${CONFIG}.txt
python tools/pmc_k1_json.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE $CONFIG $TAG > gpurun_out/${TAG}_pmc_k1_c${CONFIG}.json
grep -E "k1a|k1b|k4_sage" gpurun_out/${TAG}_pmc_summary_c${CONFIG}.txt; cat gpurun_out/${TAG}_pmc_k1_c${CONFIG}.json
