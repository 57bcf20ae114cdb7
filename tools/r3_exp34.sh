#!/bin/bash
# round 3, GPU call 34: the command with pass 1 taking one code per turn (the library of tools/var_lits1 found through LD_LIBRARY_PATH:
# the only copy of the engine in the process) against the build (two codes, vote). Call 31's A/B preloaded the variant over the product's
# library: both register the same host stubs, and the kernels that ran were the product's in every variant — void.
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3ww
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
# the one-code library, a copy of the engine with only the decoder compiled differently
[ -f tools/var_lits1/libiteres_amd.so ] || { bash tools/build_inflate_variant.sh lits1 -DITXI_LITS=1 >> $O/build.txt 2>&1 && mkdir -p tools/var_lits1 && cp tools/lits1.so tools/var_lits1/libiteres_amd.so; }
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 6 \
  one_code:LD_LIBRARY_PATH=tools/var_lits1 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3ww/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
