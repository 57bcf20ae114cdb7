#!/bin/bash
# round 3, GPU call 31: pass 1 with one / two / three literal-length codes per turn (-DITXI_LITS): the decoder alone on three contents, then the command
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r3qq
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for L in 1 2 3; do
  for c in hiseq legacy novaseq; do
    ITX_LIB=$PWD/tools/lits$L.so timeout -k 10 200 python tools/inflate_measure.py 8000000 100 2 content=$c cigar=mixed > $O/lits${L}_${c}.txt 2>&1 || { tail -5 $O/lits${L}_${c}.txt; exit 1; }
    echo "LITS=$L $c: $(grep 'kernels only' $O/lits${L}_${c}.txt) $(grep -o 'equal zlib: [A-Za-z]*' $O/lits${L}_${c}.txt)"
  done
done
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 900 python tools/ab_cli.py 500000000 100 4 \
  l1:LD_PRELOAD=/root/repo/tools/lits1.so \
  l2:LD_PRELOAD=/root/repo/tools/lits2.so \
  l3:LD_PRELOAD=/root/repo/tools/lits3.so \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -3 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r3qq/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), "mean", round(sum(w) / len(w), 3), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
