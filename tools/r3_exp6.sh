#!/bin/bash
# round 3, GPU call 6: pass 1 as one code per turn — parity, kernels on the three contents, the command
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3f
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_bamwin.py -x -q > $O/pytest_inflate.txt 2>&1
echo "pytest rc $?"; tail -3 $O/pytest_inflate.txt
for c in legacy hiseq novaseq; do
  timeout -k 10 300 python tools/inflate_measure.py 8000000 100 2 content=$c cigar=mixed > $O/${c}_8M.txt 2>&1
  grep -v "^wrote\|^call" $O/${c}_8M.txt | tail -3
done
timeout -k 10 300 python tools/inflate_measure.py 14000000 100 2 content=hiseq cigar=mixed > $O/hiseq_14M.txt 2>&1
grep -v "^wrote\|^call" $O/hiseq_14M.txt | tail -3
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 1000 python tools/ab_cli.py 200000000 100 3 \
  c256w8:ITX_BGZF_CHUNK=268435456,ITX_RESERVE_BLOCKS=8192,ITX_RESERVE_WINDOWS=8 \
  c384w8:ITX_BGZF_CHUNK=402653184,ITX_RESERVE_BLOCKS=12288,ITX_RESERVE_WINDOWS=8 \
  c128w8:ITX_RESERVE_WINDOWS=8 \
  > $O/cli_hiseq_200M.json 2> $O/cli_hiseq_200M.err
echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3f/cli_hiseq_200M.json"))
for k in d["walls_s"]:
    print(k, d["walls_s"][k], d["scan_s"][k], [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
