#!/bin/bash
# For the next round (nothing of round 3 ran this): the measurements DESIGN.md's open questions name, in one GPU call of about six minutes.
#   1. the command as built against the decoder with one 16-bit symbol table (-DITXI_SYM16), 5 runs each, interleaved
#   2. the host->device link alone, the reader alone, and both together, on the same 56 GB file
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r4probes
mkdir -p $O
export OMP_NUM_THREADS=16
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
bash tools/build_inflate_variant.sh sym16 -DITXI_SYM16 >> $O/build.txt 2>&1 && mkdir -p tools/var_sym16 && cp tools/sym16.so tools/var_sym16/libiteres_amd.so
ITX_LIB=$PWD/tools/sym16.so timeout -k 10 200 python tools/inflate_measure.py 8000000 100 2 content=hiseq cigar=mixed > $O/sym16_alone.txt 2>&1; grep "kernels only\|equal zlib" $O/sym16_alone.txt
ITX_AB_MKBAM="content=hiseq cigar=mixed" timeout -k 10 700 python tools/ab_cli.py 500000000 100 5 \
  sym16:LD_LIBRARY_PATH=tools/var_sym16 \
  > $O/cli_hiseq_500M.json 2> $O/cli_hiseq_500M.err
echo "rc $?"; tail -2 $O/cli_hiseq_500M.err
python - <<'PY'
import json, statistics
d = json.load(open("gpurun_out/r4probes/cli_hiseq_500M.json"))
print(d["same_outputs_as_base"])
for k in d["walls_s"]:
    w = d["walls_s"][k]
    print(k, w, "median", statistics.median(w), [l for l in d["notes"][k] if "device decoder" in l or "record loop" in l])
PY
bam=$(ls -d /tmp/itx_bench_r500000000_s100_t5500000_c0_hiseq_mixed 2>/dev/null | head -1)/reads.bam
[ -f "$bam" ] && timeout -k 10 120 python tools/link_vs_reader.py "$bam" 384 64 8 | tee $O/link_vs_reader.txt
