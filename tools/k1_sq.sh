cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/k1sq; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/raw -- python tools/pmc_k1.py 8192 10 > $O/log 2>$O/err; python tools/sq_summarise.py $(ls $O/raw/*/*counter_collection.csv) > $O/sq.json; rm -rf $O/raw
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/raw -- python tools/pmc_k1.py 8192 10 > $O/log2 2>$O/err2; python tools/sq_summarise.py $(ls $O/raw/*/*counter_collection.csv) > $O/sq2.json; rm -rf $O/raw
python - <<'PY'
import json
for f in ('gpurun_out/k1sq/sq.json','gpurun_out/k1sq/sq2.json'):
    d=json.load(open(f))
    for k,v in d.items():
        if 'fft_bank' in k: print(k[:30], {a:(round(b/1e6,1) if isinstance(b,float) and b>1000 else round(b,3)) for a,b in v.items()})
PY
