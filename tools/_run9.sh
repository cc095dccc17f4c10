cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pa -o p -- $R/tools/probes/attn_bwd_bench_w8 968 6 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pa/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': n[k]+=1
for k,v in agg.items():
    print(k, 'dispatches', n[k])
    for c,x in sorted(v.items()): print('   %-24s %14.0f per dispatch' % (c, x/max(n[k],1)))
PY
