#!/bin/bash
# Runs ON THE GPU BOX (via gpurun):  tools/gpu_call.sh <tag> <job> [<job> ...]     -- one parameterised runner for every call
# of a round; outputs land in gpurun_out/<tag>_*.  Jobs (arguments after the first ':' are comma-separated, none contain spaces):
#   probe:<name>                 tools/probes/<name>_probe                                  -> <tag>_probe_<name>.txt
#   bench:<cfg>[:ENV=v,ENV2=v]   one short bench line of a configuration (no CPU / counter / loop legs) -> <tag>_<cfg>[_k]_bench.json
#   ab:<cfg>:<envA>:<envB>...    same, once per environment variant (e.g. SBR_LIB=tools/probes/variants/libsbr_x.so), A/B on one box
#   full                         the driver's command, `python bench.py`                    -> <tag>_bench.json
#   timeline:<cfg>[:ENV=v,..]    rocprofv3 --kernel-trace --stats of a short run            -> <tag>_<cfg>_kernel_stats.csv, _timeline.txt
#   pmc:<cfg>                    FETCH_SIZE / WRITE_SIZE passes + tools/pmc_summary.py       -> <tag>_<cfg>_pmc.json
#   phases                       in-kernel phase counters of the C2 chains                  -> <tag>_rec_phases.txt
#   trace                        device-side stamps of the overlapped tail's consumers      -> <tag>_tail_trace.txt
#   py:<script>[:args,..][:ENV=v,..]   python tools/<script>.py args                          -> <tag>_py_<script>.txt
#   test:<pytest -k expr|all>[:ENV=v,..][:file]   pytest -m gpu                             -> <tag>_tests_<n>.txt
#   cmp:<spec>[;<spec>...]     tools/cmp_case.py on each spec (CELL:H[,H2]:B:T:key=value..., ';' between specs) -> <tag>_cmp.txt
#   bg:<job>                     the job in the background (joined at the end of the call)
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
mkdir -p $out
cd $repo
export TMPDIR=/tmp
SHORT="--no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3"
nt=0
nab=0
line() {      # one bench line, printed compactly
  python - "$1" "$2" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ph = {k: v for k, v in (d.get("phases_us") or {}).items() if k != "note"}
    print(sys.argv[2].ljust(54), d["ms_per_step"], d["value"], ph, (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
    try:
        print(open(sys.argv[1][:-4] + "err").read()[-800:])
    except Exception:
        pass
P
}
job() {
  IFS=':' read -r kind a b c d e <<< "$1"
  case $kind in
    probe) timeout 300 tools/probes/${a}_probe > $out/${tag}_probe_${a}.txt 2>&1; cat $out/${tag}_probe_${a}.txt | cut -c1-260 ;;
    bench) envs=$(echo "$b" | tr ',' ' ')
           f=$out/${tag}_${a}${c:+_$c}_bench
           env $envs timeout 600 python bench.py --config $a $SHORT > $f.json 2> $f.err; line $f.json "$a $envs" ;;
    ab)    i=0
           for v in "$b" "$c" "$d" "$e"; do
             [ -z "$v" ] && continue
             i=$((i+1)); nab=$((nab+1)); envs=$(echo "$v" | tr ',' ' ')
             f=$out/${tag}_${a}_ab${nab}_bench
             env $envs timeout 150 python bench.py --config $a $SHORT > $f.json 2> $f.err; line $f.json "$a $envs"
           done ;;
    full)  timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 1200 $out/${tag}_bench.err
           python -c "
import json; d=json.loads(open('$out/${tag}_bench.json').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','repeats','sustained','train_loop')}); print(d.get('roofline')); print(d.get('hbm_traffic')); print(d.get('phases_us')); print(d.get('other_configs')); print(d.get('mfma_counters'))" ;;
    timeline) envs=$(echo "$b" | tr ',' ' ')
           ( cd /tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${a}_stats -o s -- python $repo/bench.py --config $a --steps 6 --warmup 2 --repeats 1 --quick > $out/${tag}_${a}_stats.log 2>&1 )
           cp $(ls $out/${tag}_${a}_stats/*kernel_stats.csv | head -1) $out/${tag}_${a}_kernel_stats.csv
           python tools/trace_gaps.py $(ls $out/${tag}_${a}_stats/*kernel_trace.csv | head -1) 3 > $out/${tag}_${a}_timeline.txt 2>&1
           rm -rf $out/${tag}_${a}_stats
           echo "== timeline $a $envs"; cut -c1-150 $out/${tag}_${a}_timeline.txt ;;
    pmc)   ( cd /tmp
             short="python $repo/bench.py --config $a --steps 8 --warmup 3 --repeats 1 --quick"
             timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${a}_st -o s -- $short > /dev/null 2>&1
             SBR_TAIL_OVERLAP=2 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_${a}_pf -o f -- $short > /dev/null 2>&1
             SBR_TAIL_OVERLAP=2 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_${a}_pw -o w -- $short > /dev/null 2>&1 )
           python tools/pmc_summary.py $out/${tag}_${a}_pf $out/${tag}_${a}_pw $out/${tag}_${a}_st > $out/${tag}_${a}_pmc.json
           rm -rf $out/${tag}_${a}_pf $out/${tag}_${a}_pw $out/${tag}_${a}_st; head -c 1500 $out/${tag}_${a}_pmc.json ;;
    phases) ( timeout 120 python tools/rec_prof.py c2; timeout 120 python tools/tail_prof.py ) > $out/${tag}_rec_phases.txt 2>&1; cat $out/${tag}_rec_phases.txt | cut -c1-220 ;;
    trace) timeout 120 python tools/tail_trace.py 8 > $out/${tag}_tail_trace.txt 2>&1; tail -30 $out/${tag}_tail_trace.txt ;;
    py)    envs=$(echo "$c" | tr ',' ' ')
           env $envs timeout 300 python tools/$a.py $(echo "$b" | tr ',' ' ') > $out/${tag}_py_$a.txt 2>&1; tail -40 $out/${tag}_py_$a.txt | cut -c1-220 ;;
    test)  nt=$((nt+1)); envs=$(echo "$b" | tr ',' ' ')
           if [ "$a" = all ]; then sel=""; else sel="-k"; fi
           env $envs timeout 2700 python -m pytest ${c:-tests} -m gpu -q --durations=8 $sel ${sel:+"$a"} > $out/${tag}_tests_$nt.txt 2>&1
           echo "== tests [$a] $envs"; tail -14 $out/${tag}_tests_$nt.txt | cut -c1-250 ;;
    cmp)   specs=$(echo "${1#cmp:}" | tr ';' ' ')
           timeout 1800 python tools/cmp_case.py $specs >> $out/${tag}_cmp.txt 2>&1; tail -20 $out/${tag}_cmp.txt | cut -c1-400 ;;
    *) echo "unknown job $1" ;;
  esac
}
for j in "$@"; do
  if [[ $j == bg:* ]]; then ( job "${j#bg:}" ) > $out/${tag}_bg_$RANDOM.log 2>&1 & else job "$j"; fi
done
wait
for f in $out/${tag}_bg_*.log; do [ -f "$f" ] && { echo "== background job"; cat $f | cut -c1-250; }; done
