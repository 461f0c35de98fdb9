#!/bin/bash
# ONE parametrised GPU-box script for the round's experiments (replaces the per-experiment gpu_r4*.sh of round 4): a sequence of
# steps given as arguments, each appending to gpurun_out/<tag>_*.  Usage (one gpurun call):
#   gpurun -- bash tools/gpu_run.sh <tag> <step> [<step> ...]
# steps:
#   pytest[:<pytest args>]            the -m gpu suite (default: whole suite)            -> <tag>_pytest.log
#   envab:<specs>:<envset>|<envset>.. tools/gpu_env_ab.sh: bench.py A/B of environment switches (specs "c3 30 20;c4 8 6")
#   libab:<variants>[:<bench args>]   tools/gpu_ab.sh: A/B of variant libraries build/ablate/lib_<v>.so
#   bench[:<bench args>]              the default bench line                              -> <tag>_bench.json
#   evidence[:<parts>]                tools/gpu_evidence.sh <tag> <parts>
#   sh:<command>                      anything else
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" == "$step" ] && rest=""
  echo "=== [$TAG] $step  ($(date +%T))"
  case $kind in
    pytest)
      timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -q -m gpu ${rest:--x} > /tmp/${TAG}_pytest_full.log 2>&1
      { grep -E "^(FAILED|ERROR)|^E  |passed|failed| error" /tmp/${TAG}_pytest_full.log | head -80; echo "-----"; tail -${PYTEST_TAIL:-60} /tmp/${TAG}_pytest_full.log; } > gpurun_out/${TAG}_pytest.log
      head -40 gpurun_out/${TAG}_pytest.log ;;
    envab)
      specs=${rest%%:*}; sets=${rest#*:}
      IFS='|' read -r -a ES <<< "$sets"
      AB_REPS=${AB_REPS:-1} bash tools/gpu_env_ab.sh ${TAG} "$specs" "${ES[@]}" ;;
    libab)
      vars=${rest%%:*}; bargs=${rest#*:}; [ "$bargs" == "$rest" ] && bargs=""
      bash tools/gpu_ab.sh ${TAG} "$vars" $bargs ;;
    bench)
      timeout 900 python bench.py $rest 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('value'), v.get('latency_ms_single_env')) for k,v in d['extra'].get('configs',{}).items()})" ;;
    evidence)
      bash tools/gpu_evidence.sh ${TAG} $rest ;;
    sh)
      bash -c "$rest" 2>&1 | tail -40 ;;
    *) echo "unknown step $kind" ;;
  esac
done
