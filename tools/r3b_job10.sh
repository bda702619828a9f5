#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
MI_RTJ_LIB=$PWD/gmerlin-avdecoder_amd/lib/ab/lib_chunk3072.so timeout -k 10 300 python -m pytest tests/test_gpu_spec_index.py tests/test_gpu_parity.py -m gpu -x -q > $O/t_chunk3072.log 2>&1; echo "chunk3072 pytest rc=$?"; tail -2 $O/t_chunk3072.log
bash tools/ab_libs.sh 2 --no-stress --no-e2e --no-sweep --steps 10 -- product lib_chunk3072.so lib_chunk2560.so lib_chunk3072_lead640.so | tee $O/ab_walker_chunk.txt
timeout -k 10 400 python bench.py > $O/bench2.json 2> $O/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench2.json').read().strip().splitlines()[-1])
print(d['value'], {k:v['ms'] for k,v in d['kernels'].items()})
print({k:(v['frames_per_s'] if isinstance(v,dict) else v) for k,v in d['by_batch'].items() if k!='note'})
print(d['end_to_end']['fps'], d['speculative_index'], d['stress_amp64']['frames_per_s'])
PY
timeout -k 10 300 python bench.py --config streams4k > $O/bench_streams4k.json 2>$O/bench_streams4k.err; echo "streams4k rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_streams4k.json').read().strip().splitlines()[-1]); print('streams4k', d['value'], d['parity_checked'], d['parity_mismatches'])"
