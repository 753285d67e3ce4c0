set -o pipefail
for cfg in "2 2" "3 3" "1 1" "2 4" "3 1" "2 1" "1 2"; do
  set -- $cfg
  NEMO_AMD_SCALE_LANES=$1 NEMO_AMD_SCALE_GROUP=$2 timeout -k 10 120 python bench.py --cpu-procs 0 --steps 20 --warmup 5 --no-profile > gpurun_out/sw_$1_$2.json 2> gpurun_out/sw_$1_$2.err || exit 1
  python -c "import json;d=json.loads(open('gpurun_out/sw_$1_$2.json').read().strip().splitlines()[-1]);print('lanes $1 group $2', d['value'], d['ms_per_step'])"
done
