run() { # label, env fraction, args...
  l=$1; f=$2; shift 2
  UBDPT_MEM_FRACTION=$f timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-other-configs "$@" > gpurun_out/sw.json 2> gpurun_out/sw.err || { echo "$l failed"; tail -3 gpurun_out/sw.err; return 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]); print('$l', round(d['value'],1), d['kernels']['k_wf_gen']['launches'], d['config']['spp_total'])"
}
run C2_p16 0.8 --config C2 &&
run C2_p24 0.8 --config C2 --passes-in-flight 24 &&
run C2_p32 0.8 --config C2 --passes-in-flight 32 &&
run C2_p16 0.8 --config C2 &&
run C3_f.5 0.5 --config C3 &&
run C3_f.8 0.8 --config C3 &&
run C5_f.5 0.5 --config C5 --steps 4 &&
run C5_f.8 0.8 --config C5 --steps 4 &&
run C4_pps15 0.8 --passes-per-step 15 &&
run C4_pps16 0.8
