set -e
for cfg in "s1d6:--streams 1 --wavefront-depth 6" "s2d6:--streams 2 --wavefront-depth 6" "s3d6:--streams 3" "s2d4:--streams 2 --wavefront-depth 4" "s2d8:--streams 2 --wavefront-depth 8" "s2d12:--streams 2 --wavefront-depth 12" "s1d10:--streams 1 --wavefront-depth 10" "s2p3:--streams 2 --pipeline 3" "s2pif2:--streams 2 --passes-in-flight 2" "s4pif2:--streams 4 --passes-in-flight 2" "s2pif8:--streams 2 --passes-in-flight 8"; do
  n=${cfg%%:*}; a=${cfg#*:}
  timeout -k 10 200 python bench.py --steps 8 --warmup 1 --no-cpu-baseline $a > gpurun_out/sw_$n.log 2> gpurun_out/sw_$n.err
done
