mkdir -p gpurun_out/r3h
export DRT_MEGA_MAX_LOG2=0
for cfg in "noshare DRT_SHARE=0" "s4 DRT_SHARE_MIN=4" "s16 DRT_SHARE_MIN=16" "s32 DRT_SHARE_MIN=32" "s48 DRT_SHARE_MIN=48"; do
  set -- $cfg; name=$1; shift
  for V in 9 72; do
  env "$@" timeout 600 python bench.py --views $V --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r3h/f${V}_$name.err | tail -1 > gpurun_out/r3h/f${V}_$name.json
  echo "== $name views $V"; python tools/benchsum.py gpurun_out/r3h/f${V}_$name.json | grep -E "Mrays|trace2|trace3"
  done
done
