for v in base slab2 base slab2; do
  DRT_HIP_LIB=$PWD/drt_amd/_ab/$v.so python bench.py --no-cpu-baseline > gpurun_out/ab_$v.json 2>/dev/null
  echo "== $v"; python tools/benchsum.py gpurun_out/ab_$v.json | grep -E "Mrays|trace2|trace3"
done
