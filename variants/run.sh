L=motion_primitive_library_amd/csrc/libmplx.so
cp $L /tmp/orig.so
for v in variants/lib_*.so; do
  cp $v $L
  for bc in 0 529; do
  if [ $bc = 0 ]; then unset MPLX_GRID_BOXCAP; else export MPLX_GRID_BOXCAP=$bc; fi
  timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 2 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); print('$v boxcap=$bc', round(j['roofline']['kernel_ms'],4))"
  done
done
cp /tmp/orig.so $L
