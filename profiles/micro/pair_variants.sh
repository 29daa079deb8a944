for v in "" ny3 pub8; do
  if [ -z "$v" ]; then lib=""; else lib="MPLX_LIB=$PWD/profiles/micro/libmplx_$v.so"; fi
  echo "variant=$v"; env $lib python profiles/micro/c5_sweep.py C5 - 2>/dev/null
done
