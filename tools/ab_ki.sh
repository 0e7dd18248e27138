export TMPDIR=/tmp
python -c 'import __graft_entry__ as g; g.build()' > /dev/null 2>&1
for i in 1 2; do
for L in base ki1 ki3 ki4; do
  if [ $L = base ]; then unset LF_HIP_LIB; else export LF_HIP_LIB=$PWD/liberate_fhe_amd/csrc/variants/lib_$L.so; fi
  echo "$L $(python tools/eo.py quick 2>/dev/null | head -1)"
done; done
