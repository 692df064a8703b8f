for L in "" tools/ab/libnosort.so; do
  echo "lib=$L"; SPBLAS_GFX950_LIB=${L:+$PWD/$L} SPBLAS_GFX950_TRACE_INSPECT=1 python tools/inspect_repeat.py 2>&1 | tail -9 | grep -E "scatter|work lists"
done
