for round in 1 2 3; do
  for v in order noorder; do
    if [ $v = noorder ]; then export DOOMGPU_FE_NO_ORDER=1; else unset DOOMGPU_FE_NO_ORDER; fi
    echo -n "$v: "; python3 tests/manual/gpu_kbench.py 2>&1 | grep "B=" | sed 's/.*setup \([0-9.]*\) ms raster \([0-9.]*\) ms.*/\1\/\2/' | tr '\n' ' '; echo
  done
done
