for rep in 1 2; do for v in nosmall small; do echo "== $v rep $rep"; SEMSEG_HIP_LIB=gpurun_variants/lib_$v.so timeout 200 python scripts/host_overhead.py 2>&1 | grep -E "^B= [248]"; done; done
