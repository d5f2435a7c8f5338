for L in ${LS:-8 10 12 14 16 18 19 20}; do echo "L=$L"; ZKHIP_SLOT_L=$L timeout -k 10 120 python tools/msm_bench.py 19 5 2>&1 | tail -1 || exit 1; done
