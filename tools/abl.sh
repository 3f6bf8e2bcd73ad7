for a in ${ABLS:-0}; do
  WGNN_HIPCC_FLAGS="-DPG_ABL=$a" python -m windgnn_amd.build --force > /dev/null 2>&1 || exit 1
  echo "ABL=$a"; timeout -k 10 120 python tools/kernel_times.py pgemm 2>/dev/null | tail -1 || exit 1
done
